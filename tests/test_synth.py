"""Generator spec + writer/reader round trip + IVF invariants (index.rs:2011-2054's checks).  CPU only."""
import numpy as np

from helpers import synth
from oracle import npy_index


def test_generator_is_deterministic_and_shardable():
    spec = synth.SynthSpec(num_docs=500, num_centroids=128, dim=64, nbits=4, doc_len_min=3, doc_len_max=40, seed=11)
    c1, r1, l1 = synth.doc_tokens(spec, 0, 500)
    c2, r2, l2 = synth.doc_tokens(spec, 0, 500)
    assert np.array_equal(c1, c2) and np.array_equal(r1, r2) and np.array_equal(l1, l2)
    ca, ra, la = synth.doc_tokens(spec, 0, 200)
    cb, rb, lb = synth.doc_tokens(spec, 200, 500)
    assert np.array_equal(np.concatenate([ca, cb]), c1) and np.array_equal(np.concatenate([ra, rb]), r1)
    assert c1.min() >= 0 and c1.max() < 128 and l1.min() >= 3 and l1.max() <= 40
    assert abs(np.unpackbits(r1).mean() - 0.5) < 0.01          # residual bytes are uniform


def test_ivf_invariants_and_disk_roundtrip(tmp_path):
    spec = synth.SynthSpec(num_docs=300, num_centroids=64, dim=64, nbits=2, doc_len_min=0, doc_len_max=25, seed=2)
    a = synth.generate_arrays(spec)
    off = np.concatenate([[0], np.cumsum(a["ivf_lengths"])])
    assert off[-1] == a["ivf"].size
    for c in range(64):
        lst = a["ivf"][off[c]:off[c + 1]]
        assert np.all(np.diff(lst) > 0)                          # ascending, no duplicate ids per bucket
    iv2, il2 = npy_index.build_ivf(a["codes"], a["doc_lengths"], 64)   # oracle-side builder agrees
    assert np.array_equal(iv2, a["ivf"]) and np.array_equal(il2, a["ivf_lengths"])
    synth.write_index(str(tmp_path), a, chunk_docs=128)
    b = npy_index.read_index(str(tmp_path))
    for k in ("centroids", "bucket_weights", "ivf", "ivf_lengths", "doc_lengths", "codes", "residuals"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    assert b["metadata"]["num_chunks"] == 3 and b["metadata"]["num_documents"] == 300


def test_queries_hit_their_source_centroids():
    spec = synth.SynthSpec(num_docs=200, num_centroids=256, dim=128, doc_len_min=30, doc_len_max=30, seed=4)
    cen = synth.centroids(spec)
    qs, src = synth.make_queries(spec, 3, cen=cen)
    for q in qs:
        assert q.shape == (32, 128) and np.allclose(np.linalg.norm(q, axis=1), 1, atol=1e-5)
        assert ((q @ cen.T).max(1) > 0.6).all()


def test_lognormal_length_table_is_the_config3_shape():
    """document lengths ~ clipped LogNormal(mean ~ 73, max 180); the table is the generator's
    ragged-length mode (host statement; the device generator indexes the same table with the same hash)."""
    tab = synth.lognormal_len_table()
    assert tab.dtype == np.int32 and tab.size == 1024 and tab.min() >= 1 and tab.max() == 180
    assert np.all(np.diff(tab) >= 0) and 68 < tab.mean() < 76
    spec = synth.SynthSpec(num_docs=4000, num_centroids=64, dim=64, nbits=2, doc_len_min=1, doc_len_max=180, seed=3,
                           len_table=tab)
    l1 = synth.doc_lengths(spec, 0, 4000)
    assert np.array_equal(l1[1000:3000], synth.doc_lengths(spec, 1000, 3000))      # shardable
    assert 65 < l1.mean() < 80 and l1.max() <= 180 and l1.min() >= 1 and np.unique(l1).size > 60
    codes, res, lens = synth.doc_tokens(spec, 0, 50)
    assert np.array_equal(lens, l1[:50]) and codes.size == lens.sum()
