"""Every index geometry the crate writes is searchable, not only the four kernel widths.

codec.rs:161-166 accepts nbits in {1, 2, 4, 8}; any dim with dim * nbits % 8 == 0 packs (codec.rs:356-411).  The HIP
kernels are instantiated at dim 32/64/96/128 and nbits 2/4 (+ 8 in the all-f32 S6 kernel); the library stores every other
index with dim <= 128 in one of those shapes (np_internal.h storage_dim: zero-padded rows, 1-bit buckets widened to 2-bit
segments) and still answers in the FILE geometry.  Each case is compared with the oracle exactly like the native shapes:
stage traces bit-equal (dense and batched probe), every S6 arithmetic within its tolerance, export / decompress / encode
in file geometry.
"""
import numpy as np
import pytest

from helpers import (O, RTOL_BF16, RTOL_BF16_PLAIN, RTOL_F32, assert_ranking_close, hip_index, make_arrays, oracle_index,
                     synth, to_oracle_params)

import importlib.util
import os

import next_plaid_amd as npa
from helpers import GOLDEN

pytestmark = pytest.mark.gpu

_s = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
MG = importlib.util.module_from_spec(_s)
_s.loader.exec_module(MG)

# (dim, nbits): padded rows (48, 50: 25-byte rows, 24, 100, 8), 1-bit (64 unpadded, 40 padded: 5-byte rows -> 16), 8-bit
# (32 and 128 unpadded, 72 padded to 96)
CASES = [(48, 4), (50, 4), (24, 2), (100, 2), (8, 4), (64, 1), (40, 1), (32, 8), (72, 8), (128, 8)]


def P(**kw):
    return npa.SearchParameters(**kw)


@pytest.fixture(scope="module", params=CASES, ids=[f"d{d}b{b}" for d, b in CASES])
def geo(request):
    dim, nbits = request.param
    spec, a = make_arrays(num_docs=3000, num_centroids=512, dim=dim, nbits=nbits, doc_len_min=6, doc_len_max=40,
                          seed=500 + dim + nbits)
    assert a["residuals"].shape[1] == dim * nbits // 8
    hx = hip_index(a, max_batch=16)
    qs, src = synth.make_queries(spec, 16, n_tokens=32, cen=a["centroids"])
    q40, _ = synth.make_queries(spec, 2, n_tokens=40, cen=a["centroids"], first_query=50)
    yield spec, a, oracle_index(a), hx, qs, src, q40
    hx.close()


def trace_equal(hx, ox, q, p, rtol, what):
    tr = hx.debug_trace(q, p)
    t = ox.search(q, to_oracle_params(p), trace=True).trace
    assert np.array_equal(tr["cells"], t.cells), f"{what}: S2 cells"
    assert np.array_equal(tr["cand"], t.cand), f"{what}: S3 candidates"
    bad = np.nonzero(tr["approx"].view(np.uint32) != t.approx.view(np.uint32))[0]
    assert bad.size == 0, f"{what}: S4 approx not bit-exact at {bad[:5]}: {tr['approx'][bad[:5]]} vs {t.approx[bad[:5]]}"
    assert np.array_equal(tr["sel"], t.sel), f"{what}: S5 selection / order"
    tol = rtol * np.maximum(np.abs(t.sel_exact), 1.0)
    assert np.all(np.abs(tr["sel_exact"] - t.sel_exact) <= tol), \
        f"{what}: S6 max rel {np.max(np.abs(tr['sel_exact'] - t.sel_exact) / np.maximum(np.abs(t.sel_exact), 1))}"


def test_file_geometry_at_the_boundary(geo):
    spec, a, ox, hx, qs, src, q40 = geo
    dim, nbits = spec.dim, spec.nbits
    assert hx.embedding_dim() == dim and int(hx.info.nbits) == nbits
    e = hx.export()
    for k in ("doc_lengths", "codes", "residuals", "ivf", "ivf_lengths"):
        assert np.array_equal(e[k], np.asarray(a[k])), k
    ids = [0, 17, 2999, 5]
    embs, lens = hx.decompress_documents(ids)
    ref = np.concatenate([ox.get_document_embeddings(i) for i in ids])
    assert embs.shape == ref.shape == (int(lens.sum()), dim) and np.allclose(embs, ref, rtol=0, atol=3e-7)
    with pytest.raises(npa.ShapeError):
        hx.search(np.zeros((4, dim + 8), np.float32), P())
    with pytest.raises(npa.ShapeError):   # the STORAGE width is not a query width
        hx.search(np.zeros((4, 128 if dim != 128 else 96), np.float32), P())


def test_stage_traces_bit_equal(geo):
    spec, a, ox, hx, qs, src, q40 = geo
    for thr in (0.4, None):
        p = P(n_full_scores=256, top_k=10, n_ivf_probe=8, centroid_score_threshold=thr)
        for qi in (0, 1, 2):
            trace_equal(hx, ox, qs[qi], p, RTOL_F32, f"d{spec.dim}b{spec.nbits} thr={thr} q{qi}")
    p = P(n_full_scores=128, top_k=5, n_ivf_probe=4)
    trace_equal(hx, ox, qs[3][:7], p, RTOL_F32, "7-token query")
    trace_equal(hx, ox, q40[0], p, RTOL_F32, "40-token query (two tiles)")
    # batched probe + mat-vec approximate scores in unrolled_dot order (search.rs:140-254, 259-302): the padded row's
    # last dim % 8 products are added one by one after the eight partial sums
    for cbs in (100, 200):
        p = P(n_full_scores=256, top_k=10, n_ivf_probe=8, centroid_score_threshold=0.4, centroid_batch_size=cbs)
        for qi in (4, 5):
            trace_equal(hx, ox, qs[qi], p, RTOL_F32, f"batched cbs={cbs} q{qi}")


def test_batch_matches_oracle_in_every_precision(geo):
    spec, a, ox, hx, qs, src, q40 = geo
    for prec, rtol in ((2, RTOL_F32), (0, RTOL_F32), (1, RTOL_BF16), (3, RTOL_BF16_PLAIN)):
        p = P(n_full_scores=512, top_k=10, n_ivf_probe=16, precision=prec)
        batch = qs + [q40[1], qs[0][:3]]
        res = hx.search_batch(batch, p)
        ref = ox.search_batch(batch, to_oracle_params(p))
        for i, (r, o) in enumerate(zip(res, ref)):
            assert_ranking_close(r.passage_ids, r.scores, o.passage_ids, o.scores, rtol, f"prec={prec} q{i}")
    # the two-level filter and the plain one select the same documents
    p = P(n_full_scores=512, top_k=128, n_ivf_probe=16)
    got = hx.search_batch(qs, p)
    for knob, v in (("s4_hot", 0), ("s4_filter", 0)):
        hx.tune(knob, v)
        try:
            for g, o in zip(hx.search_batch(qs, p), got):
                assert np.array_equal(g.passage_ids, o.passage_ids) and np.array_equal(g.scores, o.scores), knob
        finally:
            hx.tune(knob, 100 if knob == "s4_hot" else 1)


def test_encode_in_file_geometry(geo):
    spec, a, ox, hx, qs, src, q40 = geo
    dim, nbits, K = spec.dim, spec.nbits, spec.num_centroids
    rng = np.random.default_rng(93)
    n = 300
    x = a["centroids"][rng.integers(0, K, n)] + (0.3 / np.sqrt(dim)) * rng.standard_normal((n, dim)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    x[5] = 0.0
    cut, _ = synth.bucket_tables(spec)
    codes, packed = hx.encode_tokens(x, cut)
    rc, rp = O.encode_tokens(x, a["centroids"], nbits, cut)
    assert packed.shape == (n, dim * nbits // 8)
    assert np.array_equal(codes, rc) and codes[5] == K - 1
    assert np.array_equal(packed, rp)


@pytest.mark.parametrize("geo", [MG.geo_name(k) for k in MG.GEO_SPECS])
def test_committed_golden_vectors(geo):
    """tests/golden/search_geometry.npz (minted by the numpy restatement): cells and candidates equal, selected set equal,
    ranking within the f32 tolerance -- through search() and through the batched entry."""
    kw = next(k for k in MG.GEO_SPECS if MG.geo_name(k) == geo)
    spec = synth.SynthSpec(**kw)
    hx = hip_index(synth.generate_arrays(spec))
    gold = np.load(os.path.join(GOLDEN, "search_geometry.npz"))
    qs = [q for q in gold[f"{geo}_queries"]]
    for name, pk in MG.GEO_CASES:
        p = P(**pk)
        batch = hx.search_batch(qs, p)
        for qi, q in enumerate(qs):
            k = f"{geo}_{name}_q{qi}"
            tr = hx.debug_trace(q, p)
            assert np.array_equal(tr["cells"], gold[k + "_cells"]), f"{k} cells"
            assert np.array_equal(tr["cand"], gold[k + "_cand"]), f"{k} candidates"
            assert set(tr["sel"].tolist()) == set(gold[k + "_sel"].tolist()), f"{k} selection"
            for r in (hx.search(q, p), batch[qi]):
                assert_ranking_close(r.passage_ids, r.scores, gold[k + "_ids"], gold[k + "_scores"], RTOL_F32, k)
    hx.close()


def test_on_disk_and_sharded_open(tmp_path):
    """The loader path (chunk files, shard ranges) through the repacking upload: dim 48 / 4-bit and dim 40 / 1-bit."""
    for dim, nbits in ((48, 4), (40, 1)):
        spec, a = make_arrays(num_docs=700, num_centroids=64, dim=dim, nbits=nbits, doc_len_min=3, doc_len_max=30, seed=31)
        d = tmp_path / f"d{dim}b{nbits}"
        d.mkdir()
        synth.write_index(str(d), a, chunk_docs=150)
        hd, ha = npa.MmapIndex.load(str(d)), hip_index(a)
        assert hd.embedding_dim() == dim
        e1, e2 = hd.export(), ha.export()
        for k in ("doc_lengths", "codes", "residuals", "ivf", "ivf_lengths"):
            assert np.array_equal(e1[k], e2[k]) and np.array_equal(e1[k], np.asarray(a[k])), k
        qs, _ = synth.make_queries(spec, 3, n_tokens=32, cen=a["centroids"])
        p = P(n_full_scores=128, top_k=10, n_ivf_probe=8)
        ref = oracle_index(a).search_batch(qs, to_oracle_params(p))
        for r1, r2, o in zip(hd.search_batch(qs, p), ha.search_batch(qs, p), ref):
            assert np.array_equal(r1.passage_ids, r2.passage_ids) and np.array_equal(r1.scores, r2.scores)
            assert_ranking_close(r1.passage_ids, r1.scores, o.passage_ids, o.scores, RTOL_F32, f"d{dim}b{nbits}")
        hs = npa.MmapIndex.load(str(d), shard_rank=1, shard_count=3)
        b0, b1 = hs.info.shard_doc_begin, hs.info.shard_doc_end
        off = np.concatenate([[0], np.cumsum(a["doc_lengths"])])
        es = hs.export()
        assert np.array_equal(es["residuals"], a["residuals"][off[b0]:off[b1]])
        assert np.array_equal(es["codes"], a["codes"][off[b0]:off[b1]])
        for h in (hd, ha, hs):
            h.close()


def test_wider_than_the_kernels_is_a_shape_error():
    spec, a = make_arrays(num_docs=50, num_centroids=16, dim=160, nbits=4, doc_len_min=4, doc_len_max=4, seed=1)
    h = hip_index(a)      # loads (decompress / export work); no search kernel is that wide -> Shape, never a silent fallback
    assert h.embedding_dim() == 160
    embs, lens = h.decompress_documents([3])
    assert np.allclose(embs, oracle_index(a).get_document_embeddings(3), rtol=0, atol=3e-7)
    with pytest.raises(npa.ShapeError):
        h.search(np.zeros((4, 160), np.float32), P())
    h.close()
