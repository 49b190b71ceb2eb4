"""BASELINE.json configs[0]: a ColGREP-style index over THIS repo's source tree, used as a
plumbing case for the whole create -> write -> load -> search chain.

No encoder model is available offline, so a token's embedding is a seeded unit vector derived from the token text
(equal tokens get equal vectors, so lexical overlap drives MaxSim exactly like a real late-interaction index).
Index creation follows the reference's create path with the pieces restated in oracle/: codebook = sampled
embeddings (stand-in for k-means, which is outside the search path), nearest-centroid codes + residual
quantisation = encode_index_chunk (index.rs:289-371), bucket cutoffs/weights = residual quantiles
(index.rs:260-270), IVF = index.rs:479-499, files = write_index_from_encoded_chunks (index.rs:373-528).

CPU test: the C-ABI host loader and the oracle loader agree on the directory; searching for a code unit's own
tokens returns that unit; a path filter (ColGREP's subset) is respected.  GPU test: the HIP path loads the same
directory, encodes the same tokens to the same bits and returns the oracle's rankings."""
import hashlib
import os
import re

import numpy as np
import pytest

from helpers import O, RTOL_F32, ROOT, assert_ranking_close, synth, to_oracle_params

import next_plaid_amd as npa

DIM, NBITS, K = 128, 4, 256
TOK = re.compile(r"[A-Za-z_][A-Za-z0-9_]*|\d+")


def _units():
    """(path, first_line, tokens) for ~24-line units of the repo's own sources (ColGREP indexes code units)."""
    out = []
    for sub in ("next-plaid_amd", "oracle", "tests", "include", "tools"):
        for dp, dn, fn in sorted(os.walk(os.path.join(ROOT, sub))):
            dn[:] = sorted(d for d in dn if d not in ("__pycache__", "golden", "probes"))
            for f in sorted(fn):
                if not f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".c")):
                    continue
                p = os.path.join(dp, f)
                lines = open(p, errors="replace").read().splitlines()
                for i in range(0, len(lines), 24):
                    toks = TOK.findall(" ".join(lines[i:i + 24]))[:48]
                    if len(toks) >= 4:
                        out.append((os.path.relpath(p, ROOT), i + 1, toks))
    return out[:1500]


def _tok_vec(tok: str) -> np.ndarray:
    seed = int.from_bytes(hashlib.blake2b(tok.encode(), digest_size=8).digest(), "little")
    v = np.random.default_rng(seed).standard_normal(DIM).astype(np.float32)
    return v / np.linalg.norm(v)


@pytest.fixture(scope="module")
def source_index(tmp_path_factory):
    units = _units()
    assert len(units) > 300, "the repo should yield a few hundred code units"
    cache = {}
    doc_emb = []
    for _, _, toks in units:
        doc_emb.append(np.stack([cache.setdefault(t, _tok_vec(t)) for t in toks]))
    lens = np.array([e.shape[0] for e in doc_emb], np.int64)
    x = np.concatenate(doc_emb, 0)
    vocab = np.stack([cache[t] for t in sorted(cache)])
    rng = np.random.default_rng(2026)
    cen = vocab[rng.choice(vocab.shape[0], K, replace=False)].copy()
    codes0 = O.compress_into_codes(x, cen)
    res = x - cen[codes0]
    n = 1 << NBITS
    cut = np.quantile(res, [i / n for i in range(1, n)]).astype(np.float32)          # index.rs:260-270
    wts = np.quantile(res, [(i + 0.5) / n for i in range(n)]).astype(np.float32)
    codes, packed = O.encode_tokens(x, cen, NBITS, cut)
    assert np.array_equal(codes, codes0)
    ivf, ivf_lengths = synth.build_ivf(codes, lens, K)
    a = dict(centroids=cen, bucket_weights=wts, bucket_cutoffs=cut, ivf=ivf, ivf_lengths=ivf_lengths, doc_lengths=lens,
             codes=codes, residuals=packed, nbits=NBITS)
    path = str(tmp_path_factory.mktemp("c1") / "index")
    synth.write_index(path, a, chunk_docs=400)
    return dict(path=path, units=units, doc_emb=doc_emb, x=x, a=a, cut=cut)


def _params(**kw):
    kw.setdefault("n_full_scores", 256)
    kw.setdefault("top_k", 5)
    kw.setdefault("n_ivf_probe", 4)
    kw.setdefault("centroid_score_threshold", None)
    return npa.SearchParameters(**kw)


def test_c1_cpu_plumbing(source_index):
    s = source_index
    info = npa.probe_index_dir(s["path"])                               # C-ABI host loader
    assert info.num_documents == len(s["units"]) and info.num_partitions == K and info.embedding_dim == DIM
    assert info.num_embeddings == s["x"].shape[0] and info.nbits == NBITS
    ox = O.OracleIndex.load(s["path"])                                  # oracle loader, same directory
    assert ox.N == len(s["units"]) and np.array_equal(ox.codes, s["a"]["codes"])
    p = to_oracle_params(_params())
    hits = 0
    for d in range(0, len(s["units"]), 97):
        r = ox.search(s["doc_emb"][d], p)
        assert r.passage_ids.size == 5 and np.all(np.diff(r.scores) <= 0)
        hits += int(d in r.passage_ids[:2])        # an identical neighbouring unit may tie for rank 1
    assert hits >= 0.9 * len(range(0, len(s["units"]), 97))
    # ColGREP path filter == search(subset): only units from tests/
    subset = np.array([i for i, u in enumerate(s["units"]) if u[0].startswith("tests/")], np.int64)
    r = ox.search(s["doc_emb"][int(subset[3])], p, subset)
    assert r.passage_ids.size > 0 and set(r.passage_ids.tolist()) <= set(subset.tolist())
    assert int(subset[3]) in r.passage_ids


@pytest.mark.gpu
def test_c1_hip_matches_oracle(source_index):
    s = source_index
    ox = O.OracleIndex.load(s["path"])
    hx = npa.MmapIndex.load(s["path"])
    assert hx.num_documents() == ox.N and hx.num_partitions() == K
    codes, packed = hx.encode_tokens(s["x"], s["cut"])                  # N3 on the real token stream
    assert np.array_equal(codes, s["a"]["codes"]) and np.array_equal(packed, s["a"]["residuals"])
    qs = [s["doc_emb"][d] for d in range(0, len(s["units"]), 61)]       # ragged query lengths (4..48 tokens)
    subset = np.array([i for i, u in enumerate(s["units"]) if u[0].startswith("tests/")], np.int64)
    for p, sub in ((_params(), None), (_params(centroid_score_threshold=0.4, n_ivf_probe=8), None), (_params(), subset)):
        res = hx.search_batch(qs, p, subset=sub)
        ref = ox.search_batch(qs, to_oracle_params(p), subset=sub)
        for i, (r, o) in enumerate(zip(res, ref)):
            assert_ranking_close(r.passage_ids, r.scores, o.passage_ids, o.scores, RTOL_F32, f"C1 q{i}")
