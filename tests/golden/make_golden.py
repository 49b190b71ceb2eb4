"""Mint the golden fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference (Rust) cannot be built here, and its own tests pin no search() rankings (SURVEY.md
F7), so these vectors are minted from oracle/plaid_numpy.py -- the independent numpy restatement --
on seeded synthetic indices, and the C oracle (and through it the HIP path) is checked against
them.  Known answers that DO come from the reference's tests are hard-coded in
tests/test_oracle_known_answers.py, not here.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "next-plaid_amd"))

from next_plaid_amd import synth  # noqa: E402
from oracle import plaid_numpy as PN  # noqa: E402
from oracle.oracle import SearchParameters  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

GOLDEN_SPEC = dict(num_docs=2000, num_centroids=512, dim=128, nbits=4, doc_len_min=20, doc_len_max=60, seed=20260925)
CASES = [  # name, params, subset kind
    ("dense_thr", dict(n_full_scores=256, top_k=10, n_ivf_probe=8, centroid_score_threshold=0.4), None),
    ("dense_nothr", dict(n_full_scores=256, top_k=10, n_ivf_probe=8, centroid_score_threshold=None), None),
    ("batched_thr", dict(n_full_scores=256, top_k=10, n_ivf_probe=8, centroid_score_threshold=0.4, centroid_batch_size=100), None),
    ("batched_nothr", dict(n_full_scores=256, top_k=10, n_ivf_probe=8, centroid_score_threshold=None, centroid_batch_size=100), None),
    ("dense_subset", dict(n_full_scores=256, top_k=5, n_ivf_probe=4, centroid_score_threshold=None), "even"),
    ("batched_subset", dict(n_full_scores=256, top_k=5, n_ivf_probe=4, centroid_score_threshold=None, centroid_batch_size=100), "even"),
]

# index geometries without a kernel instantiation (tests/test_gpu_geometry.py): rows that are not a kernel width, 1- and
# 8-bit residuals (codec.rs:161-166)
GEO_SPECS = [dict(num_docs=800, num_centroids=128, dim=50, nbits=4, doc_len_min=8, doc_len_max=30, seed=20260926),
             dict(num_docs=800, num_centroids=128, dim=40, nbits=1, doc_len_min=8, doc_len_max=30, seed=20260927),
             dict(num_docs=800, num_centroids=128, dim=72, nbits=8, doc_len_min=8, doc_len_max=30, seed=20260928)]
GEO_CASES = [("dense_thr", dict(n_full_scores=128, top_k=10, n_ivf_probe=8, centroid_score_threshold=0.4)),
             ("batched_nothr", dict(n_full_scores=128, top_k=10, n_ivf_probe=8, centroid_score_threshold=None, centroid_batch_size=50))]


def geo_name(spec):
    return f"d{spec['dim']}b{spec['nbits']}"


def mint_geometry():
    g = np.random.Generator(np.random.PCG64(11))
    cen = g.standard_normal((16, 24), dtype=np.float32)
    cen /= np.linalg.norm(cen, axis=1, keepdims=True)
    for nbits in (1, 8):
        w = np.sort(g.standard_normal(1 << nbits).astype(np.float32) * 0.1)
        packed = g.integers(0, 256, size=(8, 24 * nbits // 8), dtype=np.uint8)
        codes = g.integers(0, 16, size=8).astype(np.int64)
        out = PN.decompress(packed, codes, cen, w, nbits)
        np.savez(os.path.join(OUT, f"decompress_nbits{nbits}.npz"), centroids=cen, weights=w, packed=packed,
                 codes=codes, out=out)
    gold = {}
    for kw in GEO_SPECS:
        spec = synth.SynthSpec(**kw)
        a = synth.generate_arrays(spec)
        nx = PN.NumpyIndex(a["centroids"], a["bucket_weights"], a["ivf"], a["ivf_lengths"], a["doc_lengths"],
                           a["codes"], a["residuals"], spec.nbits)
        qs, src = synth.make_queries(spec, 3, n_tokens=32, cen=a["centroids"])
        tag = geo_name(kw)
        gold[f"{tag}_queries"] = np.stack(qs)
        for name, pk in GEO_CASES:
            p = SearchParameters(**pk)
            for qi, q in enumerate(qs):
                ids, sc, tr = nx.search(q, p, None, return_trace=True)
                gold[f"{tag}_{name}_q{qi}_ids"] = ids
                gold[f"{tag}_{name}_q{qi}_scores"] = sc
                gold[f"{tag}_{name}_q{qi}_cells"] = tr["cells"]
                gold[f"{tag}_{name}_q{qi}_cand"] = tr["cand"]
                gold[f"{tag}_{name}_q{qi}_sel"] = tr.get("sel", np.zeros(0, np.int64))
    np.savez_compressed(os.path.join(OUT, "search_geometry.npz"), **gold)


def main():
    # (i) bit-unpack tables for every byte value: bucket ids per dim, straight from the packing rule
    tabs = {}
    for nbits in (1, 2, 4, 8):
        allb = np.arange(256, dtype=np.uint8)[:, None]
        tabs[f"nbits{nbits}"] = PN.bucket_indices(allb, nbits).astype(np.int32)
    np.savez(os.path.join(OUT, "unpack_tables.npz"), **tabs)

    # (ii) decompress of a seeded 8-token document
    g = np.random.Generator(np.random.PCG64(7))
    cen = g.standard_normal((16, 32), dtype=np.float32)
    cen /= np.linalg.norm(cen, axis=1, keepdims=True)
    for nbits in (2, 4):
        w = np.sort(g.standard_normal(1 << nbits).astype(np.float32) * 0.1)
        packed = g.integers(0, 256, size=(8, 32 * nbits // 8), dtype=np.uint8)
        codes = g.integers(0, 16, size=8).astype(np.int64)
        out = PN.decompress(packed, codes, cen, w, nbits)
        np.savez(os.path.join(OUT, f"decompress_nbits{nbits}.npz"), centroids=cen, weights=w, packed=packed,
                 codes=codes, out=out)

    # (iii) full search on a seeded clustered 2000-doc index
    spec = synth.SynthSpec(**GOLDEN_SPEC)
    a = synth.generate_arrays(spec)
    nx = PN.NumpyIndex(a["centroids"], a["bucket_weights"], a["ivf"], a["ivf_lengths"], a["doc_lengths"],
                       a["codes"], a["residuals"], spec.nbits)
    qs, src = synth.make_queries(spec, 6, n_tokens=32, cen=a["centroids"])
    gold = dict(queries=np.stack(qs), src=src)
    for name, kw, sub in CASES:
        p = SearchParameters(**kw)
        subset = None if sub is None else np.arange(0, spec.num_docs, 2, dtype=np.int64)
        for qi, q in enumerate(qs):
            ids, sc, tr = nx.search(q, p, subset, return_trace=True)
            gold[f"{name}_q{qi}_ids"] = ids
            gold[f"{name}_q{qi}_scores"] = sc
            gold[f"{name}_q{qi}_cells"] = tr["cells"]
            gold[f"{name}_q{qi}_cand"] = tr["cand"]
            gold[f"{name}_q{qi}_sel"] = tr.get("sel", np.zeros(0, np.int64))
    np.savez_compressed(os.path.join(OUT, "search_2000.npz"), **gold)
    mint_geometry()
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
