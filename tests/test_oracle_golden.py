"""The C oracle against the committed golden fixtures (minted by tests/golden/make_golden.py from the
independent numpy restatement) and against the numpy restatement live.  CPU only."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, O, RTOL_F32, assert_ranking_close, oracle_index, synth
from oracle import plaid_numpy as PN

import importlib.util

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
MG = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(MG)


@pytest.mark.parametrize("nbits", [1, 2, 4, 8])
def test_unpack_tables(nbits):
    gold = np.load(os.path.join(GOLDEN, "unpack_tables.npz"))[f"nbits{nbits}"]
    rev = O.byte_reversed_bits_map(nbits)
    lut = O.bucket_weight_indices_lookup(nbits)
    got = lut[rev[np.arange(256)]]            # decompress's two-LUT composition (codec.rs:449-451)
    assert np.array_equal(got, gold)


@pytest.mark.parametrize("nbits", [1, 2, 4, 8])
def test_decompress_golden(nbits):
    g = np.load(os.path.join(GOLDEN, f"decompress_nbits{nbits}.npz"))
    out = O.decompress(g["packed"], g["codes"], g["centroids"], g["weights"], nbits)
    assert np.allclose(out, g["out"], rtol=0, atol=2e-7)
    assert np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-6)


@pytest.fixture(scope="module")
def golden_index():
    spec = synth.SynthSpec(**MG.GOLDEN_SPEC)
    a = synth.generate_arrays(spec)
    return spec, a, oracle_index(a), np.load(os.path.join(GOLDEN, "search_2000.npz"))


@pytest.mark.parametrize("case", [c[0] for c in MG.CASES])
def test_search_golden(golden_index, case):
    spec, a, ix, gold = golden_index
    name, kw, sub = next(c for c in MG.CASES if c[0] == case)
    p = O.SearchParameters(**kw)
    subset = None if sub is None else np.arange(0, spec.num_docs, 2, dtype=np.int64)
    for qi, q in enumerate(gold["queries"]):
        r = ix.search(q, p, subset, trace=True)
        assert np.array_equal(r.trace.cells, gold[f"{name}_q{qi}_cells"]), f"{name} q{qi} cells"
        assert np.array_equal(r.trace.cand, gold[f"{name}_q{qi}_cand"]), f"{name} q{qi} candidates"
        assert set(r.trace.sel.tolist()) == set(gold[f"{name}_q{qi}_sel"].tolist()), f"{name} q{qi} selection"
        assert_ranking_close(r.passage_ids, r.scores, gold[f"{name}_q{qi}_ids"], gold[f"{name}_q{qi}_scores"],
                             RTOL_F32, f"{name} q{qi}")
        assert r.passage_ids[0] == gold["src"][qi] or sub is not None  # the source doc is found


@pytest.mark.parametrize("geo", [MG.geo_name(k) for k in MG.GEO_SPECS])
def test_search_golden_other_geometries(geo):
    """dim 50 / 4-bit, dim 40 / 1-bit, dim 72 / 8-bit (codec.rs:161-166 geometries without a kernel instantiation)."""
    kw = next(k for k in MG.GEO_SPECS if MG.geo_name(k) == geo)
    spec = synth.SynthSpec(**kw)
    ix = oracle_index(synth.generate_arrays(spec))
    gold = np.load(os.path.join(GOLDEN, "search_geometry.npz"))
    for name, pk in MG.GEO_CASES:
        p = O.SearchParameters(**pk)
        for qi, q in enumerate(gold[f"{geo}_queries"]):
            r = ix.search(q, p, None, trace=True)
            k = f"{geo}_{name}_q{qi}"
            assert np.array_equal(r.trace.cells, gold[k + "_cells"]), f"{k} cells"
            assert np.array_equal(r.trace.cand, gold[k + "_cand"]), f"{k} candidates"
            assert set(r.trace.sel.tolist()) == set(gold[k + "_sel"].tolist()), f"{k} selection"
            assert_ranking_close(r.passage_ids, r.scores, gold[k + "_ids"], gold[k + "_scores"], RTOL_F32, k)


def test_oracle_vs_numpy_ragged_and_edges():
    # ragged docs incl. empty ones, K not a multiple of 32, short queries, top_k > candidates
    spec = synth.SynthSpec(num_docs=300, num_centroids=70, dim=64, nbits=2, doc_len_min=0, doc_len_max=9, seed=5)
    a = synth.generate_arrays(spec)
    ix = oracle_index(a)
    nx = PN.NumpyIndex(a["centroids"], a["bucket_weights"], a["ivf"], a["ivf_lengths"], a["doc_lengths"],
                       a["codes"], a["residuals"], spec.nbits)
    assert (a["doc_lengths"] == 0).any()
    qs, _ = synth.make_queries(spec, 5, n_tokens=3, cen=a["centroids"])
    for thr in (None, 0.3):
        for cbs in (100_000, 16):
            p = O.SearchParameters(n_full_scores=64, top_k=400, n_ivf_probe=3, centroid_score_threshold=thr,
                                   centroid_batch_size=cbs)
            for q in qs:
                r = ix.search(q, p, trace=True)
                ids, sc, tr = nx.search(q, p, return_trace=True)
                assert np.array_equal(r.trace.cells, tr["cells"])
                assert np.array_equal(r.trace.cand, tr["cand"])
                assert_ranking_close(r.passage_ids, r.scores, ids, sc, RTOL_F32)
                assert len(r.passage_ids) == min(max(p.n_full_scores // 4, p.top_k), p.n_full_scores, len(r.trace.cand))


def test_structural_pins_from_reference_integration_tests():
    # filtering_integration.rs:69-117 (results subset of the filter), :320-349 (empty subset -> empty),
    # integration_tests.rs:706-707 (scores non-increasing)
    spec = synth.SynthSpec(num_docs=10, num_centroids=8, dim=64, nbits=4, doc_len_min=8, doc_len_max=8, seed=42)
    a = synth.generate_arrays(spec)
    ix = oracle_index(a)
    q = ix.get_document_embeddings(0)
    p = O.SearchParameters(top_k=3, n_ivf_probe=4)
    subset = [0, 2, 4, 6, 8]
    r = ix.search(q, p, subset)
    assert all(pid in subset for pid in r.passage_ids) and len(r.passage_ids) > 0
    assert len(ix.search(q, p, []).passage_ids) == 0
    r = ix.search(q, O.SearchParameters(top_k=10, n_ivf_probe=4, centroid_score_threshold=None))
    assert np.all(np.diff(r.scores) <= 0) and r.passage_ids[0] == 0


def test_search_batch_matches_search_and_sets_query_id():
    # search.rs:643-675
    spec = synth.SynthSpec(num_docs=400, num_centroids=64, dim=64, nbits=4, doc_len_min=5, doc_len_max=30, seed=9)
    a = synth.generate_arrays(spec)
    ix = oracle_index(a)
    qs, _ = synth.make_queries(spec, 6, n_tokens=8, cen=a["centroids"])
    p = O.SearchParameters(n_full_scores=128, top_k=5, n_ivf_probe=4)
    for par in (True, False):
        rs = ix.search_batch(qs, p, parallel=par)
        for i, (q, r) in enumerate(zip(qs, rs)):
            one = ix.search(q, p)
            assert r.query_id == i and one.query_id == 0
            assert np.array_equal(r.passage_ids, one.passage_ids) and np.array_equal(r.scores, one.scores)
