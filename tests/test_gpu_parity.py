"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Needs a real MI355X.

Bar (prompt section 3): bit-exact for integer / index work -- probed cells, candidate ids,
approximate scores (fp32, same k-ordered FMA chain and q-ordered sum as the oracle) and the selected
documents -- and fp tolerance for the exact MaxSim scores: RTOL_F32 = 2e-5 relative in fp32 mode
(summation order only), RTOL_BF16 = 1e-3 in bf16 mode (north_star's bound).
"""
import os
import threading

import numpy as np
import pytest

from helpers import (GOLDEN, O, RTOL_BF16, RTOL_BF16_PLAIN, RTOL_F32, assert_ranking_close, hip_index, make_arrays, oracle_index,
                     synth, to_oracle_params)

import next_plaid_amd as npa

pytestmark = pytest.mark.gpu

import importlib.util

_s = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
MG = importlib.util.module_from_spec(_s)
_s.loader.exec_module(MG)


def P(**kw):
    return npa.SearchParameters(**kw)


# Every exact-MaxSim arithmetic the library ships is compared with the oracle in the edge-case tests below, not only
# the strict f32 form: 2 is the default AND what bench.py times (exact_qct_kernel / exact_qc_kernel, split-bf16),
# 0 the exact-f32 MFMA kernel, 1 and 3 the bf16 forms at north_star's 1e-3 bound.
PRECISIONS = [2, 0, 1, 3]


def rtol_of(prec):
    return RTOL_F32 if prec in (0, 2) else (RTOL_BF16 if prec == 1 else RTOL_BF16_PLAIN)


@pytest.fixture(params=PRECISIONS, ids=[f"prec{p}" for p in PRECISIONS])
def prec(request):
    return request.param


def check_trace(hx, ox, q, p, subset=None, what="", exact_rtol=None, bitexact_probe=True):
    """Stage-by-stage comparison for one query; returns the oracle result."""
    exact_rtol = rtol_of(p.precision) if exact_rtol is None else exact_rtol
    tr = hx.debug_trace(q, p, subset)
    r = ox.search(q, to_oracle_params(p), subset, trace=True)
    t = r.trace
    if bitexact_probe:
        assert np.array_equal(tr["cells"], t.cells), f"{what}: S2 cells differ: hip {tr['cells'][:20]} oracle {t.cells[:20]}"
        assert np.array_equal(tr["cand"], t.cand), f"{what}: S3 candidates differ ({tr['cand'].size} vs {t.cand.size})"
        bad = np.nonzero(tr["approx"].view(np.uint32) != t.approx.view(np.uint32))[0]
        assert bad.size == 0, f"{what}: S4 approx not bit-exact at {bad[:5]}: {tr['approx'][bad[:5]]} vs {t.approx[bad[:5]]}"
        assert np.array_equal(tr["sel"], t.sel), f"{what}: S5 selection/order differs"
        tol = exact_rtol * np.maximum(np.abs(t.sel_exact), 1.0)
        assert np.all(np.abs(tr["sel_exact"] - t.sel_exact) <= tol), \
            f"{what}: S6 exact scores differ, max rel {np.max(np.abs(tr['sel_exact'] - t.sel_exact) / np.maximum(np.abs(t.sel_exact), 1))}"
    return r


@pytest.fixture(scope="module")
def mid():
    spec, a = make_arrays(num_docs=20000, num_centroids=4096, dim=128, nbits=4, doc_len_min=40, doc_len_max=120, seed=77)
    qs, src = synth.make_queries(spec, 64, n_tokens=32, cen=a["centroids"])
    return spec, a, oracle_index(a), hip_index(a), qs, src


def test_device_present():
    assert npa.device_count() >= 1


def test_default_precision_is_the_benched_one():
    assert npa.SearchParameters().precision == 2


def test_stages_mid(mid, prec):
    spec, a, ox, hx, qs, src = mid
    for thr in (0.4, None):
        p = P(n_full_scores=512, top_k=10, n_ivf_probe=8, centroid_score_threshold=thr, precision=prec)
        for qi in range(4):
            check_trace(hx, ox, qs[qi], p, what=f"thr={thr} q{qi}")


@pytest.fixture
def tuned(mid):
    """Restores the default kernel-selection knobs after a test changed them on the shared handle."""
    hx = mid[3]
    yield hx
    for k, v in (("s4_mode", 4), ("s4_minb", 8), ("s4_swz", 1), ("s4_filter", 1), ("s6_xcd", 1), ("s6_lds", 1), ("s6_tiles", 1), ("exact_rowmax", 0), ("ub_nt", 2), ("ub_steal", 16384), ("ub_nbx", 96), ("s4_hot", 60), ("ub_direct", 8), ("ub_static", 0), ("hot_static", 1), ("s4_planes", 1), ("s4_lpd", 2), ("s4_qm", 1), ("s4_warm", 0), ("s4_hot_auto", 200000), ("s3_bisect", 1), ("s3_gain", 1), ("s3_gain_mult", 3),
                 ("s3_gain_direct", 16)):
        hx.tune(k, v)


@pytest.mark.parametrize("filt", [0, 1])
@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_s4_kernel_variants_bit_exact(mid, tuned, mode, filt):
    """S4 has three exact kernels (all XCDs on one query / one XCD per query in 8,4,2,1 phases over the centroid range,
    lockstep or streamed), with or without the u8 upper-bound filter in front; np_hip_index_tune selects them.
    Every variant must reproduce the oracle's approximate scores bit for bit -- also for one-query calls (minb=1),
    ragged query lengths and the bpermute code broadcast -- and the same selection through the batched entry."""
    spec, a, ox, hx, qs, src = mid
    hx.tune("s4_mode", mode)
    hx.tune("s4_minb", 1)
    hx.tune("s4_filter", filt)
    p = P(n_full_scores=512, top_k=10, n_ivf_probe=8, centroid_score_threshold=0.4)
    for qi in (0, 5):
        check_trace(hx, ox, qs[qi], p, what=f"S4 mode {mode} q{qi}")
    check_trace(hx, ox, qs[7][:13], p, what=f"S4 mode {mode} short query")
    hx.tune("s4_swz", 0)
    check_trace(hx, ox, qs[9], p, what=f"S4 mode {mode} bpermute")
    # whole batch through the batched entry point (B >= 8 takes the per-XCD kernel when mode > 0)
    hx.tune("s4_minb", 8)
    res = hx.search_batch(qs[:16], p)
    ref = ox.search_batch(qs[:16], to_oracle_params(p))
    for i, (r, o) in enumerate(zip(res, ref)):
        assert_ranking_close(r.passage_ids, r.scores, o.passage_ids, o.scores, RTOL_F32, f"S4 mode {mode} batch q{i}")


def test_batch_mid_all_precisions(mid):
    # precision 0: exact-f32 MFMA; 2: QC-reuse + split-bf16 (f32-class); 1: QC-reuse + bf16; 3: plain bf16
    spec, a, ox, hx, qs, src = mid
    for prec, rtol in ((0, RTOL_F32), (2, RTOL_F32), (1, RTOL_BF16), (3, RTOL_BF16_PLAIN)):
        p = P(n_full_scores=1024, top_k=10, n_ivf_probe=16, precision=prec)
        res = hx.search_batch(qs, p)
        ref = ox.search_batch(qs, to_oracle_params(p))
        assert len(res) == 64
        for i, (r, o) in enumerate(zip(res, ref)):
            assert r.query_id == i
            assert_ranking_close(r.passage_ids, r.scores, o.passage_ids, o.scores, rtol, f"prec={prec} q{i}")
            assert r.passage_ids[0] == src[i]
        st = hx.last_stats
        assert st["n_queries"] == 64 and st["n_candidates"] > 0 and st["n_exact_docs"] > 0 and st["ms_total"] > 0


def test_search_equals_batch_of_one(mid):
    spec, a, ox, hx, qs, src = mid
    p = P(n_full_scores=256, top_k=7, n_ivf_probe=4)
    assert p.precision == 2
    r1 = hx.search(qs[3], p)
    rb = hx.search_batch(qs[:8], p)[3]
    assert r1.query_id == 0 and np.array_equal(r1.passage_ids, rb.passage_ids) and np.array_equal(r1.scores, rb.scores)


@pytest.mark.parametrize("case", [c[0] for c in MG.CASES])
def test_golden_search(case, prec):
    spec = synth.SynthSpec(**MG.GOLDEN_SPEC)
    a = synth.generate_arrays(spec)
    hx = hip_index(a)
    gold = np.load(os.path.join(GOLDEN, "search_2000.npz"))
    name, kw, sub = next(c for c in MG.CASES if c[0] == case)
    p = P(**kw, precision=prec)
    subset = None if sub is None else np.arange(0, spec.num_docs, 2, dtype=np.int64)
    for qi, q in enumerate(gold["queries"]):
        tr = hx.debug_trace(q, p, subset)
        assert np.array_equal(tr["cells"], gold[f"{name}_q{qi}_cells"]), f"{name} q{qi} cells"
        assert np.array_equal(tr["cand"], gold[f"{name}_q{qi}_cand"]), f"{name} q{qi} cand"
        r = hx.search(q, p, subset)
        assert_ranking_close(r.passage_ids, r.scores, gold[f"{name}_q{qi}_ids"], gold[f"{name}_q{qi}_scores"],
                             rtol_of(prec), f"{name} q{qi}")


@pytest.mark.parametrize("dim,nbits,K", [(64, 4, 100), (64, 2, 70), (96, 4, 257), (96, 2, 64), (32, 4, 33),
                                         (128, 2, 1000)])
def test_shapes_ragged_edges(dim, nbits, K, prec):
    # ragged docs incl. EMPTY ones, K not a multiple of 32/64, short and long queries (Lq 65/100/200 take the
    # row-max exact_qc_kernel at precision 1/2), top_k > candidates
    spec, a = make_arrays(num_docs=700, num_centroids=K, dim=dim, nbits=nbits, doc_len_min=0, doc_len_max=70, seed=dim + nbits)
    assert (a["doc_lengths"] == 0).any()
    ox, hx = oracle_index(a), hip_index(a)
    for ntok in (1, 5, 32, 48, 65, 100, 200):
        qs, _ = synth.make_queries(spec, 3, n_tokens=ntok, cen=a["centroids"])
        for thr, nprobe, nfs, topk in ((None, 3, 64, 5), (0.35, 8, 256, 300), (None, 64, 40, 10)):
            p = P(n_full_scores=nfs, top_k=topk, n_ivf_probe=nprobe, centroid_score_threshold=thr, precision=prec)
            for qi, q in enumerate(qs):
                check_trace(hx, ox, q, p, what=f"d{dim} b{nbits} K{K} Lq{ntok} thr{thr} np{nprobe} q{qi}")
            res = hx.search_batch(qs, p)
            ref = ox.search_batch(qs, to_oracle_params(p))
            for r, o in zip(res, ref):
                assert_ranking_close(r.passage_ids, r.scores, o.passage_ids, o.scores, rtol_of(prec))


def test_mixed_length_batch_and_slicing(prec):
    spec, a = make_arrays(num_docs=3000, num_centroids=512, dim=128, nbits=4, doc_len_min=10, doc_len_max=60, seed=8)
    ox = oracle_index(a)
    hx = hip_index(a, max_batch=5)          # forces 13 queries through 3 slices
    g = np.random.default_rng(1)
    qs = []
    for i in range(13):
        q, _ = synth.make_queries(spec, 1, n_tokens=int(g.integers(1, 70)), cen=a["centroids"], first_query=i)
        qs.append(q[0])
    p = P(n_full_scores=128, top_k=6, n_ivf_probe=6, precision=prec)
    res = hx.search_batch(qs, p)
    ref = ox.search_batch(qs, to_oracle_params(p))
    for i, (r, o) in enumerate(zip(res, ref)):
        assert_ranking_close(r.passage_ids, r.scores, o.passage_ids, o.scores, rtol_of(prec), f"q{i}")


def test_batched_path_selection_at_size(mid, tuned):
    """K > centroid_batch_size on the production path (u8 bound -> GEMM-valued scores -> margin cut -> mat-vec scores of
    what is left): with top_k = n_sel the whole selected set comes back, and it must be the oracle's (whose approximate
    scores are all mat-vec valued); the GEMM-valued dense path is allowed to select differently at near-ties."""
    spec, a, ox, hx, qs, src = mid
    for nfs, nprobe, thr in ((512, 32, None), (2048, 16, 0.4)):
        k = nfs // 4
        p = P(n_full_scores=nfs, top_k=k, n_ivf_probe=nprobe, centroid_score_threshold=thr, centroid_batch_size=1000)
        got = hx.search_batch(qs[:24], p)
        st = dict(hx.last_stats)
        ref = ox.search_batch(qs[:24], to_oracle_params(p))
        for i, (g, o) in enumerate(zip(got, ref)):
            assert set(g.passage_ids.tolist()) == set(o.passage_ids.tolist()), f"nfs={nfs} q{i}: selected set differs"
            assert_ranking_close(g.passage_ids, g.scores, o.passage_ids, o.scores, RTOL_F32, f"batched nfs={nfs} q{i}")
        assert 0 < st["n_survivors"] < st["n_candidates"]
        hx.tune("s4_filter", 0)     # unfiltered: every candidate gets the mat-vec score
        for g, o in zip(hx.search_batch(qs[:24], p), got):
            assert np.array_equal(g.passage_ids, o.passage_ids) and np.array_equal(g.scores, o.scores)
        hx.tune("s4_filter", 1)


def test_s6_kernel_variants_identical(mid, tuned):
    """S6 launch shapes (one XCD per query or not) and end-of-launch query sharing in the filter only move documents
    between waves: scores are bit-identical."""
    spec, a, ox, hx, qs, src = mid
    for prec in (2, 1):
        p = P(n_full_scores=1024, top_k=32, n_ivf_probe=16, precision=prec)
        ref = None
        for xcd, rep, steal in ((1, 0, 16384), (0, 0, 1), (0, 0, 0x7FFFFFFF), (1, 1, 16384), (0, 2, 16384), (1, 4, 16384)):
            hx.tune("s6_xcd", xcd)
            hx.tune("ub_steal", steal)
            # query fragments in registers (exact_qct_kernel) or in LDS (exact_qcl_kernel; 4: the 4-waves-per-SIMD instantiation)
            hx.tune("s6_lds", {0: 0, 1: 1, 2: 1, 4: 2}[rep])
            hx.tune("s6_tiles", 0 if rep == 2 else 1)    # multi-tile kernels or one launch per 32-token query tile
            got = hx.search_batch(qs[:24], p)
            if ref is None:
                ref = got
                continue
            for i, (g, r) in enumerate(zip(got, ref)):
                assert np.array_equal(g.passage_ids, r.passage_ids), f"prec={prec} xcd={xcd} rep={rep} q{i}"
                assert np.array_equal(g.scores, r.scores), f"prec={prec} xcd={xcd} rep={rep} q{i}"
    orc = ox.search_batch(qs[:24], to_oracle_params(p))
    for g, o in zip(ref, orc):
        assert_ranking_close(g.passage_ids, g.scores, o.passage_ids, o.scores, RTOL_BF16)


def test_rerank_window_up_to_16384_documents(mid):
    """search.rs:26-69 puts no bound on n_full_scores; the HIP path orders a query's re-rank window in LDS, 16384 documents
    since round 5 (n_full_scores = 65536; 8192 before).  20000 documents, no threshold, a wide probe: nearly the whole corpus
    is a candidate and 16384 of them are exact-scored; the top of the ranking equals the oracle's.  One document more is a
    Search error, not a silent truncation."""
    spec, a, ox, hx, qs, src = mid
    p = P(n_full_scores=65536, top_k=50, n_ivf_probe=64, centroid_score_threshold=None)
    got = hx.search_batch(qs[:4], p)
    assert hx.last_stats["n_exact_docs"] > 4 * 12000, hx.last_stats
    for i, (g, o) in enumerate(zip(got, ox.search_batch(qs[:4], to_oracle_params(p)))):
        assert_ranking_close(g.passage_ids, g.scores, o.passage_ids, o.scores, RTOL_F32, f"n_sel 16384 q{i}")
        assert g.passage_ids[0] == src[i]
    with pytest.raises(npa.SearchError):
        hx.search_batch(qs[:1], P(n_full_scores=65540, top_k=10, n_ivf_probe=8), parallel=False)
    assert all(r.passage_ids.size == 0 for r in hx.search_batch(qs[:2], P(n_full_scores=65540, top_k=10, n_ivf_probe=8)))   # search.rs:656-660


def test_huge_norm_query_keeps_reference_semantics(mid):
    """A query whose norm exceeds 1e12 is flagged like a non-finite one: the S4 filter is skipped and S6 keeps its
    non-finite guard (products could overflow).  Scores scale with the query, rankings equal the oracle's."""
    spec, a, ox, hx, qs, src = mid
    big = [qs[0] * np.float32(3e13), qs[1], qs[2] * np.float32(1e-20)]
    p = P(n_full_scores=512, top_k=10, n_ivf_probe=8)
    res = hx.search_batch(big, p)
    ref = ox.search_batch(big, to_oracle_params(p))
    for i, (r, o) in enumerate(zip(res, ref)):
        assert_ranking_close(r.passage_ids, r.scores, o.passage_ids, o.scores, RTOL_F32, f"scaled query {i}")


def test_s4_filter_preserves_selection(mid, tuned):
    """The u8 upper-bound filter in front of S4 must not change WHICH documents are selected nor their order:
    with top_k = n_sel the whole selected set is returned, so filtered and unfiltered runs must agree bit for bit
    (ids and exact scores), for plain, ragged-length and non-finite queries; and it must actually prune."""
    spec, a, ox, hx, qs, src = mid
    batch = list(qs[:24]) + [qs[30][:7], qs[31][:1]]
    bad = qs[32].copy()
    bad[3, 5] = np.nan
    batch.append(bad)
    for nfs, nprobe, thr in ((512, 32, None), (2048, 16, 0.4), (64, 64, None)):
        k = max(nfs // 4, 1)
        p = P(n_full_scores=nfs, top_k=k, n_ivf_probe=nprobe, centroid_score_threshold=thr)
        hx.tune("s4_filter", 0)
        ref = hx.search_batch(batch, p)
        st0 = dict(hx.last_stats)
        hx.tune("s4_filter", 1)
        surv = set()
        hx.tune("s4_hot", 0)           # this test is about the single-level filter (the two-level one has its own below)
        for ub_mode in (0, 1, 2):      # plain / non-temporal / bounds-checked buffer loads of the u8 table
            hx.tune("ub_nt", ub_mode)
            hx.tune("ub_static", ub_mode & 1)
            hx.tune("ub_nbx", (96, 8, 64)[ub_mode])    # workgroups per XCD: any number walks the same claims
            got = hx.search_batch(batch, p)
            st1 = dict(hx.last_stats)
            for i, (g, r) in enumerate(zip(got, ref)):
                assert np.array_equal(g.passage_ids, r.passage_ids), f"nfs={nfs} ub={ub_mode} q{i}: selected set / order changed"
                assert np.array_equal(g.scores, r.scores), f"nfs={nfs} ub={ub_mode} q{i}"
            assert st1["n_candidates"] == st0["n_candidates"] and st1["n_cand_tokens"] == st0["n_cand_tokens"]
            assert 0 < st1["n_survivors"] <= st1["n_candidates"]
            if nfs == 512:
                assert st1["n_survivors"] < st1["n_candidates"] // 2, st1
            surv.add(st1["n_survivors"])
        assert len(surv) == 1, surv       # the integer bounds U(d) do not depend on the load flavour
    orc = ox.search_batch(batch[:8], to_oracle_params(p))
    for g, o in zip(got[:8], orc):
        assert_ranking_close(g.passage_ids, g.scores, o.passage_ids, o.scores, RTOL_F32)


def test_s4_two_level_filter_preserves_selection(mid, tuned):
    """Round 3: a HOT bound (LDS bitmap of the query's hot centroids, table rows only for a document's hot codes) runs in
    front of the exact u8 bound.  Whatever share of the centroids is hot (s4_hot per-mille; 0 = the single-level filter),
    the selected documents, their order and their exact scores are those of the unfiltered path, bit for bit -- plain,
    ragged-length and non-finite queries -- and the first level must actually spare work: fewer table rows gathered than
    the single-level filter needs, and only part of the candidates reach the exact bound."""
    spec, a, ox, hx, qs, src = mid
    batch = list(qs[:24]) + [qs[30][:7], qs[31][:1]]
    bad = qs[32].copy()
    bad[3, 5] = np.nan
    batch.append(bad)
    for nfs, nprobe, thr in ((512, 32, None), (2048, 16, 0.4), (64, 64, None)):
        p = P(n_full_scores=nfs, top_k=max(nfs // 4, 1), n_ivf_probe=nprobe, centroid_score_threshold=thr)
        hx.tune("s4_filter", 0)
        ref = hx.search_batch(batch, p)
        hx.tune("s4_filter", 1)
        rows = {}
        # (hot share, first level in bit-plane form (round 4: approx_hotp_kernel) or as byte maxima (approx_hot_kernel))
        for hot, planes in ((0, 1), (10, 1), (100, 1), (300, 1), (500, 1), (10, 0), (100, 0), (300, 0)):
            hx.tune("s4_hot", hot)
            hx.tune("s4_planes", planes)
            hx.tune("s4_lpd", 2 if hot in (10, 100, 500) else 4) # plane kernel: claims of 32 documents (2 lanes each) or 16 (4 lanes)
            hx.tune("s4_qm", 0 if hot == 500 else 1)             # ... hot codes as a position mask or compacted in place
            hx.tune("s3_bisect", 0 if hot == 100 else 1)         # S3: bitmap ranges sweep the posting lists or bisect them
            hx.tune("s4_hot_auto", 50 if hot == 300 else 200000) # ... the share scaled down by the candidate count (300 -> >= 8 per mille)
            hx.tune("ub_direct", 0 if hot == 300 else 8)   # short-list launch: per-XCD hand-out or one group of workgroups per query
            hx.tune("ub_static", 1 if hot in (10, 500) else 0)   # claims from a cursor (with stealing) or round-robin
            hx.tune("hot_static", 0 if hot in (10, 300) else 1)
            # exact level of the S2 list: every row (1000), or only the rows of the warmest 50 / 30 / 11 % of the centroids with
            # the other tokens floored (upper bound for the cut, lower bound for its threshold)
            hx.tune("s4_warm", {0: 500, 10: 110, 100: 500, 300: 1000, 500: 300}[hot])
            got = hx.search_batch(batch, p)
            st = dict(hx.last_stats)
            for i, (g, r) in enumerate(zip(got, ref)):
                assert np.array_equal(g.passage_ids, r.passage_ids), f"nfs={nfs} hot={hot} q{i}: selected set / order changed"
                assert np.array_equal(g.scores, r.scores), f"nfs={nfs} hot={hot} q{i}"
            assert 0 < st["n_survivors"] <= st["n_candidates"]
            assert st["n_cand_dcodes"] > 0 and st["n_cand_tokens"] >= st["n_cand_dcodes"]
            if planes:
                rows[hot] = st["n_cand_codes"]
            if hot == 0:
                assert st["n_level2"] == 0 and st["n_cand_codes"] <= st["n_cand_dcodes"]
            elif nfs == 512:
                assert 0 < st["n_level2"] < st["n_candidates"], st
        if nfs == 512:
            assert rows[100] < rows[0], rows        # the hot level gathers fewer table rows than the exact bound alone
    hx.tune("s4_planes", 1)
    # the floored exact level really skips table rows (n_cand_codes counts the rows requested) and keeps the results
    # (the zeroth level pinned off: its run / skip rule may decide differently for the two calls whose row counts are compared)
    hx.tune("s3_gain", 0)
    hx.tune("s4_hot", 100)
    p2 = P(n_full_scores=512, top_k=128, n_ivf_probe=32, centroid_score_threshold=None)
    hx.tune("s4_warm", 1000)
    full = hx.search_batch(batch, p2)
    rows_full, lvl2 = hx.last_stats["n_cand_codes"], hx.last_stats["n_level2"]
    hx.tune("s4_warm", 300)
    part = hx.search_batch(batch, p2)
    assert hx.last_stats["n_cand_codes"] <= rows_full, (hx.last_stats["n_cand_codes"], rows_full)
    if lvl2 > len(batch) * 128:      # more documents at the exact level than the S1 lists hold: some S2 list is not empty
        assert hx.last_stats["n_cand_codes"] < rows_full, (hx.last_stats["n_cand_codes"], rows_full, lvl2)
    for g, r in zip(part, full):
        assert np.array_equal(g.passage_ids, r.passage_ids) and np.array_equal(g.scores, r.scores)
    hx.tune("s4_warm", 0)
    orc = ox.search_batch(batch[:8], to_oracle_params(p))
    for g, o in zip(got[:8], orc):
        assert_ranking_close(g.passage_ids, g.scores, o.passage_ids, o.scores, RTOL_F32)


def test_zeroth_level_preserves_selection():
    """Round 6: where no centroid_score_threshold is set, S3 sums the probed cells' gains per document (an upper
    bound of the approximate score from the posting lists alone, gain_sweep_kernel) and only the candidates whose bound reaches
    tau0 -- the n_sel-th best exact lower bound among the ~3 n_sel documents with the largest sums -- enter the filter.  Several
    32768-document ranges, ragged and non-finite queries, every size of the S0 list and both launch forms of its exact bound: the
    selected documents, their order and their exact scores are those of the unfiltered path bit for bit, the level really
    prunes -- with a threshold too, where the removed cells are swept as bound-only cells -- and n_candidates stays the size of the
    posting-list union."""
    spec, a = make_arrays(num_docs=150000, num_centroids=4096, dim=128, nbits=4, doc_len_min=30, doc_len_max=70, seed=79)
    hx = hip_index(a)
    ox = oracle_index(a)
    qs, src = synth.make_queries(spec, 40, n_tokens=32, cen=a["centroids"])
    batch = list(qs[:32]) + [qs[33][:7], qs[34][:1]]
    bad = qs[35].copy()
    bad[3, 5] = np.nan
    batch.append(bad)
    pruned_somewhere = False
    for nfs, nprobe, thr in ((256, 32, None), (1024, 8, None), (64, 64, None), (256, 32, 0.4), (512, 8, 0.35)):
        p = P(n_full_scores=nfs, top_k=max(nfs // 4, 1), n_ivf_probe=nprobe, centroid_score_threshold=thr)
        hx.tune("s4_filter", 0)
        ref = hx.search_batch(batch, p)
        n_union = hx.last_stats["n_candidates"]
        hx.tune("s4_filter", 1)
        for gain, mult, direct in ((0, 3, 16), (1, 3, 16), (1, 1, 0), (1, 8, 4)):
            hx.tune("s3_gain", 2 * gain)   # 2 = whenever it applies (1 = with the run / skip policy)
            hx.tune("s3_gain_mult", mult)
            hx.tune("s3_gain_direct", direct)
            got = hx.search_batch(batch, p)
            st = dict(hx.last_stats)
            for i, (g, r) in enumerate(zip(got, ref)):
                assert np.array_equal(g.passage_ids, r.passage_ids), f"nfs={nfs} nprobe={nprobe} gain={gain}/{mult}/{direct} q{i}: selection changed"
                assert np.array_equal(g.scores, r.scores), f"nfs={nfs} gain={gain} q{i}"
            assert st["n_candidates"] == n_union, (st["n_candidates"], n_union)
            if gain == 0:
                assert st["n_level0"] == 0, st
            else:     # with a threshold too: the cells it removes are swept as bound-only cells
                assert 0 < st["n_level0"] <= st["n_candidates"], st
                pruned_somewhere |= st["n_level0"] < st["n_candidates"] // 2
    assert pruned_somewhere
    hx.tune("s3_gain", 2)
    hx.tune("s3_gain_mult", 3)
    hx.tune("s3_gain_direct", 16)
    p = P(n_full_scores=256, top_k=64, n_ivf_probe=32, centroid_score_threshold=None)
    got = hx.search_batch(batch[:8], p)
    orc = ox.search_batch(batch[:8], to_oracle_params(p))
    for g, o in zip(got, orc):
        assert_ranking_close(g.passage_ids, g.scores, o.passage_ids, o.scores, RTOL_F32)
    one = hx.search(batch[0], p)                       # MmapIndex::search: a batch of one takes the level too
    assert hx.last_stats["n_level0"] > 0
    assert np.array_equal(one.passage_ids, got[0].passage_ids) and np.array_equal(one.scores, got[0].scores)
    hx.close()


def test_long_documents_overflow_blocks_and_windows():
    """Distinct-code lists live in fixed-stride blocks sized for 99 % of the documents; longer lists sit in an overflow
    region, and lists longer than the filter's 128-code staging window take several passes.  A corpus of 150-700-token
    documents makes EVERY list overflow and most of them span 2-4 windows: filtered (two-level and single-level) and
    unfiltered searches must still agree bit for bit, and with the oracle."""
    spec, a = make_arrays(num_docs=3000, num_centroids=2048, dim=128, nbits=4, doc_len_min=150, doc_len_max=700, n_topics=40,
                          rand256=160, seed=97)
    ox = oracle_index(a)
    qs, src = synth.make_queries(spec, 12, n_tokens=32, cen=a["centroids"])
    qs = list(qs) + [qs[0][:9]]
    p = P(n_full_scores=256, top_k=64, n_ivf_probe=4, centroid_score_threshold=None)
    # Round 4: the block size is chosen at OPEN from s4_planes -- with the bit-plane first level (default) blocks go up to
    # 512 bytes (a whole wave stages one: LPD = 4), without it 256 bytes.  Both layouts, and on the 256-byte one both
    # first-level kernels (the plane kernel then runs LPD = 2 with every list in the overflow region).
    for open_planes in ("1", "0"):
        os.environ["NP_S4_PLANES"] = open_planes
        try:
            hx = hip_index(a)
        finally:
            del os.environ["NP_S4_PLANES"]
        hx.tune("s4_filter", 0)
        ref = hx.search_batch(qs, p)
        hx.tune("s4_filter", 1)
        for hot, planes in ((100, 1), (0, 1), (400, 1)) + (((100, 0), (400, 0)) if open_planes == "0" else ()):
            hx.tune("s4_hot", hot)
            hx.tune("s4_planes", planes)
            hx.tune("s4_lpd", 2 if hot == 400 else 4)            # (512-byte blocks take 4 lanes per document whatever the knob)
            hx.tune("s4_warm", 1000 if hot == 400 else 400)      # floored exact level over several staging windows per list
            got = hx.search_batch(qs, p)
            st = dict(hx.last_stats)
            for i, (g, r) in enumerate(zip(got, ref)):
                assert np.array_equal(g.passage_ids, r.passage_ids) and np.array_equal(g.scores, r.scores), \
                    f"open={open_planes} hot={hot} planes={planes} q{i}"
            # the lists really are longer than one window (n_cand_dcodes counts the candidates the filter saw: with no threshold
            # set, those the zeroth level handed over)
            assert st["n_cand_dcodes"] / max(st["n_level0"] or st["n_candidates"], 1) > 128, st
            assert 0 < st["n_survivors"] < st["n_candidates"]
            if hot:
                assert 0 < st["n_level2"] < st["n_candidates"], (open_planes, hot, planes, st)   # the two-level filter really ran
        for g, o in zip(got, ox.search_batch(qs, to_oracle_params(p))):
            assert_ranking_close(g.passage_ids, g.scores, o.passage_ids, o.scores, RTOL_F32)
        check_trace(hx, ox, qs[1], P(n_full_scores=128, top_k=10, n_ivf_probe=4), what="long documents")


def test_candidate_pool_rounds(mid):
    """The candidate arrays are one pool sized by workspace_bytes, not B x n_docs: a batch whose candidates do not
    fit it together is processed in rounds (first-fit in query order).  A deliberately small budget with a wide probe
    forces several rounds; results must equal the oracle's and the roomy handle's bit for bit."""
    spec, a, ox, hx, qs, src = mid
    small = hip_index(a, workspace_bytes=11 << 20, max_batch=16)
    p = P(n_full_scores=512, top_k=10, n_ivf_probe=64, centroid_score_threshold=None)
    # (round 6: with no threshold the zeroth filter level prunes the candidates BEFORE the pool is planned -- that is half its
    # point: t_cs = None at nprobe 32 needed two rounds at 10 M documents, now one -- so the rounds are forced without it first)
    gained = small.search_batch(qs[:16], p)
    rounds_gained = small.last_stats["n_rounds"]
    small.tune("s3_gain", 0)
    res = small.search_batch(qs[:16], p)
    assert small.last_stats["n_rounds"] >= 2, small.last_stats
    assert rounds_gained <= small.last_stats["n_rounds"]
    for r, g in zip(res, gained):
        assert np.array_equal(r.passage_ids, g.passage_ids) and np.array_equal(r.scores, g.scores)
    ref = hx.search_batch(qs[:16], p)
    assert hx.last_stats["n_rounds"] == 1
    assert small.last_stats["n_candidates"] == hx.last_stats["n_candidates"]
    orc = ox.search_batch(qs[:16], to_oracle_params(p))
    for i, (r, f, o) in enumerate(zip(res, ref, orc)):
        assert np.array_equal(r.passage_ids, f.passage_ids) and np.array_equal(r.scores, f.scores), f"q{i}"
        assert_ranking_close(r.passage_ids, r.scores, o.passage_ids, o.scores, RTOL_F32, f"rounds q{i}")
    for mode in (0, 2, 6):          # every S4 kernel family walks the rounds
        small.tune("s4_mode", mode)
        for r, f in zip(small.search_batch(qs[:16], p), ref):
            assert np.array_equal(r.passage_ids, f.passage_ids) and np.array_equal(r.scores, f.scores), f"mode {mode}"


def test_nan_inf_query_tokens(prec):
    spec, a = make_arrays(num_docs=1500, num_centroids=256, dim=128, nbits=4, doc_len_min=10, doc_len_max=40, seed=21)
    ox, hx = oracle_index(a), hip_index(a)
    qs, _ = synth.make_queries(spec, 2, n_tokens=8, cen=a["centroids"])
    q = qs[0].copy()
    q[2, :] = np.nan            # a NaN token: contributes 0, never selects cells ahead of finite scores
    q[5, 7] = np.inf
    for thr in (None, 0.4):
        p = P(n_full_scores=128, top_k=5, n_ivf_probe=4, centroid_score_threshold=thr, precision=prec)
        r = hx.search(q, p)
        o = ox.search(q, to_oracle_params(p))
        assert_ranking_close(r.passage_ids, r.scores, o.passage_ids, o.scores, rtol_of(prec), f"nan thr={thr}")
        check_trace(hx, ox, q, p, what=f"nan thr={thr}")
        assert np.all(np.isfinite(r.scores))


def test_subset_filter_and_empty_subset(prec):
    # filtering_integration.rs:69-117, :320-349 through the HIP path
    spec, a = make_arrays(num_docs=2000, num_centroids=256, dim=128, nbits=4, doc_len_min=10, doc_len_max=40, seed=31)
    ox, hx = oracle_index(a), hip_index(a)
    qs, _ = synth.make_queries(spec, 4, n_tokens=16, cen=a["centroids"])
    p = P(n_full_scores=256, top_k=5, n_ivf_probe=4, centroid_score_threshold=None, precision=prec)
    for subset in (np.arange(0, 2000, 2), np.array([5, 17, 1999, 4000, -3]), np.arange(100)):
        subset = subset.astype(np.int64)
        for qi, q in enumerate(qs):
            check_trace(hx, ox, q, p, subset, what=f"subset n={subset.size} q{qi}")
            r = hx.search(q, p, subset)
            assert all(pid in set(subset.tolist()) for pid in r.passage_ids)
    r = hx.search(qs[0], p, np.zeros(0, np.int64))
    assert r.passage_ids.size == 0 and r.scores.size == 0


def test_decompress_documents_matches_codec():
    # N2 row: index.rs:1159-1245 / codec.rs:423-470
    spec, a = make_arrays(num_docs=300, num_centroids=128, dim=96, nbits=2, doc_len_min=0, doc_len_max=30, seed=13)
    ox, hx = oracle_index(a), hip_index(a)
    ids = [0, 7, 299, 12, 5000]
    embs, lens = hx.decompress_documents(ids)
    assert list(lens[:4]) == [int(a["doc_lengths"][i]) for i in ids[:4]] and lens[4] == 0
    ref = np.concatenate([ox.get_document_embeddings(i) for i in ids[:4]])
    assert embs.shape == ref.shape and np.allclose(embs, ref, rtol=0, atol=3e-7)


def test_on_disk_index_equals_arrays(tmp_path):
    spec, a = make_arrays(num_docs=900, num_centroids=128, dim=128, nbits=4, doc_len_min=5, doc_len_max=50, seed=17)
    synth.write_index(str(tmp_path), a, chunk_docs=250)
    hd, ha = npa.MmapIndex.load(str(tmp_path)), hip_index(a)
    assert hd.num_documents() == 900 and hd.num_partitions() == 128 and hd.embedding_dim() == 128
    assert hd.num_embeddings() == int(a["doc_lengths"].sum()) and abs(hd.avg_doclen() - a["doc_lengths"].mean()) < 1e-9
    e1, e2 = hd.export(), ha.export()
    for k in ("doc_lengths", "codes", "residuals", "ivf", "ivf_lengths"):
        assert np.array_equal(e1[k], e2[k]) and np.array_equal(e1[k], np.asarray(a[k])), k
    qs, _ = synth.make_queries(spec, 3, n_tokens=32, cen=a["centroids"])
    p = P(n_full_scores=128, top_k=10, n_ivf_probe=8)
    for r1, r2 in zip(hd.search_batch(qs, p), ha.search_batch(qs, p)):
        assert np.array_equal(r1.passage_ids, r2.passage_ids) and np.array_equal(r1.scores, r2.scores)


@pytest.mark.parametrize("bad", [-1, 128, -(1 << 62), 1 << 40])
def test_open_rejects_an_out_of_range_code(bad):
    """The device-side narrowing of the i64 codes keeps the loader's range check (mmap.rs / index.rs:1026-1139 reject nothing
    here, but a code >= K would index past the centroid table): ANY out-of-range value fails the open with IndexLoad --
    -1 included, which a value-as-sentinel check would let through."""
    spec, a = make_arrays(num_docs=300, num_centroids=128, dim=64, nbits=4, doc_len_min=5, doc_len_max=30, seed=31)
    b = dict(a)
    b["codes"] = np.array(a["codes"], np.int64, copy=True)
    b["codes"][b["codes"].size // 2] = bad
    with pytest.raises(npa.IndexLoadError) as e:
        hip_index(b)
    assert str(bad) in str(e.value) and "out of range" in str(e.value)
    hip_index(a).close()                                   # the untouched arrays still open


def test_reload_after_the_directory_changed(tmp_path):
    """MmapIndex::reload (index.rs:1767-1775): delete() rewrites the chunk files and re-sequences the ids
    (delete.rs:66-120); reload() must serve the new directory -- here: the same corpus without its first 100 documents."""
    spec, a = make_arrays(num_docs=500, num_centroids=128, dim=128, nbits=4, doc_len_min=5, doc_len_max=40, seed=23)
    synth.write_index(str(tmp_path), a, chunk_docs=200)
    hx = npa.MmapIndex.load(str(tmp_path), max_batch=8)
    qs, src = synth.make_queries(spec, 4, n_tokens=32, cen=a["centroids"])
    p = P(n_full_scores=128, top_k=5, n_ivf_probe=8)
    before = hx.search_batch(qs, p)
    lens = np.asarray(a["doc_lengths"], np.int64)
    t0 = int(lens[:100].sum())
    npa.write_index_dir(str(tmp_path), a["centroids"], a["bucket_weights"], lens[100:], a["codes"][t0:], a["residuals"][t0:],
                        4, chunk_docs=200)
    assert hx.num_documents() == 500                      # nothing changes until reload
    hx.reload()
    assert hx.num_documents() == 400 and hx.num_embeddings() == int(lens[100:].sum())
    ivf, il = synth.build_ivf(a["codes"][t0:], lens[100:], 128)
    ox = O.OracleIndex(a["centroids"], a["bucket_weights"], ivf, il, lens[100:], a["codes"][t0:], a["residuals"][t0:], 4)
    for i, (g, o) in enumerate(zip(hx.search_batch(qs, p), ox.search_batch(qs, to_oracle_params(p)))):
        assert_ranking_close(g.passage_ids, g.scores, o.passage_ids, o.scores, RTOL_F32, f"reloaded q{i}")
        if src[i] >= 100:
            assert g.passage_ids[0] == src[i] - 100 == before[i].passage_ids[0] - 100   # ids re-sequenced
    with pytest.raises(npa.IndexLoadError):
        hip_index(a).reload()                             # not opened from a directory
    hx.close()


def test_float16_index_files(tmp_path):
    """fast-plaid '<f2' centroids / bucket_weights (mmap.rs:1757-1778): same results as the widened '<f4' files."""
    spec, a = make_arrays(num_docs=600, num_centroids=96, dim=64, nbits=2, doc_len_min=5, doc_len_max=40, seed=19)
    a16 = dict(a)
    a16["centroids"] = a["centroids"].astype(np.float16).astype(np.float32)
    a16["bucket_weights"] = a["bucket_weights"].astype(np.float16).astype(np.float32)
    synth.write_index(str(tmp_path), a16, chunk_docs=250)
    np.save(os.path.join(str(tmp_path), "centroids.npy"), a["centroids"].astype("<f2"))
    np.save(os.path.join(str(tmp_path), "bucket_weights.npy"), a["bucket_weights"].astype("<f2"))
    hd, ha = npa.MmapIndex.load(str(tmp_path)), hip_index(a16)
    qs, _ = synth.make_queries(spec, 3, n_tokens=20, cen=a16["centroids"])
    p = P(n_full_scores=128, top_k=10, n_ivf_probe=8)
    for r1, r2 in zip(hd.search_batch(qs, p), ha.search_batch(qs, p)):
        assert np.array_equal(r1.passage_ids, r2.passage_ids) and np.array_equal(r1.scores, r2.scores)


def test_token_sorted_layout_is_transparent(monkeypatch):
    """NP_TOK_SORT=1 stores every document's tokens ordered by code (opt-in S6 gather locality): search results,
    decompress_documents and export are unchanged."""
    spec, a = make_arrays(num_docs=700, num_centroids=128, dim=64, nbits=4, doc_len_min=3, doc_len_max=60, seed=23)
    plain = hip_index(a)
    monkeypatch.setenv("NP_TOK_SORT", "1")
    srt = hip_index(a)
    monkeypatch.delenv("NP_TOK_SORT")
    assert srt.info.device_bytes > plain.info.device_bytes          # the position array
    qs, _ = synth.make_queries(spec, 4, n_tokens=24, cen=a["centroids"])
    p = P(n_full_scores=256, top_k=10, n_ivf_probe=8)
    for r1, r2 in zip(plain.search_batch(qs, p), srt.search_batch(qs, p)):
        assert np.array_equal(r1.passage_ids, r2.passage_ids)
        np.testing.assert_allclose(r1.scores, r2.scores, rtol=2e-6)     # max over tokens is order-free; sums are per query token
    ids = np.array([0, 5, 699, 17], np.int64)
    e1, l1 = plain.decompress_documents(ids)
    e2, l2 = srt.decompress_documents(ids)
    assert np.array_equal(l1, l2) and np.array_equal(e1, e2)
    x1, x2 = plain.export(), srt.export()
    for k in ("codes", "residuals", "ivf", "ivf_lengths", "doc_lengths"):
        assert np.array_equal(x1[k], x2[k]), k


def test_synth_device_generator_matches_numpy_spec():
    spec = synth.SynthSpec(num_docs=5000, num_centroids=1024, dim=128, nbits=4, doc_len_min=20, doc_len_max=90, seed=99)
    a = synth.generate_arrays(spec)
    hx = npa.MmapIndex.synth(spec, centroids=a["centroids"])
    e = hx.export()
    for k in ("doc_lengths", "codes", "residuals", "ivf", "ivf_lengths"):
        assert np.array_equal(e[k], np.asarray(a[k])), k
    # a shard of it
    hs = npa.MmapIndex.synth(spec, centroids=a["centroids"], shard_rank=1, shard_count=3)
    b0, b1 = hs.info.shard_doc_begin, hs.info.shard_doc_end
    assert (b0, b1) == (5000 * 1 // 3, 5000 * 2 // 3)
    es = hs.export()
    assert np.array_equal(es["doc_lengths"], a["doc_lengths"][b0:b1])
    off = np.concatenate([[0], np.cumsum(a["doc_lengths"])])
    assert np.array_equal(es["codes"], a["codes"][off[b0]:off[b1]])
    assert np.all((es["ivf"] >= b0) & (es["ivf"] < b1))


def test_errors_through_the_abi(mid):
    spec, a, ox, hx, qs, src = mid
    with pytest.raises(npa.ShapeError):
        hx.search(np.zeros((4, 64), np.float32), P())
    with pytest.raises(npa.SearchError):
        hx.search(qs[0], P(n_ivf_probe=0))
    spec2, a2 = make_arrays(num_docs=50, num_centroids=16, dim=192, nbits=4, doc_len_min=4, doc_len_max=4, seed=1)
    h192 = hip_index(a2)     # loads fine; dim > 128 has no HIP kernel -> Shape error, never a silent fallback
    with pytest.raises(npa.ShapeError):   # (every dim <= 128 is searchable: tests/test_gpu_geometry.py)
        h192.search(np.zeros((4, 192), np.float32), P())


def test_concurrent_calls_share_one_index(mid):
    spec, a, ox, hx, qs, src = mid
    p = P(n_full_scores=256, top_k=5, n_ivf_probe=8)
    ref = hx.search_batch(qs[:16], p)
    out, errs = {}, []

    def work(t):
        try:
            out[t] = hx.search_batch(qs[:16], p)
        except Exception as e:  # pragma: no cover
            errs.append(e)

    ts = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
    for t in range(6):
        for r, o in zip(out[t], ref):
            assert np.array_equal(r.passage_ids, o.passage_ids) and np.array_equal(r.scores, o.scores)


def test_batched_probe_semantics():
    """K > centroid_batch_size (search.rs:140-254, 521-640): per-slab heaps; the threshold sees only
    (token, centroid) pairs that entered a slab-local heap.  nprobe small + low threshold makes the
    heap-only max differ from the true column max, which exercises the slab-prefix count."""
    spec, a = make_arrays(num_docs=3000, num_centroids=700, dim=64, nbits=4, doc_len_min=5, doc_len_max=40, seed=41)
    ox, hx = oracle_index(a), hip_index(a)
    qs, _ = synth.make_queries(spec, 6, n_tokens=24, cen=a["centroids"], sigma_q=1.0)
    # every other token is "weak" (scaled down): its best centroids score below t_cs while a strong
    # token scores above t_cs on the same centroid without having it in its own top-nprobe
    qs = [q * np.where(np.arange(24) % 2 == 0, 0.35, 1.0)[:, None].astype(np.float32) for q in qs]
    n_diff = 0
    for thr in (0.2, 0.3):
        for nprobe in (1, 5):
            for cbs in (64, 333):
                pb = P(n_full_scores=128, top_k=8, n_ivf_probe=nprobe, centroid_score_threshold=thr, centroid_batch_size=cbs)
                pd_ = P(n_full_scores=128, top_k=8, n_ivf_probe=nprobe, centroid_score_threshold=thr)
                for qi, q in enumerate(qs):
                    tb = hx.debug_trace(q, pb)
                    ob = ox.search(q, to_oracle_params(pb), trace=True)
                    assert ob.trace.used_batched
                    assert np.array_equal(tb["cells"], ob.trace.cells), f"thr={thr} np={nprobe} cbs={cbs} q{qi}: cells"
                    assert np.array_equal(tb["cand"], ob.trace.cand)
                    # approx scores in the reference's batched arithmetic: mat-vec = unrolled_dot order (search.rs:259-272),
                    # NOT the probe GEMM's -- bit for bit, and therefore the same selection in the same order
                    bad = np.nonzero(tb["approx"].view(np.uint32) != ob.trace.approx.view(np.uint32))[0]
                    assert bad.size == 0, f"batched approx not bit-exact at {bad[:5]}: {tb['approx'][bad[:5]]} vs {ob.trace.approx[bad[:5]]}"
                    assert np.array_equal(tb["sel"], ob.trace.sel), f"thr={thr} np={nprobe} cbs={cbs} q{qi}: selection"
                    r = hx.search(q, pb)
                    assert_ranking_close(r.passage_ids, r.scores, ob.passage_ids, ob.scores, RTOL_F32, f"batched q{qi}")
                    n_diff += int(not np.array_equal(ob.trace.cells, ox.search(q, to_oracle_params(pd_), trace=True).trace.cells))
    assert n_diff > 0, "test data never separated batched from dense threshold semantics"
    # subset in batched mode only filters candidates (search.rs:542-545)
    sub = np.arange(0, 3000, 3, dtype=np.int64)
    pb = P(n_full_scores=64, top_k=5, n_ivf_probe=4, centroid_score_threshold=None, centroid_batch_size=100)
    for q in qs[:3]:
        tb = hx.debug_trace(q, pb, sub)
        ob = ox.search(q, to_oracle_params(pb), sub, trace=True)
        assert np.array_equal(tb["cells"], ob.trace.cells) and np.array_equal(tb["cand"], ob.trace.cand)


# ---- index-time encode, N4: /rerank MaxSim -------------------------------------------------------
@pytest.mark.parametrize("dim,nbits,K,n", [(128, 4, 1024, 3001), (64, 2, 300, 257)])
def test_encode_tokens_matches_oracle(dim, nbits, K, n):
    """codec.rs:297-411 / index.rs:289-371: nearest-centroid codes (last of equal maxima, non-finite below finite)
    and packed residual buckets, bit for bit; more tokens than one workspace slice, ragged tail."""
    spec, a = make_arrays(num_docs=400, num_centroids=K, dim=dim, nbits=nbits, doc_len_min=4, doc_len_max=12, seed=91)
    a = dict(a)
    cen = a["centroids"].copy()
    cen[7] = cen[K - 5]                      # identical centroids: the later index must win
    a["centroids"] = cen
    hx = hip_index(a)
    rng = np.random.default_rng(92)
    z = rng.integers(0, K, n)
    x = cen[z] + 0.2 * rng.standard_normal((n, dim)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    x[5] = 0.0                               # every score +0.0: all equal -> last centroid
    x[6, 3] = np.nan                         # every score non-finite -> last centroid
    x[n - 1] = cen[K - 5]                    # hits the duplicated pair exactly
    cut, _ = synth.bucket_tables(spec)
    codes, packed = hx.encode_tokens(x, cut)
    rc, rp = O.encode_tokens(x, cen, nbits, cut)
    bad = np.nonzero(codes != rc)[0]
    assert bad.size == 0, f"codes differ at {bad[:8]}: hip {codes[bad[:8]]} oracle {rc[bad[:8]]}"
    assert codes[5] == K - 1 and codes[6] == K - 1 and codes[n - 1] == K - 5
    assert np.array_equal(packed, rp)
    with pytest.raises(npa.ShapeError):
        hx.encode_tokens(x[:, : dim - 1], cut)
    c0, p0 = hx.encode_tokens(x[:0], cut)
    assert c0.shape == (0,) and p0.shape[0] == 0


def test_index_build_path_encode_write_load(tmp_path):
    """The index-build path end to end (index.rs:289-528): embeddings -> np_hip_encode_tokens (codes + packed residuals on
    the GPU) -> np_hip_index_write_dir (the crate's file set, posting lists built from the codes) -> MmapIndex::load of
    that directory.  The loaded index must BE the oracle's index of the same embeddings: files equal to the oracle's
    encode, search results equal to the oracle's search."""
    dim, nbits, K, N = 96, 4, 200, 600
    spec, a = make_arrays(num_docs=N, num_centroids=K, dim=dim, nbits=nbits, doc_len_min=3, doc_len_max=24, seed=97)
    cen, (cut, wts) = a["centroids"], synth.bucket_tables(spec)
    lens = np.asarray(a["doc_lengths"], np.int64)
    emb = synth.reconstruct(a["codes"], a["residuals"], cen, wts, nbits)       # the documents' token embeddings
    rng = np.random.default_rng(98)
    emb = emb + (0.05 / np.sqrt(dim)) * rng.standard_normal(emb.shape).astype(np.float32)
    emb = (emb / np.linalg.norm(emb, axis=1, keepdims=True)).astype(np.float32)
    enc = hip_index(a)                                       # any handle with these centroids encodes (codec.rs:297-411)
    codes, packed = enc.encode_tokens(emb, cut)
    rc, rp = O.encode_tokens(emb, cen, nbits, cut)
    assert np.array_equal(codes, rc) and np.array_equal(packed, rp)
    npa.write_index_dir(str(tmp_path), cen, wts, lens, codes, packed, nbits, bucket_cutoffs=cut, chunk_docs=250)
    hx = npa.MmapIndex.load(str(tmp_path))
    ivf, il = synth.build_ivf(rc, lens, K)
    ox = O.OracleIndex(cen, wts, ivf, il, lens, rc, rp, nbits)
    e = hx.export()
    assert np.array_equal(e["codes"], rc) and np.array_equal(e["residuals"], rp)
    assert np.array_equal(e["ivf"], ivf) and np.array_equal(e["ivf_lengths"], il)
    qs = [emb[int(o):int(o) + min(int(l), 16)] for o, l in zip(np.cumsum(lens)[:8] - lens[:8], lens[:8])]
    p = P(n_full_scores=128, top_k=10, n_ivf_probe=8)
    for i, (g, o) in enumerate(zip(hx.search_batch(qs, p), ox.search_batch(qs, to_oracle_params(p)))):
        assert_ranking_close(g.passage_ids, g.scores, o.passage_ids, o.scores, RTOL_F32, f"built index q{i}")
        assert g.passage_ids[0] == i                        # a document's own tokens find it
    for h in (enc, hx):
        h.close()


def test_rerank_maxsim_matches_handler():
    """next-plaid-api handlers/rerank.rs:57-170: scores bit-identical to the sequential multiply-then-add loop,
    stable descending order, the integration test's 2.0 / 1.0 / 0.0 answers, and the two BadRequest cases."""
    rng = np.random.default_rng(93)
    for lq, dim, lens in ((32, 128, [300, 1, 37, 0, 513, 64]), (5, 100, [3, 9]), (2, 4, [2, 1, 1])):
        q = rng.standard_normal((lq, dim)).astype(np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        docs = []
        for n in lens:
            d = rng.standard_normal((n, dim)).astype(np.float32)
            docs.append((d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9)).astype(np.float32))
        docs.append(docs[0].copy())          # equal scores: the stable sort keeps input order
        order, scores = npa.rerank_maxsim(q, docs)
        ref = np.array([O.rerank_maxsim(q, d) if d.shape[0] else 0.0 for d in docs], np.float32)
        assert np.array_equal(scores.view(np.uint32), ref.view(np.uint32)), f"{scores} vs {ref}"
        assert np.array_equal(order, np.argsort(-ref.astype(np.float64), kind="stable"))
    # integration_tests.rs:2301-2376
    q = np.array([[1, 0, 0, 0], [0, 1, 0, 0]], np.float32)
    d_both = np.array([[1, 0, 0, 0], [0, 1, 0, 0]], np.float32)
    d_one = np.array([[1, 0, 0, 0], [0, 0, 1, 0]], np.float32)
    d_none = np.array([[0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    order, scores = npa.rerank_maxsim(q, [d_none, d_both, d_one])
    assert np.allclose(scores, [0.0, 2.0, 1.0], atol=1e-6) and order.tolist() == [1, 2, 0]
    with pytest.raises(ValueError, match="non-finite"):
        npa.rerank_maxsim(np.array([[np.nan, 0, 0, 0]], np.float32), [d_both])
    with pytest.raises(ValueError, match="No documents"):
        npa.rerank_maxsim(q, [])
    with pytest.raises(npa.ShapeError):
        npa.rerank_maxsim(q, [np.zeros((2, 5), np.float32)])


def test_unsorted_posting_lists_take_the_full_sweep():
    """S3 bisects a posting list for a document range (and, round 6, the zeroth level reads its range's part through
    a table built at open) only when every list of ivf.npy ascends -- what the crate writes (index.rs:479-504), not what a
    third-party writer must.  An index whose lists are permuted (one reversed, one rotated) must be detected at open and served
    by the full sweep: results equal the sorted index's bit for bit, and the zeroth level -- which needs ascending lists --
    does not run."""
    spec, a = make_arrays(num_docs=60000, num_centroids=1024, dim=64, nbits=4, doc_len_min=10, doc_len_max=40, seed=83)
    qs, src = synth.make_queries(spec, 16, n_tokens=32, cen=a["centroids"])
    b = dict(a)
    ivf = a["ivf"].copy()
    off = np.concatenate([[0], np.cumsum(a["ivf_lengths"].astype(np.int64))])
    longest = np.argsort(-a["ivf_lengths"])[:40]
    for j, c in enumerate(longest):   # the longest lists: every query probes some of them
        s, e = int(off[c]), int(off[c + 1])
        ivf[s:e] = ivf[s:e][::-1] if j % 2 == 0 else np.roll(ivf[s:e], 7)
    b["ivf"] = ivf
    hs, hu = hip_index(a), hip_index(b)
    hs.tune("s3_gain", 2)
    hu.tune("s3_gain", 2)
    try:
        for thr, nprobe in ((None, 8), (0.4, 32), (None, 32)):
            p = P(n_full_scores=256, top_k=64, n_ivf_probe=nprobe, centroid_score_threshold=thr)
            rs = hs.search_batch(qs, p)
            st_s = dict(hs.last_stats)
            ru = hu.search_batch(qs, p)
            st_u = dict(hu.last_stats)
            for i, (x, y) in enumerate(zip(rs, ru)):
                assert np.array_equal(x.passage_ids, y.passage_ids) and np.array_equal(x.scores, y.scores), (thr, nprobe, i)
            assert st_s["n_candidates"] == st_u["n_candidates"]
            assert st_u["n_level0"] == 0                       # unsorted lists: no range table, no zeroth level
            if thr is None:
                assert st_s["n_level0"] > 0                    # ... which the sorted index does run
        tr_s, tr_u = hs.debug_trace(qs[0], p), hu.debug_trace(qs[0], p)
        assert np.array_equal(tr_s["cand"], tr_u["cand"]) and np.array_equal(tr_s["sel"], tr_u["sel"])
    finally:
        hs.close()
        hu.close()
