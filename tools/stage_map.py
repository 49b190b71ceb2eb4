#!/usr/bin/env python3
"""Which pipeline stage every kernel of the search path belongs to -- ONE table for tools/make_traffic.py (PMC bytes per
stage), tools/prof_summary.py and the tests.

The stage of a kernel is the pair of HIP events of np_search.hip that brackets its launch (phase_a_once / phase_b:
ev[0] .. ev[7]; np_stats.ms_centroid .. ms_topk are the differences).  Round 4 renamed the dominant S4 kernel
(approx_hot_kernel -> approx_hotp_kernel) without touching the table that lived in make_traffic.py, and the bench line's
physical roofline silently lost half of the stage's bytes.  So the table is now CHECKED against the sources
(tests/test_bench_contract.py::test_stage_map_covers_every_launch_site): every kernel np_search.hip launches must be listed
here (a pipeline stage, or OTHER with the reason), and every name listed must still be a __global__ kernel of np_kernels.h.
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "next-plaid_amd", "csrc")

S1, S2, S3, S4, S5, S6, S7 = "qc_gemm(S1)", "probe(S2)", "candidates(S3)", "approx(S4)", "select(S5)", "exact(S6)", "topk(S7)"
STAGES = (S1, S2, S3, S4, S5, S6, S7)

STAGE = {
    # ev[0] .. ev[1]: queries -> Q.C^T, the u8 table, its per-centroid maxima
    "clear_regions_kernel": S1, "pad_rows_kernel": S1, "prep_queries_kernel": S1, "qc_gemm_kernel": S1, "qc_gemm_b3_kernel": S1,
    "hot_prep_kernel": S1, "hot_lam_kernel": S1,
    # ev[1] .. ev[2]: subset pre-filter, per-token top-nprobe, threshold, cell list
    "subset_kernel": S2, "subset_nprobe_kernel": S2, "masked_gmax_kernel": S2, "probe_mark_kernel": S2, "probe_finish_kernel": S2,
    # ev[2] .. ev[3]: posting-list union, round plan, compaction
    # (round 5: the hot level's thresholds / bitmaps / plane rows follow the round plan -- the hot share depends on the candidate count)
    "mark_slices_kernel": S3, "mark_candidates_kernel": S3, "count_chunks_kernel": S3, "plan_rounds_kernel": S3, "compact_kernel": S3,
    "hot_levels_kernel": S3, "hot_planes_kernel": S3,
    # (round 6: the zeroth filter level -- three sweeps of the probed posting lists with per-document gain sums, thresholds, counts.
    # Its S0 list takes approx_ub_kernel / ub_thr_kernel inside these events too; they stay listed under S4 by name: the level
    # runs only without a centroid_score_threshold, i.e. never on the default workload the PMC traffic file is keyed to)
    "gain_prep_kernel": S3, "gain_sweep_kernel": S3, "cells_to_bits_kernel": S2, "gain_emit_kernel": S3, "gain_thr_kernel": S3, "gain_count_kernel": S3,
    # ev[3] .. ev[4]: the two-level upper-bound filter and the exact approximate scores of the survivors
    "approx_hotp_kernel": S4, "approx_hot_kernel": S4, "approx_ub_kernel": S4, "ub_thr_kernel": S4, "ub_cut_kernel": S4,
    "approx_xcd_kernel": S4, "approx_kernel": S4, "approx_stream_kernel": S4, "gcut_kernel": S4, "approx_matvec_kernel": S4,
    "count_work_kernel": S4,
    # ev[4] .. ev[5]
    "select_kernel": S5,
    # ev[5] .. ev[6]
    "exact_qcl_kernel": S6, "exact_qct_kernel": S6, "exact_qc_kernel": S6, "exact_f32_kernel": S6, "exact_bf16_kernel": S6,
    # ev[6] .. ev[7]
    "topk_kernel": S7,
}

# launched from np_search.hip but not part of a single-GPU batch pass
OTHER = {
    "select_cut_kernel": "sharded protocol (global cut)", "merge_topk_kernel": "sharded protocol (merge)",
    "set_status_kernel": "sharded protocol (status word)", "or_reduce_kernel": "sharded protocol (subset bitmaps)",
    "decompress_kernel": "N2 decompress_documents", "encode_argmax_kernel": "N3 encode", "encode_pack_kernel": "N3 encode",
    "rerank_kernel": "N4 /rerank",
}


def launched_kernels(path=None):
    """Names of every kernel np_search.hip launches: `name<<<` or `name<template args><<<` (macros included)."""
    src = open(path or os.path.join(CSRC, "np_search.hip")).read()
    names = set()
    for m in re.finditer(r"\b([a-z][a-z0-9_]*_kernel)\b\s*(<|<<<)", src):
        names.add(m.group(1))
    return names


def defined_kernels(path=None):
    src = open(path or os.path.join(CSRC, "np_kernels.h")).read()
    return set(re.findall(r"\b([a-z][a-z0-9_]*_kernel)\s*\(", src))


def kernel_of(full_name):
    """'void np::approx_hotp_kernel<32, unsigned short, 2, 2, 4, 1>(...)' -> 'approx_hotp_kernel' (None: not ours)."""
    m = re.search(r"np::(\w+)", full_name)
    return m.group(1) if m else None


if __name__ == "__main__":
    lk, dk = launched_kernels(), defined_kernels()
    print("launched but unmapped:", sorted(lk - set(STAGE) - set(OTHER)))
    print("mapped but not defined:", sorted((set(STAGE) | set(OTHER)) - dk))
