#!/usr/bin/env python3
"""profiles/traffic.json from rocprofv3 PMC passes of the bench command.

Usage: make_traffic.py <pmc_dir> <docs_per_gpu> [out.json]
<pmc_dir>/p*/..counter_collection.csv hold one pass each (FETCH_SIZE and WRITE_SIZE in separate passes, as
MI355X_MICROARCH.md prescribes).  Per stage: HBM-side bytes per batch = (2 * FETCH_SIZE + WRITE_SIZE) KiB summed over
every dispatch of the stage's kernels, divided by the number of batches the run processed (= dispatches of
prep_queries_kernel; a stage launches some kernels once per candidate-pool round and the empty rounds return at once).
FETCH_SIZE is doubled per the guide's gfx950 note (128-B read requests tallied at 64 B); it is calibrated for wide
streaming reads only, so the figure is an upper estimate for the gather kernels."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

STAGE = {
    "qc_gemm_kernel": "qc_gemm(S1)", "prep_queries_kernel": "qc_gemm(S1)",
    "probe_mark_kernel": "probe(S2)", "probe_finish_kernel": "probe(S2)",
    "mark_slices_kernel": "candidates(S3)", "mark_candidates_kernel": "candidates(S3)", "count_chunks_kernel": "candidates(S3)",
    "plan_rounds_kernel": "candidates(S3)", "compact_kernel": "candidates(S3)",
    "approx_ub_kernel": "approx(S4)", "ub_thr_kernel": "approx(S4)", "ub_cut_kernel": "approx(S4)",
    "approx_hot_kernel": "approx(S4)", "hot_prep_kernel": "qc_gemm(S1)",
    "approx_xcd_kernel": "approx(S4)", "approx_kernel": "approx(S4)", "approx_stream_kernel": "approx(S4)",
    "approx_matvec_kernel": "approx(S4)", "gcut_kernel": "approx(S4)",
    "select_kernel": "select(S5)",
    "exact_qct_kernel": "exact(S6)", "exact_qcl_kernel": "exact(S6)", "exact_qc_kernel": "exact(S6)", "exact_f32_kernel": "exact(S6)", "exact_bf16_kernel": "exact(S6)",
    "topk_kernel": "topk(S7)",
}


def main():
    d, docs = sys.argv[1], int(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(__file__), "..", "profiles", "traffic.json")
    tot = defaultdict(lambda: defaultdict(float))     # counter -> kernel -> sum
    calls = defaultdict(lambda: defaultdict(int))
    for f in sorted(glob.glob(os.path.join(d, "p*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            m = re.search(r"np::(\w+)", r["Kernel_Name"])
            if not m:
                continue
            tot[r["Counter_Name"]][m.group(1)] += float(r["Counter_Value"])
            calls[r["Counter_Name"]][m.group(1)] += 1
    res, per_kernel = defaultdict(float), {}
    for ctr, mult in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        nb = calls[ctr].get("prep_queries_kernel", 0)
        if nb == 0:
            print(f"no {ctr} pass found under {d}", file=sys.stderr)
            return 1
        for k, v in tot[ctr].items():
            per_kernel.setdefault(k, {})[ctr + "_KiB_per_batch"] = round(v / nb, 1)
            if k in STAGE:
                res[STAGE[k]] += mult * v * 1024.0 / nb
    j = {k: int(v) for k, v in res.items()}
    j["docs_per_gpu"] = docs
    # the counters are valid for ONE build of the kernels: bench.py drops them when the sources differ
    try:
        import subprocess
        root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
        sys.path.insert(0, root)
        import bench
        j["kernels_sha"] = bench.kernels_sha()
        j["commit"] = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception as e:   # noqa: BLE001
        j["kernels_sha"], j["commit"] = None, None
    j["per_kernel"] = per_kernel
    j["_note"] = ("HBM-side bytes per batch = (2*FETCH_SIZE + WRITE_SIZE) KiB over all dispatches of the stage's kernels / batches, "
                  "separate rocprofv3 --pmc passes of the default bench command; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                  "(calibrated for streaming reads; an upper estimate for gathers)")
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in j.items() if k not in ("per_kernel", "_note")}, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
