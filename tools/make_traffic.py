#!/usr/bin/env python3
"""profiles/traffic.json from rocprofv3 PMC passes of the bench command.

Usage: make_traffic.py <pmc_dir> <docs_per_gpu> [out.json]
<pmc_dir>/p*/..counter_collection.csv hold one pass each (FETCH_SIZE and WRITE_SIZE in separate passes, as
MI355X_MICROARCH.md prescribes).  Per kernel: HBM-side bytes per batch = (f * FETCH_SIZE + WRITE_SIZE) KiB summed over every
dispatch, divided by the number of batches the run processed (= dispatches of prep_queries_kernel; a stage launches some
kernels once per candidate-pool round and the empty rounds return at once).  f = 2 on gfx950 (128-byte read requests tallied
at 64 bytes): the guide calibrates that for wide streaming reads, tools/probes/fetch_probe.hip (profiles/r05_fetch_probe.md)
for the random list-block and row patterns of S4 -- the factor is the same for every pattern that misses whole lines.
Per stage: the sum over ALL kernels tools/stage_map.py assigns to the stage (the table is checked against the launch sites
of np_search.hip by tests/test_bench_contract.py); a kernel of the search library that is neither mapped nor an index-build
kernel and moves more than 1 % of the batch's bytes FAILS the script instead of silently dropping out of the roofline."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import stage_map  # noqa: E402

FETCH_FACTOR = 2.0


def build_kernels():
    """__global__ kernels of np_index.hip: the one-off index build (synth, derived structures), not part of a batch."""
    src = open(os.path.join(stage_map.CSRC, "np_index.hip")).read()
    return set(re.findall(r"__global__[^;{]*?\b([a-z][a-z0-9_]*_kernel)\s*\(", src))


def main():
    d, docs = sys.argv[1], int(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(__file__), "..", "profiles", "traffic.json")
    tot = defaultdict(lambda: defaultdict(float))     # counter -> kernel -> sum
    calls = defaultdict(lambda: defaultdict(int))
    for f in sorted(glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = stage_map.kernel_of(r["Kernel_Name"])
            if not k:
                continue
            tot[r["Counter_Name"]][k] += float(r["Counter_Value"])
            calls[r["Counter_Name"]][k] += 1
    per_kernel = {}
    for ctr, mult in (("FETCH_SIZE", FETCH_FACTOR), ("WRITE_SIZE", 1.0)):
        nb = calls[ctr].get("prep_queries_kernel", 0)
        if nb == 0:
            print(f"no {ctr} pass found under {d}", file=sys.stderr)
            return 1
        for k, v in tot[ctr].items():
            e = per_kernel.setdefault(k, {"stage": stage_map.STAGE.get(k), "bytes_per_batch": 0})
            e[ctr + "_KiB_per_batch"] = round(v / nb, 1)
            e["bytes_per_batch"] += int(round(mult * v * 1024.0 / nb))
            e["dispatches_per_batch"] = round(calls[ctr][k] / nb, 2)
    build = build_kernels()
    batch_total = sum(e["bytes_per_batch"] for k, e in per_kernel.items() if e["stage"])
    unmapped = {k: e["bytes_per_batch"] for k, e in per_kernel.items()
                if not e["stage"] and k not in build and k not in stage_map.OTHER and e["bytes_per_batch"] > 0.01 * batch_total}
    if unmapped:
        print(f"kernels without a stage in tools/stage_map.py that move > 1 % of a batch's bytes: {unmapped}", file=sys.stderr)
        return 2
    j = {s: 0 for s in stage_map.STAGES}
    for k, e in per_kernel.items():
        if e["stage"]:
            j[e["stage"]] += e["bytes_per_batch"]
    j["docs_per_gpu"] = docs
    j["fetch_factor"] = FETCH_FACTOR
    # the counters are valid for ONE build of the kernels: bench.py drops them when the sources differ
    try:
        import subprocess
        root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
        sys.path.insert(0, root)
        import bench
        # (MAKE_TRAFFIC_SHA / MAKE_TRAFFIC_COMMIT: re-deriving the file from the kept passes of an EARLIER build)
        j["kernels_sha"] = os.environ.get("MAKE_TRAFFIC_SHA") or bench.kernels_sha()
        j["commit"] = os.environ.get("MAKE_TRAFFIC_COMMIT") or \
            subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:   # noqa: BLE001
        j["kernels_sha"], j["commit"] = None, None
    j["per_kernel"] = per_kernel
    j["_note"] = ("HBM-side bytes per batch = (2*FETCH_SIZE + WRITE_SIZE) KiB over all dispatches of the stage's kernels / batches, "
                  "separate rocprofv3 --pmc passes of the default bench command; a stage is the sum of per_kernel[*].bytes_per_batch over "
                  "the kernels tools/stage_map.py assigns to it; the x2 on FETCH_SIZE (gfx950: 128-B requests tallied at 64 B) is "
                  "calibrated for streaming reads by MI355X_MICROARCH.md and for random list blocks / rows by tools/probes/fetch_probe.hip")
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in j.items() if k not in ("per_kernel", "_note")}, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
