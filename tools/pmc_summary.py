#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (gpurun_out/pmc/p*/p_counter_collection.csv) per kernel of the search library.

Usage: pmc_summary.py <pmc_dir> [out.md]
Two figures per (kernel, counter): the SUM PER BATCH (sum over every dispatch of the pass / batches of the pass, batches =
dispatches of prep_queries_kernel) -- the figure tools/make_traffic.py, profiles/traffic.json and the bench line's
`roofline.traffic` are built from, so they can be checked by hand: bytes per batch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 --
and the mean per dispatch.  Kernels launched once per candidate-pool round have `max_rounds` dispatches per batch, the
empty ones returning at once: their mean per dispatch is diluted, the per-batch sum is not.  Kernels of the one-off index
build (no dispatch per batch) only get the mean."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
rows = defaultdict(lambda: defaultdict(list))
batches = defaultdict(int)          # counter -> dispatches of prep_queries_kernel in that counter's pass
for f in sorted(glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            if "np::" not in k:
                continue
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "prep_queries_kernel" in k:
                batches[r["Counter_Name"]] += 1
lines = ["# rocprofv3 --pmc passes of the bench command, per kernel", "",
         "`per batch` = sum over all dispatches / batches of the pass; HBM-side bytes per batch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024",
         "(FETCH_SIZE x 2 on gfx950: profiles/r05_fetch_probe.md).", ""]
for k in sorted(rows):
    lines.append(f"### {k}")
    for c, v in sorted(rows[k].items()):
        nb = batches.get(c, 0)
        per_batch = f"per batch {sum(v)/nb:,.1f}; " if nb and len(v) >= nb else ""
        lines.append(f"- {c}: {per_batch}mean {sum(v)/len(v):,.1f} over {len(v)} dispatches")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
