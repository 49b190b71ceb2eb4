#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (gpurun_out/pmc/p*/p_counter_collection.csv) per kernel:
mean counter value per dispatch for the np:: kernels.  Usage: pmc_summary.py <pmc_dir> [out.md]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
rows = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "p*", "*counter_collection.csv"))):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            if "np::" not in k:
                continue
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k in sorted(rows):
    lines.append(f"### {k}")
    for c, v in sorted(rows[k].items()):
        lines.append(f"- {c}: mean {sum(v)/len(v):,.1f} over {len(v)} dispatches")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
