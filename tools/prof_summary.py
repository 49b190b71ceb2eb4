#!/usr/bin/env python3
"""Turn rocprofv3 --kernel-trace --stats CSV output into a compact markdown table (names truncated).
Usage: prof_summary.py <s_kernel_stats.csv> <out.md> [title]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
title = sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --kernel-trace --stats"
# batches processed = dispatches of prep_queries_kernel (once per batch); kernels launched once per candidate-pool round
# (compact, approx_ub, ub_thr, ub_cut, approx_xcd, select) have max_rounds dispatches per batch of which the empty rounds
# return at once, so "us per batch" (total / batches) is the figure that matches the bench's HIP-event stage times
nb = max([int(r["Calls"]) for r in rows if "prep_queries_kernel" in r["Name"]] + [1])
out = [f"# {title}", "", f"batches in this run: {nb}", "",
       "| kernel | calls | avg us | min us | max us | us per batch | % of GPU time |", "|---|---:|---:|---:|---:|---:|---:|"]
for r in rows:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    if len(n) > 70:
        n = n[:67] + "..."
    out.append(f"| `{n}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
               f"{float(r['MaxNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e3/nb:.1f} | {float(r['Percentage']):.2f} |")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))
