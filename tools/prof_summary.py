#!/usr/bin/env python3
"""Turn rocprofv3 --kernel-trace --stats CSV output into a compact markdown table (names truncated).
Usage: prof_summary.py <s_kernel_stats.csv> <out.md> [title]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
title = sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 --kernel-trace --stats"
out = [f"# {title}", "", "| kernel | calls | avg us | min us | max us | % of GPU time |", "|---|---:|---:|---:|---:|---:|"]
for r in rows:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    if len(n) > 70:
        n = n[:67] + "..."
    out.append(f"| `{n}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
               f"{float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[:14]))
