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
# kernels of the index BUILD (np_hip_index_synth / build_device_index: they run once, before the first batch) and library
# kernels outside the np:: namespace (hipcub scans of the build, torch fills) are not per-batch work: their "us per batch"
# cell says "one-off" instead of total / batches
BUILD = ("synth_", "unique_codes_kernel", "ublock_layout_kernel", "ulen_", "useg_kernel", "doc_meta_kernel", "inv_norm_kernel",
         "sort_doc_tokens_kernel", "ivf_", "centroid_bound_kernel", "repack_rows_kernel", "unpack_rows_kernel", "cmax_kernel", "finite_", "narrow_", "rebase_", "lens_", "DeviceScan", "hipcub",
         "rocprim", "at::native", "elementwise", "fill")
for r in rows:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    one_off = "np::" not in r["Name"] or any(b in r["Name"] for b in BUILD) or int(r["Calls"]) < nb
    if len(n) > 70:
        n = n[:67] + "..."
    per = "one-off" if one_off else f"{float(r['TotalDurationNs'])/1e3/nb:.1f}"
    out.append(f"| `{n}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
               f"{float(r['MaxNs'])/1e3:.1f} | {per} | {float(r['Percentage']):.2f} |")
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))
