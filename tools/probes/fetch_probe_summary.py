#!/usr/bin/env python3
"""Join tools/probes/fetch_probe's known byte counts with the rocprofv3 --pmc passes of the same binary.

Usage: fetch_probe_summary.py <probe_stdout.log> <pmc_dir> [out.md]
<pmc_dir>/p*/**/*counter_collection.csv: one --pmc pass each (FETCH_SIZE; TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; ...).
Every pattern is launched twice (a short TLB warm-up, then the measured run): the dispatch with the larger counter is the
measured one.  Output: per pattern the payload bytes, the bytes of the 128-byte lines touched, FETCH_SIZE (KiB -> bytes),
the read requests, and the factor FETCH_SIZE must be multiplied with to give the line bytes -- the calibration
MI355X_MICROARCH.md asks for ("calibrate on a known byte count in your own access pattern")."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    log, pmc = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else None
    probes = {}
    for line in open(log):
        if not line.startswith("PROBE "):
            continue
        t = line.split()
        d = {t[i]: float(t[i + 1]) for i in range(2, len(t) - 1, 2)}
        probes[t[1]] = d
    ctr = defaultdict(dict)      # pattern -> counter -> value of the measured dispatch
    for f in sorted(glob.glob(os.path.join(pmc, "p*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "stream_kernel" in k:
                name = "stream"
            else:
                m = re.search(r"blk_kernel<(\d+),\s*(\d+)", k)
                if not m:
                    continue
                name = f"blk{m.group(1)}_s{m.group(2)}"
            v = float(r["Counter_Value"])
            c = r["Counter_Name"]
            ctr[name][c] = max(ctr[name].get(c, 0.0), v)
    rows = ["| pattern | payload GB | 128-B-line GB | ms | payload GB/s | line GB/s | M blocks/s | FETCH_SIZE GB (raw) | line bytes / FETCH_SIZE | "
            "TCC_EA0_RDREQ (M) | of them 32 B (M) | line bytes / request |",
            "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for name, d in probes.items():
        c = ctr.get(name, {})
        fetch = c.get("FETCH_SIZE")
        fb = fetch * 1024.0 if fetch is not None else None
        rd = c.get("TCC_EA0_RDREQ_sum")
        rd32 = c.get("TCC_EA0_RDREQ_32B_sum")
        line_b = d["line_GB"] * 1e9
        rows.append("| `%s` | %.3f | %.3f | %.3f | %.0f | %.0f | %.0f | %s | %s | %s | %s | %s |" % (
            name, d["payload_GB"], d["line_GB"], d["ms"], d["payload_GBps"], d["line_GBps"], d["Mblocks_per_s"],
            "%.3f" % (fb / 1e9) if fb else "-", "%.2f" % (line_b / fb) if fb else "-",
            "%.2f" % (rd / 1e6) if rd else "-", "%.2f" % (rd32 / 1e6) if rd32 is not None else "-",
            "%.1f" % (line_b / rd) if rd else "-"))
    text = "\n".join(rows) + "\n"
    if out:
        with open(out, "w") as f:
            f.write("# fetch_probe: FETCH_SIZE calibration and random-block HBM rates (tools/probes/fetch_probe.hip)\n\n")
            f.write("Table 8 GiB (far beyond the 32 MiB of L2 and the 256 MiB Infinity Cache), blocks chosen by a hash: every read is a miss.\n"
                    "`128-B-line GB` = blocks x the 128-byte lines a block touches x 128 -- what the fabric has to deliver if the L2 fills\n"
                    "whole lines; `line bytes / FETCH_SIZE` is the factor that turns rocprofv3's FETCH_SIZE (KiB) into those bytes.\n\n")
            f.write(text)
    print(text)
    return 0


if __name__ == "__main__":
    sys.exit(main())
