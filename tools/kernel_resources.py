#!/usr/bin/env python3
"""Print VGPR/AGPR/scratch/occupancy/LDS per kernel of a .hip file (hipcc -Rpass-analysis)."""
import re
import subprocess
import sys

src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src,
                      "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
keys = {"VGPRs": "V", "AGPRs": "A", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occ",
        "LDS Size [bytes/block]": "lds", "TotalSGPRs": "S"}
rows, cur = [], None
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for k, s in keys.items():
        m = re.search(re.escape(k) + r": (\d+)", l)
        if m and cur is not None and s not in cur:
            cur[s] = m.group(1)
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(.*", "", n)[:52]
    print(f"{n:54s} V={r.get('V'):>4} A={r.get('A'):>3} S={r.get('S'):>4} scratch={r.get('scratch'):>5} occ={r.get('occ'):>2} lds={r.get('lds')}")
