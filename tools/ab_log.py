#!/usr/bin/env python3
"""Append one line per bench run of a GPU call (gpurun_out/<tag>/b_*.json) to profiles/<round>_ab_runs.jsonl.
Usage: ab_log.py <round, e.g. r03> <tag> [<tag> ...]"""
import glob
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
rnd, tags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "profiles", f"{rnd}_ab_runs.jsonl")
seen = set()
if os.path.exists(out):
    seen = {(json.loads(l)["call"], json.loads(l)["name"]) for l in open(out) if l.strip()}
with open(out, "a") as f:
    for tag in tags:
        for p in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "b_*.json")), key=os.path.getmtime):
            name = os.path.basename(p)[2:-5]
            if (tag, name) in seen:
                continue
            try:
                d = json.load(open(p))
            except Exception:
                continue
            s = d["stages"]
            f.write(json.dumps({
                "call": tag, "name": name, "workload": d["config"]["workload"], "queries_per_s": d["value"],
                "p50_ms": d["p50_batch_latency_ms"], "streams": d.get("streams"),
                "ms": {k[3:]: round(v, 3) for k, v in s.items() if k.startswith("ms_")},
                "counters": {k[2:]: int(v) for k, v in s.items() if k.startswith("n_") and k not in ("n_queries",)},
                "parity": d.get("parity_vs_oracle"), "cpu_qps": (d.get("cpu_baseline") or {}).get("value"),
                "hbm_bytes_per_token": d.get("hbm_bytes_per_token")}) + "\n")
print(out, sum(1 for _ in open(out)), "lines")
