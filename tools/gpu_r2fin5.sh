#!/bin/bash
# Round 2, GPU call FIN5: the two headline lines at the round's last code commit (clear kernel).
mkdir -p gpurun_out/r2fin5
O=gpurun_out/r2fin5
run() {
  local name=$1; shift
  timeout 600 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S3', round(s['ms_candidates'],3), 'S4', round(s['ms_approx'],3), 'S5', round(s['ms_select'],3), 'S6', round(s['ms_exact'],3), d['parity_vs_oracle'], d['cpu_baseline'] and d['cpu_baseline']['value'], d['roofline']['frac'], d['roofline']['traffic'])" || tail -3 $O/b_$name.err
}
run default_10m
run 1m --docs 1000000 --steps 40 --warmup 4
