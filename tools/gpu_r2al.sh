#!/bin/bash
# Round 2, GPU call AL: one clear kernel instead of a dozen stream memsets per batch: tests + the two lines.
mkdir -p gpurun_out/r2al
O=gpurun_out/r2al
timeout 600 python -m pytest tests/ -x -q -m gpu --timeout 300 > $O/test_gpu_all.log 2>&1
grep -E "passed|failed|error" $O/test_gpu_all.log | tail -3
run() {
  local name=$1; shift
  timeout 300 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S3', round(s['ms_candidates'],3), 'S4', round(s['ms_approx'],3), 'S5', round(s['ms_select'],3), 'S6', round(s['ms_exact'],3), d['parity_vs_oracle'])" || tail -3 $O/b_$name.err
}
run 1m --docs 1000000 --steps 40 --warmup 4 --cpu-queries 0 --parity-queries 64
run 10m --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 64
