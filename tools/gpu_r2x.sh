#!/bin/bash
# Round 2, GPU call X: which exact-score kernel (NP_S4_MODE 0..8) suits the survivor lists (1.6 k - 4.3 k documents per query)?
mkdir -p gpurun_out/r2x
O=gpurun_out/r2x
run() {
  local name=$1; shift
  env $NPENV timeout 900 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S4', round(s['ms_approx'],3), 'S6', round(s['ms_exact'],3))" || tail -3 $O/b_$name.err
}
for m in 0 1 2 3 4 5 6 7 8; do
  NPENV="NP_S4_MODE=$m" run 1m_mode$m --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 0
done
for nbx in 64 96 160; do
  NPENV="NP_S4_NBX=$nbx" run 1m_nbx$nbx --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 0
done
for m in 0 3 4; do
  NPENV="NP_S4_MODE=$m" run 10m_mode$m --steps 8 --warmup 2 --cpu-queries 0 --parity-queries 0
done
