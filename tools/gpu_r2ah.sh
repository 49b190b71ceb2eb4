#!/bin/bash
# Round 2, GPU call AH: filter workgroups per XCD (the CU's vector-memory pipe is saturated at 96: fewer waves?).
mkdir -p gpurun_out/r2ah
O=gpurun_out/r2ah
run() {
  local name=$1; shift
  env $NPENV timeout 900 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S4', round(s['ms_approx'],3))" || tail -3 $O/b_$name.err
}
for nbx in 32 48 64 80 96; do
  NPENV="NP_UB_NBX=$nbx" run 1m_nbx$nbx --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 0
done
for nbx in 48 64 80; do
  NPENV="NP_UB_NBX=$nbx" run 10m_nbx$nbx --steps 8 --warmup 2 --cpu-queries 0 --parity-queries 0
done
