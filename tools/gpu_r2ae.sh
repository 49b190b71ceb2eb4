#!/bin/bash
# Round 2, GPU call AE: cheap A/Bs -- S1 with two centroid fragments per wave, 2 / 4 streams.
mkdir -p gpurun_out/r2ae
O=gpurun_out/r2ae
run() {
  local name=$1; shift
  env $NPENV timeout 900 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S4', round(s['ms_approx'],3), 'S6', round(s['ms_exact'],3))" || tail -3 $O/b_$name.err
}
NPENV="NP_GEMM_CPW=2" run cpw2_1m --docs 1000000 --steps 40 --warmup 4 --cpu-queries 0 --parity-queries 0
NPENV="X=1" run s2_1m --docs 1000000 --steps 40 --warmup 4 --cpu-queries 0 --parity-queries 0 --streams 2
NPENV="X=1" run s4_1m --docs 1000000 --steps 40 --warmup 4 --cpu-queries 0 --parity-queries 0 --streams 4
NPENV="X=1" run s6_1m --docs 1000000 --steps 40 --warmup 4 --cpu-queries 0 --parity-queries 0 --streams 6
NPENV="X=1" run s2_10m --steps 12 --warmup 2 --cpu-queries 0 --parity-queries 0 --streams 2
NPENV="X=1" run s4_10m --steps 12 --warmup 2 --cpu-queries 0 --parity-queries 0 --streams 4
NPENV="X=1" run b128_1m --docs 1000000 --steps 20 --warmup 4 --cpu-queries 0 --parity-queries 0 --batch 128
NPENV="X=1" run b32_1m --docs 1000000 --steps 60 --warmup 4 --cpu-queries 0 --parity-queries 0 --batch 32
