#!/bin/bash
# Round 2, GPU call AC: where a filter wave's cycles go (build with -DNP_UB_TIMING: two workgroups print their wave-0 clocks).
mkdir -p gpurun_out/r2ac
timeout 300 python bench.py --docs 1000000 --steps 2 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1 > gpurun_out/r2ac/b1.json 2> gpurun_out/r2ac/b1.err
grep "ub block" gpurun_out/r2ac/b1.err | tail -6
timeout 300 python bench.py --steps 2 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1 > gpurun_out/r2ac/b10.json 2> gpurun_out/r2ac/b10.err
grep "ub block" gpurun_out/r2ac/b10.err | grep -v "claims 0" | tail -6
