#!/bin/bash
# Round 2, GPU call AA: cycle breakdown of probe_mark_kernel (build with -DNP_PROBE_TIMING: block (0,0) prints its phase clocks).
mkdir -p gpurun_out/r2aa
timeout 300 python bench.py --docs 1000000 --steps 2 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1 > gpurun_out/r2aa/b.json 2> gpurun_out/r2aa/b.err
grep "probe_mark cycles" gpurun_out/r2aa/b.err | tail -4
