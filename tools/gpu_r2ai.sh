#!/bin/bash
# Round 2, GPU call AI: where a qc_gemm wave's cycles go (build with -DNP_GEMM_TIMING).
mkdir -p gpurun_out/r2ai
timeout 300 python bench.py --docs 1000000 --steps 2 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1 > gpurun_out/r2ai/b1.json 2> gpurun_out/r2ai/b1.err
grep "gemm block" gpurun_out/r2ai/b1.err | tail -6
