#!/bin/bash
# Round 2, GPU call O: batched-path mat-vec scores (a10): tests; access-order probe.
mkdir -p gpurun_out/r2o
O=gpurun_out/r2o
timeout 60 tools/probes/gather_probe4 > $O/gather_probe4.txt 2>&1; cat $O/gather_probe4.txt
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 2>&1 | tail -30 > $O/test_gpu_all.log
tail -n 25 $O/test_gpu_all.log | cut -c1-300
