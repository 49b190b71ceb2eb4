#!/bin/bash
# Round 2, GPU call B: full GPU tests, metric-config (10M docs) bench line.
mkdir -p gpurun_out/r2b
O=gpurun_out/r2b
timeout 1500 python -m pytest tests/ -q -m gpu --timeout 600 2>&1 | tail -15 > $O/test_gpu_all.log
tail -n 8 $O/test_gpu_all.log
( while true; do rocm-smi --showmemuse --showuse 2>/dev/null | grep -E "GPU\[0\]" | tr '\n' ' '; free -g | awk '/Mem:/{print " host_used_GB="$3}'; sleep 10; done ) > $O/mon.log 2>&1 &
MON=$!
SECONDS=0
timeout 1200 python bench.py --steps 10 --warmup 2 > $O/bench_10m.json 2> $O/bench_10m.err; echo "bench_10m rc=$? wall=${SECONDS}s"
kill $MON
cut -c1-3000 $O/bench_10m.json; tail -n 25 $O/bench_10m.err; tail -n 12 $O/mon.log
