#!/bin/bash
# Round 2, GPU call A: full GPU tests at HEAD, smoke, config-2 and metric-config (10M docs) bench lines, short-row gather probe.
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || tail -20 $O/build.log
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 --durations=10 2>&1 | tail -40 > $O/test_gpu_all.log
tail -n 15 $O/test_gpu_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py --docs 1000000 --steps 20 --warmup 3 --cpu-queries 16 > $O/bench_1m.json 2> $O/bench_1m.err; echo "bench_1m rc=$?"
cut -c1-1800 $O/bench_1m.json
timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench_10m.json 2> $O/bench_10m.err; echo "bench_10m rc=$?"
cut -c1-2500 $O/bench_10m.json; tail -n 25 $O/bench_10m.err
timeout 120 tools/probes/gather_probe2 > $O/gather_probe2.txt 2>&1; tail -n 50 $O/gather_probe2.txt
