#!/bin/bash
# One gpurun call: host facts, GPU parity tests (each module in its own process), smoke, short bench.
mkdir -p gpurun_out
{
  echo "== host"; nproc; free -g | head -2; lscpu | grep -E "Model name|Socket|Thread" ; df -h /tmp | tail -1
  echo "== gpu"; /opt/rocm/bin/rocm-smi --showmeminfo vram 2>/dev/null | head -8
} > gpurun_out/host.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
# exactly the driver's round-end command: ONE process for every GPU test (a per-file loop hid a two-HIP-runtimes
# failure that only shows when torch initialises after libnextplaid_hip.so in the same interpreter)
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 --durations=15 2>&1 | tail -60 > gpurun_out/test_gpu_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python bench.py --docs-per-gpu 100000 --steps 5 --warmup 2 --cpu-queries 16 > gpurun_out/bench_small.json 2> gpurun_out/bench_small.err
echo "bench_small rc=$?" >> gpurun_out/bench_small.err
tail -n 5 gpurun_out/*.log
