#!/bin/bash
# One gpurun call: host facts, GPU parity tests (each module in its own process), smoke, short bench.
mkdir -p gpurun_out
{
  echo "== host"; nproc; free -g | head -2; lscpu | grep -E "Model name|Socket|Thread" ; df -h /tmp | tail -1
  echo "== gpu"; /opt/rocm/bin/rocm-smi --showmeminfo vram 2>/dev/null | head -8
} > gpurun_out/host.txt 2>&1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
for f in tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_cpp_host.py; do
  timeout 1500 python -m pytest $f -q -m gpu --timeout 600 -rA 2>&1 | tail -150 > gpurun_out/$(basename $f .py).log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python bench.py --docs-per-gpu 100000 --steps 5 --warmup 2 --cpu-queries 16 > gpurun_out/bench_small.json 2> gpurun_out/bench_small.err
echo "bench_small rc=$?" >> gpurun_out/bench_small.err
tail -n 5 gpurun_out/*.log
