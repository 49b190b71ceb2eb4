#!/bin/bash
# Trimmed end-of-round evidence (fits ~6 GPU-minutes): full tests, smoke, default bench, rocprofv3 kernel stats,
# two PMC passes (FETCH_SIZE; L2 hit/miss).
mkdir -p gpurun_out/final
REPO=$(pwd)
for f in tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_cpp_host.py; do
  timeout 400 python -m pytest $f -q -m gpu --timeout 300 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5
done
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1
timeout 400 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 2 --cpu-queries 0 --streams 1"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/final/stats -o s -- $CMD > $REPO/gpurun_out/final/stats_bench.json 2> /dev/null
CMD="python $REPO/bench.py --steps 3 --warmup 1 --cpu-queries 0 --streams 1"
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/gpurun_out/final/p$i -o p -- $CMD > /dev/null 2> /dev/null
  echo "$set" > $REPO/gpurun_out/final/p$i.set
done
cd $REPO
python3 tools/pmc_summary.py gpurun_out/final 2>/dev/null | grep -A3 -E "approx" | head -12
find gpurun_out/final -name "*kernel_trace.csv" -size +20M -delete
cut -c1-1500 gpurun_out/final/bench_default.json
