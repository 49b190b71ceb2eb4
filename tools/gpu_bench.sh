#!/bin/bash
# Full-size bench (config 2) + rocprofv3 kernel stats of the same command (fewer steps).
mkdir -p gpurun_out
REPO=$(pwd)
timeout 1500 python bench.py --steps ${STEPS:-20} --warmup 3 ${BENCH_ARGS} > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
echo "rc=$?" >> gpurun_out/bench_full.err
if [ "${PROF:-1}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 1500 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof -o r1 -- python $REPO/bench.py --steps 5 --warmup 2 --cpu-queries 0 ${BENCH_ARGS} > $REPO/gpurun_out/prof_bench.json 2> $REPO/gpurun_out/prof_bench.err
  echo "rc=$?" >> $REPO/gpurun_out/prof_bench.err
  cd $REPO
  find gpurun_out/prof -name "*stats*" | head
  # keep only the small summaries
  find gpurun_out/prof -name "*kernel_trace*" -size +8M -delete
fi
tail -n 3 gpurun_out/bench_full.err; cat gpurun_out/bench_full.json
