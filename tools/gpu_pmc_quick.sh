#!/bin/bash
# two quick PMC passes: L2 hit/miss + SQ activity
mkdir -p gpurun_out/pmcq
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 3 --warmup 1 --cpu-queries 0 ${BENCH_ARGS}"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $REPO/gpurun_out/pmcq/p$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/gpurun_out/pmcq/p$i -o p -- $CMD > /dev/null 2> $REPO/gpurun_out/pmcq/p$i.err
done
cd $REPO
python3 tools/pmc_summary.py gpurun_out/pmcq | grep -A12 -E "${KERNELS:-approx}"
