#!/bin/bash
# PMC passes (each its own run, --kernel-trace only) + kernel stats CSV for the bench command.
mkdir -p gpurun_out/pmc
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $REPO/gpurun_out/pmc/counters_list.txt 2>&1
CMD="python $REPO/bench.py --steps 3 --warmup 1 --cpu-queries 0 ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/pmc/stats -o s -- $CMD > $REPO/gpurun_out/pmc/stats.json 2> $REPO/gpurun_out/pmc/stats.err
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/gpurun_out/pmc/p$i -o p -- $CMD > /dev/null 2> $REPO/gpurun_out/pmc/p$i.err
  echo "$set" > $REPO/gpurun_out/pmc/p$i.set
done
cd $REPO
find gpurun_out/pmc -name "*.csv" | head -40
du -sh gpurun_out/pmc
