#!/bin/bash
# End-of-round evidence: full tests, smoke, default bench, bench variants, rocprofv3 kernel stats + PMC passes.
mkdir -p gpurun_out/final
REPO=$(pwd)
bash tools/gpu_ci.sh > /dev/null 2>&1
grep -E "passed|failed|FAILED|Error" gpurun_out/test_gpu_*.log | head
tail -n 1 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
for V in "--precision 0 --cpu-queries 0" "--precision 1 --cpu-queries 0" "--precision 2 --streams 1 --cpu-queries 0" "--threshold -1 --cpu-queries 0 --steps 5"; do
  timeout 900 python bench.py $V >> gpurun_out/final/bench_variants.jsonl 2>> gpurun_out/final/bench_variants.err
done
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 2 --cpu-queries 0 --streams 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/final/stats -o s -- $CMD > $REPO/gpurun_out/final/stats_bench.json 2> /dev/null
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/gpurun_out/final/p$i -o p -- $CMD > /dev/null 2> /dev/null
  echo "$set" > $REPO/gpurun_out/final/p$i.set
done
cd $REPO
find gpurun_out/final -name "*kernel_trace.csv" -delete
cat gpurun_out/final/bench_default.json
