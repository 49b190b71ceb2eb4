#!/bin/bash
# Round 2, GPU call N: reverted kernels (sanity) + what bounds S6 / the S4 filter: LDS conflicts, TA FIFO stalls, VMEM levels, MFMA busy.
mkdir -p gpurun_out/r2n
O=gpurun_out/r2n
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 2>&1 | tail -5 > $O/test_gpu_all.log
tail -n 3 $O/test_gpu_all.log
run() {
  local name=$1; shift
  env $NPENV timeout 600 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S3', round(s['ms_candidates'],3), 'S4', round(s['ms_approx'],3), 'S5', round(s['ms_select'],3), 'S6', round(s['ms_exact'],3), d['parity_vs_oracle'])"
}
NPENV="X=1" run base_1m --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 64
NPENV="X=1" run base_10m --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 0
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --docs 1000000 --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1"
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /root/repo/$O/p$i -o p -- $CMD > /dev/null 2> /root/repo/$O/p$i.err
  echo "$set" > /root/repo/$O/p$i.set; grep -iE "error|invalid" /root/repo/$O/p$i.err | head -2
done
cd /root/repo
python3 tools/pmc_summary.py $O 2>/dev/null | grep -A30 "approx_ub_kernel\|exact_qct_kernel\|qc_gemm" | grep -vE "^--" | head -110
find $O -name "*kernel_trace.csv" -delete
