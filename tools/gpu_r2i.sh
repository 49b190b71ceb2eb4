#!/bin/bash
# Round 2, GPU call I: what bounds the S4 filter kernel: walk probe (occupancy x VALU folding x pipelining), nt loads, TCC request mix.
mkdir -p gpurun_out/r2i
O=gpurun_out/r2i
timeout 100 tools/probes/gather_probe3 > $O/gather_probe3.txt 2>&1; cat $O/gather_probe3.txt
rocprofv3 -L > $O/counters.txt 2>&1; grep -ciE "TCC_" $O/counters.txt
run() {
  local name=$1; shift
  env $NPENV timeout 600 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S4', round(s['ms_approx'],3), 'S6', round(s['ms_exact'],3))"
}
NPENV="NP_UB_NT=0" run nt0 --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 0 --streams 1
NPENV="NP_UB_NT=1" run nt1 --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 0 --streams 1
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --docs 1000000 --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1"
i=0
for set in "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /root/repo/$O/p$i -o p -- $CMD > /dev/null 2> /root/repo/$O/p$i.err
  echo "$set" > /root/repo/$O/p$i.set; tail -n 2 /root/repo/$O/p$i.err
done
cd /root/repo
python3 tools/pmc_summary.py $O 2>/dev/null | grep -A26 "approx_ub_kernel" | head -60
find $O -name "*kernel_trace.csv" -delete
