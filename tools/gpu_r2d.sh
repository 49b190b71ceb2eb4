#!/bin/bash
# Round 2, GPU call D: tests after the survivor-gap fix + C sharded entry; S4 filter diagnosis (PMC, nt / nbx variants).
mkdir -p gpurun_out/r2d
O=gpurun_out/r2d
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 2>&1 | tail -40 > $O/test_gpu_all.log
tail -n 12 $O/test_gpu_all.log
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 0 --streams 1 > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'S4', round(s['ms_approx'],3), 'S5', round(s['ms_select'],3), 'S1', round(s['ms_centroid'],3), 'surv', s['n_survivors'])"
}
run base NP_S4_FILTER=1
run nt NP_S4_FILTER=1 NP_UB_NT=1
run nbx64 NP_S4_FILTER=1 NP_S4_NBX=64
run nbx96 NP_S4_FILTER=1 NP_S4_NBX=96
run nbx160 NP_S4_FILTER=1 NP_S4_NBX=160
run mode0 NP_S4_FILTER=1 NP_S4_MODE=0
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --docs 1000000 --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAVES" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /root/repo/$O/p$i -o p -- $CMD > /dev/null 2> /root/repo/$O/p$i.err
  echo "$set" > /root/repo/$O/p$i.set
done
cd /root/repo
python3 tools/pmc_summary.py $O 2>/dev/null | head -80
