#!/bin/bash
# Round 2, GPU call E: LDS-staged filter kernel: tests, 1M / 10M bench, L2 hit/miss + stall counters of the new kernel.
mkdir -p gpurun_out/r2f
O=gpurun_out/r2f
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 2>&1 | tail -30 > $O/test_gpu_all.log
tail -n 8 $O/test_gpu_all.log
run() {  # name, args..., env via NPENV
  local name=$1; shift
  env $NPENV timeout 600 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S3', round(s['ms_candidates'],3), 'S4', round(s['ms_approx'],3), 'S5', round(s['ms_select'],3), 'S6', round(s['ms_exact'],3), 'surv', s['n_survivors'], d['parity_vs_oracle'])"
}
NPENV="NP_S4_FILTER=1" run f1_1m --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 64
NPENV="NP_S4_FILTER=1" run f1_1m_s1 --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 0 --streams 1
NPENV="NP_S4_FILTER=1" run f1_10m --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 64
NPENV="NP_S4_FILTER=1 NP_S4_MODE=0" run f1_1m_mode0 --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 0 --streams 1
NPENV="NP_S4_FILTER=1" run f1_1m_dist --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 64 --force-dist
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --docs 1000000 --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -o s -- $CMD > /dev/null 2>&1
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /root/repo/$O/p$i -o p -- $CMD > /dev/null 2> /root/repo/$O/p$i.err
  echo "$set" > /root/repo/$O/p$i.set
done
cd /root/repo
python3 tools/pmc_summary.py $O 2>/dev/null | grep -A12 "approx_ub_kernel\|exact_qct\|qc_gemm" | head -60
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2f/stats/*kernel_stats.csv')
if f:
    for r in list(csv.DictReader(open(f[0])))[:22]:
        print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
find $O -name "*kernel_trace.csv" -delete
