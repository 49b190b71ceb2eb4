#!/bin/bash
# Round 2, GPU call R: filter waves progress independently (histogram in its own kernel); S6 16-copy LUT A/B.
mkdir -p gpurun_out/r2r
O=gpurun_out/r2r
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 > $O/test_gpu_all.log 2>&1
grep -E "passed|failed|error" $O/test_gpu_all.log | tail -3
run() {
  local name=$1; shift
  env $NPENV timeout 600 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S3', round(s['ms_candidates'],3), 'S4', round(s['ms_approx'],3), 'S5', round(s['ms_select'],3), 'S6', round(s['ms_exact'],3), d['parity_vs_oracle'])"
}
NPENV="NP_S6_REP=0" run rep0_1m --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 64
NPENV="NP_S6_REP=1" run rep1_1m --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 64
NPENV="NP_S6_REP=0" run rep0_10m --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 64
NPENV="NP_S6_REP=0 NP_UB_STEAL=2147483647" run nosteal_10m --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 0
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --docs 1000000 --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats1m -o s -- $CMD > /dev/null 2>&1
cd /root/repo
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2r/stats1m/*kernel_stats.csv')
if f:
    for r in list(csv.DictReader(open(f[0])))[:30]:
        if int(r['Calls']) < 4: continue
        print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
find $O -name "*kernel_trace.csv" -delete
