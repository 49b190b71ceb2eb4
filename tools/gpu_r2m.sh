#!/bin/bash
# Round 2, GPU call M: S6 independent MFMA chains (2 vs 3 waves per SIMD), filter length-sorted claims, S3 record table.
mkdir -p gpurun_out/r2m
O=gpurun_out/r2m
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 2>&1 | tail -25 > $O/test_gpu_all.log
tail -n 6 $O/test_gpu_all.log
run() {
  local name=$1; shift
  env $NPENV timeout 600 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S3', round(s['ms_candidates'],3), 'S4', round(s['ms_approx'],3), 'S5', round(s['ms_select'],3), 'S6', round(s['ms_exact'],3), d['parity_vs_oracle'])"
}
NPENV="NP_S6_WAVES=2" run w2_1m --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 64
NPENV="NP_S6_WAVES=3" run w3_1m --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 0
NPENV="NP_S6_PIPE=0" run qct_1m --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 0
NPENV="NP_S6_WAVES=2" run w2_10m --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 64
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --docs 1000000 --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -o s -- $CMD > /dev/null 2>&1
cd /root/repo
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2m/stats/*kernel_stats.csv')
if f:
    for r in list(csv.DictReader(open(f[0])))[:26]:
        if 'rocprim' in r['Name'] or 'synth' in r['Name'] or 'inv_norm' in r['Name'] or 'unique' in r['Name'] or 'sort_doc' in r['Name']: continue
        print(r['Name'][:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
find $O -name "*kernel_trace.csv" -delete
