#!/bin/bash
# Round 2, GPU call S: S3 bitmap ranges in LDS (mark_slices_kernel) vs atomicOr in memory; filter back to workgroup-synchronised queries.
mkdir -p gpurun_out/r2s
O=gpurun_out/r2s
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 > $O/test_gpu_all.log 2>&1
grep -E "passed|failed|error" $O/test_gpu_all.log | tail -3
run() {
  local name=$1; shift
  env $NPENV timeout 600 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S3', round(s['ms_candidates'],3), 'S4', round(s['ms_approx'],3), 'S5', round(s['ms_select'],3), 'S6', round(s['ms_exact'],3), d['parity_vs_oracle'])"
}
NPENV="NP_S3_SLICES=0" run s3old_1m --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 0
NPENV="NP_S3_SLICES=1" run s3new_1m --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 64
NPENV="NP_S3_SLICES=0" run s3old_10m --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 0
NPENV="NP_S3_SLICES=1" run s3new_10m --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 64
NPENV="NP_S3_SLICES=1" run s3new_10m_b1 --steps 40 --warmup 4 --cpu-queries 0 --parity-queries 0 --batch 1 --streams 1
NPENV="NP_S3_SLICES=0" run s3old_10m_b1 --steps 40 --warmup 4 --cpu-queries 0 --parity-queries 0 --batch 1 --streams 1
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats10m -o s -- $CMD > /dev/null 2>&1
cd /root/repo
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2s/stats10m/*kernel_stats.csv')
if f:
    for r in list(csv.DictReader(open(f[0])))[:34]:
        if int(r['Calls']) < 8: continue
        print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
find $O -name "*kernel_trace.csv" -delete
