#!/bin/bash
# Round 2, GPU call P: filter table read through a bounds-checked buffer (padding positions issue no request): A/B at 1M / 10M;
# token sort now opt-in; bench stdout = one JSON line also in dist mode; per-kernel stats at 10M.
mkdir -p gpurun_out/r2p
O=gpurun_out/r2p
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 2>&1 | tail -5 > $O/test_gpu_all.log
tail -n 3 $O/test_gpu_all.log
run() {
  local name=$1; shift
  env $NPENV timeout 600 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S3', round(s['ms_candidates'],3), 'S4', round(s['ms_approx'],3), 'S5', round(s['ms_select'],3), 'S6', round(s['ms_exact'],3), d['parity_vs_oracle'], d['hbm_bytes_per_token'])"
}
NPENV="NP_UB_NT=0" run nt0_1m --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 0
NPENV="NP_UB_NT=2" run nt2_1m --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 64
NPENV="NP_UB_NT=0" run nt0_10m --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 0
NPENV="NP_UB_NT=2" run nt2_10m --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 64
NPENV="NP_UB_NT=2" run nt2_1m_dist --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 64 --force-dist
wc -l $O/b_nt2_1m_dist.json
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1"
NP_UB_NT=2 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats10m -o s -- $CMD > /dev/null 2>&1
cd /root/repo
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r2p/stats10m/*kernel_stats.csv')
if f:
    for r in list(csv.DictReader(open(f[0])))[:26]:
        print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
find $O -name "*kernel_trace.csv" -delete
