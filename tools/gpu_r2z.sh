#!/bin/bash
# Round 2, GPU call Y: probe_mark group maxima staged in LDS once; NP_S4_MODE default 4: tests + 1M / 1.25M / 10M lines + kernel stats at 1M.
mkdir -p gpurun_out/r2z
O=gpurun_out/r2z
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 > $O/test_gpu_all.log 2>&1
grep -E "passed|failed|error" $O/test_gpu_all.log | tail -3
run() {
  local name=$1; shift
  env $NPENV timeout 900 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S3', round(s['ms_candidates'],3), 'S4', round(s['ms_approx'],3), 'S5', round(s['ms_select'],3), 'S6', round(s['ms_exact'],3), d['parity_vs_oracle'])" || tail -3 $O/b_$name.err
}
NPENV="X=1" run 1m --docs 1000000 --steps 40 --warmup 4 --cpu-queries 0 --parity-queries 64
NPENV="X=1" run 10m --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 64
NPENV="X=1" run 10m_b1 --steps 40 --warmup 4 --cpu-queries 0 --parity-queries 0 --batch 1 --streams 1
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats1m -o s -- $CMD --docs 1000000 > /dev/null 2>&1
cd /root/repo
python3 tools/prof_summary.py $O/stats1m/s_kernel_stats.csv $O/kernel_stats_1m.md "1M" | grep -E "probe|qc_gemm|select|plan_rounds|topk|fill"
find $O -name "*kernel_trace.csv" -delete
