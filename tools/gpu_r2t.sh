#!/bin/bash
# Round 2, GPU call T: batched compaction gathers; evidence pass: tests, smoke, default bench (10M) + 1M line,
# rocprofv3 kernel stats and FETCH_SIZE / WRITE_SIZE / TCC passes of the default bench command.
mkdir -p gpurun_out/r2t
O=gpurun_out/r2t
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 > $O/test_gpu_all.log 2>&1
grep -E "passed|failed|error" $O/test_gpu_all.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
run() {
  local name=$1; shift
  env $NPENV timeout 900 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S3', round(s['ms_candidates'],3), 'S4', round(s['ms_approx'],3), 'S5', round(s['ms_select'],3), 'S6', round(s['ms_exact'],3), d['parity_vs_oracle'], d['cpu_baseline'])"
}
NPENV="X=1" run default_10m
NPENV="X=1" run 1m --docs 1000000 --steps 30 --warmup 3 --cpu-queries 0 --parity-queries 64
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats10m -o s -- $CMD > /dev/null 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /root/repo/$O/pmc10m/p$i -o p -- $CMD > /dev/null 2> /root/repo/$O/pmc10m/p$i.err
  echo "$set" > /root/repo/$O/pmc10m/p$i.set
done
cd /root/repo
python3 tools/prof_summary.py $O/stats10m/s_kernel_stats.csv $O/kernel_stats_10m.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1 (10M docs)" | head -40
python3 tools/make_traffic.py $O/pmc10m 10000000 $O/traffic.json
python3 tools/pmc_summary.py $O/pmc10m $O/pmc_10m.md > /dev/null
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +20M -delete
