#!/bin/bash
# Round 2, GPU call FIN3: main lines, kernel stats and PMC passes at the final commit of the round (S1 epilogue through LDS) -- tests, smoke, default bench (10M docs, the metric config at N=1),
# config-2 line (1M), per-GPU shard sizes of the 2/4/8-way split, precision sweep + t_cs=None on 10M (config 5),
# a config-3-shaped run (8.84M ragged docs, 2^18 centroids, nbits=2, batched probe), kernel stats and PMC passes.
mkdir -p gpurun_out/r2fin3/pmc10m
O=gpurun_out/r2fin3
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 > $O/test_gpu_all.log 2>&1
grep -E "passed|failed|error" $O/test_gpu_all.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
run() {
  local name=$1; shift
  env $NPENV timeout 900 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S3', round(s['ms_candidates'],3), 'S4', round(s['ms_approx'],3), 'S5', round(s['ms_select'],3), 'S6', round(s['ms_exact'],3), d['parity_vs_oracle'], d['cpu_baseline'] and d['cpu_baseline']['value'], d['value_pcie_inclusive'])" || tail -3 $O/b_$name.err
}
NPENV="X=1" run default_10m
NPENV="X=1" run 1m --docs 1000000 --steps 40 --warmup 4
NPENV="X=1" run shard_5m --docs 5000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 0
NPENV="X=1" run shard_2500k --docs 2500000 --steps 30 --warmup 3 --cpu-queries 0 --parity-queries 0
NPENV="X=1" run shard_1250k --docs 1250000 --steps 40 --warmup 4 --cpu-queries 0 --parity-queries 0
NPENV="X=1" run shard_1250k_rccl --docs 1250000 --steps 40 --warmup 4 --cpu-queries 0 --parity-queries 64 --force-dist
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats10m -o s -- $CMD > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats1m -o s -- $CMD --docs 1000000 > /dev/null 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /root/repo/$O/pmc10m/p$i -o p -- $CMD > /dev/null 2> /root/repo/$O/pmc10m/p$i.err
  echo "$set" > /root/repo/$O/pmc10m/p$i.set
done
cd /root/repo
python3 tools/prof_summary.py $O/stats10m/s_kernel_stats.csv $O/kernel_stats_10m.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1 (default workload: 10M docs)" > /dev/null
python3 tools/prof_summary.py $O/stats1m/s_kernel_stats.csv $O/kernel_stats_1m.md "rocprofv3 --kernel-trace --stats -- python bench.py --docs 1000000 --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1" > /dev/null
python3 tools/make_traffic.py $O/pmc10m 10000000 $O/traffic.json > /dev/null
python3 tools/pmc_summary.py $O/pmc10m $O/pmc_10m.md > /dev/null
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +20M -delete
