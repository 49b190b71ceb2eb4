#!/bin/bash
# Round 2, GPU call U: L1 cache-policy probe for the row gathers; PMC passes (FETCH_SIZE / WRITE_SIZE / TCC) of the default bench command.
mkdir -p gpurun_out/r2u/pmc10m
O=gpurun_out/r2u
timeout 120 tools/probes/gather_probe5 > $O/gather_probe5.txt 2>&1; cat $O/gather_probe5.txt
run() {
  local name=$1; shift
  env $NPENV timeout 900 python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
  python3 -c "
import json; d=json.load(open('$O/b_$name.json')); s=d['stages']; print('$name', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S3', round(s['ms_candidates'],3), 'S4', round(s['ms_approx'],3), 'S5', round(s['ms_select'],3), 'S6', round(s['ms_exact'],3), d['parity_vs_oracle'])"
}
NPENV="X=1" run 10m --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 0
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /root/repo/$O/pmc10m/p$i -o p -- $CMD > /dev/null 2> /root/repo/$O/pmc10m/p$i.err
  echo "$set" > /root/repo/$O/pmc10m/p$i.set
done
cd /root/repo
python3 tools/make_traffic.py $O/pmc10m 10000000 $O/traffic.json
python3 tools/pmc_summary.py $O/pmc10m $O/pmc_10m.md | grep -A4 "approx_ub\|compact\|exact_qct\|mark_slices\|approx_xcd" | head -60
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +20M -delete
