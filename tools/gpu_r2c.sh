#!/bin/bash
# Round 2, GPU call C: S4 filter: tests, config-2 and metric-config bench with the filter on / off, rocprof stats.
mkdir -p gpurun_out/r2c
O=gpurun_out/r2c
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 2>&1 | tail -40 > $O/test_gpu_all.log
tail -n 12 $O/test_gpu_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
for F in 1 0; do
  NP_S4_FILTER=$F timeout 600 python bench.py --docs 1000000 --steps 20 --warmup 3 --cpu-queries 0 --parity-queries 64 > $O/bench_1m_f$F.json 2> $O/bench_1m_f$F.err; echo "bench_1m filter=$F rc=$?"
  python3 -c "
import json; d=json.load(open('$O/bench_1m_f$F.json')); print(d['value'], d['p50_batch_latency_ms'], d['parity_vs_oracle']); print({k:round(v,3) for k,v in d['stages'].items()})"
done
NP_S4_FILTER=1 timeout 900 python bench.py --steps 10 --warmup 2 --cpu-queries 0 --parity-queries 64 > $O/bench_10m_f1.json 2> $O/bench_10m_f1.err; echo "bench_10m rc=$?"
python3 -c "
import json; d=json.load(open('$O/bench_10m_f1.json')); print(d['value'], d['p50_batch_latency_ms'], d['parity_vs_oracle']); print({k:round(v,3) for k,v in d['stages'].items()})"
tail -n 5 $O/*.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/stats -o s -- python /root/repo/bench.py --docs 1000000 --steps 5 --warmup 2 --cpu-queries 0 --parity-queries 0 --streams 1 > /dev/null 2>&1
cd /root/repo
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -r head -25
find $O/stats -name "*kernel_trace.csv" -delete
