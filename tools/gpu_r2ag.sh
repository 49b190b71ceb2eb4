#!/bin/bash
# Round 2, GPU call AG: probe_mark templated on tokens per block (8 when K > 65536): tests + the config-3-shaped line.
mkdir -p gpurun_out/r2ag
O=gpurun_out/r2ag
timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 600 > $O/test_gpu_all.log 2>&1
grep -E "passed|failed|error" $O/test_gpu_all.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --docs 8841823 --doc-len 120 --doc-len-min 20 --centroids 262144 --nbits 2 --steps 10 --warmup 2 --cpu-queries 8 --cpu-repeats 1 --parity-queries 16 > $O/b_c3_shape.json 2> $O/b_c3_shape.err
python3 -c "
import json; d=json.load(open('$O/b_c3_shape.json')); s=d['stages']; print('c3', d['value'], 'p50', d['p50_batch_latency_ms'], 'S1', round(s['ms_centroid'],3), 'S2', round(s['ms_probe'],3), 'S4', round(s['ms_approx'],3), d['parity_vs_oracle'])"
timeout 300 python bench.py --docs 1000000 --steps 30 --warmup 3 --cpu-queries 0 --parity-queries 0 > $O/b_1m.json 2> $O/b_1m.err
python3 -c "
import json; d=json.load(open('$O/b_1m.json')); s=d['stages']; print('1m', d['value'], 'S2', round(s['ms_probe'],3))"
