#!/bin/bash
# bench at several precisions, compact output
for P in ${PRECS:-0 1 2}; do
  timeout 900 python bench.py --steps ${STEPS:-20} --warmup 3 --precision $P ${BENCH_ARGS} > gpurun_out/bench_p$P.json 2> gpurun_out/bench_p$P.err
  python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_p$P.json"))
    st=d["stages"]
    print("prec $P qps", d["value"], "ms/step", d["ms_per_step"], {k: round(v,3) for k,v in st.items() if k.startswith("ms_")}, d["parity_vs_oracle"], d["roofline"]["kernel"], d["roofline"]["frac"])
except Exception as e:
    print("prec $P failed", e); print(open("gpurun_out/bench_p$P.err").read()[-2000:])
PY
done
