#!/bin/bash
# One parameterised driver for every GPU call of a round (replaces the one-off tools/gpu_r2*.sh scripts).
#
#   gpurun --timeout S -- 'bash tools/gpu.sh <tag> "<job>" "<job>" ...'
#
# Results land in gpurun_out/<tag>/.  A job is one string "<kind> <name> [args...]":
#   host                         host + GPU facts                                   -> host.txt
#   build                        __graft_entry__.build()                            -> build.log
#   test  <name> [pytest args]   python -m pytest tests -m gpu <args>               -> test_<name>.log
#   smoke                        __graft_entry__.smoke()                            -> smoke.log
#   bench <name> [bench args]    python bench.py <args>                             -> b_<name>.json (+ .err), one summary line
#   stats <name> [bench args]    rocprofv3 --kernel-trace --stats of a short bench  -> stats_<name>/ (+ kernel table .md)
#   pmc   <name> [bench args]    separate rocprofv3 --pmc passes of a short bench   -> pmc_<name>/  (+ traffic json)
#   probe <name> <binary> [args] tools/probes/<binary> alone (stdout -> probe_<name>.log), then under separate rocprofv3 --pmc passes
#                                (FETCH_SIZE; TCC_EA0_RDREQ_sum; TCC_EA0_RDREQ_32B_sum)            -> probe_<name>/ + probe_<name>.md
#   env   K=V                    export for the following jobs
#   sh    <name> <command...>    anything else                                      -> sh_<name>.log
# Per-job time limits: JOB_TIMEOUT (default 900 s).
TAG=$1; shift
REPO=$(pwd)
O=$REPO/gpurun_out/$TAG
mkdir -p "$O"
T=${JOB_TIMEOUT:-600}
summ() {
  python3 - "$1" "$2" <<'EOF'
import json, sys
name, path = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(path)); s = d["stages"]; r = d["roofline"]
    print(name, "q/s", d["value"], "p50", d["p50_batch_latency_ms"], "| S1", round(s["ms_centroid"], 3), "S2", round(s["ms_probe"], 3),
          "S3", round(s["ms_candidates"], 3), "S4", round(s["ms_approx"], 3), "S5", round(s["ms_select"], 3), "S6", round(s["ms_exact"], 3),
          "| parity", d.get("parity_vs_oracle"), "| cpu", (d.get("cpu_baseline") or {}).get("value"),
          "| roof", r["kernel"], r["frac"], "traffic", r["traffic"], "| B/tok", d.get("hbm_bytes_per_token"),
          "| surv", s.get("n_survivors"), "codes", s.get("n_cand_codes"), "cand", s.get("n_candidates"), "lvl0", s.get("n_level0"))
except Exception as e:
    print(name, "FAILED", type(e).__name__, e)
EOF
}
for job in "$@"; do
  set -- $job
  kind=$1; name=$2
  case $kind in
    env) export "$name";;
    host)
      { echo "== host"; nproc; free -g | head -2; lscpu | grep -E "Model name|Socket|Thread"; df -h /tmp | tail -1
        echo "== gpu"; /opt/rocm/bin/rocm-smi --showmeminfo vram 2>/dev/null | head -8; } > $O/host.txt 2>&1;;
    build) python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; echo "build rc=$?";;
    test) shift 2
      timeout $T python -m pytest tests/ -x -q -m gpu --timeout 600 --durations=25 "$@" > $O/test_$name.log 2>&1
      echo "test $name rc=$? : $(tail -n 1 $O/test_$name.log)";;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?";;
    bench) shift 2
      timeout $T python bench.py "$@" > $O/b_$name.json 2> $O/b_$name.err
      summ $name $O/b_$name.json | tee -a $O/summary.txt;;
    stats) shift 2
      ( cd /tmp && export TMPDIR=/tmp && timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -o s -- \
          python $REPO/bench.py --steps 6 --warmup 2 --cpu-queries 0 --parity-queries 0 "$@" > $O/stats_$name.json 2> $O/stats_$name.err )
      csv=$(find $O/stats_$name -name "*kernel_stats.csv" | head -1)
      python3 tools/prof_summary.py "$csv" $O/stats_$name.md "rocprofv3 --kernel-trace --stats: bench.py $*" | head -n 34
      cp "$csv" $O/stats_$name.csv 2>/dev/null;;
    pmc) shift 2
      mkdir -p $O/pmc_$name
      i=0
      sets=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS")
      [ -n "$PMC_SQ_ONLY" ] && sets=("${sets[@]:4}")      # PMC_SQ_ONLY=1: the two SQ passes only (what is a kernel waiting for)
      for set in "${sets[@]}"; do
        i=$((i+1))
        ( cd /tmp && export TMPDIR=/tmp && timeout $T rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$name/p$i -o p -- \
            python $REPO/bench.py --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 "$@" > /dev/null 2> $O/pmc_$name/p$i.err )
        echo "$set" > $O/pmc_$name/p$i.set
      done
      python3 tools/pmc_summary.py $O/pmc_$name $O/pmc_$name.md > /dev/null 2>&1
      python3 tools/make_traffic.py $O/pmc_$name ${PMC_DOCS:-10000000} $O/traffic_$name.json > $O/traffic_$name.log 2>&1
      find $O/pmc_$name -name "*.csv" -size +20M -delete; du -sh $O/pmc_$name;;
    probe) shift 2
      bin=$REPO/tools/probes/$1; shift
      timeout $T $bin "$@" > $O/probe_$name.log 2>&1
      mkdir -p $O/probe_$name
      i=0
      for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE"; do
        i=$((i+1))
        ( cd /tmp && export TMPDIR=/tmp && timeout $T rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/probe_$name/p$i -o p -- \
            $bin "$@" > /dev/null 2> $O/probe_$name/p$i.err )
        echo "$set" > $O/probe_$name/p$i.set
      done
      python3 tools/probes/fetch_probe_summary.py $O/probe_$name.log $O/probe_$name $O/probe_$name.md | head -n 12
      find $O/probe_$name -name "*.csv" -size +20M -delete;;
    sh) shift 2; timeout $T bash -c "$*" > $O/sh_$name.log 2>&1; echo "sh $name rc=$? : $(tail -n 2 $O/sh_$name.log)";;
    *) echo "unknown job kind: $kind";;
  esac
done
du -sh $O
