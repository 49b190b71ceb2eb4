#!/bin/bash
# Derived rocprofv3 metrics (busy / stalled / latency) of a short bench run, one --pmc pass per group (no other tracing).
#   bash tools/pmc_derived.sh <out-dir> [bench args...]      -> <out-dir>/d<i>/..., <out-dir>/derived.txt (per kernel means)
REPO=$(pwd)
O=$REPO/$1; shift
mkdir -p $O
i=0
for set in "VALUBusy SALUBusy" "MemUnitBusy MemUnitStalled" "LdsUtil LDSBankConflict" "LdsLatency VmemLatency" "OccupancyPercent MeanOccupancyPerActiveCU"; do
  i=$((i+1))
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/d$i -o p -- \
      python $REPO/bench.py --steps 3 --warmup 1 --cpu-queries 0 --parity-queries 0 --streams 1 "$@" > /dev/null 2> $O/d$i.err )
done
python3 - $O <<'PY'
import csv, glob, sys, re, collections
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(o + "/d*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"np::(\w+)", r["Kernel_Name"])
        if m:
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(o + "/derived.txt", "w") as out:
    for k in sorted(acc):
        # dispatches of empty candidate-pool rounds return at once: keep the upper half by value count via max-based filter
        line = k + ": " + ", ".join(f"{c} mean {sum(v)/len(v):.2f} max {max(v):.2f} (n={len(v)})" for c, v in sorted(acc[k].items()))
        out.write(line + "\n")
print(open(o + "/derived.txt").read())
PY
find $O -name "*.csv" -size +5M -delete
