#!/bin/bash
# GPU box: PMC counters for the bench (separate passes, --pmc only with kernel-trace as the guide prescribes)
# usage: [WORKLOAD=glyphs|dashed] tools/pmc.sh <tag> <counter set> [<counter set> ...]
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
shift
i=0
for set in "$@"; do
  i=$((i+1))
  timeout ${PMC_TIMEOUT:-420} rocprofv3 --kernel-trace --pmc $set -f csv -d $out/p$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-animated --no-side-workloads --repeats 0 --workload ${WORKLOAD:-cubic} > $out/p$i.log 2>&1
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import sys, csv, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0][-40:]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k, d in agg.items():
    print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
done
