#!/bin/bash
# GPU box: raster kernel time (alone) and FETCH_SIZE per launch for tile orders: CRH_HEAVY_FIRST = 0 | sort | b<shift> | <factor>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in ${WORKLOADS:-cubic glyphs}; do for o in "$@"; do
  t=$(CRH_HEAVY_FIRST=$o python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); k=[v for n,v in d["kernels"].items() if n in ("raster_tiles","raster_rows")][0]; print(round(d["ms_per_step"],4), round(k["avg_ms"],4), round(k["alone_ms"],4))')
  rm -rf /tmp/fo; CRH_HEAVY_FIRST=$o rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d /tmp/fo -o p -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --workload $w > /dev/null 2>&1
  f=$(find /tmp/fo -name "*counter_collection.csv" | head -1)
  fetch=$(python - "$f" <<'PY'
import sys, csv, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "k_raster_edges" in k or "k_raster_rows" in k: acc[k.split("(")[0][-34:]].append(float(row["Counter_Value"]))
print({k: round(sum(v[-4:]) / len(v[-4:]) * 2 * 1024 / 1e6, 1) for k, v in acc.items()})
PY
)
  echo "$w order=$o step/in-run/alone ms: $t  fetch MB (x2 corrected, last launches): $fetch"
done; done
