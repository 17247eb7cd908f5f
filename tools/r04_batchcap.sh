#!/bin/bash
# GPU box: the runs of k_bin_flat closed at a predicted life as well (CRH_BIN_BATCH_TICKS), and in item order (CRH_BIN_BATCH_ORDER=1)
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), d.get("latency_ms_per_step") and round(d["latency_ms_per_step"],4), d["check"] and d["check"]["frame_equals_oracle"], {k:(round(v["avg_ms"],4), v["alone_ms"] and round(v["alone_ms"],4)) for k,v in d["kernels"].items() if k.startswith("raster_bin")})'
for w in ${1:-cubic}; do
  for cap in ${2:-0 320000 280000 240000 200000}; do
    echo "== $w cap $cap"
    CRH_BIN_BATCH_TICKS=$cap CRH_PASS_VERBOSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | grep -a "batches of\|^{" | sort -u | sed 's/^{.*/JSON&/' | while read -r line; do case "$line" in JSON*) echo "${line#JSON}" | python -c "$fmt";; *) echo "$line";; esac; done
  done
  echo "== $w in item order"
  CRH_BIN_BATCH_ORDER=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
done
