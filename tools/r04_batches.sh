#!/bin/bash
# GPU box: k_bin_flat with batches by cost (default) and with equal numbers of items (CRH_NO_BIN_BATCHES=1), in the run and alone
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), d.get("latency_ms_per_step") and round(d["latency_ms_per_step"],4), d["check"] and d["check"]["frame_equals_oracle"], {k:(round(v["avg_ms"],4), v["alone_ms"] and round(v["alone_ms"],4)) for k,v in d["kernels"].items() if k.startswith("raster_bin") or k.startswith("raster_tiles") or k.startswith("raster_rows")})'
for w in ${1:-cubic}; do
  echo "== $w batches by cost / by number"
  CRH_PASS_VERBOSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | grep -a "batches of\|^{" | sort -u | sed 's/^{.*/JSON&/' | while read -r line; do case "$line" in JSON*) echo "${line#JSON}" | python -c "$fmt";; *) echo "$line";; esac; done
  CRH_NO_BIN_BATCHES=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
done
