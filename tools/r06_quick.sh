#!/bin/bash
# GPU box: one line per workload and environment setting (ENVS="A=1 B=2|C=3": settings separated by |, "-" = none)
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["spread"]["ms_per_step_median"],4) if d.get("spread") else None, round(d.get("latency_ms_per_step") or 0,3), {k:(round(v["avg_ms"],4), round(v["alone_ms"],4) if v.get("alone_ms") else None) for k,v in d["kernels"].items()}, "recount", round((d.get("recount") or {}).get("ms_per_step") or 0,4), d["check"]["frame_equals_oracle"] if d.get("check") else None)'
IFS='|' read -ra SETS <<< "${ENVS:--}"
for w in ${WORKLOADS:-cubic}; do
  for set in "${SETS[@]}"; do
    echo "== $w [$set]"
    if [ "$set" = "-" ]; then set=""; fi
    env $set timeout 300 python bench.py --workload $w --no-cpu-baseline --no-side-workloads --repeats 3 $EXTRA 2>/dev/null | tail -1 | python -c "$fmt"
  done
done
