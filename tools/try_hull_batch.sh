#!/bin/bash
# GPU box: k_hull_small with 4 (default), 8, 16, 32 Shapes per single-wave workgroup (phase 2 keeps 2 lanes per Shape busy)
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), {k:(round(v["avg_ms"],4), round(v["alone_ms"],4) if v["alone_ms"] else None) for k,v in d["kernels"].items() if "hull" in k})'
for b in ${BATCHES:-4 8 16 32}; do
  CRH_EXTRA_FLAGS=-DCRH_HULL_BATCH=$b python contrast_renderer_amd/build.py --force > /dev/null 2>&1
  for w in glyphs cubic; do
    echo "== CRH_HULL_BATCH=$b $w"
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
  done
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
