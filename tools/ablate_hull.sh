#!/bin/bash
# GPU box: k_hull_small without its phase 2 (-DCRH_ABLATE_HULL=1: sorts only; the hulls are wrong) and k_hull_large without queued Shapes (=2)
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print({k:(round(v["avg_ms"],4)) for k,v in d["kernels"].items() if "hull" in k})'
for a in 0 1 2; do
  CRH_EXTRA_FLAGS=-DCRH_ABLATE_HULL=$a python contrast_renderer_amd/build.py --force > /dev/null 2>&1
  echo "CRH_ABLATE_HULL=$a"; CRH_NO_PIPELINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload glyphs 2>&1 | tail -1 | python -c "$fmt"
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
