#!/bin/bash
# GPU box: rebuilds the library with each set of extra compiler flags and prints the stand-alone kernel times. Usage: tools/try_flags.sh "<flags1>" "<flags2>" ...
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k.startswith("raster")})'
for flags in "$@"; do
  echo "== flags: $flags"
  CRH_EXTRA_FLAGS="$flags" python contrast_renderer_amd/build.py --force > /dev/null 2>&1 || echo BUILD FAILED
  for w in ${WORKLOADS:-cubic}; do CRH_NO_PIPELINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"; done
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
