#!/bin/bash
# GPU box: the pipelined step with the compute units divided between the lanes (CRH_CU_SPLIT=n: n CUs for tessellation + binning,
# the rest for the raster kernel) against the shared machine
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()})'
for w in ${WORKLOADS:-cubic}; do
for n in ${SPLITS:-0 32 48 64 80 96 128}; do
  echo "== $w CRH_CU_SPLIT=$n"
  CRH_CU_SPLIT=$n python bench.py --steps 40 --warmup 10 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
  if [ "$n" != 0 ] && [ -n "$BINCUS" ]; then
    echo "== $w CRH_CU_SPLIT=$n CRH_BIN_CUS=$n"
    CRH_BIN_CUS=$n CRH_CU_SPLIT=$n python bench.py --steps 40 --warmup 10 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
  fi
done; done
