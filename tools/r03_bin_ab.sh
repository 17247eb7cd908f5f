#!/bin/bash
# GPU box: the binning kernels A/B — k_bin_flat (default) against k_bin_edges for every item (CRH_BIN_ITEMWISE=1), pipelined and stand-alone
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), {k:(round(v["avg_ms"],4), round(v["alone_ms"],4) if v["alone_ms"] else None) for k,v in d["kernels"].items() if k.startswith("raster")})'
for w in ${WORKLOADS:-cubic glyphs dashed s100k}; do
  for mode in flat itemwise; do
    if [ $mode = itemwise ]; then export CRH_BIN_ITEMWISE=1; else unset CRH_BIN_ITEMWISE; fi
    echo "== $w $mode pipelined (in-run, alone)"
    CRH_EDGE_PASS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
    echo "== $w $mode stand-alone"
    CRH_EDGE_PASS=1 CRH_NO_PIPELINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
  done
done
unset CRH_BIN_ITEMWISE
