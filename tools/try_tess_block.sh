#!/bin/bash
# GPU box: workgroup size of k_count / k_emit (256 default, 128, 64): single-wave workgroups find a free slot next to the raster kernel sooner
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["latency_ms_per_step"],4), {k:(round(v["avg_ms"],4), round(v["alone_ms"],4) if v["alone_ms"] else None) for k,v in d["kernels"].items() if k.startswith("tess")})'
for b in 256 128 64; do
  CRH_EXTRA_FLAGS=-DCRH_TESS_BLOCK=$b python contrast_renderer_amd/build.py --force > /dev/null 2>&1
  for w in ${WORKLOADS:-cubic glyphs}; do
    echo "== CRH_TESS_BLOCK=$b $w (ms/step pipelined, latency; tess kernels in-run / alone)"
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
  done
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
