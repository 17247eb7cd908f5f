#!/bin/bash
# GPU box: amdgpu_waves_per_eu of k_bin_edges (register budget vs spills)
cd $GRAFT_REPO_ROOT
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k.startswith("raster")})'
for bw in ${BW:-4 5 6}; do
  CRH_EXTRA_FLAGS=-DCRH_BIN_WAVES=$bw python contrast_renderer_amd/build.py --force > /dev/null 2>&1
  for w in cubic glyphs dashed; do
    echo "== bin waves $bw $w pipelined / stand-alone"
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
    CRH_NO_PIPELINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
  done
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
