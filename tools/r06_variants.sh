#!/bin/bash
# GPU box: one bench line per library variant (contrast_renderer_amd/build/variants/lib_<name>.so, built in the container by tools/build_variant.sh) and workload;
# VARIANTS="a b" picks some, "shipped" is the tree's own library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
cp contrast_renderer_amd/libcontrast_hip.so /tmp/lib_shipped.so
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["spread"]["ms_per_step_median"],4) if d.get("spread") else None, round(d.get("latency_ms_per_step") or 0,3), {k:(round(v["avg_ms"],4), round(v["alone_ms"],4) if v.get("alone_ms") else None) for k,v in d["kernels"].items() if k.startswith("raster_bin") or k.startswith("raster_tiles") or k.startswith("raster_rows") or k.startswith("raster_setup")}, d["check"]["frame_equals_oracle"] if d.get("check") else None, "animated", round((d.get("animated") or {}).get("ms_per_step") or 0, 4), "recount", round((d.get("recount") or {}).get("ms_per_step") or 0, 4))'
for name in ${VARIANTS:-shipped $(ls contrast_renderer_amd/build/variants/ | sed 's/^lib_//; s/\.so$//')}; do
  if [ $name = shipped ]; then cp /tmp/lib_shipped.so contrast_renderer_amd/libcontrast_hip.so; else cp contrast_renderer_amd/build/variants/lib_$name.so contrast_renderer_amd/libcontrast_hip.so; fi
  for w in ${WORKLOADS:-cubic}; do
    echo "== $name $w"
    timeout 300 python bench.py --workload $w --no-cpu-baseline --no-side-workloads --repeats 3 $EXTRA 2>/dev/null | tail -1 | tee gpurun_out/r06/variant_${name}_$w.json | python -c "$fmt"
  done
done
cp /tmp/lib_shipped.so contrast_renderer_amd/libcontrast_hip.so
