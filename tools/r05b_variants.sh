# round 5: libraries built with other compile-time shapes (contrast_renderer_amd/build/variants/lib_<name>.so, built in the container), one bench line per workload each
mkdir -p gpurun_out/r05b
cp contrast_renderer_amd/libcontrast_hip.so /tmp/lib_shipped.so
for lib in contrast_renderer_amd/build/variants/lib_*.so; do
  name=$(basename $lib .so); name=${name#lib_}
  cp $lib contrast_renderer_amd/libcontrast_hip.so
  for w in ${WORKLOADS:-cubic glyphs s100k}; do
    timeout 300 python bench.py --workload $w --no-cpu-baseline --no-animated 2>/dev/null | tail -1 > gpurun_out/r05b/variant_${name}_$w.json
  done
done
cp /tmp/lib_shipped.so contrast_renderer_amd/libcontrast_hip.so
