# round 5 (experiment): a resident raster grid (CRH_RASTER_PERSISTENT = workgroups) with the next frame's binning / tessellation beside it;
# lib_bin128v.so = k_bin_flat held to 128 registers (-DCRH_FLAT_VGPRS=128 -DCRH_FLAT_WAVES=4 -DCRH_FLAT_POOL=4) so that its waves fit beside four raster waves per SIMD
mkdir -p gpurun_out/r05b
cp contrast_renderer_amd/libcontrast_hip.so /tmp/lib_shipped.so
for lib in shipped bin128v; do
  if [ $lib = shipped ]; then cp /tmp/lib_shipped.so contrast_renderer_amd/libcontrast_hip.so; else cp contrast_renderer_amd/build/variants/lib_$lib.so contrast_renderer_amd/libcontrast_hip.so; fi
  for n in off 5120 4096 3584 3072; do
    for w in ${WORKLOADS:-cubic}; do
      if [ $n = off ]; then timeout 300 python bench.py --workload $w --no-cpu-baseline --no-animated 2>/dev/null | tail -1 > gpurun_out/r05b/persist_${lib}_${n}_$w.json
      else CRH_RASTER_PERSISTENT=$n timeout 300 python bench.py --workload $w --no-cpu-baseline --no-animated 2>/dev/null | tail -1 > gpurun_out/r05b/persist_${lib}_${n}_$w.json; fi
    done
  done
done
cp /tmp/lib_shipped.so contrast_renderer_amd/libcontrast_hip.so
