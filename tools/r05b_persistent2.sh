mkdir -p gpurun_out/r05b
for n in off 5120 4096 3072; do
  if [ $n = off ]; then timeout 300 python bench.py --no-cpu-baseline --no-animated 2>/dev/null | tail -1 > gpurun_out/r05b/persist2_${n}.json
  else CRH_RASTER_PERSISTENT=$n timeout 300 python bench.py --no-cpu-baseline --no-animated 2>/dev/null | tail -1 > gpurun_out/r05b/persist2_${n}.json; fi
done
