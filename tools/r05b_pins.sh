# round 5: the GPU suite under each of the round's pins (every one must be bit-equal to the oracle too)
mkdir -p gpurun_out/r05b
for pin in CRH_TESS_COUNT_EVERY_RUN=1 CRH_TESS_RUN_BLOCK=128 CRH_BIN_FLAT_THREADS=64 CRH_NO_OPTIMISTIC_UPLOAD=1 "CRH_LANE_PRIORITY=0 0 0"; do
  echo "== $pin"
  env "$pin" timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_tess_one_pass.py::test_new_paths_of_the_same_structure_keep_the_capacities 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
done > gpurun_out/r05b/pins.log 2>&1
cat gpurun_out/r05b/pins.log
