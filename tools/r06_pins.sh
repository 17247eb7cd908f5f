# round 6: the GPU suite under the pins that send it down the other code paths (every one must be bit-equal to the oracle too)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for pin in "CRH_TESS_TWO_PASS=1" "CRH_TESS_COUNT_EVERY_RUN=1" "CRH_TESS_RUN_BLOCK=128" "CRH_BIN_FLAT_THREADS=64" "CRH_NO_OPTIMISTIC_UPLOAD=1" "CRH_NO_DIRECT_LISTS=1" "CRH_FILL_KERNEL=0"; do
  echo "== $pin"
  env "$pin" timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_tess_one_pass.py::test_new_paths_of_the_same_structure_keep_the_capacities 2>&1 | grep -E "passed|failed|rror" | tail -3
done > gpurun_out/r06_pins.log 2>&1
cat gpurun_out/r06_pins.log
