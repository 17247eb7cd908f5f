# round 6: the GPU suite under more pins — every formulation and every fallback must draw the oracle's bytes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for pin in "CRH_TRIANGLE_PASS=1" "CRH_ROWS=1" "CRH_EDGE_PASS=1" "CRH_NO_BIN_BATCHES=1" "CRH_BIN_ITEMWISE=1" "CRH_LONG_LISTS=1" "CRH_LONG_LISTS=0" "CRH_HEAVY_FIRST=0" "CRH_NO_PIPELINE=1" "CRH_NO_LINEAGE=1" "CRH_NO_DIRECT_LISTS=1"; do
  echo "== $pin"
  env "$pin" timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_tess_one_pass.py::test_new_paths_of_the_same_structure_keep_the_capacities 2>&1 | grep -E "^FAILED|passed|failed|rror" | cut -c1-220 | tail -12
done > gpurun_out/r06_pins2.log 2>&1
cat gpurun_out/r06_pins2.log
