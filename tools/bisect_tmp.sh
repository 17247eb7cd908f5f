cd $GRAFT_REPO_ROOT
for flags in "-DCRH_FLAT_ROUNDS=3 -DCRH_FLAT_STAGE=128" "-DCRH_FLAT_ROUNDS=2 -DCRH_FLAT_STAGE=128" "-DCRH_FLAT_ROUNDS=3 -DCRH_FLAT_STAGE=256"; do
  CRH_EXTRA_FLAGS="$flags" python contrast_renderer_amd/build.py --force > /dev/null 2>&1
  echo "== $flags"; python -m pytest tests/test_gpu_parity.py -m gpu -q -k "golden" 2>&1 | tail -1
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
