# round 5: k_bin_flat by workgroup shape (threads per workgroup; tables scale with it) — binning alone and the pipelined step, S10k / glyphs / S100k
mkdir -p gpurun_out/r05b
for shape in 256 128 64; do
  CRH_EXTRA_FLAGS="-DCRH_FLAT_THREADS=$shape" python -c "
from contrast_renderer_amd import build as b
b.build_library(force=True)" > /dev/null 2>&1 || { echo "build failed for $shape"; continue; }
  for w in cubic glyphs s100k; do
    timeout 300 python bench.py --workload $w --no-cpu-baseline --no-animated 2>/dev/null | tail -1 > gpurun_out/r05b/flat${shape}_$w.json
  done
done
