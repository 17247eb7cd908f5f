# round 5: where does the tessellation of frame i + 1 run — beside the binning of frame i (as ever) or beside its raster kernel — and with which stream priorities
mkdir -p gpurun_out/r05b
run() { # name, env...
  name=$1; shift
  for w in cubic glyphs s100k; do
    env "$@" timeout 300 python bench.py --workload $w --no-cpu-baseline --no-animated 2>/dev/null | tail -1 > gpurun_out/r05b/sched_${name}_$w.json
  done
}
run base CRH_NOP=1
run behind CRH_TESS_BEHIND_BINNING=1
run behind_thigh CRH_TESS_BEHIND_BINNING=1 "CRH_LANE_PRIORITY=-1 0 0"
run behind_thigh_rlow CRH_TESS_BEHIND_BINNING=1 "CRH_LANE_PRIORITY=-1 -1 1"
run tlow "CRH_LANE_PRIORITY=1 0 0"
run tlow_bhigh "CRH_LANE_PRIORITY=1 -1 0"
