cd $GRAFT_REPO_ROOT
CRH_EXTRA_FLAGS=-DCRH_ABLATE python contrast_renderer_amd/build.py --force > /dev/null 2>&1
fmt='import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if k.startswith("raster_bin")})'
for dbg in 0 2097152 8388608 10485760; do
  echo "debug $dbg (2097152 no pair stores, 8388608 no record stores)"
  CRH_RASTER_DEBUG=$dbg CRH_EDGE_PASS=1 CRH_NO_PIPELINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt"
done
python contrast_renderer_amd/build.py --force > /dev/null 2>&1
