#!/bin/bash
# GPU box: parity subset, then the bench of the given workloads as the library's trial decides and with the row-span kernel pinned / excluded.
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_raster_rows.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "${1:-rows or random_scene or random_paths or quadratic or mixed or cubic or glyphs_600 or config2 or config3 or occlude or chunks}" 2>&1 | tail -12
fmt='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), d["check"] and d["check"]["frame_equals_oracle"], d["roofline"]["pass"][:40], {k:(round(v["avg_ms"],4), v["alone_ms"] and round(v["alone_ms"],4)) for k,v in d["kernels"].items() if k.startswith("raster")})'
for w in ${2:-cubic}; do
  echo "== $w trial / rows / no rows"
  CRH_PASS_VERBOSE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | grep -a "pass trial\|^{" | sed 's/^{.*/JSON&/' | while read -r line; do case "$line" in JSON*) echo "${line#JSON}" | python -c "$fmt";; *) echo "$line";; esac; done
  CRH_ROWS=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
  CRH_NO_ROWS=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w 2>&1 | tail -1 | python -c "$fmt"
done
