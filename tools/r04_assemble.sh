#!/bin/bash
# container: gpurun_out/ of tools/r04_collect.sh -> profiles/r04_*
cd /root/repo
for w in cubic glyphs dashed s100k; do python tools/collect_profiles.py r04_$w r04 gpurun_out/bench_r04_$w.json $w > /dev/null 2>&1 || echo "collect_profiles failed for $w"; done
python - <<'PY'
import json
def last(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith('{')][-1])
for n in ['s100k_loop8', 's100k_loop8_serial', 'reupload', 'gpus2_same_device']:
    json.dump(last(f'gpurun_out/bench_r04_{n}.json'), open(f'profiles/r04_bench_line_{n}.json', 'w'), indent=1)
for w, sfx in (('cubic', ''), ('glyphs', '_glyphs'), ('dashed', '_dashed'), ('s100k', '_s100k')):
    json.dump(last(f'gpurun_out/bench_r04_{w}_standalone.json'), open(f'profiles/r04_bench_line_standalone{sfx}.json', 'w'), indent=1)
PY
python tools/isa_histogram.py r04 > /dev/null 2>&1 || echo "isa_histogram failed"
python - <<'PY'
import bench
print("hash", bench.kernel_source_hash(), "traffic", bench.measured_traffic("raster_tiles", "cubic"))
PY
