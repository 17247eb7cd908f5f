#!/bin/bash
# container: gpurun_out/ of tools/r06_collect.sh -> profiles/r06_*
cd /root/repo
for w in cubic glyphs dashed s100k; do python tools/collect_profiles.py r06_$w r06 gpurun_out/bench_r06_$w.json $w > /dev/null 2>&1 || echo "collect_profiles failed for $w"; done
python - <<'PY'
import json, shutil
def last(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith('{')][-1])
names = ['reupload', 'gpus2_same_device'] + [f'{w}_loop8_{sp}_serial' for w in ('cubic', 's100k') for sp in ('path', 'tile')]
for n in names:
    try:
        json.dump(last(f'gpurun_out/bench_r06_{n}.json'), open(f'profiles/r06_bench_line_{n}.json', 'w'), indent=1)
    except Exception as e:
        print("missing", n, e)
for w, sfx in (('cubic', ''), ('glyphs', '_glyphs'), ('dashed', '_dashed'), ('s100k', '_s100k')):
    json.dump(last(f'gpurun_out/bench_r06_{w}_standalone.json'), open(f'profiles/r06_bench_line_standalone{sfx}.json', 'w'), indent=1)
for n in ('r06_slab_step_cubic.txt', 'r06_slab_step_s100k.txt', 'r06_animated_check.txt', 'pytest_r06.log'):
    try:
        shutil.copy(f'gpurun_out/{n}', f'profiles/{n if n.startswith("r06") else "r06_" + n}')
    except Exception as e:
        print("missing", n, e)
PY
python tools/isa_histogram.py r06 > /dev/null 2>&1 || echo "isa_histogram failed"
python - <<'PY'
import bench
print("hash", bench.kernel_source_hash(), "traffic", bench.measured_traffic("raster_tiles", "cubic"))
PY
