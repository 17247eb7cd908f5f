#!/bin/bash
# container: gpurun_out/ of tools/r03_collect.sh (+ tools/r03_lines.sh, if run) -> profiles/r03_*
cd /root/repo
for w in cubic glyphs dashed s100k; do python tools/collect_profiles.py r03_$w r03 gpurun_out/bench_r03_$w.json $w > /dev/null 2>&1; done
python - <<'PY'
import json, os
def last(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith('{')][-1])
for n in ['s100k_loop8', 's100k_loop8_serial', 's100k_loop8_16f_serial', 's10k_strong_loop8_serial', 's10k_weak_loop8_serial']:
    json.dump(last(f'gpurun_out/bench_r03_{n}.json'), open(f'profiles/r03_bench_line_{n}.json', 'w'), indent=1)
m = {'line_r03': 'r03_bench_line', 'line_r03_glyphs': 'r03_bench_line_glyphs', 'line_r03_dashed': 'r03_bench_line_dashed', 'line_r03_s100k': 'r03_bench_line_s100k',
     'line_r03_reupload': 'r03_bench_line_reupload', 'line_r03_standalone_cubic': 'r03_bench_line_standalone', 'line_r03_standalone_glyphs': 'r03_bench_line_standalone_glyphs',
     'line_r03_standalone_dashed': 'r03_bench_line_standalone_dashed'}
if os.environ.get('LINES'):
    for a, b in m.items():
        json.dump(last(f'gpurun_out/{a}.json'), open(f'profiles/{b}.json', 'w'), indent=1)
PY
