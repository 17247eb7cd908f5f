mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05b/pytest.log
cat gpurun_out/r05b/pytest.log
for w in "" "--workload glyphs" "--workload dashed" "--workload s100k"; do
  timeout 300 python bench.py $w 2>gpurun_out/r05b/bench_err.log | tail -1 > "gpurun_out/r05b/bench$(echo $w | tr -d ' -').json"
done
CRH_TESS_TWO_PASS=1 timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/r05b/bench_twopass.json
CRH_TESS_TWO_PASS=1 timeout 300 python bench.py --workload glyphs 2>/dev/null | tail -1 > gpurun_out/r05b/bench_twopass_glyphs.json
