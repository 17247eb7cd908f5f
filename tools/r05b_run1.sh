# round 5 (second half): GPU tests, then the four workloads' bench lines, the re-upload line (and, with "ab", the two-pass tessellation / the old upload beside them)
mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05b/pytest.log
cat gpurun_out/r05b/pytest.log
for w in cubic glyphs dashed s100k; do
  timeout 300 python bench.py --workload $w 2>gpurun_out/r05b/bench_err_$w.log | tail -1 > gpurun_out/r05b/bench_$w.json
  if [ "$1" = "ab" ]; then CRH_TESS_TWO_PASS=1 timeout 300 python bench.py --workload $w 2>/dev/null | tail -1 > gpurun_out/r05b/bench_twopass_$w.json; fi
done
timeout 300 python bench.py --reupload 2>/dev/null | tail -1 > gpurun_out/r05b/bench_reupload.json
if [ "$1" = "ab" ]; then CRH_NO_OPTIMISTIC_UPLOAD=1 timeout 300 python bench.py --reupload 2>/dev/null | tail -1 > gpurun_out/r05b/bench_reupload_old.json; fi
true
