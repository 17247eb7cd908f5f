mkdir -p gpurun_out/r05c
i=0
for p in "1 0 0" "1 0 -1" "1 -1 -1" "1 1 0" "1 -1 0" "0 0 -1"; do
  i=$((i+1))
  CRH_LANE_PRIORITY="$p" timeout 300 python bench.py --no-cpu-baseline --no-animated 2>/dev/null | tail -1 > gpurun_out/r05c/prio_$i.json
  echo "$p" > gpurun_out/r05c/prio_$i.txt
done
