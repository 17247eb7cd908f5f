#!/bin/bash
# GPU box: kernel + copy timeline of a command (rocprofv3 --kernel-trace --memory-copy-trace, no counters) -> gpurun_out/trace_<tag>/; prints the last steps
# usage: tools/r06_trace.sh <tag> <command...>
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$tag
rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -f csv -d $out -o t -- "$@" > $out/run.log 2>&1
python - $out <<'PY'
import sys, csv, glob
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("void ", "").replace("crh::", "")[:44], r.get("Queue_Id", "")))
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", ""), ""))
rows.sort()
if not rows:
    print("no rows"); sys.exit(0)
tail = rows[-110:]
t0 = tail[0][0]
for s, e, n, q in tail:
    print("%9.1f %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
PY
