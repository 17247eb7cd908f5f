#!/bin/bash
# GPU box: rocprofv3 evidence for the bench command ([WORKLOAD=glyphs|dashed] tools/profile.sh <tag>), written under gpurun_out/prof_<tag>/:
#   stats/   --kernel-trace --stats (per-kernel average duration; must agree with the HIP-event times bench.py prints)
#   fetch/, write/   separate --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), kernel trace only
#   summary.json     per-kernel: calls, avg ns, FETCH_SIZE / WRITE_SIZE per launch (raw counter units = KiB... see below)
# The guide's gfx950 correction (FETCH_SIZE reports half of a wide coalesced read) is applied in summary.json as fetch_bytes_x2.
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-side-workloads --no-animated --repeats 0 --workload ${WORKLOAD:-cubic}"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $out/stats -o r -- $BENCH > $out/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $out/fetch -o r -- $BENCH > $out/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $out/write -o r -- $BENCH > $out/write.log 2>&1
python - $out <<'PY'
import sys, csv, glob, json, collections
out = sys.argv[1]
def short(name):
    return name.split("(")[0].replace("void ", "").strip()
summary = collections.OrderedDict()
f = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    summary[short(row["Name"])] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "total_ns": float(row["TotalDurationNs"]), "pct": float(row["Percentage"])}
for key, sub in (("fetch_size", "fetch"), ("write_size", "write")):
    f = glob.glob(out + f"/{sub}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        summary.setdefault(k, {})[key + "_per_launch_raw"] = sum(v) / len(v)
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
for k, v in summary.items():
    print(k, v)
PY
