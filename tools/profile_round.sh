#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace --stats of the default bench (and of the other bench workloads when
# a second argument is given): per-kernel call counts and durations.  (The HBM traffic counters are tools/pmc_round.sh,
# the SQ counters tools/sq_round.sh: counters are collected in runs of their own, never with --stats.)
# usage: tools/profile_round.sh <tag> [all]
TAG=${1:-r5}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
stats() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python3 $R/bench.py --no-cpu --no-extras --verify 0 "$@" > $O/bench_under_rocprof_$name.json 2> $O/trace_$name.log
  python3 - "$O" "$name" <<'PY'
import csv, glob, sys
O, name = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(O + "/trace_" + name + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hrfd::" in r["Name"]:
            rows.append(r)
with open(O + "/kernel_stats_" + name + ".csv", "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
    for r in rows: w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
print(open(O + "/kernel_stats_" + name + ".csv").read())
PY
  rm -rf $O/trace_$name
}
stats wbfm256x16 --steps 200 --warmup 100
if [ "$2" = "all" ]; then
  stats wbfm1024x16 --steps 60 --warmup 40 --channels 1024
  stats mixed --steps 100 --warmup 50 --workload mixed
  stats ssbmod1024 --steps 100 --warmup 50 --workload ssbmod
  stats wbfmmod1024 --steps 20 --warmup 10 --workload wbfmmod
  stats wbfm256x16_quiet25 --steps 100 --warmup 50 --quiet-fraction 0.25 --threshold -30
  stats wbfm256x16_iqdump --steps 100 --warmup 50 --iqdump
  stats am256x16 --steps 100 --warmup 50 --workload am
  stats fm256x16 --steps 100 --warmup 50 --workload fm
  stats ssb256x16 --steps 100 --warmup 50 --workload ssb
  stats ammod1024 --steps 100 --warmup 50 --workload ammod
  stats fmmod1024 --steps 100 --warmup 50 --workload fmmod
fi
