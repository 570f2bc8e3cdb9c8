#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace --stats of the default bench (and of the other bench workloads when
# a second argument is given): per-kernel call counts and durations.  (The HBM traffic counters are tools/pmc_round.sh,
# the SQ counters tools/sq_round.sh: counters are collected in runs of their own, never with --stats.)
# Round 6: two summaries per workload.  kernel_stats_<name>.csv is rocprofv3's own --stats table over EVERY launch of the
# process (settle launches on the clock ramp included: ~140 of them run cold); kernel_stats_<name>_timed.csv
# is made from the same run's kernel trace over the launches of the bench's TIMED REGION only (the dispatches behind
# settle + warm-up, `steps` of them per kernel and step): its mean duration x the algorithmic bytes reproduces the
# line's roofline.frac, and (last end - first start) / steps is the line's region_ms_per_launch.
# usage: tools/profile_round.sh <tag> [all]
TAG=${1:-r6}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
stats() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python3 $R/bench.py --no-cpu --no-extras --verify 0 "$@" > $O/bench_under_rocprof_$name.json 2> $O/trace_$name.log
  python3 - "$O" "$name" <<'PY'
import csv, glob, json, sys
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
# the timed region alone, from the trace
try:
    line = json.loads([l for l in open(O + "/bench_under_rocprof_" + name + ".json") if l.startswith("{")][-1])
except Exception as e:
    print("no bench line:", e); sys.exit(0)
steps, skip = line["steps"], line.get("settle_steps", 0) + line["warmup"]
tr = []
for f in glob.glob(O + "/trace_" + name + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "hrfd::" in n and "build_atan" not in n and "k_membw" not in n and "k_sig" not in n and "k_gen" not in n:
            tr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
tr.sort()
names = sorted({t[2] for t in tr})
with open(O + "/kernel_stats_" + name + "_timed.csv", "w") as f:
    f.write(f"# timed region only: per kernel, the dispatches behind the first settle + warm-up = {skip} steps, {steps} steps' worth; bench line of this run: "
            f"ms_per_step {line['ms_per_step']}, region_ms_per_launch {line['roofline'].get('region_ms_per_launch')}, frac {line['roofline']['frac']}, "
            f"algorithmic bytes per launch {line['roofline']['algorithmic_bytes_per_launch']}\n")
    w = csv.writer(f); w.writerow(["Name", "CallsPerStep", "Calls", "AverageNs", "MinNs", "MaxNs", "SumPerStepNs", "FracOf8TBps_bytes_over_SumPerStep"])
    tot_first, tot_last = None, None
    for n in names:
        mine = [t for t in tr if t[2] == n]
        per = round(len(mine) / (skip + steps + line['roofline'].get('kernel_launches_sampled', 0)))   # launches of this kernel per step
        if per < 1:
            continue
        sel = mine[skip * per:(skip + steps) * per]
        if not sel:
            continue
        d = [b - a for a, b, _ in sel]
        sum_step = sum(d) / steps
        frac = line['roofline']['algorithmic_bytes_per_launch'] / (sum_step * 1e-9) / 8e12
        w.writerow([n, per, len(sel), round(sum(d) / len(d), 1), min(d), max(d), round(sum_step, 1), round(frac, 4)])
        tot_first = sel[0][0] if tot_first is None else min(tot_first, sel[0][0])
        tot_last = sel[-1][1] if tot_last is None else max(tot_last, sel[-1][1])
    if tot_first is not None:
        f.write(f"# (last end - first start) / steps = {(tot_last - tot_first) / steps / 1e6:.4f} ms per step under the profiler\n")
print(open(O + "/kernel_stats_" + name + "_timed.csv").read())
PY
  rm -rf $O/trace_$name
}
stats wbfm256x16 --steps 200 --warmup 100
if [ "$2" = "all" ]; then
  stats wbfm1024x16 --steps 60 --warmup 40 --channels 1024
  stats mixed --steps 100 --warmup 50 --workload mixed
  stats ssbmod1024 --steps 100 --warmup 50 --workload ssbmod
  stats wbfmmod1024 --steps 20 --warmup 10 --workload wbfmmod
  stats wbfmmod8192 --steps 6 --warmup 2 --workload wbfmmod --channels 8192
  stats wbfm256x16_quiet25 --steps 100 --warmup 50 --quiet-fraction 0.25 --threshold -30
  stats wbfm256x16_iqdump --steps 100 --warmup 50 --iqdump
  stats am256x16 --steps 100 --warmup 50 --workload am
  stats fm256x16 --steps 100 --warmup 50 --workload fm
  stats ssb256x16 --steps 100 --warmup 50 --workload ssb
  stats ammod1024 --steps 100 --warmup 50 --workload ammod
  stats fmmod1024 --steps 100 --warmup 50 --workload fmmod
fi
