#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of the default bench (and of the other bench workloads when
# a second argument is given), then the HBM traffic counters in separate PMC passes (MI355X_MICROARCH.md,
# "rocprofv3 PMC slots").  The PMC summary carries the tag of the kernel sources it was taken with
# (bench.kernel_source_tag), so that bench.py reports it only for the code it measured.
# usage: tools/profile_round.sh <tag> [all]
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
stats() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -- python3 $R/bench.py --no-cpu --no-extras "$@" > $O/bench_under_rocprof_$name.json 2> $O/trace_$name.log
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
fi
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/pmc_$CNT -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extras > /dev/null 2> $O/pmc_$CNT.log
done
python3 - "$O" "$R" <<'PY'
import csv, glob, collections, json, sys
O, R = sys.argv[1], sys.argv[2]
sys.path.insert(0, R)
acc = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(O + "/pmc_" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rx_wbfm" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {"n": len(v), "mean": sum(v) / len(v)} for k, v in acc.items()}
import importlib.util
spec = importlib.util.spec_from_file_location("bench", R + "/bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
summ["kernel_source_tag"] = b.kernel_source_tag()
summ["kernel"] = "hrfd::k_rx_wbfm_flow"
json.dump(summ, open(O + "/pmc_traffic.json", "w"), indent=1)
print(summ)
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
