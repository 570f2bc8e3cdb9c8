#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of the default bench, then the HBM
# traffic counters in separate PMC passes (MI355X_MICROARCH.md, "rocprofv3 PMC slots").
# usage: tools/profile_round.sh <tag>
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 200 --warmup 100 --no-cpu > $O/bench_under_rocprof.json 2> $O/trace.log
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/pmc_$CNT -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu > /dev/null 2> $O/pmc_$CNT.log
done
python3 - <<PY
import csv, glob, collections, json
O="$O"
rows=[]
for f in glob.glob(O+"/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hrfd::" in r["Name"]:
            rows.append(r)
with open(O+"/kernel_stats_hrfd.csv","w") as f:
    w=csv.writer(f); w.writerow(["Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs","StdDev"])
    for r in rows: w.writerow([r["Name"],r["Calls"],r["TotalDurationNs"],r["AverageNs"],r["MinNs"],r["MaxNs"],r["StdDev"]])
acc=collections.defaultdict(list)
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob(O+"/pmc_"+c+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rx_wbfm" in r["Kernel_Name"]:   # k_rx_wbfm_stream<..> (batches) or k_rx_wbfm<3,..>
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
summ={k:{"n":len(v),"mean":sum(v)/len(v)} for k,v in acc.items()}
json.dump(summ, open(O+"/pmc_traffic.json","w"), indent=1)
print(open(O+"/kernel_stats_hrfd.csv").read())
print(summ)
print(open(O+"/bench_under_rocprof.json").read()[:300])
PY
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
