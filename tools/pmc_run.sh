#!/bin/bash
# usage: tools/pmc_run.sh "<counters>" <tag>   (runs on the GPU box; PMC only, no trace flags besides kernel-trace)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$2
timeout 200 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $OUT.log 2>&1
python3 - <<PY
import csv,glob,collections
fs=glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
acc=collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        if 'k_rx_wbfm' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
