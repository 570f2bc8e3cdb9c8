#!/bin/bash
# usage: tools/pmc_any.sh "<counters>" <tag> <kernel substring> <bench args...>   (on the GPU box; PMC only)
CNT=$1; TAG=$2; KERN=$3; shift 3
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu "$@" > $OUT.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
