#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of ONE bench workload.  usage: tools/prof_one.sh <name> <bench args...>
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_one
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$NAME -- python3 $R/bench.py --no-cpu --no-extras "$@" > $O/bench_$NAME.json 2> $O/trace_$NAME.log
python3 - "$O" "$NAME" <<'PY'
import csv, glob, sys
O, name = sys.argv[1], sys.argv[2]
for f in glob.glob(O + "/trace_" + name + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hrfd::" in r["Name"]:
            print("%-70s calls %5s avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $O/trace_$NAME
