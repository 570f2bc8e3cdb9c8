#!/bin/bash
# Runs on the GPU box: start / end times (rocprofv3 --kernel-trace) of the hrfd kernels of the LAST step of a bench
# workload, relative to the step's first kernel, with the hardware queue each ran on: which kernels overlap.
# usage: tools/kernel_timeline.sh <kernels per step> <bench args...>  > gpurun_out/timeline.txt
PER=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tl_any
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu ${HRFD_TL_EXTRAS:---no-extras} "$@" > $O/bench.json 2> $O/log.txt
python3 - "$O" "$PER" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hrfd::" in r["Kernel_Name"] and "build_atan" not in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void hrfd::", "").replace("hrfd::", ""), r.get("Queue_Id", "?")))
rows.sort()
per = int(sys.argv[2])
chunk = rows[-per:]
t0 = chunk[0][0]
for a, b, n, q in chunk:
    print("   %-34s queue %-3s start %9.1f us  end %9.1f us  (%.1f us)" % (n[:34], q, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3))
print("   span %.1f us" % ((max(r[1] for r in chunk) - t0) / 1e3))
PY
rm -rf $O
