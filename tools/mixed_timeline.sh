#!/bin/bash
# Runs on the GPU box: kernel start / end times (rocprofv3 --kernel-trace) of the last steps of the mixed-bank bench,
# relative to the step's first kernel: which kernels overlap (the AM+SSB recurrences and the finisher run on the side
# stream).  usage: tools/mixed_timeline.sh > gpurun_out/mixed_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tl_mixed
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --workload mixed --steps 30 --warmup 20 --no-cpu --no-extras "$@" > $O/bench.json 2> $O/log.txt
python3 - "$O" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hrfd::" in r["Kernel_Name"] and "build_atan" not in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void hrfd::", "").replace("hrfd::", ""), r.get("Queue_Id", "?")))
rows.sort()
# a step of the mixed bank is PER kernels (2: k_rx_wbfm_flow and k_rx_fir<15>), whatever their order
import os
per = int(os.environ.get("PER", "1"))
n_steps = len(rows) // per
for k in range(max(0, n_steps - 3), n_steps):
    chunk = rows[len(rows) - (n_steps - k) * per: len(rows) - (n_steps - k - 1) * per]
    t0 = chunk[0][0]
    nxt = len(rows) - (n_steps - k - 1) * per
    print("step:")
    for a, b, n, q in chunk:
        print("   %-28s queue %-3s start %8.1f us  end %8.1f us  (%.1f us)" % (n, q, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3))
    print("   step span %.1f us; next step's first kernel starts at %s" % ((max(r[1] for r in chunk) - t0) / 1e3, ("%.1f us" % ((rows[nxt][0] - t0) / 1e3)) if nxt < len(rows) else "-"))
PY
rm -rf $O
