#!/bin/bash
# Runs on the GPU box: is the headline kernel bound by VALU issue?  Derived and raw SQ counters in PMC-only passes.
cd /tmp && export TMPDIR=/tmp
export HRFD_BENCH_SETTLE=0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/valu_probe
mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -i -E "VALU|Busy|SQ_INST_CYCLES|SQ_BUSY_CU|SQ_CYCLES|SQ_THREAD|SQ_IFETCH|SQ_INSTS_VALU" $O/counters_list.txt | cut -c1-160 | sort -u | head -80 > $O/counters_of_interest.txt
pass() {
  local name=$1 cnt=$2; shift 2
  timeout -k 10 300 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/raw_$name -- \
    python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --no-extras --verify 0 "$@" > /dev/null 2> $O/$name.log || echo "FAILED $name"
}
pass d1 "VALUBusy SALUBusy VALUUtilization"
pass d2 "MemUnitBusy MemUnitStalled LDSBankConflict"
pass r1 "SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_WAVES"
pass r2 "SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT"
pass r3 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_SALU SQ_INSTS_SMEM"
python3 - "$O" <<'PY'
import csv, glob, collections, os, sys
O = sys.argv[1]
with open(O + "/valu_probe.txt", "w") as out:
    for d in sorted(glob.glob(O + "/raw_*")):
        name = os.path.basename(d)[4:]
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "k_rx_wbfm_flow" in k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out.write("== pass %s\n" % name)
        for k, cs in sorted(acc.items()):
            out.write("  %s\n" % k[:110])
            for c, v in sorted(cs.items()):
                out.write("    %-28s n=%3d mean=%.6g\n" % (c, len(v), sum(v) / len(v)))
print(open(O + "/valu_probe.txt").read())
# the summary bench.py quotes in its line (roofline.issue), keyed by the device code like the PMC traffic
import json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
want = ("VALUBusy", "VALUUtilization", "SALUBusy", "MemUnitStalled", "LDSBankConflict", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAVES")
summ = {}
txt = open(O + "/valu_probe.txt").read().split("\n")
for line in txt:
    f = line.split()
    if len(f) >= 3 and f[0] in want and f[-1].startswith("mean="):
        summ[f[0]] = float(f[-1][5:])
json.dump({"kernel_code_tag": bench.kernel_code_tag(), "kernel": "k_rx_wbfm_flow<4, false, false, 3>, 256 channels x 16 blocks",
           "how": "tools/valu_probe.sh: rocprofv3 --pmc (derived and raw SQ counters, PMC-only passes of bench.py --steps 6 --warmup 2), mean over the launches", "counters": summ},
          open(O + "/valu_probe.json", "w"), indent=1)
PY
rm -rf $O/raw_*
