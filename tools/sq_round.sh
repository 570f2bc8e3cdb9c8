#!/bin/bash
# Runs on the GPU box.  SQ counters of the headline kernel (and of the mixed bank and the SSB modulator) in PMC-only
# passes, eight SQ slots per pass: where the waves' cycles go.  usage: tools/sq_round.sh <tag>
TAG=${1:-r5}
cd /tmp && export TMPDIR=/tmp
export HRFD_BENCH_SETTLE=0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sq_$TAG
mkdir -p $O
pass() {   # name, counters, bench args
  local name=$1 cnt=$2; shift 2
  timeout -k 10 300 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/raw_$name -- \
    python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --no-extras --verify 0 "$@" > /dev/null 2> $O/$name.log || echo "FAILED $name"
}
pass wbfm_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM"
pass wbfm_b "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
pass wbfm_c "GRBM_GUI_ACTIVE GRBM_COUNT"
pass mixed_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY" --workload mixed
pass ssbmod_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY" --workload ssbmod
pass wbfmmod_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" --workload wbfmmod
python3 - "$O" <<'PY'
import csv, glob, collections, os, sys
O = sys.argv[1]
with open(O + "/sq_counters.txt", "w") as out:
    for d in sorted(glob.glob(O + "/raw_*")):
        name = os.path.basename(d)[4:]
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "hrfd::" in k and "k_membw" not in k and "k_build_atan" not in k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        out.write("== pass %s\n" % name)
        for k, cs in sorted(acc.items()):
            out.write("  %s\n" % k[:110])
            for c, v in sorted(cs.items()):
                out.write("    %-24s n=%3d mean=%.6g\n" % (c, len(v), sum(v) / len(v)))
print(open(O + "/sq_counters.txt").read())
PY
rm -rf $O/raw_*
