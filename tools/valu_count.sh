#!/bin/bash
# Runs on the GPU box: SQ_INSTS_VALU / SQ_WAVE_CYCLES of the headline kernel for library variants (a noise-free measure of an
# instruction-count change).  usage: tools/valu_count.sh OUT VARIANT...   ("ship" = hackrfdiags_amd/lib/libhrfd.so)
# HRFD_VC_KERNEL: substring of the kernel's name (default k_rx_wbfm_flow); HRFD_VC_ARGS: bench arguments (default: the headline),
# e.g. HRFD_VC_KERNEL='k_mod<1>' HRFD_VC_ARGS='--workload ssbmod' (round 6: the x8 tail's instruction diet)
cd /tmp && export TMPDIR=/tmp
export HRFD_BENCH_SETTLE=0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/valu_count
mkdir -p $O
out=$R/$1; shift
: > $out
for v in "$@"; do
  lib=$R/hackrfdiags_amd/lib/variants/$v/libhrfd.so
  [ "$v" = ship ] && lib=$R/hackrfdiags_amd/lib/libhrfd.so
  export HRFD_LIB=$lib
  rm -rf $O/raw
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/raw -- \
    python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --no-extras --verify 0 $HRFD_VC_ARGS > /dev/null 2> $O/$v.log || echo "FAILED $v" >> $out
  python3 - "$O/raw" "$v" "${HRFD_VC_KERNEL:-k_rx_wbfm_flow}" >> $out <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[3] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], " ".join("%s %.5g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
done
cat $out
