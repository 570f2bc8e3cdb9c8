#!/bin/bash
# Runs on the GPU box.  HBM traffic of EVERY bench workload from the PMC counters, collected as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE each in a rocprofv3 run of its own
# (--pmc with --kernel-trace only), of the very bench command whose line quotes them.  The summary carries the hash of
# the device code it measured (bench.kernel_code_tag: the .hip_fatbin of libhrfd.so); bench.py reports `traffic` only
# for that code.  Per workload: the sum over ALL hrfd:: kernels of a step (the batch kernel and whatever runs behind it),
# per step; the two plain stream kernels of the denominators (k_membw_*) are not counted.
# usage: tools/pmc_round.sh <tag>            -> gpurun_out/pmc_<tag>/pmc_traffic.json (+ per-kernel table)
TAG=${1:-r5}
cd /tmp && export TMPDIR=/tmp
export HRFD_BENCH_SETTLE=0          # counters, not clocks: no settling launches
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
STEPS=6; WARM=2
one() {   # name, bench args...   (ONLY=<name>: that workload alone, merged into profiles/latest_pmc_traffic.json of the same device code)
  local name=$1; shift
  if [ -n "$ONLY" ] && [ "$ONLY" != "$name" ]; then return; fi
  for CNT in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/raw_${name}_$CNT -- \
      python3 $R/bench.py --steps $STEPS --warmup $WARM --no-cpu --no-extras --verify 0 "$@" > $O/${name}_$CNT.json 2> $O/${name}_$CNT.log || echo "FAILED $name $CNT"
  done
  echo "done $name" >> $O/progress.txt
}
one wbfm_256x16
one wbfm_1024x16 --channels 1024
one mixed_256x16 --workload mixed
one wbfm_256x16_random --signal random
one wbfm_256x16_quiet25 --quiet-fraction 0.25 --threshold -30
one wbfm_256x16_iqdump --iqdump
one am_256x16 --workload am
one fm_256x16 --workload fm
one ssb_256x16 --workload ssb
one ssbmod_1024x16 --workload ssbmod
one ammod_1024x16 --workload ammod
one fmmod_1024x16 --workload fmmod
one wbfmmod_1024x16 --workload wbfmmod
one wbfm_4096x16 --channels 4096
one wbfm_256x64 --blocks 64
one wbfmmod_8192x16 --workload wbfmmod --channels 8192
python3 - "$O" "$R" $STEPS $WARM <<'PY'
import csv, glob, collections, json, os, sys
O, R, STEPS, WARM = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
import importlib.util
spec = importlib.util.spec_from_file_location("bench", R + "/bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
out = {"kernel_code_tag": b.kernel_code_tag(), "kernel_source_tag": b.kernel_source_tag(),
       "how": "rocprofv3 --pmc <counter> --kernel-trace, one counter per run, bench.py --steps %d --warmup %d --no-cpu --no-extras <workload>; "
              "KiB per step = sum over the step's hrfd:: kernels of (mean per dispatch x dispatches per step)" % (STEPS, WARM),
       "workloads": {}}
table = []
for d in sorted(glob.glob(O + "/raw_*_FETCH_SIZE")):
    name = os.path.basename(d)[4:-len("_FETCH_SIZE")]
    w = {}
    for cnt in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob(O + "/raw_%s_%s/**/*counter_collection.csv" % (name, cnt), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "hrfd::" in k and "k_membw" not in k and "k_build_atan" not in k and "k_atan_eval" not in k and r["Counter_Name"] == cnt:
                    acc[k].append(float(r["Counter_Value"]))
        if not acc:
            continue
        # HRFD_BENCH_SETTLE=0: warm-up + timed steps are all the steps the bench ran -- plus, for the receive workloads, the
        # min(8, steps) single launches it samples behind the timed region (bench.measure_rx, round 5)
        # (round 6: the modulator workloads sample min(4, steps) single calls behind theirs, bench.measure_mod)
        steps_total = STEPS + WARM + (min(4, STEPS) if name.endswith(("mod_1024x16", "mod_8192x16")) else min(8, STEPS))
        tot = 0.0
        for k, v in sorted(acc.items()):
            per_step = len(v) / steps_total
            mean = sum(v) / len(v)
            tot += mean * per_step
            table.append((name, cnt, k[:90], len(v), round(per_step, 2), round(mean, 1)))
        w[cnt + "_KiB"] = tot
    if len(w) == 2:
        out["workloads"][name] = w
if os.environ.get("ONLY"):
    prev = json.load(open(R + "/profiles/latest_pmc_traffic.json"))
    assert prev["kernel_code_tag"] == out["kernel_code_tag"], "the summary to merge into is of another device code"
    prev["workloads"].update(out["workloads"])
    out = prev
json.dump(out, open(O + "/pmc_traffic.json", "w"), indent=1)
with open(O + "/pmc_per_kernel.txt", "w") as f:
    f.write("workload, counter, kernel, dispatches, per step, mean KiB per dispatch\n")
    for t in table:
        f.write(", ".join(str(x) for x in t) + "\n")
print(json.dumps(out["workloads"], indent=1))
PY
rm -rf $O/raw_*
