#!/bin/bash
# A/B of libhrfd variants on ONE box, alternating (boxes differ by +-3 %, runs on a box by +-1 %): the default bench
# headline (256 WBFM channels x 16 blocks, back-to-back launches, one event pair around the timed region) and the
# 1024-channel shape.  usage: tools/flow_ab.sh OUT NAME[=path] ...   ("ship" = hackrfdiags_amd/lib/libhrfd.so)
set -e -o pipefail
mkdir -p gpurun_out
out=$1; shift
: > $out
for rep in 1 2 3; do
  for v in "$@"; do
    lib=$PWD/hackrfdiags_amd/lib/variants/$v/libhrfd.so
    [ "$v" = ship ] && lib=$PWD/hackrfdiags_amd/lib/libhrfd.so
    for C in ${AB_CHANNELS:-256 1024}; do
      HRFD_LIB=$lib python3 bench.py --no-extras --no-cpu --steps 100 --warmup 100 --channels $C --verify ${AB_VERIFY:-4} ${AB_EXTRA} > gpurun_out/_line.json
      python3 - "$v" "$C" <<'PY' >> $out
import json, sys
l = json.load(open("gpurun_out/_line.json"))
r = l["roofline"]
print(f"{sys.argv[1]:10s} C {sys.argv[2]:>4s} ms_per_step {l['ms_per_step']:.4f} region_kernel_ms {r['kernel_ms_mean']:.4f} single_min {r['kernel_ms_min']:.4f} frac {r['frac']:.4f} "
      f"oracle_ok {l['verification'].get('oracle_channels_checked')} repaired {l['verification']['tiles_repaired_in_place']} uncommitted {l['verification']['uncommitted_launches']}")
PY
    done
  done
done
cat $out
