cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "1024 16 10 1" "8192 16 4 1"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wbprof_$1 -- python3 $R/tools/wbmod_ab.py $1 $2 $3 $4 > $R/gpurun_out/wbprof_$1.log 2>&1
  f=$(find $R/gpurun_out/wbprof_$1 -name "*kernel_stats.csv" | head -1)
  echo "== $cfg"; cat $f | cut -c1-200
  rm -rf $R/gpurun_out/wbprof_$1
done
