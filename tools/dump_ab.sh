# A/B on one box: the iq dump's counted waits (shipped) against round 3's plain counts (variant "dumpold"), alternating
for i in 1 2 3; do
  for v in shipped dumpold; do
    if [ $v = shipped ]; then L=$PWD/hackrfdiags_amd/lib/libhrfd.so; else L=$PWD/hackrfdiags_amd/lib/variants/dumpold/libhrfd.so; fi
    echo -n "$v iqdump: "; HRFD_LIB=$L timeout -k 10 100 python bench.py --iqdump --no-cpu --no-extras 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  done
done
echo -n "shipped plain: "; timeout -k 10 100 python bench.py --no-cpu --no-extras 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
