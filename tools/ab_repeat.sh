# usage: tools/ab_repeat.sh N   -- every variant under hackrfdiags_amd/lib/variants, N rounds, interleaved
for i in $(seq 1 ${1:-3}); do python tools/gpu_ab.py run; done
