echo "full blocks, 16 per channel"; python tools/gpu_ab.py run base
for v in half640 half1024; do for nb in 24 32 48; do echo "$v half blocks x $nb"; HRFD_BLK=131072 HRFD_B=$nb python tools/gpu_ab.py run $v; done; done
