for rl in 1 2 4 8 16; do echo "runlen $rl"; HRFD_RUNLEN=$rl python tools/gpu_time.py 1; done
for st in 0 1 2 8 16; do echo "stagger $st"; HRFD_STAGGER=$st python tools/gpu_time.py 1; done
