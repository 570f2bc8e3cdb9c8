import os, sys, time
os.environ["HRFD_DEBUG_HOOKS"]="1"
os.environ.setdefault("GPU_MAX_HW_QUEUES","2")
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from hackrfdiags_amd import api
dev=torch.device("cuda:0")
C, NB = 1024, 16
pcm=torch.randint(-20000,20000,(C,NB*512),dtype=torch.int16,device=dev)
out=torch.zeros((C,NB*512*512),dtype=torch.int8,device=dev)
for kind,name in ((api.MOD_FM,"fm"),(api.MOD_WBFM,"wbfm"),(api.MOD_AM,"am")):
    for sl in (1,0):
        for scan in (0,1):
            m=api.Mod(kind,C)
            m.debug_set_sliced(sl)
            if kind!=api.MOD_AM: m.debug_set_scan(scan)
            for _ in range(30): m.process_device(pcm.data_ptr(),NB*512,out.data_ptr())
            torch.cuda.synchronize()
            t0=time.perf_counter()
            N=30
            for _ in range(N): m.process_device(pcm.data_ptr(),NB*512,out.data_ptr())
            torch.cuda.synchronize()
            print(f"{name} sliced={sl} scan={'old' if scan else 'rows'}: {(time.perf_counter()-t0)/N*1e3:.4f} ms",flush=True)
            del m
