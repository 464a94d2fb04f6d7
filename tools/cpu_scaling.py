import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, os
from cvgpuspeedup_amd import cvgs
from cvgpuspeedup_amd import workloads as W
from oracle import oracle_binding as ob
frame=W.random_u8((2160,3840,3), W.SEED)
crops=W.random_crops(50,3840,2160,seed=W.SEED+500000)
b=np.zeros((50,3*64*128),np.float32)
ch=cvgs.lower(W.k1_chain(cvgs.GpuMat.from_array(frame,cvgs.CV_8UC3),crops,cvgs.GpuMat.from_array(b,cvgs.CV_32FC1)))
lib=ob.load_oracle(); print("max threads", lib.oracle_max_threads(), "nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for th in (1,2,4,8,16,32,64,128):
    lib.oracle_set_threads(th)
    reps=max(8,2*th)
    ob.execute_k1_fast(ch, reps)
    t0=time.perf_counter(); n=0
    while time.perf_counter()-t0<1.5:
        ob.execute_k1_fast(ch, reps); n+=reps
    dt=time.perf_counter()-t0
    print(th, "threads: %.1f Mpix/s"%(n*50*8192/dt/1e6))
