import sys; sys.path.insert(0,'/root/repo')
import ctypes as C, numpy as np, torch
from cvgpuspeedup_amd import capi, cvgs
from cvgpuspeedup_amd import workloads as W
dev=torch.device('cuda:0'); lib=capi.load_library()
f=cvgs.CV_32FC3
n=50; dst=(112,112)
chains=[];keep=[]
for i in range(16):
    fr=W.random_u8_torch((2160,3840,3),600+i,dev)
    out=torch.zeros((n,3*dst[0]*dst[1]),dtype=torch.float32,device=dev)
    m=cvgs.GpuMat.from_tensor(fr,cvgs.CV_8UC3)
    ms=[]
    for k in range(n):
        a=0.1*k; sc=0.3+0.02*k
        ms.append([[sc*np.cos(a),-sc*np.sin(a),-200.0-10*k],[sc*np.sin(a),sc*np.cos(a),-100.0-5*k]])
    ops=[cvgs.warp(cvgs.WARP_AFFINE,cvgs.CV_8UC3,[m]*n,ms,dst),cvgs.multiply(f,[0.3]*3),cvgs.subtract(f,W.K1_SUB[3]),cvgs.divide(f,W.K1_DIV[3]),cvgs.split(f,cvgs.GpuMat.from_tensor(out,cvgs.CV_32FC1),dst)]
    chains.append(cvgs.lower(ops)); keep+=[fr,out]
print(cvgs.kernel_name(*ops))
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    for ch in chains: capi.check(lib.cvgs_execute(C.byref(ch.desc), s.cuda_stream))
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for k in range(64): capi.check(lib.cvgs_execute(C.byref(chains[k%16].desc), torch.cuda.current_stream().cuda_stream))
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("us per launch", e0.elapsed_time(e1)*1e3/(20*64), "Mpix/s", n*dst[0]*dst[1]/(e0.elapsed_time(e1)*1e-3/(20*64))/1e6)
