"""Quick per-source-type timing of the K1 chain (50 crops of a 4K frame, one resident frame): which kernel each type selects and its time per launch."""
import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvgpuspeedup_amd import cvgs, workloads as W
import ctypes as C
from cvgpuspeedup_amd import capi
dev=torch.device('cuda:0'); lib=capi.load_library()
def run(depth, cn, tdtype):
    fw,fh=W.FRAME_4K
    frame=(torch.rand((fh,fw,cn),device=dev)*200).to(tdtype)
    crops=W.random_crops(50,fw,fh)
    out=torch.zeros((50,cn*64*128),dtype=torch.float32,device=dev)
    ops=W.k1_chain(cvgs.GpuMat.from_tensor(frame,cvgs.make_type(depth,cn)),crops,cvgs.GpuMat.from_tensor(out,cvgs.CV_32FC1),cn=cn,src_depth=depth)
    ch=cvgs.lower(ops); s=torch.cuda.current_stream().cuda_stream
    for _ in range(20): lib.cvgs_execute(C.byref(ch.desc), s)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500): lib.cvgs_execute(C.byref(ch.desc), s)
    e1.record(); torch.cuda.synchronize()
    print(cvgs.kernel_name(*ops), depth, cn, round(e0.elapsed_time(e1)/500*1000,2),'us')
run(cvgs.CV_8U,3,torch.uint8); run(cvgs.CV_8U,4,torch.uint8); run(cvgs.CV_16U,3,torch.uint16); run(cvgs.CV_16S,4,torch.int16); run(cvgs.CV_32F,3,torch.float32)
