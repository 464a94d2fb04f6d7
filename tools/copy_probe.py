import sys; sys.path.insert(0,'/root/repo')
import torch
from cvgpuspeedup_amd import capi
lib=capi.load_library()
dev=torch.device('cuda:0')
s=torch.cuda.current_stream().cuda_stream
for mib in (128,256,373,512,1024,2048):
    n=mib<<20
    a=torch.empty(n,dtype=torch.uint8,device=dev); b=torch.empty(n,dtype=torch.uint8,device=dev); a.zero_()
    for name,fn in (("torch",lambda: b.copy_(a)),("cvgs",lambda: capi.check(lib.cvgs_stream_copy(b.data_ptr(),a.data_ptr(),n,s)))):
        fn(); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print(mib,name,round(2*n*20/(e0.elapsed_time(e1)*1e-3)/1e9,1))
    del a,b; torch.cuda.empty_cache()
