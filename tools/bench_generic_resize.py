import sys; sys.path.insert(0,'/root/repo')
import ctypes as C, torch, json
from cvgpuspeedup_amd import capi, cvgs
from cvgpuspeedup_amd import workloads as W
dev=torch.device('cuda:0'); lib=capi.load_library()
def run(name, dtype, depth, cn, src_wh, dst):
    sw,sh=src_wh; st=cvgs.make_type(depth,cn); f=cvgs.make_type(cvgs.CV_32F,cn)
    chains=[];keep=[]
    for i in range(6):
        src=(torch.rand((sh,sw,cn),device=dev)*200).to(dtype)
        out=torch.zeros((dst[1],dst[0],cn),dtype=torch.float32,device=dev)
        ops=[cvgs.resize(st,cvgs.INTER_LINEAR,cvgs.GpuMat.from_tensor(src,st),dst), cvgs.write(f,cvgs.GpuMat.from_tensor(out,f))]
        chains.append(cvgs.lower(ops)); keep+=[src,out]
    s=torch.cuda.current_stream().cuda_stream; st_={'i':0}
    def launch():
        capi.check(lib.cvgs_execute(C.byref(chains[st_['i']%6].desc), s)); st_['i']+=1
    for _ in range(3): launch()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): launch()
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)*1e-3/30
    print(json.dumps({"case":name,"kernel":cvgs.kernel_name(*ops),"us":round(t*1e6,2),"out_Mpix_per_s":round(dst[0]*dst[1]/t/1e6,1)}))
run("4K->1080p 32FC3", torch.float32, cvgs.CV_32F, 3, W.FRAME_4K, (1920,1080))
run("1080p->4K 32FC3", torch.float32, cvgs.CV_32F, 3, W.FRAME_1080P, (3840,2160))
run("4K->1080p 32FC1", torch.float32, cvgs.CV_32F, 1, W.FRAME_4K, (1920,1080))
run("4K->1080p 16UC1", torch.int16, cvgs.CV_16U, 1, W.FRAME_4K, (1920,1080))
run("4K->1080p 8UC3 (k1)", torch.uint8, cvgs.CV_8U, 3, W.FRAME_4K, (1920,1080))
