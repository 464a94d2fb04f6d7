import sys; sys.path.insert(0,'/root/repo')
import ctypes as C, torch, json
from cvgpuspeedup_amd import capi, cvgs
from cvgpuspeedup_amd import workloads as W
dev=torch.device('cuda:0'); lib=capi.load_library()
w,h=W.FRAME_4K
def run(name, code, it, ot, out_cn):
    chains=[];keep=[]
    for i in range(12):
        src=W.random_u8_torch((h,w,cvgs.type_cn(it)),900+i,dev)
        out=torch.zeros((h,w,out_cn),dtype=torch.uint8,device=dev)
        ops=[cvgs.ReadIOp(capi.READ_PIXEL,it,[cvgs.GpuMat.from_tensor(src,it)],1), cvgs.cvtColor(code,it,ot), cvgs.write(ot,cvgs.GpuMat.from_tensor(out,ot))]
        chains.append(cvgs.lower(ops)); keep+=[src,out]
    s=torch.cuda.current_stream().cuda_stream
    st={'i':0}
    def launch():
        capi.check(lib.cvgs_execute(C.byref(chains[st['i']%12].desc), s)); st['i']+=1
    for _ in range(5): launch()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): launch()
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)*1e-3/50
    alg=w*h*(cvgs.type_cn(it)+out_cn)
    print(json.dumps({"case":name,"kernel":cvgs.kernel_name(*ops),"us":round(t*1e6,2),"GB_per_s":round(alg/t/1e9,1),"frac":round(alg/t/8e12,4)}))
run("4K BGR2RGB u8", cvgs.COLOR_BGR2RGB, cvgs.CV_8UC3, cvgs.CV_8UC3, 3)
run("4K BGR2GRAY u8", cvgs.COLOR_BGR2GRAY, cvgs.CV_8UC3, cvgs.CV_8UC1, 1)
run("4K BGR2BGRA u8", cvgs.COLOR_BGR2BGRA, cvgs.CV_8UC3, cvgs.CV_8UC4, 4)
run("4K BGRA2BGR u8", cvgs.COLOR_BGRA2BGR, cvgs.CV_8UC4, cvgs.CV_8UC3, 3)
