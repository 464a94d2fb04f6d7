#!/usr/bin/env python3
"""A decoder surface -> detector input: NV12 1080p / 4K -> 640x640 (stretch and aspect-ratio-preserving letterbox) -> RGB-order
normalize -> NCHW fp32, one K4 launch; device time per launch from a replayed HIP graph, 24 surfaces in rotation."""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvgpuspeedup_amd import capi, cvgs
from cvgpuspeedup_amd import workloads as W
dev = torch.device("cuda:0"); lib = capi.load_library()
def run(name, w, h, dst, ar, n_rot=24, iters=100, flags=0, u8_image=False, layout=capi.YUV_NV12, ref_chain=False, f32_image=False):
    chains=[]; keep=[]
    f3=cvgs.CV_32FC3
    for i in range(n_rot):
        surf = W.random_u8_torch((h*3//2, w), 100+i, dev)
        m = cvgs.GpuMat.from_tensor(surf, cvgs.CV_8UC1)
        luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, m.data, m.step, owner=m.owner)
        rd = cvgs.read_nv12(luma, dst, capi.YUV_LIMITED, capi.BT709, False, layout=layout)
        rd.ar = ar
        rd.background = cvgs._scalar([114.0,114.0,114.0])
        if ref_chain:  # the reference's own fused chain (tests/resize/test_fused_resize.cu:141-147): float4 -> SaturateCast -> VectorReorder<uchar4,2,1,0,3> -> write
            rd = cvgs.read_nv12(luma, dst, capi.YUV_FULL, capi.BT709, True, layout=layout)
            out = torch.zeros((dst[1], dst[0], 4), dtype=torch.uint8, device=dev)
            ops=[rd, cvgs.convertTo(cvgs.CV_32FC4, cvgs.CV_8UC4), cvgs.cvtColor(cvgs.COLOR_RGBA2BGRA, cvgs.CV_8UC4), cvgs.write(cvgs.CV_8UC4, cvgs.GpuMat.from_tensor(out, cvgs.CV_8UC4))]
        elif f32_image:  # -> packed CV_32FC3 image scaled to 0..1
            out = torch.zeros((dst[1], dst[0], 3), dtype=torch.float32, device=dev)
            ops=[rd, cvgs.multiply(f3,[1/255.0]*3), cvgs.write(f3, cvgs.GpuMat.from_tensor(out, f3))]
        elif u8_image:  # -> BGR u8 image (thumbnail / display path): swap, saturating cast, packed pixels
            out = torch.zeros((dst[1], dst[0], 3), dtype=torch.uint8, device=dev)
            ops=[rd, cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f3), cvgs.convertTo(f3, cvgs.CV_8UC3), cvgs.write(cvgs.CV_8UC3, cvgs.GpuMat.from_tensor(out, cvgs.CV_8UC3))]
        else:
            out = torch.zeros((1, 3*dst[0]*dst[1]), dtype=torch.float32, device=dev)
            ops=[rd, cvgs.multiply(f3,[1/255.0]*3), cvgs.subtract(f3,[0.485,0.456,0.406]), cvgs.divide(f3,[0.229,0.224,0.225]), cvgs.split(f3, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), dst)]
        chains.append(cvgs.lower(ops, flags)); keep += [surf,out]
    s=torch.cuda.current_stream().cuda_stream
    st={'i':0}
    def launch():
        capi.check(lib.cvgs_execute(C.byref(chains[st['i']%n_rot].desc), s)); st['i']+=1
    for _ in range(10): launch()
    torch.cuda.synchronize()
    # device time: 10 passes over the rotation replayed from a HIP graph (eager python launches are host-bound at ~9 us)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            sg = torch.cuda.current_stream().cuda_stream
            for i in range(10 * n_rot):
                capi.check(lib.cvgs_execute(C.byref(chains[i % n_rot].desc), sg))
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): g.replay()
        e1.record(); torch.cuda.synchronize()
    res = {"case":name,"kernel":cvgs.kernel_name(*ops, flags=flags),"us":round(e0.elapsed_time(e1)*1e3/(80*n_rot),2)}
    del chains, keep, g
    torch.cuda.empty_cache()
    return res
def run_all():
    return [run("1080p NV12 -> 640x640 stretch -> RGB normalize -> NCHW", 1920,1080,(640,640), cvgs.IGNORE_AR),
            run("1080p NV12 -> 640x640 letterbox -> RGB normalize -> NCHW", 1920,1080,(640,640), cvgs.PRESERVE_AR),
            run("4K NV12 -> 640x640 stretch -> RGB normalize -> NCHW", 3840,2160,(640,640), cvgs.IGNORE_AR),
            run("4K NV12 -> 640x640 letterbox -> RGB normalize -> NCHW", 3840,2160,(640,640), cvgs.PRESERVE_AR),
            run("4K NV12 -> 1920x1080 BGR u8 image", 3840,2160,(1920,1080), cvgs.IGNORE_AR, u8_image=True),
            run("1080p NV12 -> 640x360 BGR u8 image", 1920,1080,(640,360), cvgs.IGNORE_AR, u8_image=True),
            run("reference chain: 1080p NV12 -> 640x360 BGRA u8 (float4 -> SaturateCast -> VectorReorder -> write)", 1920,1080,(640,360), cvgs.IGNORE_AR, ref_chain=True),
            run("reference chain: 4K NV12 -> 1920x1080 BGRA u8", 3840,2160,(1920,1080), cvgs.IGNORE_AR, ref_chain=True),
            run("reference chain, interpreted kernel: 1080p NV12 -> 640x360 BGRA u8", 1920,1080,(640,360), cvgs.IGNORE_AR, ref_chain=True, flags=capi.CHAIN_FORCE_GENERIC),
            run("4K NV12 -> 1920x1080 packed fp32 RGB image (x 1/255)", 3840,2160,(1920,1080), cvgs.IGNORE_AR, f32_image=True),
            run("1080p NV12 -> 640x360 packed fp32 RGB image (x 1/255)", 1920,1080,(640,360), cvgs.IGNORE_AR, f32_image=True),
            # planar chroma (software decoders' yuv420p) on the same kernel, and what the interpreted kernel took for it before
            run("1080p I420 -> 640x640 letterbox -> RGB normalize -> NCHW", 1920,1080,(640,640), cvgs.PRESERVE_AR, layout=capi.YUV_I420),
            run("4K I420 -> 640x640 letterbox -> RGB normalize -> NCHW", 3840,2160,(640,640), cvgs.PRESERVE_AR, layout=capi.YUV_I420),
            run("4K YV12 -> 1920x1080 BGR u8 image", 3840,2160,(1920,1080), cvgs.IGNORE_AR, u8_image=True, layout=capi.YUV_YV12),
            run("6K I420 -> 1280x720 stretch -> RGB normalize -> NCHW", 6144,3160,(1280,720), cvgs.IGNORE_AR, n_rot=18, layout=capi.YUV_I420),
            run("6K NV12 -> 1280x720 stretch -> RGB normalize -> NCHW", 6144,3160,(1280,720), cvgs.IGNORE_AR, n_rot=18),
            run("1080p I420 -> 640x640 letterbox, interpreted kernel (CVGS_CHAIN_FORCE_GENERIC)", 1920,1080,(640,640), cvgs.PRESERVE_AR, flags=capi.CHAIN_FORCE_GENERIC, layout=capi.YUV_I420),
            run("6K I420 -> 1280x720, interpreted kernel (CVGS_CHAIN_FORCE_GENERIC)", 6144,3160,(1280,720), cvgs.IGNORE_AR, n_rot=18, flags=capi.CHAIN_FORCE_GENERIC, layout=capi.YUV_I420),
            run("1080p NV12 -> 640x640 letterbox, interpreted kernel (CVGS_CHAIN_FORCE_GENERIC)", 1920,1080,(640,640), cvgs.PRESERVE_AR, flags=capi.CHAIN_FORCE_GENERIC)]


if __name__ == "__main__":
    for r in run_all():
        print(json.dumps(r))
