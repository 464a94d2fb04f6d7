"""Runs one known-answer case of tests/golden/reference_kats.json on a backend:
'oracle' (host memory, CPU oracle) or 'gpu' (torch device memory, HIP library through the C-ABI)."""
import json
import os

import numpy as np

from cvgpuspeedup_amd import capi, cvgs

KAT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")
NP_DEPTH = {"8U": np.uint8, "8S": np.int8, "16U": np.uint16, "16S": np.int16, "32S": np.int32, "32F": np.float32,
            "64F": np.float64, "16F": np.float16}
CV_DEPTH = {"8U": cvgs.CV_8U, "8S": cvgs.CV_8S, "16U": cvgs.CV_16U, "16S": cvgs.CV_16S, "32S": cvgs.CV_32S,
            "32F": cvgs.CV_32F, "64F": cvgs.CV_64F, "16F": cvgs.CV_16F}
CODES = {"RGB2BGR": cvgs.COLOR_RGB2BGR, "BGR2RGB": cvgs.COLOR_BGR2RGB, "RGBA2BGRA": cvgs.COLOR_RGBA2BGRA,
         "BGRA2RGBA": cvgs.COLOR_BGRA2RGBA, "RGB2GRAY": cvgs.COLOR_RGB2GRAY, "RGBA2GRAY": cvgs.COLOR_RGBA2GRAY,
         "BGR2GRAY": cvgs.COLOR_BGR2GRAY, "BGRA2GRAY": cvgs.COLOR_BGRA2GRAY}
AR = {"IGNORE_AR": cvgs.IGNORE_AR, "PRESERVE_AR": cvgs.PRESERVE_AR}


def load_cases():
    with open(KAT_PATH) as f:
        return json.load(f)["cases"]


def parse_type(s):
    d, c = s.split("C")
    return d, int(c), cvgs.make_type(CV_DEPTH[d], int(c))


class Mem:
    """Allocates typed 2D arrays on the chosen backend and wraps them as GpuMat."""

    def __init__(self, backend):
        self.backend = backend
        if backend == "gpu":
            import torch
            self.torch = torch
            self.dev = torch.device("cuda:0")

    def alloc(self, rows, cols, type_str, fill=None):
        d, c, cvt = parse_type(type_str)
        a = np.zeros((rows, cols, c), NP_DEPTH[d])
        if fill is not None:
            a[...] = np.asarray(fill, dtype=NP_DEPTH[d])
        if self.backend == "gpu":
            t = self.torch.from_numpy(a).to(self.dev)
            return cvgs.GpuMat.from_tensor(t, cvt), (lambda: t.cpu().numpy())
        return cvgs.GpuMat.from_array(a, cvt), (lambda: a)

    def run(self, iops, flags=0):
        if self.backend == "gpu":
            cvgs.executeOperations(self.torch.cuda.current_stream(), *iops, flags=flags)
            self.torch.cuda.synchronize()
        else:
            from oracle import oracle_binding
            oracle_binding.execute(cvgs.lower(iops, flags))


def build_ops(case, cur_type):
    ops = []
    for op in case["ops"]:
        kind = op[0]
        if kind == "cvtColor":
            out_t = parse_type(op[2])[2] if len(op) > 2 else None
            code = CODES[op[1]]
            d, c, _ = (cvgs.type_depth(cur_type), cvgs.type_cn(cur_type), None)
            iop = cvgs.cvtColor(code, cur_type, out_t)
        elif kind == "convertTo":
            iop = cvgs.convertTo(cur_type, parse_type(op[1])[2])
        elif kind == "convertTo_alpha":
            iop = cvgs.convertTo(cur_type, parse_type(op[1])[2], op[2])
        elif kind == "convertTo_alpha_beta":
            iop = cvgs.convertTo(cur_type, parse_type(op[1])[2], op[2], op[3])
        else:
            iop = {"multiply": cvgs.multiply, "subtract": cvgs.subtract, "divide": cvgs.divide, "add": cvgs.add}[kind](
                cur_type, op[1])
        ops.append(iop)
        cur_type = iop.out_type
    return ops, cur_type


def run_chain_case(case, backend, flags=0, batch_limit=None):
    """Returns (outputs, layout): outputs is a numpy array [batch, channels, H, W] of the written values."""
    mem = Mem(backend)
    fw, fh = case["frame"]
    sd, scn, stype = parse_type(case["src_type"])
    od, ocn, otype = parse_type(case["out_type"])
    rd = case["read"]
    if case.get("no_thread_fusion"):
        flags |= capi.CHAIN_NO_THREAD_FUSION
    batch = rd.get("batch", 1)
    if batch_limit:
        batch = min(batch, batch_limit)
    if rd["kind"] == "pixel_batch":
        srcs = [mem.alloc(fh, fw, case["src_type"], case["init"])[0] for _ in range(batch)]
        read = cvgs.ReadIOp(capi.READ_PIXEL, stype, srcs, batch)
        W, H = fw, fh
    else:
        frame, _ = mem.alloc(fh, fw, case["src_type"], case["init"])
        if rd["kind"] == "pixel_single":
            read = cvgs.ReadIOp(capi.READ_PIXEL, stype, [frame], 1)
            W, H = fw, fh
        elif rd["kind"] == "resize_single":
            x, y, w, h = rd["roi"]
            read = cvgs.resize(stype, cvgs.INTER_LINEAR, frame.roi(x, y, w, h), rd["dst"], fx=0.0, fy=0.0)
            W, H = rd["dst"]
        else:  # resize_batch: crops of crop_wh at (i,i)
            cw, chh = rd["crop_wh"]
            mats = [frame.roi(i, i, cw, chh) for i in range(batch)]
            read = cvgs.resize(stype, cvgs.INTER_LINEAR, mats, rd["dst"], batch, rd.get("background"), AR[rd["ar"]])
            W, H = rd["dst"]
    ops, cur = build_ops(case, read.out_type())
    assert cur == otype, (case["name"], cur, otype)
    wk = case["write"]
    if wk == "tensor_split":
        out, fetch = mem.alloc(batch, W * H * ocn, od + "C1", None)
        write = cvgs.split(otype, out, (W, H))
        get = lambda: fetch().reshape(batch, ocn, H, W)
    elif wk == "write3d":
        out, fetch = mem.alloc(batch, W * H, case["out_type"], None)
        write = cvgs.write(otype, out, (W, H))
        get = lambda: fetch().reshape(batch, H, W, ocn).transpose(0, 3, 1, 2)
    elif wk == "write2d":
        out, fetch = mem.alloc(H, W, case["out_type"], None)
        write = cvgs.write(otype, out)
        get = lambda: fetch().reshape(1, H, W, ocn).transpose(0, 3, 1, 2)
    else:  # split_planes / split_planes_batch
        planes = [[mem.alloc(H, W, od + "C1", None) for _ in range(ocn)] for _ in range(batch)]
        mats = [[p[0] for p in img] for img in planes]
        write = cvgs.split(otype, mats if wk == "split_planes_batch" else mats[0])
        get = lambda: np.stack([np.stack([p[1]().reshape(H, W) for p in img]) for img in planes])
    mem.run([read] + ops + [write], flags)
    return get()


def check_chain_case(case, out):
    """Reference tolerance: float |diff| <= 1e-4, integer exact (tests/testsCommon.cuh:36-61)."""
    exp = np.asarray(case["expected"], np.float64)
    tol = case["tol"]
    full = np.empty(out.shape, np.float64)
    full[...] = exp[None, :, None, None]
    if "window" in case:
        x1, y1, x2, y2 = case["window"]
        outside = np.asarray(case["expected_outside"], np.float64)
        full[...] = outside[None, :, None, None]
        full[:, :, y1:y2 + 1, x1:x2 + 1] = exp[None, :, None, None]
    err = np.abs(out.astype(np.float64) - full)
    assert err.max() <= tol, "%s: max |diff| %g > %g (at %s)" % (case["name"], err.max(), tol,
                                                                   np.unravel_index(err.argmax(), err.shape))
