#!/usr/bin/env python3
"""Generate tests/golden/reference_kats.json: the known-answer vectors the reference's own tests hold for
the hot path.

The reference (/root/reference) cannot be built or run in this image (needs nvcc + OpenCV-CUDA + the
un-vendored FusedKernelLibrary), and it has no Python.  Its tests, however, feed CONSTANT-colour images
through fixed pointwise tables and require the fused result to equal OpenCV's within 1e-4 (float) or
exactly (integer) -- tests/testsCommon.cuh:36-61.  For a constant image the expected output is therefore a
closed-form constant per channel, which this script evaluates in float64 from the tables below.  The tables
are DATA transcribed from the cited test files (initial colours, alpha, subtract/divide scalars, crop and
target sizes); no reference source is copied.  Every case carries its citation.

Run:  python tests/golden/make_reference_kats.py   (rewrites reference_kats.json next to it)
"""
import json
import os

ALPHA = 0.3
# tests/batchresize/test_batchresize_x_split3D.cu:56-67,241-252 (per channel count)
K1_INIT = {1: [2], 2: [2, 37], 3: [5, 5, 5], 4: [2, 37, 128, 20]}
SUB = {1: [1.0], 2: [1.0, 4.0], 3: [1.0, 4.0, 3.2], 4: [1.0, 4.0, 3.2, 0.5]}
DIV = {1: [3.2], 2: [3.2, 0.6], 3: [3.2, 0.6, 11.8], 4: [3.2, 0.6, 11.8, 33.0]}
# tests/resize/test_resize_x_split.cu:36-41, tests/read/test_read_x_write.cu:38-43
K2_INIT = {1: [2], 2: [2, 37], 3: [2, 37, 128], 4: [2, 37, 128, 20]}
# tests/batchread/test_batchread_x_write3D.cu:49-53
K5_INIT = {1: [10], 2: [10, 20], 3: [10, 20, 30], 4: [10, 20, 30, 40]}
K5_SUB = {1: [1.5], 2: [1.5, 4.0], 3: [1.5, 4.0, 3.2], 4: [1.5, 4.0, 3.2, 0.5]}

F32 = lambda v: float(__import__("numpy").float32(v))  # cv::Scalar(double) -> float narrowing of the operand


def swap02(v):
    v = list(v)
    v[0], v[2] = v[2], v[0]
    return v


def chain_const(vals, ops):
    """Evaluate a pointwise op list on a per-channel constant in float64 (tolerance absorbs fp32 rounding)."""
    v = [float(x) for x in vals]
    for op, arg in ops:
        if op == "swap":
            v = swap02(v)
        elif op == "mul":
            v = [a * F32(b) for a, b in zip(v, arg)]
        elif op == "sub":
            v = [a - F32(b) for a, b in zip(v, arg)]
        elif op == "div":
            v = [a / F32(b) for a, b in zip(v, arg)]
        elif op == "add":
            v = [a + F32(b) for a, b in zip(v, arg)]
    return v


RANGE = {"8U": (0, 255), "8S": (-128, 127), "16U": (0, 65535), "16S": (-32768, 32767)}


def sat(init, depth):
    """GpuMat(rows, cols, type, Scalar) saturates the scalar to the type (e.g. 128 -> 127 for CV_8S)."""
    if depth not in RANGE:
        return list(init)
    lo, hi = RANGE[depth]
    return [min(max(v, lo), hi) for v in init]


cases = []


def k1_case(depth, cn, ar=False):
    t = "%sC%d" % (depth, cn)
    ops = []
    if cn in (3, 4):
        ops.append(["cvtColor", "RGB2BGR" if cn == 3 else "RGBA2BGRA"])
    ops += [["multiply", [ALPHA] * cn], ["subtract", SUB[cn]], ["divide", DIV[cn]]]
    pw = ([("swap", None)] if cn in (3, 4) else []) + [("mul", [ALPHA] * cn), ("sub", SUB[cn]), ("div", DIV[cn])]
    case = {
        "name": "k1%s_%s" % ("_ar" if ar else "", t), "src_type": t, "frame": [3840, 2160], "init": K1_INIT[cn],
        "read": {"kind": "resize_batch", "batch": 50, "crop_wh": [30, 120] if ar else [60, 120], "dst": [64, 128],
                 "ar": "PRESERVE_AR" if ar else "IGNORE_AR", "background": [128.0] * cn if ar else None},
        "ops": ops, "write": "tensor_split", "out_type": "32FC%d" % cn, "tol": 1e-4,
        "expected": chain_const(K1_INIT[cn], pw),
        "source": ("tests/batchresize/test_batchresize_aspectratio_x_split3D.cu:57-78,80-95,151-170" if ar else
                   "tests/batchresize/test_batchresize_x_split3D.cu:241-265,311-323,337-355"),
    }
    if ar:
        # 30x120 -> scale 128/120 -> 32x128 centred: columns 16..47 carry the image, the rest the background
        case["window"] = [16, 0, 47, 127]
        case["expected_outside"] = chain_const([128.0] * cn, pw)
    cases.append(case)


for d in ("8U", "16U", "16S"):
    for c in (3, 4):
        k1_case(d, c)
        k1_case(d, c, ar=True)

# K2: single resize of a ROI -> mul -> sub -> div -> split into C separate planes (no channel swap)
for d in ("8U", "16U", "16S"):
    for c in (3, 4):
        cases.append({
            "name": "k2_%sC%d" % (d, c), "src_type": "%sC%d" % (d, c), "frame": [3840, 2160], "init": K2_INIT[c],
            "read": {"kind": "resize_single", "roi": [200, 200, 60, 120], "dst": [64, 128]},
            "ops": [["multiply", [ALPHA] * c], ["subtract", SUB[c]], ["divide", DIV[c]]],
            "write": "split_planes", "out_type": "32FC%d" % c, "tol": 1e-4,
            "expected": chain_const(K2_INIT[c], [("mul", [ALPHA] * c), ("sub", SUB[c]), ("div", DIV[c])]),
            "source": "tests/resize/test_resize_x_split.cu:36-52,79-84,96-99"})

# K3: resize up (3870x2260) and down (300x500) of a constant 4K image, saturate back to the input type
for t, c in (("8U", 1), ("8U", 3), ("8U", 4), ("16U", 1), ("16U", 3), ("16U", 4), ("16S", 1), ("16S", 3), ("16S", 4),
             ("32F", 1)):
    for dst in ([3870, 2260], [300, 500]):
        cases.append({
            "name": "k3_%sC%d_%dx%d" % (t, c, dst[0], dst[1]), "src_type": "%sC%d" % (t, c), "frame": [3840, 2160],
            "init": K2_INIT[c], "read": {"kind": "resize_single", "roi": [0, 0, 3840, 2160], "dst": dst},
            "ops": [["convertTo", "%sC%d" % (t, c)]], "write": "write2d", "out_type": "%sC%d" % (t, c),
            "tol": 1e-4 if t == "32F" else 0, "expected": [float(v) for v in K2_INIT[c]],
            "source": "tests/resize/test_resize_write.cu:31-72"})

# K5: batch read -> convertTo(alpha=1) -> sub -> div -> write3D  (thread fusion off)
K5_PAIRS = [("8U", 1), ("8S", 1), ("16U", 1), ("16S", 1), ("32S", 1), ("32F", 1), ("8U", 2), ("8U", 3), ("8U", 4),
            ("8S", 2), ("8S", 3), ("8S", 4), ("16U", 2), ("16U", 3), ("16U", 4), ("16S", 2), ("16S", 3), ("16S", 4),
            ("32S", 2), ("32S", 3), ("32S", 4)]
for d, c in K5_PAIRS:
    cases.append({
        "name": "k5_%sC%d" % (d, c), "src_type": "%sC%d" % (d, c), "frame": [60, 120], "init": sat(K5_INIT[c], d),
        "read": {"kind": "pixel_batch", "batch": 50},
        "ops": [["convertTo_alpha", "32FC%d" % c, 1.0], ["subtract", K5_SUB[c]], ["divide", DIV[c]]],
        "write": "write3d", "out_type": "32FC%d" % c, "tol": 1e-4, "no_thread_fusion": True,
        "expected": chain_const(sat(K5_INIT[c], d), [("mul", [1.0] * c), ("sub", K5_SUB[c]), ("div", DIV[c])]),
        "source": "tests/batchread/test_batchread_x_write3D.cu:45-58,60-96,202-224"})

# K5 on CV_64F outputs (test_batchread_x_write3D.cu:208-209,222-224): arithmetic in double
for d, c, o in (("32F", 1, "64F"), ("64F", 1, "64F"), ("32F", 2, "64F"), ("32F", 3, "64F"), ("32F", 4, "64F")):
    cases.append({
        "name": "k5_%sC%d_to_%s" % (d, c, o), "src_type": "%sC%d" % (d, c), "frame": [60, 120], "init": K5_INIT[c],
        "read": {"kind": "pixel_batch", "batch": 50},
        "ops": [["convertTo_alpha", "%sC%d" % (o, c), 1.0], ["subtract", K5_SUB[c]], ["divide", DIV[c]]],
        "write": "write3d", "out_type": "%sC%d" % (o, c), "tol": 1e-4, "no_thread_fusion": True,
        "expected": chain_const(K5_INIT[c], [("mul", [1.0] * c), ("sub", K5_SUB[c]), ("div", DIV[c])]),
        "source": "tests/batchread/test_batchread_x_write3D.cu:45-58,60-96,208-209,222-224"})

# K6 / K7 on CV_32F -> CV_64F (test_read_x_write.cu:139-141, test_read_x_split.cu LAUNCH list)
for c in (2, 3, 4):
    sub = [0.3] * c
    cases.append({
        "name": "k6_32FC%d_to_64F" % c, "src_type": "32FC%d" % c, "frame": [3840, 2160], "init": K2_INIT[c],
        "read": {"kind": "pixel_single"},
        "ops": [["convertTo", "64FC%d" % c], ["subtract", sub], ["multiply", SUB[c]], ["divide", DIV[c]], ["add", DIV[c]]],
        "write": "write2d", "out_type": "64FC%d" % c, "tol": 1e-4,
        "expected": chain_const(K2_INIT[c], [("sub", sub), ("mul", SUB[c]), ("div", DIV[c]), ("add", DIV[c])]),
        "source": "tests/read/test_read_x_write.cu:31-73,139-141"})
    cases.append({
        "name": "k7_32FC%d_to_64F" % c, "src_type": "32FC%d" % c, "frame": [3840, 2160], "init": K2_INIT[c],
        "read": {"kind": "pixel_single"}, "ops": [["convertTo", "64FC%d" % c]], "write": "split_planes",
        "out_type": "64FC%d" % c, "tol": 1e-4, "expected": [float(v) for v in K2_INIT[c]],
        "source": "tests/read/test_read_x_split.cu:30-59"})

# K6: read -> convertTo -> sub(0.3) -> mul -> div -> add(= the divide scalar) -> write 2D, 4K image
for d, c in K5_PAIRS:
    sub = [0.3] * c
    mul = SUB[c]  # the table's third column is used as the multiplier (test_read_x_write.cu:38-50)
    cases.append({
        "name": "k6_%sC%d" % (d, c), "src_type": "%sC%d" % (d, c), "frame": [3840, 2160], "init": sat(K2_INIT[c], d),
        "read": {"kind": "pixel_single"},
        "ops": [["convertTo", "32FC%d" % c], ["subtract", sub], ["multiply", mul], ["divide", DIV[c]], ["add", DIV[c]]],
        "write": "write2d", "out_type": "32FC%d" % c, "tol": 1e-4,
        "expected": chain_const(sat(K2_INIT[c], d), [("sub", sub), ("mul", mul), ("div", DIV[c]), ("add", DIV[c])]),
        "source": "tests/read/test_read_x_write.cu:31-73,121-141"})

# K7: read -> convertTo -> split into planes
for d, c in [p for p in K5_PAIRS if p[1] >= 2]:
    cases.append({
        "name": "k7_%sC%d" % (d, c), "src_type": "%sC%d" % (d, c), "frame": [3840, 2160], "init": sat(K2_INIT[c], d),
        "read": {"kind": "pixel_single"}, "ops": [["convertTo", "32FC%d" % c]], "write": "split_planes",
        "out_type": "32FC%d" % c, "tol": 1e-4, "expected": [float(v) for v in sat(K2_INIT[c], d)],
        "source": "tests/read/test_read_x_split.cu:30-59"})

# convertTo KATs (OpenCV semantics: integral outputs round to nearest even)
cases += [
    {"name": "convertTo_8UC1_32FC1", "src_type": "8UC1", "frame": [16, 16], "init": [20], "read": {"kind": "pixel_single"},
     "ops": [["convertTo", "32FC1"]], "write": "write2d", "out_type": "32FC1", "tol": 0, "expected": [20.0],
     "source": "tests/single_operation/test_convertTo.cu:47,60-61"},
    {"name": "convertTo_8UC3_32FC3_ab", "src_type": "8UC3", "frame": [16, 16], "init": [20, 30, 40],
     "read": {"kind": "pixel_single"}, "ops": [["convertTo_alpha_beta", "32FC3", 0.5, 0.5]], "write": "write2d",
     "out_type": "32FC3", "tol": 0, "expected": [10.5, 15.5, 20.5],
     "source": "tests/single_operation/test_convertTo.cu:51,66-67"},
    {"name": "convertTo_8UC4_32SC4_ab", "src_type": "8UC4", "frame": [16, 16], "init": [20, 30, 40, 50],
     "read": {"kind": "pixel_single"}, "ops": [["convertTo_alpha_beta", "32SC4", 0.5, 0.5]], "write": "write2d",
     "out_type": "32SC4", "tol": 0, "expected": [10, 16, 20, 26],
     "source": "tests/single_operation/test_convertTo.cu:53,69-70 (cv::GpuMat::convertTo rounds half to even)"},
    {"name": "convertTo_8UC4_32SC4_a", "src_type": "8UC4", "frame": [16, 16], "init": [20, 30, 40, 50],
     "read": {"kind": "pixel_single"}, "ops": [["convertTo_alpha", "32SC4", 0.5]], "write": "write2d",
     "out_type": "32SC4", "tol": 0, "expected": [10, 15, 20, 25],
     "source": "tests/single_operation/test_convertTo.cu:55,72-73"},
]

# split KAT
cases.append({"name": "split_8UC3", "src_type": "8UC3", "frame": [16, 16], "init": [1, 2, 3],
              "read": {"kind": "pixel_single"}, "ops": [], "write": "split_planes", "out_type": "8UC3", "tol": 0,
              "expected": [1, 2, 3], "source": "tests/unit_tests/test_split.cu:21-27,64-90"})
cases.append({"name": "split_batch_8UC3", "src_type": "8UC3", "frame": [16, 16], "init": [1, 2, 3],
              "read": {"kind": "pixel_batch", "batch": 10}, "ops": [], "write": "split_planes_batch", "out_type": "8UC3",
              "tol": 0, "expected": [1, 2, 3], "source": "tests/unit_tests/test_split.cu:47-62"})

# cvtColor KATs: the 16 cases of tests/color/test_cvtColor.cu:105-123 (8U and 16U)
CVT_INIT = {3: [10, 100, 200], 4: [1, 2, 3, 4]}


def gray(r, g, b):
    return float(round(0.299 * r + 0.587 * g + 0.114 * b))


for d in ("8U", "16U"):
    for code, icn, ocn, exp in (
            ("RGB2BGR", 3, 3, [200, 100, 10]), ("RGBA2BGRA", 4, 4, [3, 2, 1, 4]),
            ("BGR2RGB", 3, 3, [200, 100, 10]), ("BGRA2RGBA", 4, 4, [3, 2, 1, 4]),
            ("RGB2GRAY", 3, 1, [gray(10, 100, 200)]), ("RGBA2GRAY", 4, 1, [gray(1, 2, 3)]),
            ("BGR2GRAY", 3, 1, [gray(200, 100, 10)]), ("BGRA2GRAY", 4, 1, [gray(3, 2, 1)])):
        cases.append({"name": "cvtColor_%s_%s" % (code, d), "src_type": "%sC%d" % (d, icn), "frame": [3840, 2160],
                      "init": CVT_INIT[icn], "read": {"kind": "pixel_single"}, "ops": [["cvtColor", code, "%sC%d" % (d, ocn)]],
                      "write": "write2d", "out_type": "%sC%d" % (d, ocn), "tol": 0, "expected": [float(v) for v in exp],
                      "no_thread_fusion": True, "source": "tests/color/test_cvtColor.cu:31-56,105-123"})

# CircularTensor ordering KATs: after ITERS updates with value i+1, slot z holds ITERS - age(z)
for name, order, mode, write, in_t, elem, cp in (
        ("circular_newest_first", "NewestFirst", "Standard", "tensor_split", "8UC3", "32FC1", 3),
        ("circular_transposed_newest_first", "NewestFirst", "Transposed", "tensor_t_split", "8UC3", "32FC1", 3),
        ("circular_transposed_oldest_first", "OldestFirst", "Transposed", "tensor_t_split", "8UC3", "32FC1", 3),
        ("circular_oldest_first_packed", "OldestFirst", "Standard", "tensor_write", "8UC4", "32FC4", 1)):
    cases.append({"name": name, "kind": "circular", "in_type": in_t, "elem_type": elem, "color_planes": cp, "batch": 15,
                  "width": 128, "height": 128, "iters": 100, "order": order, "mode": mode, "write": write,
                  "expected_slot": [100 - z if order == "NewestFirst" else 100 - (14 - z) for z in range(15)],
                  "source": "tests/batchread/test_circularbatchread_x_write3D.cu:224-282,284-340,342-398,400-460"})

# CircularBatchRead<Ascendent>: out[z] = in[(z + first) mod BATCH], first = 4, BATCH = 15, value of plane i = i
cases.append({"name": "circular_batch_read", "kind": "circular_batch_read", "batch": 15, "first": 4, "width": 32,
              "height": 32, "expected_plane": [(z + 4) % 15 for z in range(15)],
              "source": "tests/batchread/test_circularbatchread_x_write3D.cu:24-87"})

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
with open(out, "w") as f:
    json.dump({"comment": "generated by make_reference_kats.py from the reference tests' parameter tables; see its docstring",
               "cases": cases}, f, indent=1)
print("wrote %d cases to %s" % (len(cases), out))
