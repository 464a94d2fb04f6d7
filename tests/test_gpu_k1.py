"""K1 parity on the GPU: N crops -> bilinear resize -> cvtColor/mul/sub/div -> planar fp32 tensor, through the
C-ABI, bit-exact against the CPU oracle.  Mirrors reference tests/batchresize/test_batchresize_x_split3D.cu and
test_batchresize_aspectratio_x_split3D.cu, but on NON-constant images (the reference only uses constant colours)."""
import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

pytestmark = pytest.mark.gpu


def run_both(oracle, frame_np, crops, batch_out, dst=(64, 128), cn=3, flags=0, **kw):
    import torch
    dev = torch.device("cuda:0")
    src_type = cvgs.make_type(cvgs.CV_8U, cn)
    frame_t = torch.from_numpy(frame_np).to(dev)
    out_t = torch.full((batch_out, cn * dst[0] * dst[1]), -777.0, dtype=torch.float32, device=dev)
    g_src = cvgs.GpuMat.from_tensor(frame_t, src_type)
    g_out = cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1)
    ops = H.k1_chain(g_src, crops, g_out, dst, cn, **kw)
    name = cvgs.kernel_name(*ops, flags=flags)
    cvgs.executeOperations(torch.cuda.current_stream(), *ops, flags=flags)
    torch.cuda.synchronize()
    gpu = out_t.cpu().numpy()

    ref = np.full((batch_out, cn * dst[0] * dst[1]), -777.0, dtype=np.float32)
    h_src = cvgs.GpuMat.from_array(frame_np, src_type)
    h_out = cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1)
    oracle.execute(cvgs.lower(H.k1_chain(h_src, crops, h_out, dst, cn, **kw)))
    return gpu, ref, name


@pytest.mark.parametrize("batch", [10, 50])
def test_k1_reference_fixed_crops(oracle, batch):
    """cfg #2(a): 4K frame, 60x120 crops at (i,i) -> 64x128, kernel-argument descriptors."""
    frame = H.random_u8((2160, 3840, 3))
    gpu, ref, name = run_both(oracle, frame, H.fixed_crops(batch), batch)
    assert name.startswith("k1_u8c3_swap_mul_sub_div"), name
    H.assert_bit_exact(gpu, ref, "K1 fixed crops")


def test_k1_variable_crops(oracle):
    """cfg #2(b): 50 variable-size crops (down- and up-scaling in both axes)."""
    frame = H.random_u8((2160, 3840, 3))
    crops = H.random_crops(50, 3840, 2160)
    gpu, ref, _ = run_both(oracle, frame, crops, 50)
    H.assert_bit_exact(gpu, ref, "K1 variable crops")


def test_k1_generic_kernel_agrees(oracle):
    """The interpreted kernel and the specialised kernel give identical bits."""
    frame = H.random_u8((1080, 1920, 3), seed=7)
    crops = H.random_crops(12, 1920, 1080, seed=9)
    gpu_fast, ref, n1 = run_both(oracle, frame, crops, 12)
    gpu_gen, _, n2 = run_both(oracle, frame, crops, 12, flags=cvgs.capi.CHAIN_FORCE_GENERIC)
    assert n1 != n2 and n2.startswith("generic")
    H.assert_bit_exact(gpu_gen, ref, "generic kernel")
    H.assert_bit_exact(gpu_fast, gpu_gen, "fast vs generic")


def test_k1_four_channels(oracle):
    """K1-C4: uchar4 -> RGBA2BGRA -> float4 (reference :315-319)."""
    frame = H.random_u8((720, 1280, 4), seed=11)
    crops = H.random_crops(20, 1280, 720, seed=12, wmax=300, hmax=600)
    gpu, ref, name = run_both(oracle, frame, crops, 20, cn=4)
    assert name.startswith("k1_u8c4")
    H.assert_bit_exact(gpu, ref, "K1 u8c4")


@pytest.mark.parametrize("ar", [cvgs.PRESERVE_AR, cvgs.PRESERVE_AR_RN_EVEN, cvgs.PRESERVE_AR_LEFT])
def test_k1_aspect_ratio(oracle, ar):
    """K1-AR: padding with the background value pushed through the chain (reference
    test_batchresize_aspectratio_x_split3D.cu:151-157), incl. the tested 30x120 crop."""
    frame = H.random_u8((1080, 1920, 3), seed=21)
    crops = [(i, i, 30, 120) for i in range(8)] + H.random_crops(16, 1920, 1080, seed=22)
    gpu, ref, _ = run_both(oracle, frame, crops, 24, ar=ar, background=[128.0] * 3)
    H.assert_bit_exact(gpu, ref, "K1 AR mode %d" % ar)


def test_k1_used_planes_default_value(oracle):
    """usedPlanes < N: the remaining planes carry the default value pushed through the chain."""
    frame = H.random_u8((480, 640, 3), seed=31)
    crops = H.random_crops(10, 640, 480, seed=32, wmax=200, hmax=300)
    gpu, ref, _ = run_both(oracle, frame, crops, 10, used=6, background=[7.0, 8.0, 9.0])
    H.assert_bit_exact(gpu, ref, "K1 usedPlanes")
    t = gpu.reshape(10, 3, 128, 64)
    assert np.all(t[6:, 0] == t[6, 0, 0, 0])


def test_k1_edge_geometries(oracle):
    """1- and 2-pixel-wide crops (byte-gather path), crops touching the last row/column of the frame, a target
    that is not a multiple of the 64-lane tile."""
    frame = H.random_u8((96, 160, 3), seed=41)
    crops = [(0, 0, 1, 1), (159, 95, 1, 1), (158, 0, 2, 96), (0, 94, 160, 2), (100, 50, 60, 46), (0, 0, 160, 96),
             (157, 93, 3, 3)]
    for dst in [(64, 128), (100, 37), (7, 5)]:
        gpu, ref, _ = run_both(oracle, frame, crops, len(crops), dst=dst)
        H.assert_bit_exact(gpu, ref, "K1 edges dst=%s" % (dst,))


def test_k1_device_plane_table(oracle):
    """Large batches: descriptors in a device-resident plane table instead of kernel arguments."""
    import torch
    frame = H.random_u8((1080, 1920, 3), seed=51)
    crops = H.random_crops(300, 1920, 1080, seed=52, wmax=256, hmax=256)
    # library-managed upload (batch > CVGS_KERNARG_PLANES)
    gpu, ref, name = run_both(oracle, frame, crops, 300)
    H.assert_bit_exact(gpu, ref, "K1 batch 300 (auto table)")
    # caller-managed resident table
    dev = torch.device("cuda:0")
    frame_t = torch.from_numpy(frame).to(dev)
    g_src = cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3)
    out_t = torch.zeros((300, 3 * 64 * 128), dtype=torch.float32, device=dev)
    g_out = cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1)
    ops = H.k1_chain(g_src, crops, g_out)
    table = torch.frombuffer(bytearray(cvgs.build_plane_table(ops[0])), dtype=torch.uint8).to(dev)
    ops2 = H.k1_chain(g_src, crops, g_out, table=table.data_ptr())
    cvgs.executeOperations(torch.cuda.current_stream(), *ops2)
    torch.cuda.synchronize()
    H.assert_bit_exact(out_t.cpu().numpy(), ref, "K1 batch 300 (resident table)")


def test_k1_no_used_planes(oracle):
    """usedPlanes == 0 ("empty" crop list): every plane is the default value pushed through the chain."""
    frame = H.random_u8((64, 64, 3), seed=61)
    gpu, ref, _ = run_both(oracle, frame, [(0, 0, 8, 8)] * 4, 4, used=0, background=[10.0, 20.0, 30.0])
    H.assert_bit_exact(gpu, ref, "usedPlanes == 0")
    t = gpu.reshape(4, 3, 128, 64)
    assert (t[:, 0] == t[0, 0, 0, 0]).all() and (t[:, 2] == t[0, 2, 0, 0]).all()


def test_k1_large_batch_and_large_crops(oracle):
    """2000 crops in one launch (device table), including whole-frame crops (strong downscale, scale > 21)."""
    frame = H.random_u8((1080, 1920, 3), seed=71)
    crops = H.random_crops(1996, 1920, 1080, seed=72, wmax=1920, hmax=1080) + [(0, 0, 1920, 1080), (0, 0, 1920, 1), (0, 0, 1, 1080),
                                                                                 (1919, 1079, 1, 1)]
    gpu, ref, _ = run_both(oracle, frame, crops, 2000)
    H.assert_bit_exact(gpu, ref, "2000 crops")


def test_k1_stream_capture_guard():
    """Batches that need a library-managed descriptor upload (more than CVGS_KERNARG_PLANES_MAX = 320 planes) cannot be
    captured into a HIP graph (the copy would reference dead host memory): the call must fail loudly; the same batch with a
    resident plane table captures fine, and so does a batch of up to 320 planes with host descriptors (they travel in the
    kernel arguments)."""
    import torch
    dev = torch.device("cuda:0")
    frame_t = torch.from_numpy(H.random_u8((480, 640, 3), seed=81)).to(dev)
    g_src = cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3)
    for n, host_descriptors_capturable in ((100, True), (320, True), (400, False)):
        crops = H.random_crops(n, 640, 480, seed=82, wmax=200, hmax=200)
        out_t = torch.zeros((n, 3 * 64 * 128), dtype=torch.float32, device=dev)
        g_out = cvgs.GpuMat.from_tensor(out_t, cvgs.CV_32FC1)
        ops = H.k1_chain(g_src, crops, g_out)
        cvgs.executeOperations(torch.cuda.current_stream(), *ops)  # eager: fine
        torch.cuda.synchronize()
        eager = out_t.clone()
        assert eager.any()
        table = torch.frombuffer(bytearray(cvgs.build_plane_table(ops[0])), dtype=torch.uint8).to(dev)
        ops_t = H.k1_chain(g_src, crops, g_out, table=table.data_ptr())
        for use_table in (False, True):
            if not use_table and not host_descriptors_capturable:
                continue
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    cvgs.executeOperations(torch.cuda.current_stream(), *(ops_t if use_table else ops))
            out_t.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(out_t, eager), (n, use_table)
        if not host_descriptors_capturable:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    with pytest.raises(cvgs.capi.CvgsError):
                        cvgs.executeOperations(torch.cuda.current_stream(), *ops)
                    cvgs.executeOperations(torch.cuda.current_stream(), *ops_t)  # keep the capture non-empty


@pytest.mark.parametrize("n", [65, 128, 300, 320, 321])
@pytest.mark.parametrize("variant", ["u8c3", "u8c4", "u16c3", "s16c4", "f16out", "interp", "aspect_unused"])
def test_k1_large_kernel_argument_batches(oracle, n, variant):
    """65 .. 320 crops with host descriptors: the planar-tensor K1 kernels take them in a 16 KB kernel-argument block (the
    reference's benchmark sweeps the batch to 300, tests/batchresize/test_batchresize_x_split3D.cu:384-392); 321 goes
    through the staged device table.  Every type pair of the reference's sweep, the fp16 hand-off, an interpreted program,
    aspect-ratio padding with unused planes."""
    import torch
    dev = torch.device("cuda:0")
    cn = 4 if variant in ("u8c4", "s16c4") else 3
    depth = {"u16c3": cvgs.CV_16U, "s16c4": cvgs.CV_16S}.get(variant, cvgs.CV_8U)
    np_dt = {cvgs.CV_8U: np.uint8, cvgs.CV_16U: np.uint16, cvgs.CV_16S: np.int16}[depth]
    raw = H.random_u16((270, 480, cn), seed=300 + n)
    frame = (raw & 0xff).astype(np.uint8) if depth == cvgs.CV_8U else raw.view(np_dt)
    crops = H.random_crops(n, 480, 270, seed=77 + n, wmin=2, wmax=200, hmin=2, hmax=200)
    half = variant == "f16out"
    kw = dict(cn=cn, src_depth=depth, half=half)
    if variant == "aspect_unused":
        kw.update(used=n - 7, ar=cvgs.PRESERVE_AR, background=[3.0, 60.0, 128.0])
    out_np = np.float16 if half else np.float32
    out_t = torch.zeros((n, cn * 64 * 128), dtype=torch.float16 if half else torch.float32, device=dev)
    ref = np.zeros((n, cn * 64 * 128), out_np)
    o_type = cvgs.CV_16FC1 if half else cvgs.CV_32FC1
    ft = torch.from_numpy(frame.view(np.int16) if depth == cvgs.CV_16U else frame).to(dev)
    st = cvgs.make_type(depth, cn)

    def chain(src, out):
        ops = H.k1_chain(src, crops, out, **kw)
        if variant == "interp":  # one more stage: not one of the compile-time programs
            f = cvgs.make_type(cvgs.CV_32F, cn)
            ops = ops[:-1] + [cvgs.add(f, [0.25] * cn), ops[-1]]
        return ops

    ops = chain(cvgs.GpuMat.from_tensor(ft, st), cvgs.GpuMat.from_tensor(out_t, o_type))
    name = cvgs.kernel_name(*ops)
    assert name.startswith("k1_"), name
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    oracle.execute(cvgs.lower(chain(cvgs.GpuMat.from_array(frame, st), cvgs.GpuMat.from_array(ref, o_type))))
    assert ref.any()
    H.assert_bit_exact(out_t.cpu().numpy(), ref, "%d crops, %s via %s" % (n, variant, name))


# ---- division by a wave-uniform divisor (k_taps.hpp div_by_uniform) ---------------------------------------------------
def _div_chain(src, crops, out, mul, sub, div, swap=True):
    f = cvgs.CV_32FC3
    ops = [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, [src.roi(*c) for c in crops], (64, 128), len(crops))]
    if swap:
        ops.append(cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f))
    return ops + [cvgs.multiply(f, mul), cvgs.subtract(f, sub), cvgs.divide(f, div), cvgs.split(f, out, (64, 128))]


@pytest.mark.parametrize("name,mul,sub,div", [
    ("zero dividends, negative zero quotients", [-1.0, -0.5, 2.0], [0.0, 0.0, 0.0], [3.0, 0.7, -11.8]),
    ("integer mean: x == 0 wherever v == sub", [1.0, 1.0, 1.0], [1.0, 4.0, 6.0], [2.0, 8.0, 1.0]),
    ("divisor significand all ones -> IEEE division", [0.3, 0.3, 0.3], [1.0, 4.0, 3.2], [float(np.float32(2.0) - np.float32(2.0 ** -23)), 0.6, 11.8]),
    ("operands outside the guarded range -> IEEE division", [0.3, 0.3, 0.3], [1.0, 4.0, 3.2], [1e30, 0.6, 11.8]),
    ("tiny multiplier outside the guard", [1e-12, 0.3, 0.3], [0.0, 4.0, 3.2], [3.2, 0.6, 11.8]),
    ("negative divisors", [0.3, 0.3, 0.3], [1.0, 4.0, 3.2], [-3.2, -0.6, -11.8]),
    ("hard-to-round divisors", [0.0078125, 1.0, 255.0], [0.5, 127.5, 0.0], [0.229, 58.395, 3.0]),
])
@pytest.mark.parametrize("swap", [True, False])
def test_k1_division_paths_match_the_oracle(oracle, name, mul, sub, div, swap):
    """Every way the DIV stage can run -- FMA-corrected reciprocal (guarded operands, no zero dividend in the wave),
    the wave-level fallback for zero dividends, the host guard's refusals -- gives the IEEE quotient's bits."""
    import torch
    dev = torch.device("cuda:0")
    frame = H.random_u8((600, 800, 3), seed=91)
    frame[100:300, 100:500] = 0          # a black region: zero dividends when sub == 0
    frame[300:400, :, 0] = 1             # v == sub for the integer-mean case
    frame[300:400, :, 1] = 4
    frame[300:400, :, 2] = 6
    crops = H.random_crops(20, 800, 600, seed=92, wmax=400, hmax=500) + [(100, 100, 64, 128), (90, 290, 200, 150)]
    ft = torch.from_numpy(frame).to(dev)
    out = torch.full((len(crops), 3 * 64 * 128), -777.0, dtype=torch.float32, device=dev)
    ops = _div_chain(cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), mul, sub, div, swap)
    assert cvgs.kernel_name(*ops).startswith("k1_u8c3_swap_mul_sub_div" if swap else "k1_u8c3_mul_sub_div")
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    ref = np.full((len(crops), 3 * 64 * 128), -777.0, np.float32)
    oracle.execute(cvgs.lower(_div_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1),
                                         mul, sub, div, swap)))
    H.assert_bit_exact(out.cpu().numpy(), ref, name)
    if "zero" in name:
        assert (ref.view(np.uint32) == 0x80000000).any(), "the case must really produce negative zeros"


# ---- round 6: K1's interpreted ARITHMETIC programs (k_common.hpp: InterpProgT<true>::run_arith) ---------------------------------------------------
def _arith_programs(f, cn):
    sw = cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f) if cn == 3 else None
    k = lambda v: list(v)[:cn]  # noqa: E731
    progs = {
        "norm_then_add": [sw, cvgs.multiply(f, k([0.3] * 4)), cvgs.subtract(f, k([1.0, 4.0, 3.2, 0.5])), cvgs.divide(f, k([3.2, 0.6, 11.8, 33.0])), cvgs.add(f, k([0.5, 0.25, 0.125, 1.0]))],
        "sub_div_only": [cvgs.subtract(f, k([127.5] * 4)), cvgs.divide(f, k([58.4, 57.1, 57.4, 60.0]))],
        "div_first": [cvgs.divide(f, k([255.0] * 4)), cvgs.subtract(f, k([0.485, 0.456, 0.406, 0.5])), cvgs.divide(f, k([0.229, 0.224, 0.225, 0.25]))],  # two DIVs: one guarded, one real
        "eight_stages": [cvgs.add(f, k([1.0] * 4)), cvgs.multiply(f, k([0.5, 0.25, 2.0, 3.0])), sw, cvgs.subtract(f, k([0.1, 0.2, 0.3, 0.4])), cvgs.divide(f, k([3.0, 7.0, 9.0, 11.0])),
                         cvgs.add(f, k([5.0] * 4)), cvgs.multiply(f, k([-1.0, 1.0, -1.0, 1.0])), cvgs.subtract(f, k([2.0] * 4))],  # more than the unrolled stages
        "zeros_through_div": [cvgs.multiply(f, k([0.0, -0.0, 1.0, 0.0])), cvgs.divide(f, k([3.2, 0.6, 11.8, 33.0])), cvgs.add(f, k([1.0] * 4))],  # +0 / -0 dividends: the wave divides for real
        "tiny_and_huge": [cvgs.multiply(f, k([1e-30, 1e12, 3e38, 1.0])), cvgs.divide(f, k([3.2, 0.6, 0.5, 33.0])), cvgs.subtract(f, k([1.0] * 4))],
        "refused_divisor": [cvgs.multiply(f, k([0.3] * 4)), cvgs.divide(f, k([float(np.float32(2.0) - np.float32(2.0 ** -23)), 2.0 ** 24, 11.8, 33.0]))],
        "swap_only": [sw],
        "div_only": [cvgs.divide(f, k([255.0, 127.5, 63.75, 2.0]))],
        "scale_shift": [cvgs.multiply(f, k([1 / 255.0] * 4)), cvgs.add(f, k([-0.5, -0.25, 0.0, 0.5]))],
        "full_shape": [sw, cvgs.subtract(f, k([104.0, 117.0, 123.0, 0.0])), cvgs.multiply(f, k([0.017, 0.0175, 0.0171, 1.0])), cvgs.divide(f, k([0.9, 1.1, 1.3, 2.0])),
                       cvgs.add(f, k([0.1, 0.2, 0.3, 0.4])), cvgs.multiply(f, k([2.0, 0.5, -1.0, 1.0]))],
        "denormals": [cvgs.multiply(f, k([1e-40, 3e-41, 1e-39, 1e-42])), cvgs.add(f, k([1e-39, -1e-40, 0.0, 1e-41])), cvgs.multiply(f, k([0.5, 0.25, 1.5, 0.75]))],  # subnormal operands, products and sums
        "minus_zero_products": [cvgs.subtract(f, k([5.0] * 4)), cvgs.multiply(f, k([-0.0, 0.0, -0.0, 0.0])), cvgs.add(f, k([-0.0] * 4))],  # p * o with zero products: the fma form keeps their signs
    }
    return {n: [s for s in p if s is not None] for n, p in progs.items()}


@pytest.mark.parametrize("name", ["norm_then_add", "sub_div_only", "div_first", "eight_stages", "zeros_through_div", "tiny_and_huge", "refused_divisor", "minus_zero_products",
                                  "swap_only", "div_only", "scale_shift", "full_shape", "denormals"])
@pytest.mark.parametrize("cn,ar,half", [(3, cvgs.IGNORE_AR, False), (4, cvgs.IGNORE_AR, False), (3, cvgs.PRESERVE_AR, False), (3, cvgs.IGNORE_AR, True)])
def test_interpreted_arithmetic_programs(oracle, device, name, cn, ar, half):
    """Resize chains whose program is NOT [swap,] mul, sub, div: the canonical arithmetic shape ([swap] <= 2 linear stages [div] <= 2 linear stages) is
    rewritten on the host into the straight-line K1CanonProg (every linear stage one fma with (o, -0) / (1, o) / (1, -o); the division by reciprocal
    under the per-wave dividend check); anything else runs K1's interpreted kernel, programs of MUL / ADD / SUB / DIV / REORDER stages on its
    arithmetic path: the program's words fetched before the taps, MUL / ADD / SUB as one fma with selected operands (bit-identical to the plain operation,
    zero signs included), the vetted DIV stage by reciprocal when every dividend of the wave fits.  Against the oracle AND the forced generic kernel;
    aspect-ratio padding pushes the background value through the same program; the fp16 tensor takes the cast in the store."""
    import torch
    fh, fw = 240, 320
    u, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    frame = H.random_u8((fh, fw, cn), seed=7100 + cn)
    frame[40:80, 60:200] = 0  # a black block: zero dividends for some waves only
    crops = H.random_crops(7, fw, fh, seed=7200 + cn, wmin=3, wmax=250, hmin=3, hmax=200)
    prog = _arith_programs(f, cn)[name]
    ot, od = (torch.float16, np.float16) if half else (torch.float32, np.float32)
    o1 = cvgs.CV_16FC1 if half else cvgs.CV_32FC1

    def build(src_mat, out_mat):
        ops = [cvgs.resize(u, cvgs.INTER_LINEAR, [src_mat.roi(*c) for c in crops], (64, 128), 6, [10.0, 20.0, 30.0, 40.0][:cn], ar)] + list(prog)
        if half:
            ops.append(cvgs.convertTo(f, cvgs.make_type(cvgs.CV_16F, cn)))
        return ops + [cvgs.split(cvgs.make_type(cvgs.CV_16F, cn) if half else f, out_mat, (64, 128))]

    ft = torch.from_numpy(frame).to(device)
    out = torch.full((7, cn * 128 * 64), -777.0, dtype=ot, device=device)
    ref = np.full((7, cn * 128 * 64), -777.0, dtype=od)
    ops = build(cvgs.GpuMat.from_tensor(ft, u), cvgs.GpuMat.from_tensor(out, o1))
    # chains of the canonical shape ([swap] <= 2 linear stages [div] <= 2 linear stages) run K1CanonProg, the others K1's interpreted program
    want = "_interp" if name in ("eight_stages", "div_first") else "_arith"
    assert cvgs.kernel_name(*ops).endswith(want + ("_f16" if half else "")), (name, cvgs.kernel_name(*ops))
    with np.errstate(all="ignore"):
        oracle.execute(cvgs.lower(build(cvgs.GpuMat.from_array(frame, u), cvgs.GpuMat.from_array(ref, o1))))
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    H.assert_bit_exact(got, ref, "K1 interpreted arithmetic program %s C%d" % (name, cn))
    out.fill_(-777.0)
    cvgs.executeOperations(torch.cuda.current_stream(), *ops, flags=capi.CHAIN_FORCE_GENERIC)
    torch.cuda.synchronize()
    H.assert_bit_exact(got, out.cpu().numpy(), "K1 interpreted vs the generic kernel")


@pytest.mark.parametrize("name", ["sub_div_only", "scale_shift", "zeros_through_div", "full_shape"])
@pytest.mark.parametrize("cn", [1, 2, 3, 4])
@pytest.mark.parametrize("target", ["planar", "packed_f32", "packed_u8"])
def test_canonical_program_on_every_u8_target(oracle, device, name, cn, target):
    """The canonical arithmetic program also serves 1- / 2-channel planar tensors and packed fp32 / u8 pixel targets of u8 sources (k_k1.hip: canon_packed):
    grayscale "(x - mean) / std", brightness / contrast into a u8 image, ... -- against the oracle and the forced generic kernel."""
    import torch
    fh, fw = 200, 260
    u, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    frame = H.random_u8((fh, fw, cn), seed=7300 + cn)
    frame[30:60, 40:160] = 0
    crops = H.random_crops(5, fw, fh, seed=7400 + cn, wmin=3, wmax=200, hmin=3, hmax=150)
    prog = [s for s in _arith_programs(f, cn)[name]]
    dst = (64, 48)
    if target == "planar":
        shape, odt, tdt, ot = (5, cn * dst[0] * dst[1]), np.float32, torch.float32, cvgs.CV_32FC1
    elif target == "packed_f32":
        shape, odt, tdt, ot = (5, dst[0] * dst[1], cn), np.float32, torch.float32, f
    else:
        shape, odt, tdt, ot = (5, dst[0] * dst[1], cn), np.uint8, torch.uint8, u

    def build(src_mat, out_mat):
        ops = [cvgs.resize(u, cvgs.INTER_LINEAR, [src_mat.roi(*c) for c in crops], dst, 4, [10.0, 20.0, 30.0, 40.0][:cn], cvgs.PRESERVE_AR)] + list(prog)
        if target == "planar":
            return ops + [cvgs.split(f, out_mat, dst) if cn > 1 else cvgs.write(f, out_mat, dst)]
        if target == "packed_u8":
            ops.append(cvgs.convertTo(f, u))
        return ops + [cvgs.write(ot, out_mat, dst)]

    ft = torch.from_numpy(frame).to(device)
    out = torch.zeros(shape, dtype=tdt, device=device)
    ref = np.zeros(shape, dtype=odt)
    ops = build(cvgs.GpuMat.from_tensor(ft, u), cvgs.GpuMat.from_tensor(out, ot))
    kname = cvgs.kernel_name(*ops)
    assert kname.startswith("k1_u8c%d" % cn) and kname.endswith("_arith"), kname
    with np.errstate(all="ignore"):
        oracle.execute(cvgs.lower(build(cvgs.GpuMat.from_array(frame, u), cvgs.GpuMat.from_array(ref, ot))))
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    H.assert_bit_exact(got, ref, "canonical program %s C%d -> %s" % (name, cn, target))
    out.zero_()
    cvgs.executeOperations(torch.cuda.current_stream(), *ops, flags=capi.CHAIN_FORCE_GENERIC)
    torch.cuda.synchronize()
    H.assert_bit_exact(got, out.cpu().numpy(), "canonical vs the generic kernel")


@pytest.mark.parametrize("name", ["sub_div_only", "norm_then_add", "zeros_through_div", "full_shape", "eight_stages"])
@pytest.mark.parametrize("depth,cn", [("16U", 3), ("16U", 4), ("16S", 3), ("32F", 3), ("32F", 4)])
def test_canonical_program_on_other_source_depths(oracle, device, name, depth, cn):
    """16-bit and fp32 sources into planar tensors take the canonical arithmetic program too (fp32 frames with infinities among the pixels: the
    per-wave dividend check sends those waves to the real division)."""
    import torch
    from tests import kat_runner as K
    fh, fw = 180, 240
    st, f = cvgs.make_type(K.CV_DEPTH[depth], cn), cvgs.make_type(cvgs.CV_32F, cn)
    if depth == "32F":
        frame = (H.random_u16((fh, fw, cn), 7500 + cn).astype(np.float32) / 64.0 - 300.0).astype(np.float32)
        frame[100:110, 50:90] = np.inf
    else:
        frame = H.random_u16((fh, fw, cn), 7500 + cn).view(K.NP_DEPTH[depth]).copy()
    frame[20:50, 30:150] = 0
    crops = H.random_crops(5, fw, fh, seed=7600 + cn, wmin=3, wmax=200, hmin=3, hmax=150)
    prog = _arith_programs(f, cn)[name]
    dst = (64, 48)

    def build(src_mat, out_mat):
        return [cvgs.resize(st, cvgs.INTER_LINEAR, [src_mat.roi(*c) for c in crops], dst, 4, [10.0, 20.0, 30.0, 40.0][:cn], cvgs.PRESERVE_AR)] + list(prog) + [cvgs.split(f, out_mat, dst)]

    ft = torch.from_numpy(frame.view(np.int16) if depth == "16U" else frame).to(device)
    out = torch.zeros((5, cn * dst[0] * dst[1]), dtype=torch.float32, device=device)
    ref = np.zeros((5, cn * dst[0] * dst[1]), np.float32)
    ops = build(cvgs.GpuMat.from_tensor(ft, st), cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1))
    want = "_interp" if name == "eight_stages" else "_arith"
    assert cvgs.kernel_name(*ops).endswith(want), cvgs.kernel_name(*ops)
    with np.errstate(all="ignore"):
        oracle.execute(cvgs.lower(build(cvgs.GpuMat.from_array(frame, st), cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1))))
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    H.assert_bit_exact(got, ref, "canonical program %s on %sC%d" % (name, depth, cn))
    out.zero_()
    cvgs.executeOperations(torch.cuda.current_stream(), *ops, flags=capi.CHAIN_FORCE_GENERIC)
    torch.cuda.synchronize()
    H.assert_bit_exact(got, out.cpu().numpy(), "canonical vs the generic kernel")
