"""Arithmetic IOps on integer-typed values (cvGS::multiply / add / subtract / divide<I>, integer pixel type I) and CV_64F warp
sources -- the two drop-in holes VERDICT r2 listed (#4, #5).  The reference instantiates fk::Mul<uchar3> etc. for every type
(include/cvGPUSpeedup.cuh:131-149) but FKL, which defines them, is not in its tree and no reference test uses them; the
semantics are this build's: scalar truncated to the pixel's own type, 64-bit integer arithmetic, division
truncating toward zero with x / 0 = 0, result saturated.  Checked three ways: known answers, numpy restatement, GPU == oracle."""
import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

NP = {cvgs.CV_8U: np.uint8, cvgs.CV_8S: np.int8, cvgs.CV_16U: np.uint16, cvgs.CV_16S: np.int16, cvgs.CV_32S: np.int32}


def numpy_int_arith(img, op, scalar):
    info = np.iinfo(img.dtype)
    b = np.clip(np.trunc(np.asarray(scalar, np.float64)), info.min, info.max).astype(np.int64)
    a = img.astype(np.int64)
    if op == "mul":
        r = a * b
    elif op == "add":
        r = a + b
    elif op == "sub":
        r = a - b
    else:
        safe = np.where(b == 0, 1, b)
        r = np.where(b == 0, 0, np.sign(a) * np.sign(safe) * (np.abs(a) // np.abs(safe)))
    return np.clip(r, info.min, info.max).astype(img.dtype)


FN = {"mul": cvgs.multiply, "add": cvgs.add, "sub": cvgs.subtract, "div": cvgs.divide}


def run_oracle_chain(oracle, img, cv_t, iops):
    out = np.zeros_like(img)
    oracle.execute(cvgs.lower([cvgs.ReadIOp(capi.READ_PIXEL, cv_t, [cvgs.GpuMat.from_array(img, cv_t)], 1), *iops,
                               cvgs.write(cv_t, cvgs.GpuMat.from_array(out, cv_t))]))
    return out


def test_known_answers_on_the_oracle(oracle):
    img = np.array([[[200, 100, 50]]], np.uint8)
    assert run_oracle_chain(oracle, img, cvgs.CV_8UC3, [cvgs.multiply(cvgs.CV_8UC3, [2, 3, 1])]).tolist() == [[[255, 255, 50]]]
    assert run_oracle_chain(oracle, img, cvgs.CV_8UC3, [cvgs.subtract(cvgs.CV_8UC3, [250, 1.9, 50])]).tolist() == [[[0, 99, 0]]]
    assert run_oracle_chain(oracle, img, cvgs.CV_8UC3, [cvgs.divide(cvgs.CV_8UC3, [3, 0, 7])]).tolist() == [[[66, 0, 7]]]
    s16 = np.array([[[-30000, -7, 7]]], np.int16)
    assert run_oracle_chain(oracle, s16, cvgs.CV_16SC3, [cvgs.subtract(cvgs.CV_16SC3, [10000, 0, 0])]).tolist() == [[[-32768, -7, 7]]]
    assert run_oracle_chain(oracle, s16, cvgs.CV_16SC3, [cvgs.divide(cvgs.CV_16SC3, [2, 2, -2])]).tolist() == [[[-15000, -3, -3]]]
    s32 = np.array([[[2_000_000_000, -5]]], np.int32)
    assert run_oracle_chain(oracle, s32, cvgs.CV_32SC2, [cvgs.multiply(cvgs.CV_32SC2, [2, 400_000_000])]).tolist() == [[[2147483647, -2000000000]]]
    assert run_oracle_chain(oracle, s32, cvgs.CV_32SC2, [cvgs.add(cvgs.CV_32SC2, [2_000_000_000, -2147483648.0])]).tolist() == [[[2147483647, -2147483648]]]


@pytest.mark.parametrize("depth", [cvgs.CV_8U, cvgs.CV_8S, cvgs.CV_16U, cvgs.CV_16S, cvgs.CV_32S])
@pytest.mark.parametrize("op", ["mul", "add", "sub", "div"])
def test_oracle_matches_the_numpy_restatement(oracle, depth, op):
    rng = np.random.default_rng(depth * 10 + len(op))
    info = np.iinfo(NP[depth])
    img = rng.integers(info.min, info.max, (40, 50, 3), dtype=np.int64, endpoint=True).astype(NP[depth])
    for scalar in ([2, -3, 0], [7.9, -0.9, 1], [info.max, info.min, 5], [1e12, -1e12, 3]):
        t = cvgs.make_type(depth, 3)
        got = run_oracle_chain(oracle, img, t, [FN[op](t, scalar)])
        H.assert_bit_exact(got, numpy_int_arith(img, op, scalar), "%s %s %s" % (depth, op, scalar))


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [cvgs.CV_8U, cvgs.CV_8S, cvgs.CV_16U, cvgs.CV_16S, cvgs.CV_32S])
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_gpu_integer_arithmetic_matches_the_oracle(oracle, depth, cn):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(depth * 7 + cn)
    info = np.iinfo(NP[depth])
    img = rng.integers(info.min, info.max, (67, 131, cn), dtype=np.int64, endpoint=True).astype(NP[depth])
    t = cvgs.make_type(depth, cn)
    tt = {cvgs.CV_8U: torch.uint8, cvgs.CV_8S: torch.int8, cvgs.CV_16U: torch.int16, cvgs.CV_16S: torch.int16, cvgs.CV_32S: torch.int32}[depth]
    img_t = torch.from_numpy(img.view(np.int16) if depth == cvgs.CV_16U else img).to(dev)
    f = cvgs.make_type(cvgs.CV_32F, cn)
    chains = [
        [cvgs.multiply(t, [2, 3, 1, -1][:cn])], [cvgs.add(t, [100, -100, 7.7, 0][:cn]), cvgs.divide(t, [3, 0, -2, 5][:cn])],
        [cvgs.subtract(t, [info.max, 1, 2, 3][:cn]), cvgs.multiply(t, [-2, 2, 0, 9][:cn]), cvgs.add(t, [1, 1, 1, 1][:cn])],
        # integer arithmetic, then on to float and back: the value changes type mid-chain
        [cvgs.divide(t, [2, 2, 2, 2][:cn]), cvgs.convertTo(t, f, 0.5, 1.25), cvgs.multiply(f, [1.5] * cn), cvgs.convertTo(f, t), cvgs.add(t, [5] * cn)],
    ]
    for k, iops in enumerate(chains):
        out_t = torch.zeros_like(img_t)
        rd = cvgs.ReadIOp(capi.READ_PIXEL, t, [cvgs.GpuMat.from_tensor(img_t, t)], 1)
        ops = [rd, *iops, cvgs.write(t, cvgs.GpuMat.from_tensor(out_t, t))]
        assert "generic" in cvgs.kernel_name(*ops), cvgs.kernel_name(*ops)  # no fast kernel may take integer-typed arithmetic
        cvgs.executeOperations(torch.cuda.current_stream(), *ops)
        torch.cuda.synchronize()
        got = out_t.cpu().numpy()
        got = got.view(np.uint16) if depth == cvgs.CV_16U else got
        H.assert_bit_exact(got, run_oracle_chain(oracle, img, t, iops), "depth %d cn %d chain %d" % (depth, cn, k))


@pytest.mark.gpu
def test_gpu_integer_arithmetic_behind_a_resize_and_in_a_double_chain(oracle):
    import torch
    dev = torch.device("cuda:0")
    frame = H.random_u8((240, 320, 3), seed=61)
    frame_t = torch.from_numpy(frame).to(dev)
    f, s16, d64 = cvgs.CV_32FC3, cvgs.CV_16SC3, cvgs.CV_64FC3
    for iops, out_np, out_t_dtype, wt in (
            ([cvgs.convertTo(f, s16, 3.0, -300.0), cvgs.subtract(s16, [100, -100, 32000]), cvgs.divide(s16, [7, 0, -3])], np.int16, torch.int16, s16),
            ([cvgs.convertTo(f, s16), cvgs.multiply(s16, [200, -200, 1]), cvgs.convertTo(s16, d64, 0.25, 0.0), cvgs.add(d64, [0.1, 0.2, 0.3])], np.float64, torch.float64, d64)):
        out_t = torch.zeros((100, 150, 3), dtype=out_t_dtype, device=dev)
        ref = np.zeros((100, 150, 3), out_np)
        cvgs.executeOperations(torch.cuda.current_stream(), cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), (150, 100)),
                               *iops, cvgs.write(wt, cvgs.GpuMat.from_tensor(out_t, wt)))
        oracle.execute(cvgs.lower([cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), (150, 100)), *iops,
                                   cvgs.write(wt, cvgs.GpuMat.from_array(ref, wt))]))
        torch.cuda.synchronize()
        H.assert_bit_exact(out_t.cpu().numpy(), ref, "integer arithmetic behind a resize -> %s" % out_np.__name__)


@pytest.mark.gpu
@pytest.mark.parametrize("perspective", [False, True])
def test_gpu_warp_of_a_cv64f_source(oracle, perspective):
    """reference warp<WT, InputType> takes any input type (include/cvGPUSpeedup.cuh:285-292): taps cast to float first"""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    img = rng.uniform(-50, 300, (90, 120, 3))
    img_t = torch.from_numpy(img).to(dev)
    m = [[0.9, -0.2, 5.0], [0.15, 1.1, -3.0]] + ([[1e-4, -2e-4, 1.0]] if perspective else [])
    wtype = cvgs.WARP_PERSPECTIVE if perspective else cvgs.WARP_AFFINE
    out_t = torch.zeros((80, 100, 3), dtype=torch.float32, device=dev)
    ref = np.zeros((80, 100, 3), np.float32)
    f = cvgs.CV_32FC3
    cvgs.executeOperations(torch.cuda.current_stream(), cvgs.warp(wtype, cvgs.CV_64FC3, [cvgs.GpuMat.from_tensor(img_t, cvgs.CV_64FC3)], [m], (100, 80)),
                           cvgs.multiply(f, [0.5, 2.0, 1.0]), cvgs.write(f, cvgs.GpuMat.from_tensor(out_t, f)))
    oracle.execute(cvgs.lower([cvgs.warp(wtype, cvgs.CV_64FC3, [cvgs.GpuMat.from_array(img, cvgs.CV_64FC3)], [m], (100, 80)),
                               cvgs.multiply(f, [0.5, 2.0, 1.0]), cvgs.write(f, cvgs.GpuMat.from_array(ref, f))]))
    torch.cuda.synchronize()
    H.assert_bit_exact(out_t.cpu().numpy(), ref, "CV_64F warp source")
