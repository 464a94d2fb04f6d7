"""Pins the CPU oracle against every known-answer vector the reference's own tests hold for the hot path
(tests/golden/reference_kats.json).  Runs without a GPU."""
import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import kat_runner as K

CASES = K.load_cases()
CHAIN_CASES = [c for c in CASES if "kind" not in c]


def _small(case):
    """Keep the CPU suite quick: shrink the 4K frames of pure pixel chains (the values are position independent)."""
    c = dict(case)
    if c["read"]["kind"] in ("pixel_single",) and c["frame"] == [3840, 2160]:
        c["frame"] = [384, 216]
    return c


@pytest.mark.parametrize("case", CHAIN_CASES, ids=[c["name"] for c in CHAIN_CASES])
def test_oracle_matches_reference_kat(case, oracle):
    c = _small(case)
    out = K.run_chain_case(c, "oracle", batch_limit=10)
    K.check_chain_case(c, out)


def test_k1_strict_fp32_constants(oracle):
    """The strict-fp32 (no FMA) values of the K1 KAT, bit for bit (0.15625, -4.1666665, -0.1440678)."""
    case = [c for c in CASES if c["name"] == "k1_8UC3"][0]
    out = K.run_chain_case(case, "oracle", batch_limit=3)
    exp = np.array([0.15625, -4.1666665, -0.1440678], np.float32)
    assert (out == exp[None, :, None, None]).all()


@pytest.mark.parametrize("case", [c for c in CASES if c.get("kind") == "circular"], ids=lambda c: c["name"])
def test_oracle_circular_tensor_kat(case, oracle):
    _, icn, itype = K.parse_type(case["in_type"])
    ed, ecn, etype = K.parse_type(case["elem_type"])
    order = cvgs.NewestFirst if case["order"] == "NewestFirst" else cvgs.OldestFirst
    mode = cvgs.Transposed if case["mode"] == "Transposed" else cvgs.Standard
    W, H, B, CP = case["width"], case["height"], case["batch"], case["color_planes"]
    ct = oracle.OracleCircular(W, H, etype, CP, B, order, mode)
    ftype = cvgs.make_type(cvgs.CV_32F, icn)
    for i in range(case["iters"]):
        frame = np.full((H, W, icn), (i + 1) % 256, np.uint8)
        kind = {"tensor_split": capi.WRITE_TENSOR_SPLIT, "tensor_t_split": capi.WRITE_TENSOR_T_SPLIT,
                "tensor_write": capi.WRITE_PIXEL_3D}[case["write"]]
        chain = cvgs.lower([cvgs.ReadIOp(capi.READ_PIXEL, itype, [cvgs.GpuMat.from_array(frame, itype)], 1),
                            cvgs.convertTo(itype, ftype), cvgs.WriteIOp(kind, ftype, 16, W, H, 0, B)])
        ct.update(chain)
    t = ct.array(np.float32)
    exp = np.asarray(case["expected_slot"], np.float32)
    if case["write"] == "tensor_t_split":
        t = t.reshape(CP, B, H, W)
        assert (t == exp[None, :, None, None]).all()
    elif case["write"] == "tensor_split":
        t = t.reshape(B, CP, H, W)
        assert (t == exp[:, None, None, None]).all()
    else:
        t = t.reshape(B, H, W, ecn)
        assert (t == exp[:, None, None, None]).all()


def test_oracle_circular_batch_read(oracle):
    """fk::CircularBatchRead<Ascendent>: out[z] = in[(z + first) mod BATCH] -- lowered on the host as a rotation of
    the plane list (reference tests/batchread/test_circularbatchread_x_write3D.cu:59-83)."""
    case = [c for c in CASES if c["name"] == "circular_batch_read"][0]
    B, first, W, H = case["batch"], case["first"], case["width"], case["height"]
    planes = [np.full((H, W, 3), i, np.uint8) for i in range(B)]
    mats = [cvgs.GpuMat.from_array(p, cvgs.CV_8UC3) for p in planes]
    rotated = [mats[(z + first) % B] for z in range(B)]
    out = np.zeros((B, W * H, 3), np.uint8)
    chain = cvgs.lower([cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, rotated, B),
                        cvgs.write(cvgs.CV_8UC3, cvgs.GpuMat.from_array(out, cvgs.CV_8UC3), (W, H))])
    oracle.execute(chain)
    exp = np.asarray(case["expected_plane"], np.uint8)
    assert (out.reshape(B, H, W, 3) == exp[:, None, None, None]).all()
