"""K1's whole-frame form for packed targets (k_k1_x4.hip: four x-adjacent output pixels per lane, source rows kept unpacked
in registers across output rows): resize -> convertTo<32F, 8U> -> write (the reference's tests/resize/test_resize_write.cu
chain), bit-exact vs the oracle and identical to the one-pixel-per-lane kernel.  CVGS_K1_X4=1 makes launch_k1 pick the kernel whenever the chain is eligible, so that small, ragged
and edge-heavy shapes run through it too; the size rule itself is checked on an up-scaled 4K output."""
import os

import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H
from tests.test_gpu_chains import _both
from tests.test_gpu_k1_writes import _name

pytestmark = pytest.mark.gpu


@pytest.fixture
def forced():
    old = os.environ.get("CVGS_K1_X4")
    os.environ["CVGS_K1_X4"] = "1"
    yield
    if old is None:
        del os.environ["CVGS_K1_X4"]
    else:
        os.environ["CVGS_K1_X4"] = old


def _chain(src, cn, dst, to_u8, x_off=3, pad=6):
    u, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    pitch_w = dst[0] + x_off + pad  # a view of a wider buffer -> pitched rows that start on any byte
    odt = np.uint8 if to_u8 else np.float32

    def build(wrap, wrap_out, out):
        o = wrap_out(np.zeros((dst[1], pitch_w, cn), odt) if out is None else out, u if to_u8 else f)
        ops = [cvgs.resize(u, cvgs.INTER_LINEAR, wrap(src, u), dst)]
        if to_u8:
            ops.append(cvgs.convertTo(f, u))
        return ops + [cvgs.write(u if to_u8 else f, o.roi(x_off, 0, dst[0], dst[1]))]

    return build, (dst[1], pitch_w, cn), odt


# (source h, w) -> (dst w, dst h): 2x up, ragged up (tail lanes, several column tiles), down 2.25x, 1 < f < 2 both ways,
# identity, mixed up/down, a one-row source, a source barely one window wide, more rows than one wave takes
SHAPES = [((135, 240), (480, 270)), ((97, 211), (1031, 333)), ((540, 961), (427, 240)), ((300, 400), (263, 197)),
          ((64, 96), (96, 64)), ((200, 150), (601, 77)), ((1, 50), (130, 9)), ((9, 3), (8, 70)), ((33, 40), (260, 530))]


@pytest.mark.parametrize("cn", [1, 2, 3, 4])
@pytest.mark.parametrize("shape", SHAPES)
def test_x4_matches_the_oracle(forced, shape, cn, to_u8=True):
    (sh, sw), dst = shape
    if sw * cn < 8:
        pytest.skip("rows narrower than one 8-byte window stay with the one-pixel kernel")
    src = H.random_u8((sh, sw, cn), 900 + cn + sh)
    build, oshape, odt = _chain(src, cn, dst, to_u8)
    gpu, ref = _both(build, oshape, odt)
    H.assert_bit_exact(gpu[0], ref[0], "resize -> packed, four pixels per lane")
    assert _name(build) == "k1_u8c%d_packed_%s_x4" % (cn, "u8" if to_u8 else "f32")
    one, _ = _both(build, oshape, odt, flags=capi.CHAIN_NO_THREAD_FUSION)
    H.assert_bit_exact(one[0], gpu[0], "the one-pixel-per-lane kernel agrees")


TYPED = [("16U", 1), ("16U", 3), ("16U", 4), ("16S", 1), ("16S", 3), ("16S", 4), ("32F", 1)]


def _typed_chain(src, depth, cn, dst, x_off=3, pad=6):
    from tests import kat_runner as K
    st, f = cvgs.make_type(K.CV_DEPTH[depth], cn), cvgs.make_type(cvgs.CV_32F, cn)
    np_dt = K.NP_DEPTH[depth]
    pitch_w = dst[0] + x_off + pad

    def build(wrap, wrap_out, out):
        o = wrap_out(np.zeros((dst[1], pitch_w, cn), np_dt) if out is None else out, st)
        ops = [cvgs.resize(st, cvgs.INTER_LINEAR, wrap(src, st), dst)]
        if depth != "32F":
            ops.append(cvgs.convertTo(f, st))
        return ops + [cvgs.write(st, o.roi(x_off, 0, dst[0], dst[1]))]

    return build, (dst[1], pitch_w, cn), np_dt


def _typed_name(depth, cn):
    t = depth[-1].lower() + depth[:-1]
    return "k1_%sc%d_packed_%s_x%d" % (t, cn, t, 4 if depth == "32F" else 2)


@pytest.mark.parametrize("depth,cn", TYPED)
@pytest.mark.parametrize("shape", SHAPES)
def test_x4_16_bit_and_float_images(forced, shape, depth, cn):
    """the reference's resize_write sweep (tests/resize/test_resize_write.cu:110-123: CV_16U / CV_16S C1, C3, C4 and CV_32FC1):
    two pixels per lane for 16-bit images (16-byte tap windows), four for CV_32FC1."""
    from tests.test_gpu_chains import _random_src
    (sh, sw), dst = shape
    eb = 4 if depth == "32F" else 2
    if sw * cn * eb < (8 if depth == "32F" else 16):
        pytest.skip("rows narrower than one tap window stay with the one-pixel kernel")
    src = _random_src((sh, sw, cn), depth, 700 + cn + sh)
    if depth == "32F":
        src = (src * 50.0).astype(np.float32)
    build, oshape, odt = _typed_chain(src, depth, cn, dst)
    gpu, ref = _both(build, oshape, odt)
    H.assert_bit_exact(gpu[0], ref[0], "resize -> packed %sC%d" % (depth, cn))
    assert _name(build) == _typed_name(depth, cn), _name(build)
    one, _ = _both(build, oshape, odt, flags=capi.CHAIN_NO_THREAD_FUSION)
    H.assert_bit_exact(one[0], gpu[0], "the one-pixel-per-lane kernel agrees")


@pytest.mark.parametrize("depth,cn", [("16U", 3), ("16S", 1), ("32F", 1)])
def test_x4_reference_resize_write_size_typed(depth, cn):
    """4K -> 3870 x 2260 on the 16-bit / float types of the reference's sweep, no environment hook."""
    from tests.test_gpu_chains import _random_src
    assert "CVGS_K1_X4" not in os.environ
    src = _random_src((2160, 3840, cn), depth, 60 + cn)
    if depth == "32F":
        src = (src * 50.0).astype(np.float32)
    build, oshape, odt = _typed_chain(src, depth, cn, (3870, 2260), x_off=0, pad=0)
    gpu, ref = _both(build, oshape, odt)
    H.assert_bit_exact(gpu[0], ref[0], "4K -> 3870x2260 packed %sC%d" % (depth, cn))
    assert _name(build) == _typed_name(depth, cn)


@pytest.mark.parametrize("seed", range(6))
def test_x4_random_shapes(forced, seed):
    """15 random (source, target, channels) shapes per seed: any mix of up- and down-scaling, ragged widths, 1-row sources."""
    rng = np.random.default_rng(4200 + seed)
    for _ in range(15):
        cn = int(rng.integers(1, 5))
        sh, sw = int(rng.integers(1, 140)), int(rng.integers((8 + cn - 1) // cn, 320))
        dst = (int(rng.integers(4, 720)), int(rng.integers(1, 220)))
        src = H.random_u8((sh, sw, cn), int(rng.integers(1 << 30)))
        build, oshape, odt = _chain(src, cn, dst, True, x_off=int(rng.integers(0, 9)), pad=int(rng.integers(0, 9)))
        gpu, ref = _both(build, oshape, odt)
        H.assert_bit_exact(gpu[0], ref[0], "%dx%dx%d -> %dx%d" % (sw, sh, cn, dst[0], dst[1]))
        assert _name(build) == "k1_u8c%d_packed_u8_x4" % cn


@pytest.mark.parametrize("cn", [3, 4])
def test_x4_extreme_pixels_and_constant_rows(forced, cn):
    """0 / 255 checkerboards (every interpolated value a .5 tie or an end of the range) through the folded SaturateCast."""
    src = np.zeros((120, 200, cn), np.uint8)
    src[::2, ::2] = 255
    src[1::2, 1::2] = 255
    src[:, -1] = 255
    build, oshape, odt = _chain(src, cn, (400, 240), True)
    gpu, ref = _both(build, oshape, odt)
    H.assert_bit_exact(gpu[0], ref[0], "checkerboard 2x up")
    build, oshape, odt = _chain(src, cn, (300, 90), True)
    gpu, ref = _both(build, oshape, odt)
    H.assert_bit_exact(gpu[0], ref[0], "checkerboard mixed")


def test_x4_batch_of_images(forced, to_u8=True):
    """write<O>(GpuMat, Size): N crops of a frame resized into dense [plane][y][x] packed pixels (one launch, blockIdx.y = crop)."""
    cn, n, dst = 3, 5, (132, 75)
    src = H.random_u8((400, 600, cn), 77)
    crops = H.random_crops(n, 600, 400, seed=9, wmin=8, wmax=400, hmin=2, hmax=300)
    u, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    ot, odt = (u, np.uint8) if to_u8 else (f, np.float32)

    def build(wrap, wrap_out, out):
        frame = wrap(src, u)
        o = wrap_out(np.zeros((n, dst[0] * dst[1], cn), odt) if out is None else out, ot)
        ops = [cvgs.resize(u, cvgs.INTER_LINEAR, [frame.roi(*c) for c in crops], dst, n)]
        if to_u8:
            ops.append(cvgs.convertTo(f, u))
        return ops + [cvgs.write(ot, o, dst)]

    gpu, ref = _both(build, (n, dst[0] * dst[1], cn), odt)
    H.assert_bit_exact(gpu[0], ref[0], "batch resize -> packed")
    assert _name(build) == "k1_u8c3_packed_%s_x4" % ("u8" if to_u8 else "f32")


def test_x4_leaves_what_it_does_not_cover(forced):
    """a program between resize and write, a packed fp32 target: the one-pixel kernel."""
    cn = 3
    src = H.random_u8((100, 160, cn), 5)
    u, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    def with_program(wrap, wrap_out, out):
        o = wrap_out(np.zeros((200, 320, cn), np.float32) if out is None else out, f)
        return [cvgs.resize(u, cvgs.INTER_LINEAR, wrap(src, u), (320, 200)), cvgs.multiply(f, [0.5] * cn), cvgs.write(f, o)]

    gpu, ref = _both(with_program, (200, 320, cn), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "program between resize and write")
    assert _name(with_program) == "k1_u8c3_packed_f32_arith"  # (the canonical arithmetic program, round 6)
    build, oshape, odt = _chain(src, cn, (320, 200), False)
    gpu, ref = _both(build, oshape, odt)
    H.assert_bit_exact(gpu[0], ref[0], "packed fp32 target")
    assert _name(build) == "k1_u8c3_packed_f32"


def test_x4_is_the_default_for_whole_frames():
    """no environment hook: a 1080p -> 4K packed u8 resize (the reference's test_resize_write chain at frame size) takes
    the four-pixels-per-lane kernel; a 64 x 128 crop does not."""
    assert "CVGS_K1_X4" not in os.environ
    src = H.random_u8((1080, 1920, 3), 11)
    build, oshape, odt = _chain(src, 3, (3840, 2160), True, x_off=0, pad=0)
    gpu, ref = _both(build, oshape, odt)
    H.assert_bit_exact(gpu[0], ref[0], "1080p -> 4K packed u8")
    assert _name(build) == "k1_u8c3_packed_u8_x4"
    small, oshape, odt = _chain(src, 3, (64, 128), True, x_off=0, pad=0)
    assert _name(small) == "k1_u8c3_packed_u8"
    down, oshape, odt = _chain(H.random_u8((2160, 3840, 3), 12), 3, (1920, 1080), True, x_off=0, pad=0)
    assert _name(down) == "k1_u8c3_packed_u8"  # vertical down-scaling: no source row is shared, the one-pixel kernel is as fast


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_x4_reference_resize_write_size(cn):
    """tests/resize/test_resize_write.cu:55-56 at its own size: 4K -> 3870 x 2260 (rows of 3870 * cn bytes start on any byte)."""
    src = H.random_u8((2160, 3840, cn), 40 + cn)
    build, oshape, odt = _chain(src, cn, (3870, 2260), True, x_off=0, pad=0)
    gpu, ref = _both(build, oshape, odt)
    H.assert_bit_exact(gpu[0], ref[0], "4K -> 3870x2260 packed u8c%d" % cn)
    assert _name(build) == "k1_u8c%d_packed_u8_x4" % cn


# round 6: u8c3 / u8c4 without horizontal down-scaling take ONE 16-byte (+ 4-byte) tap window per lane and source row; the window is clamped
# back at the row's end, and a lane's pixels may start anywhere inside it: narrow sources (barely one window), identity scale (every lane's
# window offset is a multiple of 12 / 16), ragged last column tiles, fractional up-scaling, sources that END at the end of their allocation
WIDE_SHAPES = [((40, 6), (18, 80)), ((40, 7), (7, 40)), ((33, 17), (50, 70)), ((64, 256), (256, 64)), ((64, 257), (257, 64)), ((30, 300), (1023, 41)),
               ((50, 86), (258, 100)), ((25, 1000), (1000, 25)), ((19, 341), (999, 57)), ((12, 5), (12, 24))]


@pytest.mark.parametrize("cn", [3, 4])
@pytest.mark.parametrize("shape", WIDE_SHAPES)
def test_x4_wide_tap_window(forced, shape, cn):
    (sh, sw), dst = shape
    if sw * cn < (20 if cn == 4 else 16):
        # (narrower than one wide window: the four-window form serves it -- still bit-exact, still this kernel)
        assert sw * cn >= 8
    src = H.random_u8((sh, sw, cn), 4200 + cn + sw)
    build, oshape, odt = _chain(src, cn, dst, True, x_off=1, pad=3)
    gpu, ref = _both(build, oshape, odt)
    H.assert_bit_exact(gpu[0], ref[0], "resize -> packed u8, wide tap window, %dx%d -> %dx%d c%d" % (sw, sh, dst[0], dst[1], cn))
    assert _name(build) == "k1_u8c%d_packed_u8_x4" % cn
    one, _ = _both(build, oshape, odt, flags=capi.CHAIN_NO_THREAD_FUSION)
    H.assert_bit_exact(one[0], gpu[0], "the one-pixel-per-lane kernel agrees")
