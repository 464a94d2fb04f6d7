"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/cvgs_hip.h declares,
validates chains like the reference's static_asserts/asserts do, and selects the expected kernels (dry run)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(*headers):
    names = set()
    for h in headers or ("cvgs_hip.h", "cvgs_hip_ext.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"^\s*#\s*define[^\n]*(\\\n[^\n]*)*", "", src, flags=re.M)  # (function-like macros are not entry points)
        names |= set(re.findall(r"\b(cvgs_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def exported_symbols(path):
    """The dynamic symbol table of a shared library (defined symbols only)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_library_exports_every_declared_symbol(lib):
    names = header_functions()
    assert len(names) >= 15
    declared_in_binding = {s[0] for s in capi.SYMBOLS}
    for n in names:
        assert hasattr(lib, n), "libcvgs_hip.so does not export %s" % n
        assert n in declared_in_binding, "capi.SYMBOLS lacks %s" % n
    assert lib.cvgs_abi_version() == 6
    assert b"gfx950" in lib.cvgs_version_string()


def test_exported_symbol_list_is_exactly_the_c_abi():
    """the product library's dynamic symbol table IS the C-ABI -- what include/cvgs_hip.h (the drop-in boundary: what
    replaces fk::executeOperations / fk::CircularTensor, reference include/cvGPUSpeedup.cuh:464-627) and include/cvgs_hip_ext.h (engine
    extensions: queue, exchange) declare, nothing more (no C++ internals, no kernel stubs, no test aids), nothing less."""
    core = header_functions("cvgs_hip.h")
    ext = header_functions("cvgs_hip_ext.h")
    assert not [n for n in core if n.startswith(("cvgs_queue_", "cvgs_exchange_", "cvgs_debug_"))], "queue / exchange / debug entry points belong to cvgs_hip_ext.h / the test aid"
    assert not [n for n in ext if n.startswith("cvgs_debug_")]
    assert exported_symbols(capi.LIB_PATH) == sorted(set(core) | set(ext))
    # the boundary itself stays small: what replaces executeOperations + CircularTensor, plane tables, the tick launch, introspection
    assert core == sorted(["cvgs_abi_version", "cvgs_version_string", "cvgs_last_error", "cvgs_device_count", "cvgs_execute", "cvgs_execute_many",
                           "cvgs_stream_release", "cvgs_validate", "cvgs_kernel_name", "cvgs_plane_table_bytes", "cvgs_plane_table_build", "cvgs_plane_table_hull",
                           "cvgs_circular_create", "cvgs_circular_create_ex", "cvgs_circular_update", "cvgs_circular_data", "cvgs_circular_bytes",
                           "cvgs_circular_updates", "cvgs_circular_destroy", "cvgs_stream_copy", "cvgs_range_push", "cvgs_range_pop"])
    rccl = os.path.join(os.path.dirname(capi.LIB_PATH), "libcvgs_rccl.so")
    assert exported_symbols(rccl) == header_functions("cvgs_rccl.h")


def test_struct_layout_matches_c(lib):
    # sizes the C side asserts through struct_size; a mismatch makes every call fail with ERR_INVALID
    ch = capi.new_chain()
    assert lib.cvgs_validate(C.byref(ch)) == capi.ERR_INVALID  # empty chain, but NOT a size mismatch
    assert b"size mismatch" not in lib.cvgs_last_error()
    ch.struct_size = 8
    assert lib.cvgs_validate(C.byref(ch)) == capi.ERR_INVALID
    assert b"size mismatch" in lib.cvgs_last_error()


def _k1(n=4, **kw):
    frame = np.zeros((480, 640, 3), np.uint8)
    out = np.zeros((n, 3 * 64 * 128), np.float32)
    return H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), H.random_crops(n, 640, 480, wmax=100, hmax=200),
                      cvgs.GpuMat.from_array(out, cvgs.CV_32FC1), **kw), (frame, out)


def test_kernel_selection(lib):
    ops, keep = _k1()
    assert cvgs.kernel_name(*ops) == "k1_u8c3_swap_mul_sub_div"
    assert cvgs.kernel_name(*ops, flags=capi.CHAIN_FORCE_GENERIC).startswith("generic_inline")
    ops2, keep2 = _k1(swap=False)
    assert cvgs.kernel_name(*ops2) == "k1_u8c3_mul_sub_div"
    # a program the fast path does not special-case runs interpreted inside the K1 kernel
    ops3 = ops[:2] + [cvgs.add(cvgs.CV_32FC3, [1, 2, 3])] + ops[2:]
    assert cvgs.kernel_name(*ops3) == "k1_u8c3_interp"
    big, keep3 = _k1(n=100)
    assert cvgs.kernel_name(*big).startswith("k1_u8c3")


def test_validation_errors(lib):
    ops, keep = _k1()
    good = cvgs.lower(ops)
    assert lib.cvgs_validate(C.byref(good.desc)) == 0

    def bad(mut, code=capi.ERR_INVALID):
        ch = cvgs.lower(ops)
        mut(ch.desc)
        assert lib.cvgs_validate(C.byref(ch.desc)) == code, lib.cvgs_last_error()
        with pytest.raises(capi.CvgsError):
            capi.check(lib.cvgs_validate(C.byref(ch.desc)))

    bad(lambda d: setattr(d.read, "batch", 0))
    bad(lambda d: setattr(d.read, "used_planes", 99))
    bad(lambda d: setattr(d.read, "dst_width", 0))
    bad(lambda d: setattr(d.read, "aspect_ratio", 7))
    bad(lambda d: setattr(d.write, "width", 63))        # plane size mismatch (reference: assert on split shape)
    bad(lambda d: setattr(d.write, "planes", 2))        # tensor smaller than the batch
    bad(lambda d: setattr(d.write, "dst_type", cvgs.CV_32FC4))  # type produced != type written
    bad(lambda d: setattr(d.write, "data", None))
    bad(lambda d: setattr(d, "n_ops", 99))
    bad(lambda d: setattr(d.ops[0], "opcode", 77))
    bad(lambda d: setattr(d.read, "src_type", cvgs.CV_64FC3))  # a CV_64F resize source is served since round 2; these u8 rows are too short for it
    wsrc = np.zeros((16, 16, 3), np.float64)
    wout = np.zeros((1, 3 * 8 * 8), np.float32)
    wch = cvgs.lower([cvgs.warp(cvgs.WARP_AFFINE, cvgs.CV_64FC3, cvgs.GpuMat.from_array(wsrc, cvgs.CV_64FC3), [[1, 0, 0], [0, 1, 0]], (8, 8)),
                      cvgs.split(cvgs.CV_32FC3, cvgs.GpuMat.from_array(wout, cvgs.CV_32FC1), (8, 8))])
    assert lib.cvgs_validate(C.byref(wch.desc)) == 0  # CV_64F warp sources are served since round 3 (k_warp64)
    # arithmetic on an integer-typed value: served since round 3 (integer arithmetic, saturating); CV_16F values are not
    frame = np.zeros((8, 8, 3), np.uint8)
    outm = np.zeros((8, 8, 3), np.uint8)
    rd = cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3)], 1)
    ch = cvgs.lower([rd, cvgs.multiply(cvgs.CV_8UC3, [2, 2, 2]), cvgs.write(cvgs.CV_8UC3, cvgs.GpuMat.from_array(outm, cvgs.CV_8UC3))])
    assert lib.cvgs_validate(C.byref(ch.desc)) == 0
    outh = np.zeros((8, 8, 3), np.float16)
    ch = cvgs.lower([rd, cvgs.convertTo(cvgs.CV_8UC3, cvgs.CV_16FC3), cvgs.multiply(cvgs.CV_16FC3, [2, 2, 2]),
                     cvgs.write(cvgs.CV_16FC3, cvgs.GpuMat.from_array(outh, cvgs.CV_16FC3))])
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_UNSUPPORTED


def test_dimension_limits(lib):
    """The kernels index rows with 32-bit arithmetic: planes above 2^24 pixels per side, and 4:2:0 surfaces whose luma plane
    exceeds 2 GiB, are refused at validation instead of wrapping around (nothing is dereferenced by cvgs_validate)."""
    ops, keep = _k1()

    def code(mut):
        ch = cvgs.lower(ops)
        mut(ch.desc)
        return lib.cvgs_validate(C.byref(ch.desc))

    big = (1 << 24) + 1
    assert code(lambda d: setattr(d.read, "dst_width", big)) == capi.ERR_UNSUPPORTED
    assert code(lambda d: setattr(d.read, "dst_height", big)) == capi.ERR_UNSUPPORTED

    def wide_source(d):
        im = C.cast(d.read.src, C.POINTER(capi.Image2D))[0]
        im.width, im.step = big, 3 * big
    assert code(wide_source) == capi.ERR_UNSUPPORTED
    # a P010 surface 40000 x 40000: the chroma plane would start 3.2 GB after the luma plane
    fake = np.zeros((4, 4), np.uint16)
    luma = cvgs.GpuMat(40000, 40000, cvgs.CV_16UC1, fake.ctypes.data, 80000, owner=fake)
    out = np.zeros((1, 3 * 8 * 8), np.float32)
    ch = cvgs.lower([cvgs.read_nv12(luma, (8, 8), alpha=False, layout=capi.YUV_P010), cvgs.split(cvgs.CV_32FC3, cvgs.GpuMat.from_array(out, cvgs.CV_32FC1), (8, 8))])
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_UNSUPPORTED
    luma = cvgs.GpuMat(20000, 40000, cvgs.CV_16UC1, fake.ctypes.data, 80000, owner=fake)  # 1.6 GB: fine
    ch = cvgs.lower([cvgs.read_nv12(luma, (8, 8), alpha=False, layout=capi.YUV_P010), cvgs.split(cvgs.CV_32FC3, cvgs.GpuMat.from_array(out, cvgs.CV_32FC1), (8, 8))])
    assert lib.cvgs_validate(C.byref(ch.desc)) == 0


def _oracle_k1(oracle, frame, crops):
    ref = np.zeros((len(crops), 3 * 64 * 128), np.float32)
    oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1))))
    return ref


def test_mirrors_oracle_and_validation(oracle, lib):
    """The oracle interprets the same descriptor field; bad mirror lists are refused."""
    frame = H.random_u8((240, 320, 3), seed=4)
    crops = H.random_crops(3, 320, 240, seed=5, wmax=100, hmax=100)
    a = np.zeros((3, 3 * 64 * 128), np.float32)
    b = np.zeros_like(a)
    ops = H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_array(a, cvgs.CV_32FC1))
    ops[-1].mirrored_to([b.ctypes.data])
    oracle.execute(cvgs.lower(ops))
    assert a.any() and (a.view(np.uint32) == b.view(np.uint32)).all()
    ch = cvgs.lower(ops)
    ch.desc.write.n_mirrors = 8
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID
    ch = cvgs.lower(ops)
    ch.desc.write.mirrors = None
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID
    ch = cvgs.lower(ops)
    ch.desc.flags = 1 << 8  # experimental-variant bits are no longer part of the boundary
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID


def test_colour_op_channel_indices_are_validated(lib):
    """ADD_ALPHA / DROP_ALPHA / GRAY aux indices must name channels the value has (REORDER already did)."""
    frame = np.zeros((8, 8, 3), np.uint8)
    out1 = np.zeros((8, 8), np.uint8)
    rd = cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3)], 1)
    ch = cvgs.lower([rd, cvgs.cvtColor(cvgs.COLOR_RGB2GRAY, cvgs.CV_8UC3, cvgs.CV_8UC1),
                     cvgs.write(cvgs.CV_8UC1, cvgs.GpuMat.from_array(out1, cvgs.CV_8UC1))])
    assert lib.cvgs_validate(C.byref(ch.desc)) == 0
    ch.desc.ops[0].aux = 3 | (1 << 2) | (2 << 4)  # R taken from channel 3 of a 3-channel value
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID
    out4 = np.zeros((8, 8, 4), np.uint8)
    ch = cvgs.lower([rd, cvgs.cvtColor(cvgs.COLOR_RGB2RGBA, cvgs.CV_8UC3, cvgs.CV_8UC4),
                     cvgs.write(cvgs.CV_8UC4, cvgs.GpuMat.from_array(out4, cvgs.CV_8UC4))])
    assert lib.cvgs_validate(C.byref(ch.desc)) == 0
    ch.desc.ops[0].aux = 0 | (3 << 2) | (2 << 4)
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID


def test_builder_type_checks():
    """The checks the reference does with static_assert (include/cvGPUSpeedup.cuh:76,153-156,211)."""
    with pytest.raises(ValueError):
        cvgs.convertTo(cvgs.CV_8UC1, cvgs.CV_32FC2)
    with pytest.raises(ValueError):
        cvgs.cvtColor(cvgs.COLOR_RGB2BGR, cvgs.CV_32SC3)
    with pytest.raises(ValueError):
        cvgs.cvtColor(8, cvgs.CV_8UC1, cvgs.CV_8UC3)  # COLOR_GRAY2BGR: not in SupportedColorConversions
    with pytest.raises(ValueError):
        cvgs.resize(cvgs.CV_8UC3, 0, [], (64, 128))   # INTER_NEAREST: not in SupportedInterpolations
    ops, keep = _k1()
    with pytest.raises(TypeError):                     # IOp input type must equal the previous output type
        cvgs.lower([ops[0], cvgs.multiply(cvgs.CV_8UC3, [1, 1, 1])] + ops[2:])


def test_plane_table_build(lib, oracle):
    ops, keep = _k1(n=5)
    raw = cvgs.build_plane_table(ops[0])
    assert len(raw) == lib.cvgs_plane_table_bytes(5) == 5 * 48
    tab = np.frombuffer(raw, dtype=np.dtype([("data", "<u8"), ("w", "<i4"), ("h", "<i4"), ("step", "<i4"), ("fx", "<f4"),
                                             ("fy", "<f4"), ("x1", "<i4"), ("y1", "<i4"), ("x2", "<i4"), ("y2", "<i4"),
                                             ("pad", "<i4")]))
    for i, m in enumerate(ops[0].mats):
        g = oracle.resize_geometry(m.cols, m.rows, 64, 128, cvgs.IGNORE_AR)
        assert tab["data"][i] == m.data and tab["w"][i] == m.cols and tab["h"][i] == m.rows
        assert tab["fx"][i] == np.float32(g.fx) and tab["fy"][i] == np.float32(g.fy)
        assert (tab["x1"][i], tab["y1"][i], tab["x2"][i], tab["y2"][i]) == (0, 0, 63, 127)


@pytest.mark.parametrize("ar", [cvgs.IGNORE_AR, cvgs.PRESERVE_AR, cvgs.PRESERVE_AR_RN_EVEN, cvgs.PRESERVE_AR_LEFT])
def test_host_geometry_matches_oracle(lib, oracle, ar):
    """The product's host-side geometry (plane tables) and the oracle's restatement agree bit for bit."""
    rng = np.random.default_rng(3)
    frame = np.zeros((1100, 1300, 3), np.uint8)
    m = cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3)
    sizes = [(int(rng.integers(1, 1300)), int(rng.integers(1, 1100))) for _ in range(40)] + [(30, 120), (60, 120), (1, 1)]
    for dst in [(64, 128), (128, 64), (33, 77)]:
        rd = cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, [m.roi(0, 0, w, h) for w, h in sizes], dst, len(sizes), None, ar)
        raw = np.frombuffer(cvgs.build_plane_table(rd), dtype=np.uint8).reshape(len(sizes), 48)
        for i, (w, h) in enumerate(sizes):
            g = oracle.resize_geometry(w, h, dst[0], dst[1], ar)
            fx, fy = raw[i, 20:28].view(np.float32)
            win = tuple(raw[i, 28:44].view(np.int32))
            assert (fx, fy) == (np.float32(g.fx), np.float32(g.fy)), (w, h, dst)
            assert win == (g.x1, g.y1, g.x2, g.y2), (w, h, dst, win)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        capi.load_library()


def test_rccl_library_exports_every_declared_symbol():
    from cvgpuspeedup_amd import rccl
    src = open(os.path.join(ROOT, "include", "cvgs_rccl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(cvgs_[a-z0-9_]+)\s*\(", src)))
    lib = rccl.load_library()
    bound = {s[0] for s in rccl.SYMBOLS}
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), n
        assert n in bound, n


def test_device_plane_tables_are_refused_for_layouts_they_cannot_vouch_for(lib):
    """a device plane table carries no layout tag; P010 / I420 / YV12 need per-plane checks the call cannot make."""
    import numpy as np
    surf = np.zeros((96 * 3 // 2, 128), np.uint8)
    out = np.zeros((1, 3 * 32 * 32), np.float32)
    for layout, ok in ((capi.YUV_NV12, True), (capi.YUV_NV21, True), (capi.YUV_P010, False), (capi.YUV_I420, False), (capi.YUV_YV12, False)):
        rd = cvgs.read_nv12(cvgs.GpuMat.from_array(surf, cvgs.CV_8UC1), (32, 32), layout=capi.YUV_NV12, alpha=False)
        lowered = cvgs.lower([rd, cvgs.split(cvgs.CV_32FC3, cvgs.GpuMat.from_array(out, cvgs.CV_32FC1), (32, 32))])
        lowered.desc.read.flags |= 1  # CVGS_READ_FLAG_TABLE_ON_DEVICE: `src` now names a (pretend) device table
        lowered.desc.read.yuv_layout = layout
        rc = lib.cvgs_validate(C.byref(lowered.desc))
        assert (rc == 0) == ok, (layout, rc, lib.cvgs_last_error())


def test_plane_table_hull_and_the_fields_that_state_it(lib):
    """ABI 6: cvgs_plane_table_hull gives the byte range a table's planes read; a chain states it for a DEVICE table only."""
    frame = np.zeros((480, 640, 3), np.uint8)
    out = np.zeros((3, 3 * 64 * 128), np.float32)
    g = cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3)
    crops = [(10, 20, 100, 50), (300, 400, 64, 80), (5, 470, 30, 10)]
    ops = H.k1_chain(g, crops, cvgs.GpuMat.from_array(out, cvgs.CV_32FC1))
    lo, hi = cvgs.table_hull(ops[0])
    base, step = frame.ctypes.data, 640 * 3
    assert lo == base + 20 * step + 10 * 3                      # first byte of the top-most crop
    assert hi == base + 479 * step + (300 + 64) * 3             # one past the last tapped byte: row 479 is the last row of crops 1 and 2, crop 1 ends further right
    # NV12 surfaces: the chroma rows behind the luma plane belong to the range
    surf = np.zeros((480 * 3 // 2, 640), np.uint8)
    rd = cvgs.read_nv12(cvgs.GpuMat(480, 640, cvgs.CV_8UC1, surf.ctypes.data, 640, owner=surf), (64, 64), capi.YUV_FULL, capi.BT709, False)
    lo, hi = cvgs.table_hull(rd)
    assert lo == surf.ctypes.data and hi == surf.ctypes.data + 640 * 720
    # the fields describe device tables only, come in pairs, lo <= hi; unknown read flags are refused
    low = cvgs.lower(ops)
    for lo_v, hi_v, flags, msg in ((base, base + 100, 0, b"DEVICE table"), (0, 0, capi.READ_FLAG_TABLE_SOURCES_VOUCHED, b"DEVICE table"),
                                   (0, 0, 4, b"unknown read flags")):
        low.desc.read.table_src_lo, low.desc.read.table_src_hi, low.desc.read.flags = lo_v, hi_v, flags
        assert lib.cvgs_validate(C.byref(low.desc)) == capi.ERR_INVALID
        assert msg in lib.cvgs_last_error(), lib.cvgs_last_error()
    low.desc.read.flags = capi.READ_FLAG_TABLE_ON_DEVICE
    low.desc.read.src = 4096  # (a pretend device table: validation does not read it)
    for lo_v, hi_v in ((base, 0), (0, base), (base + 8, base)):
        low.desc.read.table_src_lo, low.desc.read.table_src_hi = lo_v, hi_v
        assert lib.cvgs_validate(C.byref(low.desc)) == capi.ERR_INVALID
        assert b"table_src_lo" in lib.cvgs_last_error()
    low.desc.read.table_src_lo, low.desc.read.table_src_hi = base, base + frame.nbytes
    assert lib.cvgs_validate(C.byref(low.desc)) == capi.OK, lib.cvgs_last_error()
