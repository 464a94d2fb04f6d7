"""Host-side mirror of the reference's cvGS:: operator interface for the hot path, in Python.

The reference interface is a C++ template facade (include/cvGPUSpeedup.cuh in the reference); its
C++ mirror for drop-in use lives in cvgpuspeedup_amd/include/cvGPUSpeedup.h.  This module spells the
same builders (resize / convertTo / multiply / subtract / divide / add / cvtColor / split / splitT /
write / executeOperations / CircularTensor) for the Python test and benchmark harnesses: every
builder returns a small IOp object, executeOperations() lowers the list to ONE cvgs_chain_desc and
calls cvgs_execute() -> one HIP kernel.  Nothing here computes pixels.
"""
import ctypes as C
import struct

from . import capi
from .capi import (DEPTH_8U, DEPTH_8S, DEPTH_16U, DEPTH_16S, DEPTH_32S, DEPTH_32F, DEPTH_64F, DEPTH_16F, make_type,
                   type_cn, type_depth)

# ---- OpenCV type codes / enums (numeric values of OpenCV 4.x) ---------------------------------------
for _d, _n in ((DEPTH_8U, "8U"), (DEPTH_8S, "8S"), (DEPTH_16U, "16U"), (DEPTH_16S, "16S"), (DEPTH_32S, "32S"),
               (DEPTH_32F, "32F"), (DEPTH_64F, "64F"), (DEPTH_16F, "16F")):
    globals()["CV_" + _n] = _d
    for _c in (1, 2, 3, 4):
        globals()["CV_%sC%d" % (_n, _c)] = make_type(_d, _c)

INTER_LINEAR = 1
COLOR_BGR2BGRA = COLOR_RGB2RGBA = 0
COLOR_BGRA2BGR = COLOR_RGBA2RGB = 1
COLOR_BGR2RGBA = COLOR_RGB2BGRA = 2
COLOR_RGBA2BGR = COLOR_BGRA2RGB = 3
COLOR_BGR2RGB = COLOR_RGB2BGR = 4
COLOR_BGRA2RGBA = COLOR_RGBA2BGRA = 5
COLOR_BGR2GRAY = 6
COLOR_RGB2GRAY = 7
COLOR_BGRA2GRAY = 10
COLOR_RGBA2GRAY = 11
# reference include/cv2cuda_types.cuh:77-86
SUPPORTED_COLOR_CONVERSIONS = (0, 1, 2, 3, 4, 5, 6, 7, 10, 11)
SUPPORTED_INTERPOLATIONS = (INTER_LINEAR,)

PRESERVE_AR, IGNORE_AR, PRESERVE_AR_RN_EVEN, PRESERVE_AR_LEFT = 0, 1, 2, 3  # cvGS::AspectRatio
NewestFirst, OldestFirst = capi.NEWEST_FIRST, capi.OLDEST_FIRST                # fk::CircularTensorOrder
Standard, Transposed = capi.PLANES_STANDARD, capi.PLANES_TRANSPOSED            # fk::ColorPlanes

_DEPTH_BYTES = {DEPTH_8U: 1, DEPTH_8S: 1, DEPTH_16U: 2, DEPTH_16S: 2, DEPTH_32S: 4, DEPTH_32F: 4, DEPTH_64F: 8,
                DEPTH_16F: 2}


def elem_size(cv_type):
    return _DEPTH_BYTES[type_depth(cv_type)] * type_cn(cv_type)


# ---- cv::cuda::GpuMat stand-in ------------------------------------------------------------------------
class GpuMat:
    """A typed, pitched 2D view of (device or host) memory: data/rows/cols/step/type like cv::cuda::GpuMat.

    `owner` keeps the backing storage (torch tensor, numpy array) alive; a crop is a view that shares it
    (reference: GpuMat::operator()(Rect), tests/batchresize/test_batchresize_x_split3D.cu:284)."""

    def __init__(self, rows, cols, cv_type, data, step=None, owner=None):
        self.rows, self.cols, self.cv_type = int(rows), int(cols), int(cv_type)
        self.data = int(data)
        self.step = int(step) if step is not None else self.cols * elem_size(cv_type)
        self.owner = owner

    @staticmethod
    def from_tensor(t, cv_type):
        """Wrap a torch tensor (H,W[,C]) whose rows are contiguous."""
        assert t.stride(-1) == 1 or t.dim() == 2
        return GpuMat(t.shape[0], t.shape[1], cv_type, t.data_ptr(), t.stride(0) * t.element_size(), owner=t)

    @staticmethod
    def from_array(a, cv_type):
        """Wrap a numpy array (H,W[,C]) (host memory: for the CPU oracle only)."""
        return GpuMat(a.shape[0], a.shape[1], cv_type, a.ctypes.data, a.strides[0], owner=a)

    def type(self):
        return self.cv_type

    def roi(self, x, y, w, h):
        """cv::Rect2d crop: doubles are truncated like the reference (include/cvGPUSpeedup.cuh:247-249)."""
        x, y, w, h = int(x), int(y), int(w), int(h)
        assert 0 <= x and 0 <= y and x + w <= self.cols and y + h <= self.rows, "ROI outside the image"
        return GpuMat(h, w, self.cv_type, self.data + y * self.step + x * elem_size(self.cv_type), self.step,
                      owner=self.owner)

    def row(self, i):
        return GpuMat(1, self.cols, self.cv_type, self.data + i * self.step, self.step, owner=self.owner)

    def image2d(self):
        return capi.Image2D(self.data, self.cols, self.rows, self.step, getattr(self, "uv_offset", 0))

    def nv12_roi(self, x, y, w, h):
        """Crop of an NV12 luma view (rows = luma height; the UV plane follows it, or sits at self.uv_offset): a view
        like any other crop, plus the luma -> chroma offset the read stage needs.  x, y, w, h must be even."""
        x, y, w, h = int(x), int(y), int(w), int(h)
        if (x | y | w | h) & 1:
            raise ValueError("NV12 crops need even x, y, width and height")
        if x < 0 or y < 0 or x + w > self.cols or y + h > self.rows:
            raise ValueError("ROI outside the matrix")
        parent_uv = getattr(self, "uv_offset", 0) or self.rows * self.step
        m = GpuMat(h, w, self.cv_type, self.data + y * self.step + x * elem_size(self.cv_type), self.step, owner=self.owner)
        m.uv_offset = parent_uv + (y // 2 - y) * self.step
        return m


def _scalar(vals, n=4):
    vals = list(vals) if hasattr(vals, "__len__") else [vals]
    return [float(v) for v in vals] + [0.0] * (n - len(vals))


def cvScalar_set(cv_type, value):
    """cvGS::cvScalar_set<T>(value) (reference include/cvGPUSpeedupHelpers.cuh:23-37)."""
    return [float(value)] * type_cn(cv_type)


# ---- IOps -----------------------------------------------------------------------------------------------
class ReadIOp:
    def __init__(self, kind, src_type, mats, used_planes=None, dsize=None, ar=IGNORE_AR, background=None,
                 yuv=None, table=None, batch=None):
        self.kind, self.src_type, self.mats = kind, src_type, list(mats) if mats is not None else None
        self.batch = batch if batch is not None else len(self.mats)
        self.used_planes = self.batch if used_planes is None else int(used_planes)
        self.dsize = dsize  # (width, height)
        self.ar = ar
        self.background = _scalar(background if background is not None else [0.0] * 4)
        self.yuv = yuv or (capi.YUV_FULL, capi.BT601, 0)
        self.table = table  # device pointer of a prepared plane table (or None)
        self.warp = None    # WARP kinds: batch x 9 floats, the inverse transforms

    def out_type(self):
        if self.kind == capi.READ_PIXEL:
            return self.src_type
        if self.kind in (capi.READ_RESIZE_LINEAR, capi.READ_WARP_AFFINE, capi.READ_WARP_PERSPECTIVE):
            return make_type(DEPTH_32F, type_cn(self.src_type))  # reference :227: CV_32F of same channels
        return make_type(DEPTH_32F, 4 if self.yuv[2] else 3)


class PointwiseIOp:
    """One or more pointwise stages with an input and an output type (fk::Unary / fk::Binary)."""

    def __init__(self, in_type, out_type, ops):
        self.in_type, self.out_type, self.ops = in_type, out_type, ops  # ops: [(opcode, aux, operand[4])]


class WriteIOp:
    def __init__(self, kind, dst_type, data=0, width=0, height=0, step=0, planes=0, planes2d=None, keep=None):
        self.kind, self.dst_type, self.data = kind, dst_type, data
        self.width, self.height, self.step, self.planes, self.planes2d = width, height, step, planes, planes2d
        self.keep = keep
        self.mirrors = []  # device pointers of further tensors that receive the same values (cvgs_write_desc.mirrors)

    def mirrored_to(self, pointers):
        """The P2P exchange of the sharded batched-crop path: the same values also go to these tensors (peers' copies)."""
        self.mirrors = [int(p) for p in pointers]
        return self


def resize(src_type, interp, mats, dsize, used_planes=None, background=None, ar=IGNORE_AR, fx=0.0, fy=0.0):
    """cvGS::resize<T, INTER_F, NPtr, AR>(array<GpuMat,N>, dsize, usedPlanes, background) and the single-image
    cvGS::resize<T, INTER_F>(GpuMat, dsize, fx, fy) (reference include/cvGPUSpeedup.cuh:209-245)."""
    if interp not in SUPPORTED_INTERPOLATIONS:
        raise ValueError("Interpolation type not supported yet.")
    single = isinstance(mats, GpuMat)
    mats = [mats] if single else list(mats)
    w, h = int(dsize[0]), int(dsize[1])
    if single and (w == 0 or h == 0):  # dsize from fx, fy like cv::resize
        w = int(round(mats[0].cols * fx))
        h = int(round(mats[0].rows * fy))
    return ReadIOp(capi.READ_RESIZE_LINEAR, src_type, mats, used_planes, (w, h), ar, background)


WARP_AFFINE, WARP_PERSPECTIVE = 0, 1  # fk::WarpType


def invert_affine(m):
    """cv::invertAffineTransform on a 2x3 double matrix (the formula of OpenCV's imgwarp.cpp)."""
    (a, b, tx), (c, d, ty) = [[float(v) for v in row] for row in m]
    det = a * d - b * c
    det = 1.0 / det if det != 0.0 else 0.0
    a11, a22, a12, a21 = d * det, a * det, -b * det, -c * det
    return [[a11, a12, -a11 * tx - a12 * ty], [a21, a22, -a21 * tx - a22 * ty]]


def invert_3x3(m):
    """cv::Mat::inv() of a 3x3 double matrix: OpenCV's closed form (adjugate / determinant); singular -> zeros."""
    s = [[float(v) for v in row] for row in m]
    det = (s[0][0] * (s[1][1] * s[2][2] - s[1][2] * s[2][1]) - s[0][1] * (s[1][0] * s[2][2] - s[1][2] * s[2][0]) +
           s[0][2] * (s[1][0] * s[2][1] - s[1][1] * s[2][0]))
    if det == 0.0:
        return [[0.0] * 3 for _ in range(3)]
    d = 1.0 / det
    return [[(s[1][1] * s[2][2] - s[1][2] * s[2][1]) * d, (s[0][2] * s[2][1] - s[0][1] * s[2][2]) * d,
             (s[0][1] * s[1][2] - s[0][2] * s[1][1]) * d],
            [(s[1][2] * s[2][0] - s[1][0] * s[2][2]) * d, (s[0][0] * s[2][2] - s[0][2] * s[2][0]) * d,
             (s[0][2] * s[1][0] - s[0][0] * s[1][2]) * d],
            [(s[1][0] * s[2][1] - s[1][1] * s[2][0]) * d, (s[0][1] * s[2][0] - s[0][0] * s[2][1]) * d,
             (s[0][0] * s[1][1] - s[0][1] * s[1][0]) * d]]


def warp(warp_type, src_type, mats, transforms, dsize, used_planes=None, default_value=None):
    """cvGS::warp<WT, InputType[, BATCH]>(input(s), transform_matrix(-ces) (forward, CV_64FC1), dstSize
    [, usedPlanes, defaultValue]) (reference include/cvGPUSpeedup.cuh:288-442): the matrices are inverted on the
    host in double (cv::invertAffineTransform / cv::Mat::inv) and narrowed to float."""
    single = isinstance(mats, GpuMat)
    mats = [mats] if single else list(mats)
    transforms = [transforms] if single else list(transforms)
    used = len(mats) if used_planes is None else int(used_planes)
    flat = []
    for i in range(len(mats)):
        if i < used:
            if mats[i].cv_type != src_type:
                raise RuntimeError("Input type does not match the input type of the operation.")
            if warp_type == WARP_AFFINE:
                inv = invert_affine(transforms[i]) + [[0.0, 0.0, 1.0]]
            else:
                inv = invert_3x3(transforms[i])
            flat += [struct.unpack("f", struct.pack("f", v))[0] for row in inv for v in row]
        else:
            flat += [0.0] * 9
    kind = capi.READ_WARP_AFFINE if warp_type == WARP_AFFINE else capi.READ_WARP_PERSPECTIVE
    # dsize: one (width, height) for every plane, or a list of them (the std::array<cv::Size, BATCH> overloads, reference :381-401)
    sizes = None
    if len(dsize) and hasattr(dsize[0], "__len__"):
        sizes = [(int(w), int(h)) for (w, h) in dsize]
        if len(sizes) != len(mats):
            raise ValueError("one destination size per plane")
        dsize = (max(w for w, _ in sizes), max(h for _, h in sizes))
    rd = ReadIOp(kind, src_type, mats[:used] + [mats[0]] * (len(mats) - used), used, (int(dsize[0]), int(dsize[1])),
                 IGNORE_AR, default_value)
    rd.warp = flat
    rd.warp_sizes = sizes
    return rd


def cast(in_type, out_type):
    """fk::Cast<I, O>: static_cast per channel (truncating), as the reference's warp tests use it
    (tests/warping/test_warping_opencv.cu:63)."""
    if type_cn(in_type) != type_cn(out_type):
        raise ValueError("Cast cannot change the number of channels")
    return PointwiseIOp(in_type, out_type, [(capi.OP_CAST_TRUNC, type_depth(out_type), None)])


def read_nv12(mat, dsize=None, color_range=capi.YUV_FULL, primaries=capi.BT709, alpha=True, layout=capi.YUV_NV12):
    """fk::ReadYUV<NV12> + fk::ConvertYUVToRGB<NV12, range, primaries, alpha, floatN>, optionally as the
    BackIOp of fk::Resize<INTER_LINEAR> (reference tests/resize/test_fused_resize.cu:141-143).
    `mat` is the CV_8UC1 luma view (rows = luma height); the UV plane follows it in memory.  layout = YUV_P010: the luma
    view is CV_16UC1 (10-bit codes in the high bits of 16-bit samples) and R, G, B come out on the 0..1023 scale."""
    kind = capi.READ_NV12 if dsize is None else capi.READ_NV12_RESIZE_LINEAR
    mats = [mat] if isinstance(mat, GpuMat) else list(mat)  # a list = N crops (GpuMat.nv12_roi) of decoder surfaces, one launch
    rd = ReadIOp(kind, make_type(DEPTH_16U if layout == capi.YUV_P010 else DEPTH_8U, 1), mats, len(mats), dsize, IGNORE_AR, None,
                 (color_range, primaries, 1 if alpha else 0))
    rd.yuv_layout = layout  # fk::ReadYUV<PF>: NV12 (the reference's), NV21, I420, YV12, P010
    return rd


def convertTo(in_type, out_type, alpha=None, beta=None):
    """cvGS::convertTo<I,O>([alpha[,beta]]) (reference include/cvGPUSpeedup.cuh:74-129): integral outputs go
    through float: cast -> mul -> (add) -> saturate."""
    if type_cn(in_type) != type_cn(out_type):
        raise ValueError("convertTo does not support changing the number of channels")
    od = type_depth(out_type)
    if alpha is None:
        return PointwiseIOp(in_type, out_type, [(capi.OP_CAST, od, None)])
    # CV_16F (this engine's half hand-off type) is storage only: computed in float, rounded once at the end
    integral = od in (DEPTH_8U, DEPTH_8S, DEPTH_16U, DEPTH_16S, DEPTH_32S, DEPTH_16F)
    mid = DEPTH_32F if integral else od  # float for integral outputs, the output type (32F / 64F) otherwise
    f32 = lambda v: struct.unpack("f", struct.pack("f", float(v)))[0]  # the reference's parameters are `float`
    ops = [(capi.OP_CAST, mid, None), (capi.OP_MUL, 0, [f32(alpha)] * 4)]
    if beta is not None:
        ops.append((capi.OP_ADD, 0, [f32(beta)] * 4))
    if integral:
        ops.append((capi.OP_CAST, od, None))
    return PointwiseIOp(in_type, out_type, ops)


def _binary(opcode, cv_type, scalar):
    return PointwiseIOp(cv_type, cv_type, [(opcode, 0, _scalar(scalar))])


def multiply(cv_type, scalar):
    return _binary(capi.OP_MUL, cv_type, scalar)


def subtract(cv_type, scalar):
    return _binary(capi.OP_SUB, cv_type, scalar)


def divide(cv_type, scalar):
    return _binary(capi.OP_DIV, cv_type, scalar)


def add(cv_type, scalar):
    return _binary(capi.OP_ADD, cv_type, scalar)


_SWAP3, _ID3 = 2 | (1 << 2) | (0 << 4), 0 | (1 << 2) | (2 << 4)


def _alpha_max(depth):
    return {DEPTH_8U: 255.0, DEPTH_16U: 65535.0, DEPTH_32F: 1.0}[depth]


def cvtColor(code, in_type, out_type=None):
    """cvGS::cvtColor<CODE, I, O=I>() (reference include/cvGPUSpeedup.cuh:151-161)."""
    out_type = in_type if out_type is None else out_type
    d = type_depth(in_type)
    if d not in (DEPTH_8U, DEPTH_16U, DEPTH_32F) or type_depth(out_type) != d:
        raise ValueError("Wrong CV_TYPE_DEPTH, it has to be CV_8U, or CV_16U or CV_32F")
    if code not in SUPPORTED_COLOR_CONVERSIONS:
        raise ValueError("Color conversion type not supported yet.")
    icn, ocn = type_cn(in_type), type_cn(out_type)
    if code in (0, 2):
        want, op = (3, 4), (capi.OP_ADD_ALPHA, _ID3 if code == 0 else _SWAP3, [_alpha_max(d)] * 4)
    elif code in (1, 3):
        want, op = (4, 3), (capi.OP_DROP_ALPHA, _ID3 if code == 1 else _SWAP3, None)
    elif code == 4:
        want, op = (3, 3), (capi.OP_REORDER, _SWAP3, None)
    elif code == 5:
        want, op = (4, 4), (capi.OP_REORDER, _SWAP3 | (3 << 6), None)
    elif code in (6, 10):
        want, op = (3 if code == 6 else 4, 1), (capi.OP_GRAY, _SWAP3, None)
    else:
        want, op = (3 if code == 7 else 4, 1), (capi.OP_GRAY, _ID3, None)
    if (icn, ocn) != want:
        raise ValueError("channel counts do not match the colour conversion code")
    return PointwiseIOp(in_type, out_type, [op])


def split(out_type, output, plane_dims=None):
    """cvGS::split<O>(GpuMat tensor, Size plane) -> TensorSplit (NCHW), or split<O>(vector<GpuMat>) /
    (array<vector<GpuMat>,N>) -> SplitWrite (reference include/cvGPUSpeedup.cuh:163-192)."""
    cn = type_cn(out_type)
    if isinstance(output, GpuMat):
        w, h = int(plane_dims[0]), int(plane_dims[1])
        assert output.cols % (w * h) == 0 and output.cols // (w * h) == cn, \
            "Each row of the GpuMat should contain as many planes as width / (planeDims.width * planeDims.height)"
        return WriteIOp(capi.WRITE_TENSOR_SPLIT, out_type, output.data, w, h, 0, output.rows, keep=output)
    planes = list(output)
    if planes and isinstance(planes[0], GpuMat):
        planes = [planes]
    flat = [m for img in planes for m in img]
    if cn < 2:
        raise ValueError("Split operations can only be used with types of 2, 3 or 4 channels.")
    arr = (capi.Image2D * len(flat))(*[m.image2d() for m in flat])
    return WriteIOp(capi.WRITE_SPLIT_2D, out_type, 0, flat[0].cols, flat[0].rows, 0, len(planes), planes2d=arr,
                    keep=flat)


def split_tensor(out_type, data_ptr, width, height, planes, keep=None):
    """cvGS::split<O>(fk::RawPtr<_3D, base>) (reference :194-197)."""
    return WriteIOp(capi.WRITE_TENSOR_SPLIT, out_type, int(data_ptr), width, height, 0, planes, keep=keep)


def splitT(out_type, data_ptr, width, height, planes, keep=None):
    """cvGS::splitT<O>(fk::RawPtr<T3D, base>) -> TensorTSplit, CNHW (reference :199-202)."""
    return WriteIOp(capi.WRITE_TENSOR_T_SPLIT, out_type, int(data_ptr), width, height, 0, planes, keep=keep)


def write(out_type, output, plane_dims=None):
    """cvGS::write<O>(GpuMat) -> PerThreadWrite<_2D>; write<O>(GpuMat, Size) -> PerThreadWrite<_3D>
    (reference include/cvGPUSpeedup.cuh:449-457)."""
    if plane_dims is None:
        return WriteIOp(capi.WRITE_PIXEL_2D, out_type, output.data, output.cols, output.rows, output.step, 1,
                        keep=output)
    w, h = int(plane_dims[0]), int(plane_dims[1])
    return WriteIOp(capi.WRITE_PIXEL_3D, out_type, output.data, w, h, 0, output.rows, keep=output)


def write_batch(out_type, outputs):
    """PerThreadWrite<_2D> per batch element (array of GpuMats)."""
    outs = list(outputs)
    arr = (capi.Image2D * len(outs))(*[m.image2d() for m in outs])
    return WriteIOp(capi.WRITE_PIXEL_2D_BATCH, out_type, 0, outs[0].cols, outs[0].rows, 0, len(outs), planes2d=arr,
                    keep=outs)


# ---- lowering ---------------------------------------------------------------------------------------------
class LoweredChain:
    """A cvgs_chain_desc plus everything that must stay alive while it is in use."""

    def __init__(self, desc, keep):
        self.desc, self.keep = desc, keep


def lower(iops, flags=0):
    """Type-check the IOp list like the reference's template machinery does and emit ONE chain descriptor."""
    iops = list(iops)
    if len(iops) < 2 or not isinstance(iops[0], ReadIOp) or not isinstance(iops[-1], WriteIOp):
        raise ValueError("a chain is Read, pointwise..., Write")
    rd, wr = iops[0], iops[-1]
    ch = capi.new_chain()
    ch.flags = flags
    keep = [rd.mats, wr.keep, wr.planes2d]
    r = ch.read
    r.kind, r.src_type, r.batch, r.used_planes = rd.kind, rd.src_type, rd.batch, rd.used_planes
    if rd.table is not None:
        r.src = int(rd.table)
        r.flags = capi.READ_FLAG_TABLE_ON_DEVICE
        # ABI 6: a device-table chain states the byte range its table's planes read, so that cvgs_execute_many can check the chains of a tick
        # for independence (the host cannot see into the table).  rd.table_hull = (lo, hi) states it; rd.table_vouched = True vouches instead
        # (tables rewritten in place); otherwise it is computed from the host views the table was built from (cvgs_plane_table_hull).
        hull = getattr(rd, "table_hull", None)
        if getattr(rd, "table_vouched", False):
            r.flags |= capi.READ_FLAG_TABLE_SOURCES_VOUCHED
        elif hull is None and rd.mats and all(m is not None for m in rd.mats):
            hull = table_hull(rd)
        if hull is not None:
            r.table_src_lo, r.table_src_hi = int(hull[0]), int(hull[1])
    else:
        arr = (capi.Image2D * rd.batch)()
        for i, m in enumerate(rd.mats):
            if m.cv_type != rd.src_type:
                raise RuntimeError("Input type does not match the input type of the operation.")
            arr[i] = m.image2d()
        keep.append(arr)
        r.src = C.cast(arr, C.c_void_p).value
    if rd.dsize is not None:
        r.dst_width, r.dst_height = rd.dsize
    r.aspect_ratio = rd.ar
    for i in range(4):
        r.background[i] = rd.background[i]
    r.yuv_range, r.yuv_primaries, r.yuv_alpha = rd.yuv
    r.yuv_layout = getattr(rd, "yuv_layout", 0)
    if rd.warp is not None:
        wm = (C.c_float * len(rd.warp))(*rd.warp)
        keep.append(wm)
        r.warp_matrices = C.cast(wm, C.POINTER(C.c_float))
        if getattr(rd, "warp_sizes", None):
            ws = (C.c_int32 * (2 * len(rd.warp_sizes)))(*[v for wh in rd.warp_sizes for v in wh])
            keep.append(ws)
            r.warp_dst_sizes = C.cast(ws, C.POINTER(C.c_int32))
    cur = rd.out_type()
    n = 0
    for iop in iops[1:-1]:
        if not isinstance(iop, PointwiseIOp):
            raise ValueError("only pointwise IOps may sit between the read and the write")
        if iop.in_type != cur:
            raise TypeError("IOp input type %d does not match the previous output type %d" % (iop.in_type, cur))
        for opcode, aux, operand in iop.ops:
            if n >= capi.MAX_OPS:
                raise ValueError("too many pointwise stages")
            ch.ops[n].opcode, ch.ops[n].aux = opcode, aux
            for i in range(4):
                ch.ops[n].operand[i] = operand[i] if operand is not None else 0.0
                ch.ops[n].operand_d[i] = operand[i] if operand is not None else 0.0
            n += 1
        cur = iop.out_type
    ch.n_ops = n
    if wr.dst_type != cur:
        raise TypeError("write type %d does not match the chain's output type %d" % (wr.dst_type, cur))
    w = ch.write
    w.kind, w.dst_type, w.data = wr.kind, wr.dst_type, wr.data
    w.width, w.height, w.step, w.planes = wr.width, wr.height, wr.step, wr.planes
    if wr.planes2d is not None:
        w.planes2d = C.cast(wr.planes2d, C.c_void_p).value
    if getattr(wr, "mirrors", None):
        arr = (C.c_void_p * len(wr.mirrors))(*wr.mirrors)
        keep.append(arr)
        w.mirrors = C.cast(arr, C.POINTER(C.c_void_p))
        w.n_mirrors = len(wr.mirrors)
    return LoweredChain(ch, keep)


def stream_handle(stream):
    """cv::cuda::StreamAccessor::getStream: accept a raw hipStream_t int, a torch stream, or None."""
    if stream is None:
        return 0
    if isinstance(stream, int):
        return stream
    return int(stream.cuda_stream)


_ATTACHED = {}  # stream handle -> (Queue, submit flags): cvGS::attachQueue


def attachQueue(stream, queue, deferWait=False, minGroup=0):
    """cvGS::attachQueue(stream, queue): executeOperations(stream, ...) on this stream goes through the queue from now on, stream-ordered
    (cvgs_queue_submit_on, hybrid latency policy: groups below minGroup -- default 8 -- stay launches); deferWait: the stream is not held
    on each batch, fence(stream) orders the consumer."""
    prev = _ATTACHED.get(stream_handle(stream))
    if prev is not None:
        _flush(stream, prev)  # (calls a previous attachment recorded)
    _ATTACHED[stream_handle(stream)] = [queue, Queue.HYBRID | (Queue.DEFER_WAIT if deferWait else 0) | ((minGroup & 0xff) << 8), None]


def attachQueueTicks(stream, queue, tick=16):
    """cvGS::attachQueueTicks(stream, queue, tick): executeOperations(stream, ...) calls on this stream are RECORDED and submitted `tick` at
    a time behind one gate (cvgs_queue_submit_many_on, deferred waits); fence(stream) / detachQueue(stream) submit what is pending.
    Sources and tensors of recorded calls are in flight until the fence."""
    attachQueue(stream, queue, deferWait=True)
    a = _ATTACHED[stream_handle(stream)]
    a += [max(1, min(int(tick), 4 * 64)), []]


def recordTicks(stream, tick=16):
    """cvGS::recordTicks(stream, tick): recorded ticks with NO queue -- the recorded calls are launched `tick` at a time as ONE multi-chain
    kernel (cvgs_execute_many), strictly stream-ordered; fence(stream) / stopRecording(stream) launch what is pending."""
    a = _ATTACHED.get(stream_handle(stream))
    if a is not None:
        _flush(stream, a)
    _ATTACHED[stream_handle(stream)] = [None, 0, None, max(1, min(int(tick), 128)), []]


def stopRecording(stream):
    detachQueue(stream)


def _flush(stream, a):
    if len(a) < 5 or not a[4]:
        return
    pending, a[4] = a[4], []
    if a[0] is None:  # no queue: one cvgs_execute_many launch per <= 128 chains
        lib = capi.load_library()
        for base in range(0, len(pending), 128):
            group = pending[base:base + 128]
            capi.check(lib.cvgs_execute_many(pack_chains(group), len(group), stream_handle(stream)))
        return
    for base in range(0, len(pending), 64):
        group = pending[base:base + 64]
        t = a[0].submit_many_on(stream, Queue.chain_pointers(group), len(group), a[1])
        if t != Queue.TICKET_DIRECT:
            a[2] = t


def detachQueue(stream):
    a = _ATTACHED.get(stream_handle(stream))
    if a is not None:
        _flush(stream, a)
    _ATTACHED.pop(stream_handle(stream), None)


def fence(stream):
    a = _ATTACHED.get(stream_handle(stream))
    if a:
        _flush(stream, a)
    if a and a[2] is not None:
        a[0].stream_wait(a[2], stream)
        a[2] = None


def executeOperations(stream, *iops, flags=0):
    """cvGS::executeOperations(stream, iops...) (reference include/cvGPUSpeedup.cuh:464-473): one kernel,
    asynchronous on `stream`, never synchronises.  On a stream attached to a queue (attachQueue): the same contract through
    cvgs_queue_submit_on."""
    lib = capi.load_library()
    lowered = lower(iops, flags)
    a = _ATTACHED.get(stream_handle(stream)) if _ATTACHED else None
    if a is not None and len(a) >= 5:  # recorded ticks
        a[4].append(lowered)
        if len(a[4]) >= a[3]:
            _flush(stream, a)
        return lowered
    if a is not None:
        t = a[0].submit_lowered_on(stream, lowered, a[1])
        if (a[1] & Queue.DEFER_WAIT) and t != Queue.TICKET_DIRECT:
            a[2] = t
        return lowered
    capi.check(lib.cvgs_execute(C.byref(lowered.desc), stream_handle(stream)))
    return lowered


def pack_chains(lowered_chains):
    """A contiguous cvgs_chain_desc[n] for cvgs_execute_many from LoweredChain objects (which must stay alive)."""
    arr = (capi.ChainDesc * len(lowered_chains))()
    for i, lc in enumerate(lowered_chains):
        C.memmove(C.byref(arr[i]), C.byref(lc.desc), C.sizeof(capi.ChainDesc))
    return arr


def executeMany(stream, chains, flags=0):
    """cvgs_execute_many: `chains` = list of IOp lists (independent chains: own frames, crop lists, output tensors);
    same-shape K1 chains run as ONE launch, bit-identical to one executeOperations per chain."""
    lib = capi.load_library()
    lowered = [lower(iops, flags) for iops in chains]
    arr = pack_chains(lowered)
    capi.check(lib.cvgs_execute_many(arr, len(lowered), stream_handle(stream)))
    return lowered, arr


def executeOperations_io(input_mat, output_mat, stream, *iops, flags=0):
    """executeOperations(GpuMat in, [GpuMat out,] stream, iops...) (reference :475-503): a PerThreadRead is
    prepended and, if `output_mat` is given, a PerThreadWrite appended."""
    first = iops[0]
    in_type = first.in_type if isinstance(first, PointwiseIOp) else first.dst_type
    chain = [ReadIOp(capi.READ_PIXEL, in_type, [input_mat], 1)] + list(iops)
    if output_mat is not None:
        last = iops[-1]
        chain.append(write(last.out_type, output_mat))
    return executeOperations(stream, *chain, flags=flags)


def executeOperations_batch(inputs, stream, *iops, active_batch=None, default_value=None, output=None,
                            output_plane=None, flags=0):
    """executeOperations(array<GpuMat,N> in, [activeBatch, default,] [GpuMat out, Size plane,] stream, iops...)
    (reference :506-583)."""
    inputs = list(inputs)
    first = iops[0]
    in_type = first.in_type if isinstance(first, PointwiseIOp) else first.dst_type
    chain = [ReadIOp(capi.READ_PIXEL, in_type, inputs, active_batch, None, IGNORE_AR, default_value)] + list(iops)
    if output is not None:
        chain.append(write(iops[-1].out_type, output, output_plane))
    return executeOperations(stream, *chain, flags=flags)


def kernel_name(*iops, flags=0):
    lib = capi.load_library()
    lowered = lower(iops, flags)
    buf = C.create_string_buffer(128)
    capi.check(lib.cvgs_kernel_name(C.byref(lowered.desc), buf, 128))
    return buf.value.decode()


def table_hull(read_iop):
    """(lo, hi): the byte range that holds everything the planes of a read stage read (cvgs_plane_table_hull) -- what a chain that passes
    the built device table states in read.table_src_lo / table_src_hi."""
    lib = capi.load_library()
    keep_table = read_iop.table
    read_iop.table = None
    try:
        lowered = lower([read_iop, WriteIOp(capi.WRITE_PIXEL_3D, read_iop.out_type(), 16, 1, 1, 0, read_iop.batch)])
    finally:
        read_iop.table = keep_table
    lo, hi = C.c_void_p(), C.c_void_p()
    capi.check(lib.cvgs_plane_table_hull(C.byref(lowered.desc.read), C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def build_plane_table(read_iop):
    """Host bytes of the device plane table for a read stage (cvgs_plane_table_build)."""
    lib = capi.load_library()
    lowered = lower([read_iop, WriteIOp(capi.WRITE_PIXEL_3D, read_iop.out_type(), 16, 1, 1, 0, read_iop.batch)])
    n = lib.cvgs_plane_table_bytes(read_iop.batch)
    buf = (C.c_uint8 * n)()
    capi.check(lib.cvgs_plane_table_build(C.byref(lowered.desc.read), buf))
    return bytes(buf)


class CircularTensor:
    """cvGS::CircularTensor<I, O, COLOR_PLANES, BATCH, ORDER, CP_MODE> (reference include/cvGPUSpeedup.cuh:600-627)."""

    def __init__(self, in_type, out_elem_type, color_planes, batch, order, cp_mode=Standard, width=0, height=0,
                 device_id=0, mirrored=False, capturable=False):
        """mirrored=True: the opt-in mirrored-ring variant (cvgs_circular_create_ex, CVGS_CIRCULAR_MIRRORED): no shift
        traffic, but data() moves with every update."""
        self.in_type, self.elem_type, self.color_planes, self.batch = in_type, out_elem_type, color_planes, batch
        self.order, self.cp_mode, self.mirrored, self.capturable = order, cp_mode, mirrored, capturable
        self.handle = C.c_void_p(0)
        self.lib = capi.load_library()
        if width and height:
            self.Alloc(width, height, device_id)

    def Alloc(self, width, height, device_id=0):
        self.width, self.height = width, height
        capi.check(self.lib.cvgs_circular_create_ex(C.byref(self.handle), width, height, self.elem_type,
                                                    self.color_planes, self.batch, self.order, self.cp_mode, device_id,
                                                    (capi.CIRCULAR_MIRRORED if self.mirrored else 0) | (capi.CIRCULAR_CAPTURABLE if self.capturable else 0)))

    def update(self, stream, *iops, flags=0):
        """update(stream, GpuMat input, iops..., write) or update(stream, readIOp, iops..., write)."""
        iops = list(iops)
        if isinstance(iops[0], GpuMat):
            iops[0] = ReadIOp(capi.READ_PIXEL, self.in_type, [iops[0]], 1)
        lowered = lower(iops, flags)
        capi.check(self.lib.cvgs_circular_update(self.handle, C.byref(lowered.desc), stream_handle(stream)))
        return lowered

    def write_split(self, out_type):
        """fk::Write<fk::TensorSplit<O>>{tensor.ptr()} for this tensor."""
        return WriteIOp(capi.WRITE_TENSOR_SPLIT, out_type, 0, self.width, self.height, 0, self.batch)

    def write_splitT(self, out_type):
        return WriteIOp(capi.WRITE_TENSOR_T_SPLIT, out_type, 0, self.width, self.height, 0, self.batch)

    def write_packed(self, out_type):
        """fk::Write<fk::TensorWrite<O>>{tensor.ptr()} (COLOR_PLANES == 1)."""
        return WriteIOp(capi.WRITE_PIXEL_3D, out_type, 0, self.width, self.height, 0, self.batch)

    def data(self):
        return self.lib.cvgs_circular_data(self.handle)

    def nbytes(self):
        return self.lib.cvgs_circular_bytes(self.handle)

    def updates(self):
        return self.lib.cvgs_circular_updates(self.handle)

    def release(self):
        if self.handle:
            self.lib.cvgs_circular_destroy(self.handle)
            self.handle = C.c_void_p(0)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class Queue:
    """cvgs_queue_*: a device-side descriptor queue.  `submit(*iops)` has executeOperations' call shape (one call per frame,
    the same IOps) but no kernel launch per call: a resident server grid takes the batch from a ring, and consecutive batches
    overlap on the device.  Taken: K1's hot shape (batched 8U / 16U / 16S C3 / C4 resize -> [swap,] mul, sub, div -> fp32 or fp16 planar tensor) or
    the same behind crops of NV12 / NV21 decoder surfaces (read_nv12(..., dsize), 3 channels); a queue serves the kind of its
    first submit.  capi.CvgsError(CVGS_ERR_UNSUPPORTED) otherwise."""

    def __init__(self, device=-1, depth=0, idle_us=0.0, flags=0):  # device -1: the calling thread's current device
        self.lib = capi.load_library()
        self.handle = C.c_void_p()
        capi.check(self.lib.cvgs_queue_create(C.byref(self.handle), device, depth, float(idle_us), flags))
        self._keep = {}

    def submit(self, *iops, flags=0):
        lowered = lower(iops, flags)
        return self.submit_lowered(lowered)

    def submit_lowered(self, lowered):
        t = C.c_uint64()
        capi.check(self.lib.cvgs_queue_submit(self.handle, C.byref(lowered.desc), C.byref(t)))
        return t.value

    DEFER_WAIT, HYBRID, TICKET_DIRECT = 1, 2, (1 << 64) - 1

    @staticmethod
    def MIN_GROUP(n):
        return (n & 0xff) << 8

    def submit_on(self, stream, *iops, flags=0, submit_flags=0):
        """cvgs_queue_submit_on: executeOperations(stream, iops...) on the queue -- ordered behind everything already enqueued on
        `stream`, and (unless DEFER_WAIT) in front of everything enqueued on it afterwards; no host synchronisation.  Returns the
        ticket (TICKET_DIRECT when the HYBRID policy launched the chain directly on the stream)."""
        return self.submit_lowered_on(stream, lower(iops, flags), submit_flags)

    def submit_lowered_on(self, stream, lowered, submit_flags=0):
        t = C.c_uint64()
        capi.check(self.lib.cvgs_queue_submit_on(self.handle, C.byref(lowered.desc), stream_handle(stream), submit_flags, C.byref(t)))
        return t.value

    def submit_many_on(self, stream, ptr_array, n, submit_flags=0):
        """cvgs_queue_submit_many_on: n pre-lowered chains (chain_pointers) behind ONE gate on `stream`; returns the last ticket"""
        t = C.c_uint64()
        capi.check(self.lib.cvgs_queue_submit_many_on(self.handle, ptr_array, n, stream_handle(stream), submit_flags, C.byref(t)))
        return t.value

    def recover(self):
        """cvgs_queue_recover: after the watchdog has fired; returns the number of batches that were lost"""
        n = C.c_uint64()
        capi.check(self.lib.cvgs_queue_recover(self.handle, C.byref(n)))
        return n.value

    def submit_many(self, ptr_array, n):
        """ptr_array: (POINTER(ChainDesc) * n) of pre-lowered chains (see chain_pointers); returns the last ticket"""
        t = C.c_uint64()
        capi.check(self.lib.cvgs_queue_submit_many(self.handle, ptr_array, n, C.byref(t)))
        return t.value

    @staticmethod
    def chain_pointers(lowered_chains):
        arr = (C.POINTER(capi.ChainDesc) * len(lowered_chains))()
        for i, lc in enumerate(lowered_chains):
            arr[i] = C.pointer(lc.desc)
        return arr

    def wait(self, ticket, timeout_s=10.0):
        """the batch of `ticket` and every batch submitted before it are complete on return (batches finish in any order on the device)"""
        capi.check(self.lib.cvgs_queue_wait(self.handle, ticket, float(timeout_s)))

    def stream_wait(self, ticket, stream):
        """`stream` waits (on the device) for the batch of `ticket` and every batch submitted before it"""
        capi.check(self.lib.cvgs_queue_stream_wait(self.handle, ticket, stream_handle(stream)))

    def stats(self):
        out = (C.c_uint64 * 8)()
        capi.check(self.lib.cvgs_queue_stats(self.handle, out))
        keys = ("submitted", "completed", "server_launches", "janitor_rounds", "server_ticks_100MHz", "workgroups", "ring_slots", "error")
        d = dict(zip(keys, [int(v) for v in out]))
        d["host_writes_device_memory"] = bool(d["ring_slots"] >> 32)  # slots go through the PCIe BAR (else: staged by a copy kernel)
        d["ring_slots"] &= 0xffffffff
        return d

    def stream_handle(self):
        """hipStream_t of the server grid (HIP events around its launches, profilers)"""
        return self.lib.cvgs_queue_stream(self.handle)

    def profile(self):
        """instrumentation of the last retired server: ticks are 10 ns"""
        out = (C.c_uint64 * 16)()
        capi.check(self.lib.cvgs_queue_profile(self.handle, out))
        v = [int(x) for x in out]
        return {"worker0": {"tasks": v[8], "find_us": v[9] / 100.0, "rows_us": v[10] / 100.0, "drain_us": v[11] / 100.0, "arrive_us": v[5] / 100.0, "first_to_last_us": v[4] / 100.0, "idle_polls": v[12], "slot_load_us": v[0] / 100.0, "find_iterations": v[1]},
                "host_submit_ns": {"ring_wait": v[13]},
                "ring_full_waits": v[6], "complete_behind_an_incomplete_head_avg": v[7]}

    def destroy(self):
        if self.handle:
            self.lib.cvgs_queue_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
