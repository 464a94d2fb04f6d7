// k_warp.hip -- cvGS::warp: affine / perspective warping fused in front of a pointwise chain and a write stage
// (reference include/cvGPUSpeedup.cuh:288-442 -> fk::Warping<WT, Read<PerThreadRead<_2D,T>>>, batched with
// usedPlanes / default value like the resize; tests/warping/test_warping_opencv.cu).  SURVEY.md 8(f)4.
// One thread = one output pixel of plane z: destination (x,y) -> source position through the inverse transform
// (strict fp32, products and sums in the written order, IEEE division), INTER_LINEAR taps as in the resize, zero
// outside the source.  Interpreted program and generic write stage: this is not a hot-path kernel.
#include <type_traits>

#include "k_taps.hpp"

namespace cvgs {

template <int NPL>
struct WarpKernArgs {
    ChainArgs c;
    WarpPlane planes[NPL > 0 ? NPL : 1];
};
static_assert(sizeof(WarpKernArgs<kInlineWarp>) <= 4096, "kernel-argument block must fit 4 KB");

template <int NPL>
__global__ __launch_bounds__(256) void k_warp(const WarpKernArgs<NPL> a, const WarpPlane* __restrict__ table) {
    const ChainArgs& c = a.c;
    const ReadArgs& r = c.read;
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int z = blockIdx.z;
    if (x >= r.dst_w || y >= r.dst_h) return;
    WarpPlane P;
    if constexpr (NPL == 0) P = table[z];
    else P = a.planes[z];
    if (x >= P.dw || y >= P.dh) return; // this plane's own destination size (the launch covers the largest)

    Px p;
    p.v[0] = p.v[1] = p.v[2] = p.v[3] = 0.f;
    int depth = CVGS_DEPTH_32F, cn = r.out_cn;
    if (z >= r.used) {
#pragma unroll
        for (int k = 0; k < 4; ++k) p.v[k] = r.bg[k]; // fk::BatchRead default value, then the whole chain
    } else {
        const float fx = (float)x, fy = (float)y;
        float sx = (P.m[0] * fx + P.m[1] * fy) + P.m[2];
        float sy = (P.m[3] * fx + P.m[4] * fy) + P.m[5];
        if (r.kind == CVGS_READ_WARP_PERSPECTIVE) {
            const float w = (P.m[6] * fx + P.m[7] * fy) + P.m[8];
            sx = sx / w;
            sy = sy / w;
        }
        if (sx >= 0.f && sx < (float)P.w && sy >= 0.f && sy < (float)P.h) {
            const int x1 = (int)floorf(sx), y1 = (int)floorf(sy);
            const int x2 = x1 + 1, y2 = y1 + 1;
            const int x2r = min(x2, P.w - 1), y2r = min(y2, P.h - 1);
            Px p00, p10, p01, p11;
            const uint8_t* ra = P.data + (size_t)y1 * (size_t)P.step;
            const uint8_t* rb = P.data + (size_t)y2r * (size_t)P.step;
            load_px(ra, r.depth, r.cn, x1, p00);
            load_px(ra, r.depth, r.cn, x2r, p10);
            load_px(rb, r.depth, r.cn, x1, p01);
            load_px(rb, r.depth, r.cn, x2r, p11);
            const float w00 = ((float)x2 - sx) * ((float)y2 - sy);
            const float w10 = (sx - (float)x1) * ((float)y2 - sy);
            const float w01 = ((float)x2 - sx) * (sy - (float)y1);
            const float w11 = (sx - (float)x1) * (sy - (float)y1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < r.cn) {
                    float acc = tap_f(p00.v[k], r.depth) * w00;
                    acc = acc + tap_f(p10.v[k], r.depth) * w10;
                    acc = acc + tap_f(p01.v[k], r.depth) * w01;
                    acc = acc + tap_f(p11.v[k], r.depth) * w11;
                    p.v[k] = acc;
                }
            }
        }
    }
    InterpProgInt::run(c.prog, p, depth, cn);
    const DstPlane* dst = c.write.table ? c.write.table : c.dst_inline;
    write_px(c.write, dst, x, y, z, p, depth, cn);
}

template <int NPL>
static hipError_t launch_warp_t(const ChainArgs& c, const WarpPlane* planes, int n, const WarpPlane* table, hipStream_t s) {
    WarpKernArgs<NPL> a;
    a.c = c;
    for (int i = 0; i < (NPL > 0 ? NPL : 1); ++i) a.planes[i] = (NPL > 0 && i < n) ? planes[i] : WarpPlane{};
    const dim3 block(64, 4, 1);
    const dim3 grid((c.read.dst_w + 63) / 64, (c.read.dst_h + 3) / 4, c.read.batch);
    hipLaunchKernelGGL(k_warp<NPL>, grid, block, 0, s, a, table);
    return hipGetLastError();
}

// ---- fast path: u8 C3/C4 sources -> program -> planar fp32 tensor (the batched "align N faces / N detections and
// hand them to the network" shape).  Same structure as the fast resize kernel: lane = output column, wave = one output
// row, blockIdx.y = plane; both horizontal taps of a source row arrive in ONE unaligned 8-byte load (two loads per
// pixel), compile-time pointwise program, non-temporal 256-byte planar rows.  Bit-identical to k_warp.
struct WarpGeom {
    uint32_t col_tiles;
    int32_t dst_w, dst_h, used, out_w, pad;
    int64_t img_stride, ch_stride;
    void* out; // float* or _Float16* (OT)
    int64_t row_pitch, img_pitch; // PACKED: bytes between output rows / images
};

template <int CN, int NPL, class Prog, bool PERSP, typename OT = float, bool PACKED = false>
__global__ __launch_bounds__(256) void k_warp_fast(const WarpKernArgs<NPL> a, const WarpPlane* __restrict__ table, const WarpGeom g) {
    const ChainArgs& c = a.c;
    const int z = (int)blockIdx.y;
    const int dst_w = g.dst_w, dst_h = g.dst_h, used = g.used, W = g.out_w;
    WarpPlane P;
    if constexpr (NPL == 0) P = table[z < used ? z : 0];
    else P = a.planes[z];
    asm volatile("" ::"s"(dst_w), "s"(dst_h), "s"(used), "s"(W), "s"(P.w), "s"(P.h), "s"(P.step), "s"(P.data), "s"(P.m[0]), "s"(P.m[1]),
                 "s"(P.m[2]), "s"(P.m[3]), "s"(P.m[4]), "s"(P.m[5]), "s"(P.m[6]), "s"(P.m[7]), "s"(P.m[8]));
    int col_tile = 0, row_tile = (int)blockIdx.x;
    if (g.col_tiles > 1) {
        col_tile = (int)(blockIdx.x % g.col_tiles);
        row_tile = (int)(blockIdx.x / g.col_tiles);
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int x = col_tile * 64 + lane;
    const int y = row_tile * 4 + wave;
    if (y >= dst_h || x >= dst_w) return;

    Px p;
    p.v[0] = p.v[1] = p.v[2] = p.v[3] = 0.f;
    if (z >= used) {
#pragma unroll
        for (int k = 0; k < 4; ++k) p.v[k] = c.read.bg[k];
    } else {
        const float fx = (float)x, fy = (float)y;
        float sx = (P.m[0] * fx + P.m[1] * fy) + P.m[2];
        float sy = (P.m[3] * fx + P.m[4] * fy) + P.m[5];
        if constexpr (PERSP) {
            const float w = (P.m[6] * fx + P.m[7] * fy) + P.m[8];
            sx = sx / w;
            sy = sy / w;
        }
        if (sx >= 0.f && sx < (float)P.w && sy >= 0.f && sy < (float)P.h) {
            const int x1 = (int)floorf(sx), y1 = (int)floorf(sy);
            const int x2 = x1 + 1, y2 = y1 + 1;
            const int y2r = min(y2, P.h - 1);
            const bool edge = x2 > P.w - 1;
            const int row_bytes = P.w * CN;
            const int o = x1 * CN;
            const bool tiny = row_bytes < 8; // uniform per plane
            const uint32_t ol = (uint32_t)(tiny ? o : min(o, row_bytes - 8));
            const int sh = (o - (int)ol) * 8;
            const gptr_u8 ra = (gptr_u8)P.data + (size_t)y1 * (size_t)P.step;
            const gptr_u8 rb = (gptr_u8)P.data + (size_t)y2r * (size_t)P.step;
            Win<1> va, vb;
            if (!tiny) {
                va = load_win<1>(ra + ol);
                vb = load_win<1>(rb + ol);
            } else {
                va = gather_win<CN, 1>(ra, o, row_bytes);
                vb = gather_win<CN, 1>(rb, o, row_bytes);
            }
            float p00[4], p10[4], p01[4], p11[4];
            unpack_pair<CN, SRC_U8>(shift_win<1>(va, sh), edge, p00, p10);
            unpack_pair<CN, SRC_U8>(shift_win<1>(vb, sh), edge, p01, p11);
            const float w00 = ((float)x2 - sx) * ((float)y2 - sy);
            const float w10 = (sx - (float)x1) * ((float)y2 - sy);
            const float w01 = ((float)x2 - sx) * (sy - (float)y1);
            const float w11 = (sx - (float)x1) * (sy - (float)y1);
#pragma unroll
            for (int k = 0; k < CN; ++k) {
                float acc = p00[k] * w00;
                acc = acc + p10[k] * w10;
                acc = acc + p01[k] * w01;
                acc = acc + p11[k] * w11;
                p.v[k] = acc;
            }
        }
    }
    int depth = CVGS_DEPTH_32F, cn = CN;
    Prog::run(c.prog, p, depth, cn);
    if constexpr (PACKED) { // NHWC: one 12 / 16-byte store per pixel, consecutive lanes write consecutive pixels
        uint8_t* const row = (uint8_t*)g.out + (int64_t)z * g.img_pitch + (int64_t)y * g.row_pitch;
        store_packed_px<CN, OT>((OT*)row + (int64_t)x * cn, p.v, cn);
    } else {
        OT* const orow = (OT*)g.out + (int64_t)z * g.img_stride + (int64_t)y * W;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < cn) st_nt(orow + (int64_t)k * g.ch_stride + x, p.v[k]); // OT = _Float16: the chain's trailing CAST(CV_16F)
    }
}

template <int CN, class Prog, bool PERSP, typename OT, bool PACKED = false>
static hipError_t launch_warp_fast_t(const ChainArgs& c, const WarpPlane* planes, int n, const WarpPlane* table, hipStream_t s) {
    WarpGeom g;
    g.col_tiles = (uint32_t)((c.read.dst_w + 63) / 64);
    g.dst_w = c.read.dst_w; g.dst_h = c.read.dst_h; g.used = c.read.used; g.out_w = c.write.width; g.pad = 0;
    g.img_stride = c.write.img_stride; g.ch_stride = c.write.ch_stride;
    g.out = c.write.data;
    const int64_t px_bytes = (int64_t)sizeof(OT) * c.write.cn;
    g.row_pitch = c.write.kind == CVGS_WRITE_PIXEL_2D ? c.write.step : c.write.width * px_bytes;
    g.img_pitch = c.write.kind == CVGS_WRITE_PIXEL_2D ? 0 : c.write.img_stride * px_bytes;
    const dim3 grid(g.col_tiles * (uint32_t)((c.read.dst_h + 3) / 4), (unsigned)c.read.batch);
    if (table) {
        WarpKernArgs<0> a;
        a.c = c;
        a.planes[0] = WarpPlane{};
        hipLaunchKernelGGL((k_warp_fast<CN, 0, Prog, PERSP, OT, PACKED>), grid, dim3(256), 0, s, a, table, g);
    } else {
        WarpKernArgs<kInlineWarp> a;
        a.c = c;
        for (int i = 0; i < kInlineWarp; ++i) a.planes[i] = i < n ? planes[i] : WarpPlane{};
        hipLaunchKernelGGL((k_warp_fast<CN, kInlineWarp, Prog, PERSP, OT, PACKED>), grid, dim3(256), 0, s, a, (const WarpPlane*)nullptr, g);
    }
    return hipGetLastError();
}

template <int CN, bool PERSP, typename OT>
static hipError_t launch_warp_fast_ot(int prog_id, const ChainArgs& c, const WarpPlane* planes, int n, const WarpPlane* table, hipStream_t s) {
    if (prog_id == 0) return launch_warp_fast_t<CN, ProgSwapMulSubDiv, PERSP, OT>(c, planes, n, table, s);
    if (prog_id == 1) return launch_warp_fast_t<CN, ProgMulSubDiv, PERSP, OT>(c, planes, n, table, s);
    return launch_warp_fast_t<CN, InterpProg, PERSP, OT>(c, planes, n, table, s);
}
// packed fp32 pixels (NHWC): interpreted program
template <int CN, bool PERSP>
static hipError_t launch_warp_fast_packed(const ChainArgs& c, const WarpPlane* planes, int n, const WarpPlane* table, hipStream_t s) {
    return launch_warp_fast_t<CN, InterpProg, PERSP, float, true>(c, planes, n, table, s);
}
template <int CN, bool PERSP>
static hipError_t launch_warp_fast_prog(bool f16, int prog_id, const ChainArgs& c, const WarpPlane* planes, int n, const WarpPlane* table,
                                        hipStream_t s) {
    if (f16) return launch_warp_fast_ot<CN, PERSP, _Float16>(prog_id, c, planes, n, table, s);
    return launch_warp_fast_ot<CN, PERSP, float>(prog_id, c, planes, n, table, s);
}

// 1 = took it, 0 = not eligible
static int try_warp_fast(const ChainArgs& c_in, const WarpPlane* planes, int n, const WarpPlane* table, hipStream_t s, bool dry_run,
                         LaunchInfo* info, hipError_t* err) {
    const ReadArgs& r = c_in.read;
    if (r.depth != CVGS_DEPTH_8U || (r.cn != 3 && r.cn != 4) || r.batch > 65535) return 0;
    const bool planar = c_in.write.kind == CVGS_WRITE_TENSOR_SPLIT || c_in.write.kind == CVGS_WRITE_TENSOR_T_SPLIT;
    const bool packed = c_in.write.kind == CVGS_WRITE_PIXEL_2D || c_in.write.kind == CVGS_WRITE_PIXEL_3D;
    if ((!planar && !packed) || c_in.write.data2) return 0;
    const bool f16 = c_in.write.depth == CVGS_DEPTH_16F;
    if (!f16 && c_in.write.depth != CVGS_DEPTH_32F) return 0;
    if (packed && f16) return 0;
    ChainArgs c_cut;
    if (f16) { // fp16 tensors: the trailing CAST(CV_16F) moves into the store
        if (c_in.prog.n < 1 || c_in.prog.opcode[c_in.prog.n - 1] != CVGS_OP_CAST) return 0;
        c_cut = c_in;
        c_cut.prog.n -= 1;
    }
    const ChainArgs& c = f16 ? c_cut : c_in;
    for (int k = 0; k < c.prog.n; ++k)
        if (c.prog.opcode[k] == CVGS_OP_CAST || c.prog.opcode[k] == CVGS_OP_CAST_TRUNC) return 0;
    const ProgArgs& p = c.prog;
    const int swap = r.cn == 3 ? (2 | (1 << 2) | (0 << 4)) : (2 | (1 << 2) | (0 << 4) | (3 << 6));
    int prog_id = 2;
    if (p.n == 4 && p.opcode[0] == CVGS_OP_REORDER && p.aux[0] == swap && p.opcode[1] == CVGS_OP_MUL && p.opcode[2] == CVGS_OP_SUB &&
        p.opcode[3] == CVGS_OP_DIV)
        prog_id = 0;
    else if (p.n == 3 && p.opcode[0] == CVGS_OP_MUL && p.opcode[1] == CVGS_OP_SUB && p.opcode[2] == CVGS_OP_DIV)
        prog_id = 1;
    const bool persp = r.kind == CVGS_READ_WARP_PERSPECTIVE;
    if (info) {
        static const char* names[2][2][3] = {
            {{"warp_affine_u8c3_swap_mul_sub_div", "warp_affine_u8c3_mul_sub_div", "warp_affine_u8c3_interp"},
             {"warp_affine_u8c4_swap_mul_sub_div", "warp_affine_u8c4_mul_sub_div", "warp_affine_u8c4_interp"}},
            {{"warp_perspective_u8c3_swap_mul_sub_div", "warp_perspective_u8c3_mul_sub_div", "warp_perspective_u8c3_interp"},
             {"warp_perspective_u8c4_swap_mul_sub_div", "warp_perspective_u8c4_mul_sub_div", "warp_perspective_u8c4_interp"}}};
        static const char* names16[2][2][3] = {
            {{"warp_affine_u8c3_swap_mul_sub_div_f16", "warp_affine_u8c3_mul_sub_div_f16", "warp_affine_u8c3_interp_f16"},
             {"warp_affine_u8c4_swap_mul_sub_div_f16", "warp_affine_u8c4_mul_sub_div_f16", "warp_affine_u8c4_interp_f16"}},
            {{"warp_perspective_u8c3_swap_mul_sub_div_f16", "warp_perspective_u8c3_mul_sub_div_f16", "warp_perspective_u8c3_interp_f16"},
             {"warp_perspective_u8c4_swap_mul_sub_div_f16", "warp_perspective_u8c4_mul_sub_div_f16", "warp_perspective_u8c4_interp_f16"}}};
        static const char* names_packed[2][2] = {{"warp_affine_u8c3_packed_f32", "warp_affine_u8c4_packed_f32"},
                                                 {"warp_perspective_u8c3_packed_f32", "warp_perspective_u8c4_packed_f32"}};
        info->kernel = packed ? names_packed[persp][r.cn == 4] : (f16 ? names16[persp][r.cn == 4][prog_id] : names[persp][r.cn == 4][prog_id]);
    }
    if (dry_run) return 1;
    if (packed) {
        if (r.cn == 3) *err = persp ? launch_warp_fast_packed<3, true>(c, planes, n, table, s) : launch_warp_fast_packed<3, false>(c, planes, n, table, s);
        else *err = persp ? launch_warp_fast_packed<4, true>(c, planes, n, table, s) : launch_warp_fast_packed<4, false>(c, planes, n, table, s);
        return 1;
    }
    if (r.cn == 3) *err = persp ? launch_warp_fast_prog<3, true>(f16, prog_id, c, planes, n, table, s) : launch_warp_fast_prog<3, false>(f16, prog_id, c, planes, n, table, s);
    else *err = persp ? launch_warp_fast_prog<4, true>(f16, prog_id, c, planes, n, table, s) : launch_warp_fast_prog<4, false>(f16, prog_id, c, planes, n, table, s);
    return 1;
}

int launch_warp(const ChainArgs& c, const WarpPlane* planes, int n, const WarpPlane* dev_table, uint32_t chain_flags, void* stream,
                bool dry_run, LaunchInfo* info) {
    const bool persp = c.read.kind == CVGS_READ_WARP_PERSPECTIVE;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (!(chain_flags & CVGS_CHAIN_FORCE_GENERIC) && try_warp_fast(c, planes, n, dev_table, s, dry_run, info, &e))
        return e == hipSuccess ? 0 : -(int)e - 1000;
    if (info) info->kernel = persp ? "warp_perspective_interp" : "warp_affine_interp";
    if (dry_run) return 0;
    if (dev_table) e = launch_warp_t<0>(c, nullptr, 0, dev_table, s);
    else if (n <= 4) e = launch_warp_t<4>(c, planes, n, nullptr, s);
    else e = launch_warp_t<kInlineWarp>(c, planes, n, nullptr, s);
    return e == hipSuccess ? 0 : -(int)e - 1000;
}

} // namespace cvgs
