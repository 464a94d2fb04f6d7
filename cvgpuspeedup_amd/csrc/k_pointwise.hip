// k_pointwise.hip -- thread-fused pointwise chains on u8 sources:
//   PerThreadRead<_2D, uchar{1,2,3,4}> [batched] -> SaturateCast/Mul/Sub/Div/... -> fp32 (or fp16) planar tensor
//   (TensorSplit / TensorTSplit, optionally mirrored into a second target) or packed fp32 / fp16 pixels (2D / 3D).
// The engine's counterpart of the reference's ENABLE_THREAD_FUSION=true fast path (reference
// include/cvGPUSpeedup.cuh:464-473): each thread owns FOUR x-adjacent pixels, reads them with one
// wide load (4*CN bytes) and writes 16-byte vectors.  Results are bit-identical to the interpreted kernel; chains it
// does not cover (other depths, integer outputs, SplitWrite) stay on k_generic.  Used by K5/K6/K7 and by the
// CircularTensor push of an un-resized frame (cfg #4).
#include "k_common.hpp"

namespace cvgs {

typedef uint32_t u32u __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) u32u* gp_u32;
typedef const __attribute__((address_space(1))) uint8_t* gp_u8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4u __attribute__((aligned(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16x4 f16x4u __attribute__((aligned(2)));

// four consecutive output elements in one non-temporal vector store (fp16: the chain's trailing CAST(CV_16F) is this
// round-to-nearest-even conversion)
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    f32x4 q = {a, b, c, d};
    __builtin_nontemporal_store(q, (f32x4u*)p);
}
__device__ __forceinline__ void store4(_Float16* p, float a, float b, float c, float d) {
    f16x4 q = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
    __builtin_nontemporal_store(q, (f16x4u*)p);
}
__device__ __forceinline__ void store1(float* p, float v) { *p = v; }
__device__ __forceinline__ void store1(_Float16* p, float v) { *p = (_Float16)v; }

using ProgCastMulSubDiv = StaticProg<CVGS_OP_CAST, CVGS_OP_MUL, CVGS_OP_SUB, CVGS_OP_DIV>;
using ProgCast = StaticProg<CVGS_OP_CAST>;

struct PwGeom {
    int32_t w, h, used, cn;
    int32_t packed;    // 1: packed pixels (PIXEL_2D / PIXEL_3D), 0: planar
    int32_t row_pitch; // packed: bytes between output rows
    int32_t row_pitch2, pad;
    int64_t img_stride, ch_stride, img_stride2, ch_stride2; // planar: elements; packed: img_stride in BYTES
    uint8_t* out;
    uint8_t* out2;
};

template <int CN, int NPL, class Prog, typename OT>
__global__ __launch_bounds__(256) void k_pointwise4(const KernArgs<NPL> a, const PwGeom g) {
    const ChainArgs& c = a.c;
    const int z = (int)blockIdx.z;
    const int W = g.w, H = g.h, used = g.used;
    PlaneParams P;
    if constexpr (NPL == 0) P = c.read.table[z < used ? z : 0];
    else P = a.planes[z];
    asm volatile("" ::"s"(W), "s"(H), "s"(used), "s"(P.step));
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int x0 = ((int)blockIdx.x * 64 + lane) * 4;
    const int y = (int)blockIdx.y * 4 + wave;
    if (y >= H || x0 >= W) return;
    const int npx = min(4, W - x0);

    // ---- read 4 pixels ----
    uint32_t raw[CN]; // 4*CN bytes
    if (z < used) {
        const gp_u8 row = (gp_u8)P.data + (size_t)y * (size_t)P.step + (size_t)x0 * CN;
        if (npx == 4) {
#pragma unroll
            for (int k = 0; k < CN; ++k) raw[k] = *(gp_u32)(row + 4 * k);
        } else {
#pragma unroll
            for (int k = 0; k < CN; ++k) raw[k] = 0;
#pragma unroll
            for (int b = 0; b < 4 * CN; ++b)
                if (b < npx * CN) raw[b >> 2] |= (uint32_t)row[b] << (8 * (b & 3));
        }
    }
    Px px[4];
    int depth = CVGS_DEPTH_8U, cn = CN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if (ch < CN) {
                const int b = i * CN + ch;
                px[i].v[ch] = z < used ? (float)((raw[b >> 2] >> (8 * (b & 3))) & 0xffu) : c.read.bg[ch];
            } else {
                px[i].v[ch] = 0.f;
            }
        }
    }
    Prog::run4(c.prog, px, depth, cn);

    // ---- write ----
    if (g.packed) {
        // cn floats per pixel, contiguous: 4 pixels = cn float4
        uint8_t* rows[2] = {g.out + (size_t)z * g.img_stride + (size_t)y * g.row_pitch,
                            g.out2 ? g.out2 + (size_t)z * g.img_stride2 + (size_t)y * g.row_pitch2 : nullptr};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (!rows[t]) continue;
            OT* o = (OT*)rows[t] + (size_t)x0 * cn;
            if (npx == 4) {
                float flat[16];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch)
                        if (ch < CN) flat[i * CN + ch] = px[i].v[ch];
#pragma unroll
                for (int v = 0; v < CN; ++v) store4(o + 4 * v, flat[4 * v], flat[4 * v + 1], flat[4 * v + 2], flat[4 * v + 3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch)
                        if (i < npx && ch < CN) store1(o + i * CN + ch, px[i].v[ch]);
            }
        }
    } else {
        OT* bases[2] = {(OT*)g.out + (int64_t)z * g.img_stride, g.out2 ? (OT*)g.out2 + (int64_t)z * g.img_stride2 : nullptr};
        const int64_t chs[2] = {g.ch_stride, g.ch_stride2};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (!bases[t]) continue;
            OT* o = bases[t] + (int64_t)y * W + x0;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                if (ch < cn) {
                    if (npx == 4) {
                        store4(o + (int64_t)ch * chs[t], px[0].v[ch], px[1].v[ch], px[2].v[ch], px[3].v[ch]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (i < npx) store1(o + (int64_t)ch * chs[t] + i, px[i].v[ch]);
                    }
                }
            }
        }
    }
}

template <int CN, class Prog, typename OT>
static hipError_t launch_pw(const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, hipStream_t s) {
    const dim3 grid((g.w + 255) / 256, (g.h + 3) / 4, c.read.batch);
    if (c.read.table) {
        KernArgs<0> a;
        a.c = c;
        a.planes[0] = PlaneParams{};
        hipLaunchKernelGGL((k_pointwise4<CN, 0, Prog, OT>), grid, dim3(256), 0, s, a, g);
    } else {
        KernArgs<CVGS_KERNARG_PLANES> a;
        a.c = c;
        for (int i = 0; i < CVGS_KERNARG_PLANES; ++i) a.planes[i] = i < ni ? ip[i] : PlaneParams{};
        hipLaunchKernelGGL((k_pointwise4<CN, CVGS_KERNARG_PLANES, Prog, OT>), grid, dim3(256), 0, s, a, g);
    }
    return hipGetLastError();
}

template <int CN, typename OT>
static hipError_t launch_pw_prog(int prog_id, const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, hipStream_t s) {
    if (prog_id == 0) return launch_pw<CN, ProgCastMulSubDiv, OT>(c, ip, ni, g, s);
    if (prog_id == 1) return launch_pw<CN, ProgCast, OT>(c, ip, ni, g, s);
    return launch_pw<CN, InterpProg, OT>(c, ip, ni, g, s);
}

template <typename OT>
static hipError_t launch_pw_cn(int prog_id, const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, hipStream_t s) {
    switch (c.read.cn) {
    case 1: return launch_pw_prog<1, OT>(prog_id, c, ip, ni, g, s);
    case 2: return launch_pw_prog<2, OT>(prog_id, c, ip, ni, g, s);
    case 3: return launch_pw_prog<3, OT>(prog_id, c, ip, ni, g, s);
    default: return launch_pw_prog<4, OT>(prog_id, c, ip, ni, g, s);
    }
}

// Returns 1 if it took the chain, 0 if not eligible, <0 on error.
int launch_pointwise(const ChainArgs& c_in, const PlaneParams* inline_planes, int n_inline, uint32_t chain_flags, void* stream,
                     bool dry_run, LaunchInfo* info) {
    const ReadArgs& r = c_in.read;
    const WriteArgs& w = c_in.write;
    if (chain_flags & CVGS_CHAIN_NO_THREAD_FUSION) return 0;
    if (r.kind != CVGS_READ_PIXEL || r.depth != CVGS_DEPTH_8U || r.batch > 65535) return 0;
    const bool f16 = w.depth == CVGS_DEPTH_16F;
    if (!f16 && w.depth != CVGS_DEPTH_32F) return 0;
    // fp16 targets: the chain ends with CAST(CV_16F); that conversion happens in the store
    ChainArgs c_cut;
    if (f16) {
        if (c_in.prog.n < 2 || c_in.prog.opcode[c_in.prog.n - 1] != CVGS_OP_CAST) return 0;
        c_cut = c_in;
        c_cut.prog.n -= 1;
    }
    const ChainArgs& c = f16 ? c_cut : c_in;
    const bool planar = w.kind == CVGS_WRITE_TENSOR_SPLIT || w.kind == CVGS_WRITE_TENSOR_T_SPLIT;
    const bool packed = w.kind == CVGS_WRITE_PIXEL_2D || w.kind == CVGS_WRITE_PIXEL_3D;
    if (!planar && !packed) return 0;
    // the program must turn the u8 value into fp32 with its first CAST and never change the channel count
    const ProgArgs& p = c.prog;
    if (p.n < 1 || p.opcode[0] != CVGS_OP_CAST || p.aux[0] != CVGS_DEPTH_32F) return 0;
    for (int k = 1; k < p.n; ++k)
        if (p.opcode[k] != CVGS_OP_MUL && p.opcode[k] != CVGS_OP_ADD && p.opcode[k] != CVGS_OP_SUB && p.opcode[k] != CVGS_OP_DIV &&
            p.opcode[k] != CVGS_OP_REORDER)
            return 0;
    if (w.cn != r.cn) return 0;
    if (!r.table && n_inline > CVGS_KERNARG_PLANES) return 0;

    int prog_id = 2;
    if (p.n == 4 && p.opcode[1] == CVGS_OP_MUL && p.opcode[2] == CVGS_OP_SUB && p.opcode[3] == CVGS_OP_DIV) prog_id = 0;
    else if (p.n == 1) prog_id = 1;
    if (info) {
        static const char* names[2][3] = {{"pointwise4_u8_cast_mul_sub_div", "pointwise4_u8_cast", "pointwise4_u8_interp"},
                                          {"pointwise4_u8_cast_mul_sub_div_f16", "pointwise4_u8_cast_f16", "pointwise4_u8_interp_f16"}};
        info->kernel = names[f16][prog_id];
    }
    if (dry_run) return 1;

    PwGeom g;
    g.w = r.dst_w; g.h = r.dst_h; g.used = r.used; g.cn = r.cn;
    g.packed = packed ? 1 : 0;
    g.out = w.data; g.out2 = w.data2; g.pad = 0;
    if (packed) {
        const int px_bytes = (f16 ? 2 : 4) * w.cn;
        g.row_pitch = w.kind == CVGS_WRITE_PIXEL_2D ? w.step : w.width * px_bytes;
        g.row_pitch2 = w.width * px_bytes;
        g.img_stride = w.kind == CVGS_WRITE_PIXEL_2D ? 0 : (int64_t)w.img_stride * px_bytes; // bytes
        g.img_stride2 = (int64_t)w.img_stride2 * px_bytes;
        g.ch_stride = g.ch_stride2 = 0;
    } else {
        g.row_pitch = g.row_pitch2 = 0;
        g.img_stride = w.img_stride; g.ch_stride = w.ch_stride;
        g.img_stride2 = w.img_stride2; g.ch_stride2 = w.ch_stride2;
    }
    hipStream_t s = (hipStream_t)stream;
    const hipError_t e = f16 ? launch_pw_cn<_Float16>(prog_id, c, inline_planes, n_inline, g, s)
                             : launch_pw_cn<float>(prog_id, c, inline_planes, n_inline, g, s);
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

} // namespace cvgs
