// k_nv12.hip -- K4: NV12 read-back fused into the bilinear resize: each of the 4 taps is fetched from the luma
// plane and the interleaved chroma plane, converted YCbCr -> RGB(A) in float, THEN interpolated, then pushed
// through the pointwise program and written (planar fp32 tensor, or packed pixels).  Replaces
//   fk::Resize<INTER_LINEAR>::build(fk::fuse(Read<ReadYUV<NV12>>, Unary<ConvertYUVToRGB<NV12,range,primaries,alpha,floatN>>), size)
// (reference tests/resize/test_fused_resize.cu:141-147).
//
// Mapping: lane = output column, wave = one output row (blockIdx.y = row group, blockIdx.z = plane).  Per output
// pixel and source row: ONE unaligned 2-byte load brings both luma taps, ONE unaligned 4-byte load both chroma
// pairs (4 loads per pixel instead of 12 byte loads); windows are clamped into the row.  Planar stores are
// full-wave 256-byte rows, non-temporal.
#include "k_common.hpp"

namespace cvgs {

typedef uint16_t u16_unaligned __attribute__((aligned(1)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) u16_unaligned* gptr_u16;
typedef const __attribute__((address_space(1))) u32_unaligned* gptr_u32;
typedef const __attribute__((address_space(1))) uint8_t* gptr_b;

struct N12Geom {
    int32_t dst_w, dst_h, out_w, cn; // cn: 3, or 4 with alpha
    int64_t img_stride, ch_stride;
    uint8_t* out;
    int32_t out_step; // packed 2D writes: bytes per row
    int32_t packed;   // 0: planar fp32 tensor, 1: packed pixels through the generic write stage
    // optional second planar target with its own strides (CircularTensor push: history ring + ordered tensor)
    uint8_t* out2;
    int64_t img_stride2, ch_stride2;
};

constexpr int kOpSwapRB12 = 100;
template <int... OPS>
struct N12Prog {
    static __device__ __forceinline__ void run(const ProgArgs& prog, Px& p, int& depth, int& cn) {
        int k = 0;
        ((step<OPS>(prog, k, p, depth, cn), ++k), ...);
    }
    template <int OP>
    static __device__ __forceinline__ void step(const ProgArgs& prog, int k, Px& p, int& depth, int& cn) {
        if constexpr (OP == kOpSwapRB12) {
            const float t = p.v[0];
            p.v[0] = p.v[2];
            p.v[2] = t;
        } else {
            apply_op(OP, prog.aux[k], prog.operand[k], p, depth, cn);
        }
    }
};
using N12SwapMulSubDiv = N12Prog<kOpSwapRB12, CVGS_OP_MUL, CVGS_OP_SUB, CVGS_OP_DIV>;

template <int NPL, class Prog, typename OT = float>
__global__ __launch_bounds__(256) void k4_nv12_resize(const KernArgs<NPL> a, const N12Geom g) {
    const ChainArgs& c = a.c;
    const int z = (int)blockIdx.z;
    const int dst_w = g.dst_w, dst_h = g.dst_h, W = g.out_w, CN = g.cn;
    PlaneParams P;
    if constexpr (NPL == 0) P = c.read.table[z];
    else P = a.planes[z];
    const int yuv_range = c.read.yuv_range, yuv_prim = c.read.yuv_primaries, packed = g.packed;
    const int64_t img_stride = g.img_stride, ch_stride = g.ch_stride;
    uint8_t* const out_base = g.out;
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    const f32x4s op0 = *(const f32x4s*)c.prog.operand[0], op1 = *(const f32x4s*)c.prog.operand[1],
                 op2 = *(const f32x4s*)c.prog.operand[2], op3 = *(const f32x4s*)c.prog.operand[3];
    // one batch of scalar loads, one wait (see k_k1.hip)
    asm volatile("" ::"s"(dst_w), "s"(dst_h), "s"(W), "s"(CN), "s"(P.w), "s"(P.h), "s"(P.step), "s"(P.fx), "s"(P.fy), "s"(P.data), "s"(P.uv_off),
                 "s"(yuv_range), "s"(yuv_prim), "s"(packed), "s"(img_stride), "s"(ch_stride), "s"(out_base), "s"(op0), "s"(op1),
                 "s"(op2), "s"(op3));
    const YuvK yk = yuv_matrix(yuv_range, yuv_prim);

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int x = (int)blockIdx.x * 64 + lane;
    const int y = (int)blockIdx.y * 4 + wave;
    if (y >= dst_h || x >= dst_w) return;

    // column geometry
    const float sx = (float)x * P.fx;
    const int x1 = (int)floorf(sx);
    const int x2 = x1 + 1;
    const float wxa = (float)x2 - sx, wxb = sx - (float)x1;
    const bool edge = x2 > P.w - 1;
    const int x2r = edge ? x1 : x2;
    const uint32_t yo = (uint32_t)min(x1, P.w - 2);
    const int ysh = (x1 - (int)yo) * 8;
    const int c1 = x1 >> 1, c2 = x2r >> 1;
    const uint32_t uo = (uint32_t)min(2 * c1, P.w - 4);
    const int ush = (2 * c1 - (int)uo) * 8;
    const bool same_pair = c2 == c1;
    // row geometry (wave-uniform)
    const float sy = (float)y * P.fy;
    const int y1 = (int)floorf(sy);
    const int y2 = y1 + 1;
    const int y2r = min(y2, P.h - 1);
    const float wya = (float)y2 - sy, wyb = sy - (float)y1;

    const gptr_b base = (gptr_b)P.data;
    const size_t step = (size_t)P.step;
    const gptr_b ya = base + (size_t)__builtin_amdgcn_readfirstlane(y1) * step;
    const gptr_b yb = base + (size_t)__builtin_amdgcn_readfirstlane(y2r) * step;
    const gptr_b uvp = base + (size_t)P.uv_off; // crops of a surface carry their own luma -> chroma offset
    const gptr_b ua = uvp + (size_t)__builtin_amdgcn_readfirstlane(y1 >> 1) * step;
    const gptr_b ub = uvp + (size_t)__builtin_amdgcn_readfirstlane(y2r >> 1) * step;
    const uint32_t vya = *(gptr_u16)(ya + yo);
    const uint32_t vyb = *(gptr_u16)(yb + yo);
    const uint32_t vua = *(gptr_u32)(ua + uo);
    const uint32_t vub = *(gptr_u32)(ub + uo);

    const uint32_t ya0 = (vya >> ysh) & 0xffu, ya1 = edge ? ya0 : (vya >> 8) & 0xffu;
    const uint32_t yb0 = (vyb >> ysh) & 0xffu, yb1 = edge ? yb0 : (vyb >> 8) & 0xffu;
    const uint32_t pa0 = (vua >> ush) & 0xffffu, pa1 = same_pair ? pa0 : (vua >> 16) & 0xffffu;
    const uint32_t pb0 = (vub >> ush) & 0xffffu, pb1 = same_pair ? pb0 : (vub >> 16) & 0xffffu;

    Px t00, t10, t01, t11;
    yuv_to_rgb((float)ya0, (float)(pa0 & 0xffu), (float)(pa0 >> 8), yk, t00);
    yuv_to_rgb((float)ya1, (float)(pa1 & 0xffu), (float)(pa1 >> 8), yk, t10);
    yuv_to_rgb((float)yb0, (float)(pb0 & 0xffu), (float)(pb0 >> 8), yk, t01);
    yuv_to_rgb((float)yb1, (float)(pb1 & 0xffu), (float)(pb1 >> 8), yk, t11);

    const float w00 = wxa * wya, w10 = wxb * wya, w01 = wxa * wyb, w11 = wxb * wyb;
    Px p;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float acc = t00.v[k] * w00;
        acc = acc + t10.v[k] * w10;
        acc = acc + t01.v[k] * w01;
        acc = acc + t11.v[k] * w11;
        p.v[k] = acc;
    }
    int depth = CVGS_DEPTH_32F, cn = CN;
    Prog::run(c.prog, p, depth, cn);

    if (packed) {
        write_px(c.write, c.dst_inline, x, y, z, p, depth, cn);
    } else {
        // OT = _Float16: the chain's trailing CAST(CV_16F) is this round-to-nearest-even conversion
        OT* const orow = (OT*)out_base + (int64_t)z * img_stride + (int64_t)y * W;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < cn) __builtin_nontemporal_store((OT)p.v[k], orow + (int64_t)k * ch_stride + x);
        if (g.out2) { // wave-uniform
            OT* const orow2 = (OT*)g.out2 + (int64_t)z * g.img_stride2 + (int64_t)y * W;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < cn) __builtin_nontemporal_store((OT)p.v[k], orow2 + (int64_t)k * g.ch_stride2 + x);
        }
    }
}

template <class Prog, typename OT = float>
static hipError_t launch_n12(const ChainArgs& c, const PlaneParams* ip, int ni, const N12Geom& g, hipStream_t s) {
    const dim3 grid((g.dst_w + 63) / 64, (g.dst_h + 3) / 4, c.read.batch);
    if (c.read.table) {
        KernArgs<0> a;
        a.c = c;
        a.planes[0] = PlaneParams{};
        hipLaunchKernelGGL((k4_nv12_resize<0, Prog, OT>), grid, dim3(256), 0, s, a, g);
    } else if (ni <= 8) {
        KernArgs<8> a;
        a.c = c;
        for (int i = 0; i < 8; ++i) a.planes[i] = i < ni ? ip[i] : PlaneParams{};
        hipLaunchKernelGGL((k4_nv12_resize<8, Prog, OT>), grid, dim3(256), 0, s, a, g);
    } else { // crop lists of a decoder surface: up to CVGS_KERNARG_PLANES descriptors in the kernel arguments
        KernArgs<CVGS_KERNARG_PLANES> a;
        a.c = c;
        for (int i = 0; i < CVGS_KERNARG_PLANES; ++i) a.planes[i] = i < ni ? ip[i] : PlaneParams{};
        hipLaunchKernelGGL((k4_nv12_resize<CVGS_KERNARG_PLANES, Prog, OT>), grid, dim3(256), 0, s, a, g);
    }
    return hipGetLastError();
}

// Returns 1 if it took the chain, 0 if not eligible, <0 on error.
int launch_nv12(const ChainArgs& c_in, const PlaneParams* inline_planes, int n_inline, int min_width, void* stream,
                bool dry_run, LaunchInfo* info) {
    const ReadArgs& r = c_in.read;
    // fp16 planar tensors: the trailing CAST(CV_16F) moves into the store
    const bool planar_kind = c_in.write.kind == CVGS_WRITE_TENSOR_SPLIT || c_in.write.kind == CVGS_WRITE_TENSOR_T_SPLIT;
    const bool f16 = planar_kind && c_in.write.depth == CVGS_DEPTH_16F && c_in.prog.n >= 1 &&
                     c_in.prog.opcode[c_in.prog.n - 1] == CVGS_OP_CAST;
    ChainArgs c_cut;
    if (f16) {
        c_cut = c_in;
        c_cut.prog.n -= 1;
        for (int k = 0; k < c_cut.prog.n; ++k)
            if (c_cut.prog.opcode[k] == CVGS_OP_CAST || c_cut.prog.opcode[k] == CVGS_OP_CAST_TRUNC) return 0;
    }
    const ChainArgs& c = f16 ? c_cut : c_in;
    if (r.kind != CVGS_READ_NV12_RESIZE_LINEAR) return 0;
    if (r.table || n_inline > CVGS_KERNARG_PLANES || min_width < 4) return 0; // tiny frames / resident tables: generic kernel
    if (r.used != r.batch || r.batch > 65535) return 0;
    for (int i = 0; i < n_inline; ++i) { // aspect-ratio padding is the generic kernel's business
        const PlaneParams& P = inline_planes[i];
        if (P.x1 != 0 || P.y1 != 0 || P.x2 != r.dst_w - 1 || P.y2 != r.dst_h - 1) return 0;
    }
    const WriteArgs& w = c.write;
    const bool planar = planar_kind && (w.depth == CVGS_DEPTH_32F || f16);
    const bool packed = w.kind == CVGS_WRITE_PIXEL_2D || w.kind == CVGS_WRITE_PIXEL_3D;
    if (!planar && !packed) return 0;

    N12Geom g;
    g.dst_w = r.dst_w; g.dst_h = r.dst_h; g.out_w = w.width; g.cn = r.out_cn;
    g.img_stride = w.img_stride; g.ch_stride = w.ch_stride;
    g.out = w.data; g.out_step = w.step; g.packed = packed ? 1 : 0;
    g.out2 = w.data2; g.img_stride2 = w.img_stride2; g.ch_stride2 = w.ch_stride2;

    const int swap = r.out_cn == 3 ? (2 | (1 << 2) | (0 << 4)) : (2 | (1 << 2) | (0 << 4) | (3 << 6));
    const ProgArgs& p = c.prog;
    const bool fast_prog = planar && p.n == 4 && p.opcode[0] == CVGS_OP_REORDER && p.aux[0] == swap &&
                           p.opcode[1] == CVGS_OP_MUL && p.opcode[2] == CVGS_OP_SUB && p.opcode[3] == CVGS_OP_DIV;
    if (info)
        info->kernel = f16 ? (fast_prog ? "k4_nv12_resize_swap_mul_sub_div_f16" : "k4_nv12_resize_interp_f16")
                           : (fast_prog ? "k4_nv12_resize_swap_mul_sub_div" : "k4_nv12_resize_interp");
    if (dry_run) return 1;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (f16) e = fast_prog ? launch_n12<N12SwapMulSubDiv, _Float16>(c, inline_planes, n_inline, g, s)
                           : launch_n12<InterpProg, _Float16>(c, inline_planes, n_inline, g, s);
    else e = fast_prog ? launch_n12<N12SwapMulSubDiv>(c, inline_planes, n_inline, g, s) : launch_n12<InterpProg>(c, inline_planes, n_inline, g, s);
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

} // namespace cvgs
