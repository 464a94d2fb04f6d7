// k_pointwise.hip -- thread-fused pointwise chains on u8 sources:
//   PerThreadRead<_2D, uchar{1,2,3,4}> [batched] -> SaturateCast/Mul/Sub/Div/... -> fp32 (or fp16) planar tensor
//   (TensorSplit / TensorTSplit, optionally mirrored into a second target) or packed fp32 / fp16 pixels (2D / 3D).
// The engine's counterpart of the reference's ENABLE_THREAD_FUSION=true fast path (reference
// include/cvGPUSpeedup.cuh:464-473): each thread owns FOUR x-adjacent pixels, reads them with one
// wide load (4*CN bytes) and writes 16-byte vectors.  Results are bit-identical to the interpreted kernel; chains it
// does not cover (integer outputs, CV_64F) stay on k_generic.  SplitWrite<_2D> targets (separate pitched planes) are the
// planar stores with one base and pitch per plane.  Used by K5/K6/K7 and by the
// CircularTensor push of an un-resized frame (cfg #4).
#include <memory>
#include <type_traits>

#include "k_pointwise_body.hpp"

namespace cvgs {

template <int CN, int NPL, class Prog, typename OT, int SD = CVGS_DEPTH_8U>
__global__ __launch_bounds__(256) void k_pointwise4(const KernArgs<NPL> a, const PwGeom g) {
    const ChainArgs& c = a.c;
    const int z = (int)blockIdx.z;
    PlaneParams P;
    if constexpr (NPL == 0) P = c.read.table[z < g.used ? z : 0];
    else P = a.planes[z];
    asm volatile("" ::"s"(g.w), "s"(g.h), "s"(g.used), "s"(P.step));
    if constexpr (!is_yuv_sd<SD> && CN == 1) {
        if (g.narrow < 0) { // scalar: two rows per thread (launch_pw)
            pw4_body_rows2<CN, Prog, OT, SD>(c, P, g, (int)blockIdx.x, (int)blockIdx.y, z);
            return;
        }
    }
    pw4_body<CN, Prog, OT, SD>(c, P, g, (int)blockIdx.x, (int)blockIdx.y, z);
}

// cvgs_execute_many on the pointwise shape (round 6): the planes of ALL chains in the kernel arguments, blockIdx.z = chain x max_batch + plane;
// every chain writes its own tensor (the segment's), reads its own planes, shares the geometry, the program and its operands
template <int CN, int CAP, class Prog, typename OT>
__global__ __launch_bounds__(256) void k_pointwise4_many(const KernArgsManyInline<CAP> a, const PwGeom g, const uint32_t max_batch) {
    const ChainArgs& c = a.c;
    const uint32_t chain = blockIdx.z / max_batch, zl = blockIdx.z - chain * max_batch;
    const ManySeg sg = a.seg[chain];
    if ((int)zl >= sg.batch) return; // a shorter chain of the tick
    const PlaneParams P = a.planes[(uint32_t)(uintptr_t)sg.table + (zl < (uint32_t)sg.used ? zl : 0u)];
    PwGeom gl = g;
    gl.out = sg.out;
    gl.used = sg.used;
    asm volatile("" ::"s"(gl.w), "s"(gl.h), "s"(gl.used), "s"(P.step));
    pw4_body<CN, Prog, OT, CVGS_DEPTH_8U>(c, P, gl, (int)blockIdx.x, (int)blockIdx.y, (int)zl);
}

template <int CN, class Prog, typename OT>
static hipError_t launch_pw_many(const ChainArgs& c, const PlaneParams* planes, int n_planes, const ManySeg* segs, int n_segs, int max_batch, const PwGeom& g,
                                 hipStream_t s) {
    const int px_per_wave_row = 256 >> g.narrow, rows_per_block = 4 << g.narrow;
    const dim3 grid((g.w + px_per_wave_row - 1) / px_per_wave_row, (g.h + rows_per_block - 1) / rows_per_block, (unsigned)(n_segs * max_batch));
    const uint32_t mb = (uint32_t)max_batch;
    auto go = [&](auto cap_tag) {
        constexpr int CAP = decltype(cap_tag)::value;
        // the 16 KB / 52 KB argument block: staged in a per-thread heap buffer, handed over by address (as K1's: k_k1_impl.hpp launch_t)
        static thread_local std::unique_ptr<KernArgsManyInline<CAP>> staged;
        if (!staged) staged.reset(new KernArgsManyInline<CAP>());
        KernArgsManyInline<CAP>& a = *staged;
        a.c = c;
        for (int i = 0; i < CVGS_MAX_CHAINS; ++i) a.seg[i] = i < n_segs ? segs[i] : ManySeg{nullptr, nullptr, 0, 0};
        for (int i = 0; i < n_planes && i < CAP; ++i) a.planes[i] = planes[i];
        void* args[] = {(void*)&a, (void*)&g, (void*)&mb};
        return hipLaunchKernel((const void*)&k_pointwise4_many<CN, CAP, Prog, OT>, grid, dim3(256), args, 0, s);
    };
    const hipError_t e = n_planes <= kManyInlineSmall ? go(std::integral_constant<int, kManyInlineSmall>{}) : go(std::integral_constant<int, kManyInlineLarge>{});
    return e != hipSuccess ? e : hipGetLastError();
}

template <int CN, class Prog, typename OT, int SD = CVGS_DEPTH_8U>
static hipError_t launch_pw(const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g_in, hipStream_t s) {
    PwGeom g = g_in;
    // whole one-channel frames (>= 1 Mpixel per launch): two rows per thread (pw4_body_rows2; 4K 8UC1 -> 32FC1 12.99 -> 12.14 us, 32FC1 15.5 -> 14.9;
    // with 2-4 channels a thread already moves 40-96 bytes and the second row costs 3-8 %)
    if (!is_yuv_sd<SD> && CN == 1 && g.narrow == 0 && (int64_t)g.w * g.h * c.read.batch >= (1 << 20)) g.narrow = -1;
    const int px_per_wave_row = g.narrow < 0 ? 256 : 256 >> g.narrow, rows_per_block = g.narrow < 0 ? 8 : 4 << g.narrow;
    const dim3 grid((g.w + px_per_wave_row - 1) / px_per_wave_row, (g.h + rows_per_block - 1) / rows_per_block, c.read.batch);
    if (c.read.table) {
        KernArgs<0> a;
        a.c = c;
        a.planes[0] = PlaneParams{};
        hipLaunchKernelGGL((k_pointwise4<CN, 0, Prog, OT, SD>), grid, dim3(256), 0, s, a, g);
    } else {
        KernArgs<CVGS_KERNARG_PLANES> a;
        a.c = c;
        for (int i = 0; i < CVGS_KERNARG_PLANES; ++i) a.planes[i] = i < ni ? ip[i] : PlaneParams{};
        hipLaunchKernelGGL((k_pointwise4<CN, CVGS_KERNARG_PLANES, Prog, OT, SD>), grid, dim3(256), 0, s, a, g);
    }
    return hipGetLastError();
}

template <int CN, typename OT>
static hipError_t launch_pw_prog(int prog_id, const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, hipStream_t s) {
    if (prog_id == 0) return launch_pw<CN, ProgCastMulSubDiv, OT>(c, ip, ni, g, s);
    if (prog_id == 1) return launch_pw<CN, ProgCast, OT>(c, ip, ni, g, s);
    return launch_pw<CN, ArithProg<CN, CVGS_DEPTH_8U>, OT>(c, ip, ni, g, s);
}

// other source depths, fp32 output: the normalisation chain ([cast,] mul, sub, div) as a compile-time program, anything
// else interpreted
using ProgMulSubDivPw = StaticProg<CVGS_OP_MUL, CVGS_OP_SUB, CVGS_OP_DIV>;
template <int SD, class Prog>
static hipError_t launch_pw_depth_prog(const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, hipStream_t s) {
    switch (c.read.cn) {
    case 1: return launch_pw<1, Prog, float, SD>(c, ip, ni, g, s);
    case 2: return launch_pw<2, Prog, float, SD>(c, ip, ni, g, s);
    case 3: return launch_pw<3, Prog, float, SD>(c, ip, ni, g, s);
    default: return launch_pw<4, Prog, float, SD>(c, ip, ni, g, s);
    }
}
template <int SD>
static hipError_t launch_pw_depth_arith(const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, hipStream_t s) {
    switch (c.read.cn) {
    case 1: return launch_pw<1, ArithProg<1, SD>, float, SD>(c, ip, ni, g, s);
    case 2: return launch_pw<2, ArithProg<2, SD>, float, SD>(c, ip, ni, g, s);
    case 3: return launch_pw<3, ArithProg<3, SD>, float, SD>(c, ip, ni, g, s);
    default: return launch_pw<4, ArithProg<4, SD>, float, SD>(c, ip, ni, g, s);
    }
}
template <int SD>
static hipError_t launch_pw_depth(bool normalise, const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, hipStream_t s) {
    using Norm = std::conditional_t<SD == CVGS_DEPTH_32F, ProgMulSubDivPw, ProgCastMulSubDiv>;
    if (normalise) return launch_pw_depth_prog<SD, Norm>(c, ip, ni, g, s);
    return launch_pw_depth_arith<SD>(c, ip, ni, g, s);
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

// ---- u8 -> u8 chains (colour conversions, brightness / contrast, ...): 4 pixels per thread, any program ----------------
// The standalone cvGS::cvtColor / convertTo chains of the reference's tests (tests/color/test_cvtColor.cu:55,
// tests/single_operation/test_convertTo.cu) on packed u8 images: one 4*CN-byte load, the interpreted program on four
// pixels, one 4*OCN-byte store (consecutive lanes write consecutive chunks).  The channel count may change (OCN).
template <int CN, int OCN, int NPL>
__global__ __launch_bounds__(256) void k_pointwise4_u8u8(const KernArgs<NPL> a, const PwGeom g) {
    const ChainArgs& c = a.c;
    const int z = (int)blockIdx.z;
    PlaneParams P;
    if constexpr (NPL == 0) P = c.read.table[z < g.used ? z : 0];
    else P = a.planes[z];
    const int W = g.w, H = g.h, used = g.used;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int x0 = ((int)blockIdx.x * 64 + lane) * 4;
    const int y = (int)blockIdx.y * 4 + wave;
    if (y >= H || x0 >= W) return;
    const int npx = min(4, W - x0);
    uint32_t raw[CN];
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
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
            px[i].v[ch] = ch < CN ? (z < used ? elem_value<CVGS_DEPTH_8U>(raw, i * CN + ch) : c.read.bg[ch]) : 0.f;
    InterpProg::run4(c.prog, px, depth, cn); // ends on CV_8U values with OCN channels (checked on the host)

    uint8_t* orow = g.out + (size_t)z * g.img_stride + (size_t)y * g.row_pitch + (size_t)x0 * OCN;
    if (npx == 4) {
        uint32_t q[OCN];
#pragma unroll
        for (int d = 0; d < OCN; ++d) {
            q[d] = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * d + i; // output byte e = pixel e / OCN, channel e % OCN
                q[d] = sat_u8_insert(px[e / OCN].v[e % OCN], (uint32_t)i, q[d]); // CV_8U-typed value: convert + insert in one instruction
            }
        }
        typedef uint32_t vq __attribute__((ext_vector_type(OCN == 3 ? 3 : (OCN == 4 ? 4 : (OCN == 2 ? 2 : 1)))));
        typedef uint32_t u32a1 __attribute__((aligned(1)));
        if constexpr (OCN == 1) {
            __builtin_nontemporal_store(q[0], (u32a1*)orow);
        } else {
            typedef vq vqu __attribute__((aligned(1)));
            vq v;
#pragma unroll
            for (int d = 0; d < OCN; ++d) v[d] = q[d];
            __builtin_nontemporal_store(v, (vqu*)orow);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ch = 0; ch < OCN; ++ch)
                if (i < npx) orow[i * OCN + ch] = (uint8_t)px[i].v[ch];
    }
}

template <int CN, int OCN>
static hipError_t launch_u8u8(const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, hipStream_t s) {
    const dim3 grid((g.w + 255) / 256, (g.h + 3) / 4, c.read.batch);
    if (c.read.table) {
        KernArgs<0> a;
        a.c = c;
        a.planes[0] = PlaneParams{};
        hipLaunchKernelGGL((k_pointwise4_u8u8<CN, OCN, 0>), grid, dim3(256), 0, s, a, g);
    } else {
        KernArgs<CVGS_KERNARG_PLANES> a;
        a.c = c;
        for (int i = 0; i < CVGS_KERNARG_PLANES; ++i) a.planes[i] = i < ni ? ip[i] : PlaneParams{};
        hipLaunchKernelGGL((k_pointwise4_u8u8<CN, OCN, CVGS_KERNARG_PLANES>), grid, dim3(256), 0, s, a, g);
    }
    return hipGetLastError();
}
template <int CN>
static hipError_t launch_u8u8_ocn(int ocn, const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, hipStream_t s) {
    switch (ocn) {
    case 1: return launch_u8u8<CN, 1>(c, ip, ni, g, s);
    case 2: return launch_u8u8<CN, 2>(c, ip, ni, g, s);
    case 3: return launch_u8u8<CN, 3>(c, ip, ni, g, s);
    default: return launch_u8u8<CN, 4>(c, ip, ni, g, s);
    }
}

// Returns 1 if it took the chain, 0 if not eligible, <0 on error.
static int launch_pointwise_u8u8(const ChainArgs& c, const PlaneParams* ip, int ni, uint32_t chain_flags, hipStream_t s, bool dry_run,
                                 LaunchInfo* info) {
    const ReadArgs& r = c.read;
    const WriteArgs& w = c.write;
    if (chain_flags & CVGS_CHAIN_NO_THREAD_FUSION) return 0;
    if (r.kind != CVGS_READ_PIXEL || r.depth != CVGS_DEPTH_8U || w.depth != CVGS_DEPTH_8U || r.batch > 65535) return 0;
    if (w.kind != CVGS_WRITE_PIXEL_2D && w.kind != CVGS_WRITE_PIXEL_3D) return 0;
    if (w.data2 || (!r.table && ni > CVGS_KERNARG_PLANES)) return 0;
    PwGeom g;
    g.w = r.dst_w; g.h = r.dst_h; g.used = r.used; g.cn = r.cn; g.packed = 1; g.narrow = 0;
    g.out = w.data; g.out2 = nullptr;
    g.row_pitch = w.kind == CVGS_WRITE_PIXEL_2D ? w.step : w.width * w.cn;
    g.row_pitch2 = 0;
    g.img_stride = w.kind == CVGS_WRITE_PIXEL_2D ? 0 : (int64_t)w.img_stride * w.cn; // bytes
    g.img_stride2 = g.ch_stride = g.ch_stride2 = 0;
    {
        // the colour conversions themselves (RGB <-> BGR, +-alpha, *2GRAY) as compile-time programs, 16 pixels per thread
        const int rc = launch_u8_colour16(c, ip, ni, g, s, dry_run, info);
        if (rc) return rc;
    }
    if (info) info->kernel = "pointwise4_u8_u8_interp";
    if (dry_run) return 1;
    hipError_t e;
    switch (r.cn) {
    case 1: e = launch_u8u8_ocn<1>(w.cn, c, ip, ni, g, s); break;
    case 2: e = launch_u8u8_ocn<2>(w.cn, c, ip, ni, g, s); break;
    case 3: e = launch_u8u8_ocn<3>(w.cn, c, ip, ni, g, s); break;
    default: e = launch_u8u8_ocn<4>(w.cn, c, ip, ni, g, s); break;
    }
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

// CV_16U -> CV_16U colour conversions (the reference's cvtColor test sweeps 16-bit types too): only the compile-time
// permutation / gray kernels exist for them; every other 16-bit program stays on the interpreted kernel.
static int launch_pointwise_u16u16(const ChainArgs& c, const PlaneParams* ip, int ni, uint32_t chain_flags, hipStream_t s, bool dry_run,
                                   LaunchInfo* info) {
    const ReadArgs& r = c.read;
    const WriteArgs& w = c.write;
    if (chain_flags & CVGS_CHAIN_NO_THREAD_FUSION) return 0;
    if (r.kind != CVGS_READ_PIXEL || r.depth != CVGS_DEPTH_16U || w.depth != CVGS_DEPTH_16U || r.batch > 65535) return 0;
    if (w.kind != CVGS_WRITE_PIXEL_2D && w.kind != CVGS_WRITE_PIXEL_3D) return 0;
    if (w.data2 || (!r.table && ni > CVGS_KERNARG_PLANES)) return 0;
    PwGeom g;
    g.w = r.dst_w; g.h = r.dst_h; g.used = r.used; g.cn = r.cn; g.packed = 1; g.narrow = 0;
    g.out = w.data; g.out2 = nullptr;
    g.row_pitch = w.kind == CVGS_WRITE_PIXEL_2D ? w.step : w.width * w.cn * 2;
    g.row_pitch2 = 0;
    g.img_stride = w.kind == CVGS_WRITE_PIXEL_2D ? 0 : (int64_t)w.img_stride * w.cn * 2; // bytes
    g.img_stride2 = g.ch_stride = g.ch_stride2 = 0;
    return launch_u8_colour16(c, ip, ni, g, s, dry_run, info);
}

// CV_32F -> CV_32F colour conversions: the compile-time permutation / gray kernels of k_cvtcolor_u8.hip, four pixels per thread
static int launch_pointwise_f32f32(const ChainArgs& c, const PlaneParams* ip, int ni, uint32_t chain_flags, hipStream_t s, bool dry_run,
                                   LaunchInfo* info) {
    const ReadArgs& r = c.read;
    const WriteArgs& w = c.write;
    if (chain_flags & CVGS_CHAIN_NO_THREAD_FUSION) return 0;
    if (r.kind != CVGS_READ_PIXEL || r.depth != CVGS_DEPTH_32F || w.depth != CVGS_DEPTH_32F || r.batch > 65535) return 0;
    if (w.kind != CVGS_WRITE_PIXEL_2D && w.kind != CVGS_WRITE_PIXEL_3D) return 0;
    if (w.data2 || r.table || ni > CVGS_KERNARG_PLANES) return 0;
    PwGeom g;
    g.w = r.dst_w; g.h = r.dst_h; g.used = r.used; g.cn = r.cn; g.packed = 1; g.narrow = 0;
    g.out = w.data; g.out2 = nullptr;
    g.row_pitch = w.kind == CVGS_WRITE_PIXEL_2D ? w.step : w.width * w.cn * 4;
    g.row_pitch2 = 0;
    g.img_stride = w.kind == CVGS_WRITE_PIXEL_2D ? 0 : (int64_t)w.img_stride * w.cn * 4; // bytes
    g.img_stride2 = g.ch_stride = g.ch_stride2 = 0;
    return launch_f32_colour4(c, ip, ni, g, s, dry_run, info);
}

// Eligibility + geometry of the thread-fused path, shared with the single-launch CircularTensor push (k_circular.hip).
// On success `c` is the chain to run (an fp16 target's trailing CAST is folded into the store), `g` the geometry.
bool pointwise4_plan(const ChainArgs& c_in, int n_inline, uint32_t chain_flags, ChainArgs& c, PwGeom& g, int& prog_id, bool& f16, bool* u8out) {
    const ReadArgs& r = c_in.read;
    const WriteArgs& w = c_in.write;
    if (chain_flags & CVGS_CHAIN_NO_THREAD_FUSION) return false;
    // 4:2:0 surfaces with interleaved chroma read WITHOUT a resize (the decode-side cvtColor: cvGS::cvtColorNV12 -> ... -> tensor):
    // the value arrives as CV_32F R, G, B[, A], so the chain is a CV_32F chain with out_cn channels
    const bool yuv = r.kind == CVGS_READ_NV12; // every layout: NV12 / NV21 / P010 (interleaved chroma), I420 / YV12 (planar chroma)
    if ((r.kind != CVGS_READ_PIXEL && !yuv) || r.batch > 65535) return false;
    const int sdepth = yuv ? CVGS_DEPTH_32F : r.depth, scn = yuv ? r.out_cn : r.cn;
    const bool u8src = sdepth == CVGS_DEPTH_8U;
    if (!u8src && sdepth != CVGS_DEPTH_8S && sdepth != CVGS_DEPTH_16U && sdepth != CVGS_DEPTH_16S && sdepth != CVGS_DEPTH_32S &&
        sdepth != CVGS_DEPTH_32F)
        return false;
    f16 = w.depth == CVGS_DEPTH_16F;
    // packed u8 pixels behind a 4:2:0 read (NV12 -> BGR image): the chain ends with CAST(CV_8U), which the store performs
    const bool u8o = yuv && u8out && w.depth == CVGS_DEPTH_8U && (w.kind == CVGS_WRITE_PIXEL_2D || w.kind == CVGS_WRITE_PIXEL_3D) &&
                     c_in.prog.n >= 1 && c_in.prog.opcode[c_in.prog.n - 1] == CVGS_OP_CAST && c_in.prog.aux[c_in.prog.n - 1] == CVGS_DEPTH_8U;
    if (u8out) *u8out = u8o;
    if (!f16 && !u8o && w.depth != CVGS_DEPTH_32F) return false;
    if (f16 && !u8src && !yuv) return false;
    c = c_in;
    if (u8o) c.prog.n -= 1;
    if (f16) { // fp16 targets: the chain ends with CAST(CV_16F); that conversion happens in the store
        if (c_in.prog.n < (yuv ? 1 : 2) || c_in.prog.opcode[c_in.prog.n - 1] != CVGS_OP_CAST) return false;
        c.prog.n -= 1;
    }
    const bool planar = w.kind == CVGS_WRITE_TENSOR_SPLIT || w.kind == CVGS_WRITE_TENSOR_T_SPLIT;
    const bool packed = w.kind == CVGS_WRITE_PIXEL_2D || w.kind == CVGS_WRITE_PIXEL_3D;
    const bool split2d = w.kind == CVGS_WRITE_SPLIT_2D && !w.data2;
    if (!planar && !packed && !split2d) return false;
    // the program must turn the u8 value into fp32 with its first CAST and never change the channel count
    const ProgArgs& p = c.prog;
    // (a CV_32F source needs no cast; every other depth must become fp32 before anything else happens)
    // (channel permutations on the source type may come first: cvtColor<BGR2RGB, CV_8UC3>() then convertTo<..., CV_32FC3>())
    int first = 0;
    while (first < p.n && p.opcode[first] == CVGS_OP_REORDER) ++first;
    const bool has_cast = first < p.n && p.opcode[first] == CVGS_OP_CAST && p.aux[first] == CVGS_DEPTH_32F;
    const bool starts_with_cast = has_cast && first == 0;
    if (sdepth != CVGS_DEPTH_32F && !has_cast) return false;
    for (int k = has_cast ? first + 1 : first; k < p.n; ++k) {
        // a 4:2:0 read's value is CV_32F throughout: convertTo<CV_32FCn, O>(alpha) spells cast(CV_32F) -> mul -> cast(O), and
        // that first cast is the identity (ArithProg executes nothing for it)
        if (yuv && p.opcode[k] == CVGS_OP_CAST && p.aux[k] == CVGS_DEPTH_32F) continue;
        if (p.opcode[k] != CVGS_OP_MUL && p.opcode[k] != CVGS_OP_ADD && p.opcode[k] != CVGS_OP_SUB && p.opcode[k] != CVGS_OP_DIV &&
            p.opcode[k] != CVGS_OP_REORDER)
            return false;
    }
    if (w.cn != scn) return false;
    if (!r.table && n_inline > CVGS_KERNARG_PLANES) return false;
    if (yuv && r.table) return false; // resident tables: per-plane preconditions cannot be checked on the host

    prog_id = 2;
    if (!u8src) prog_id = 3; // interpreted program, per-depth kernel
    else if (starts_with_cast && p.n == 4 && p.opcode[1] == CVGS_OP_MUL && p.opcode[2] == CVGS_OP_SUB && p.opcode[3] == CVGS_OP_DIV) prog_id = 0;
    else if (starts_with_cast && p.n == 1) prog_id = 1;

    guarded_div_setup(c.prog, scn); // the first DIV stage with fitting divisors divides by reciprocal, checked per wave (k_common.hpp: div4_guarded)
    g.w = r.dst_w; g.h = r.dst_h; g.used = r.used; g.cn = scn;
    g.packed = packed ? 1 : (split2d ? 2 : 0);
    g.out = w.data; g.out2 = w.data2; g.narrow = 0;
    if (split2d) {
        g.out = g.out2 = nullptr; // the planes are in c.dst_inline / c.write.table
        g.row_pitch = g.row_pitch2 = 0;
        g.img_stride = g.ch_stride = g.img_stride2 = g.ch_stride2 = 0;
    } else if (packed) {
        const int px_bytes = (u8o ? 1 : (f16 ? 2 : 4)) * w.cn;
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
    return true;
}

int launch_pointwise_many(const ChainArgs& c_in, const PlaneParams* planes, int n_planes, const ManySeg* segs, int n_segs, int max_batch,
                          uint32_t chain_flags, void* stream, bool dry_run) {
    const ReadArgs& r = c_in.read;
    const WriteArgs& w = c_in.write;
    // the hot pointwise shape only: u8 planes read per pixel, an fp32 program, a dense fp32 target (planar tensor or packed pixels)
    if (r.kind != CVGS_READ_PIXEL || r.depth != CVGS_DEPTH_8U || r.table || w.data2 || w.depth != CVGS_DEPTH_32F) return 0;
    if (w.kind != CVGS_WRITE_TENSOR_SPLIT && w.kind != CVGS_WRITE_TENSOR_T_SPLIT && w.kind != CVGS_WRITE_PIXEL_3D) return 0;
    if (n_segs < 2 || n_segs > CVGS_MAX_CHAINS || max_batch < 1 || (int64_t)n_segs * max_batch > 65535 || n_planes < 1 || n_planes > kManyInlineLarge) return 0;
    ChainArgs c;
    PwGeom g;
    int prog_id = 0;
    bool f16 = false;
    if (!pointwise4_plan(c_in, 1, chain_flags, c, g, prog_id, f16, nullptr) || prog_id == 3 || f16 || g.packed == 2) return 0;
    if (dry_run) return 1;
    g.narrow = g.w <= 64 ? 2 : (g.w <= 128 ? 1 : 0);
    hipStream_t s = (hipStream_t)stream;
    auto by_prog = [&](auto cn_tag) {
        constexpr int CN = decltype(cn_tag)::value;
        if (prog_id == 0) return launch_pw_many<CN, ProgCastMulSubDiv, float>(c, planes, n_planes, segs, n_segs, max_batch, g, s);
        if (prog_id == 1) return launch_pw_many<CN, ProgCast, float>(c, planes, n_planes, segs, n_segs, max_batch, g, s);
        return launch_pw_many<CN, ArithProg<CN, CVGS_DEPTH_8U>, float>(c, planes, n_planes, segs, n_segs, max_batch, g, s);
    };
    hipError_t e;
    switch (c.read.cn) {
    case 1: e = by_prog(std::integral_constant<int, 1>{}); break;
    case 2: e = by_prog(std::integral_constant<int, 2>{}); break;
    case 3: e = by_prog(std::integral_constant<int, 3>{}); break;
    default: e = by_prog(std::integral_constant<int, 4>{}); break;
    }
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

// Measured and NOT adopted (round 2): running C1 / C2 chains as C4 images of a quarter / half the width (4 / 2 x-adjacent
// pixels are one C4 pixel with periodic operands; byte-identical output, 1601 GPU tests green) so that a thread moves 16 source
// elements instead of 4: a 4K 8UC1 -> 32FC1 chain went from 12.8 to 15.0 us, 32FC1 15.2 -> 16.4 us, 50 crops of 16SC1 5.1 -> 7.7 us.
// A quarter of the waves (8,640: ONE resident round) put every wave in the same phase at the same time -- the load burst and the
// store burst no longer overlap -- whereas 32,400 small waves in four rounds pipeline them.  More waves, not fatter waves.
// Returns 1 if it took the chain, 0 if not eligible, <0 on error.
int launch_pointwise(const ChainArgs& c_in, const PlaneParams* inline_planes, int n_inline, uint32_t chain_flags, void* stream,
                     bool dry_run, LaunchInfo* info) {
    {
        const int rc = launch_pointwise_u8u8(c_in, inline_planes, n_inline, chain_flags, (hipStream_t)stream, dry_run, info);
        if (rc) return rc;
    }
    {
        const int rc = launch_pointwise_u16u16(c_in, inline_planes, n_inline, chain_flags, (hipStream_t)stream, dry_run, info);
        if (rc) return rc;
    }
    {
        const int rc = launch_pointwise_f32f32(c_in, inline_planes, n_inline, chain_flags, (hipStream_t)stream, dry_run, info);
        if (rc) return rc;
    }
    ChainArgs c;
    PwGeom g;
    int prog_id = 0;
    bool f16 = false;
    bool u8o = false;
    if (!pointwise4_plan(c_in, n_inline, chain_flags, c, g, prog_id, f16, &u8o)) return 0;
    g.narrow = g.w <= 64 ? 2 : (g.w <= 128 ? 1 : 0); // batches of small crops (the reference's 60x120 crops): several rows per wave
    const bool yuv = c.read.kind == CVGS_READ_NV12;
    if (yuv) { // the thread's 4 pixels share 2 chroma pairs: even widths (validated for every 4:2:0 plane) and x0 % 4 == 0
        const bool ten = c.read.yuv_layout == CVGS_YUV_P010;
        const bool planar_chroma = c.read.yuv_layout == CVGS_YUV_I420 || c.read.yuv_layout == CVGS_YUV_YV12;
        if (info) {
            static const char* names[3][3] = {{"pointwise4_nv12", "pointwise4_nv12_u8", "pointwise4_nv12_f16"},
                                              {"pointwise4_p010", "pointwise4_p010_u8", "pointwise4_p010_f16"},
                                              {"pointwise4_i420", "pointwise4_i420_u8", "pointwise4_i420_f16"}};
            info->kernel = names[ten ? 1 : (planar_chroma ? 2 : 0)][u8o ? 1 : (f16 ? 2 : 0)];
        }
        if (dry_run) return 1;
        const ProgArgs& p = c.prog;
        const bool norm = p.n == 3 && p.opcode[0] == CVGS_OP_MUL && p.opcode[1] == CVGS_OP_SUB && p.opcode[2] == CVGS_OP_DIV;
        hipStream_t s = (hipStream_t)stream;
        hipError_t e;
        auto go = [&](auto cn_tag, auto sd_tag) {
            constexpr int CN = decltype(cn_tag)::value, SD = decltype(sd_tag)::value;
            using Arith = ArithProg<CN, CVGS_DEPTH_32F>;
            if (u8o) return launch_pw<CN, Arith, uint8_t, SD>(c, inline_planes, n_inline, g, s);
            if (f16) return launch_pw<CN, Arith, _Float16, SD>(c, inline_planes, n_inline, g, s);
            return norm ? launch_pw<CN, ProgMulSubDivPw, float, SD>(c, inline_planes, n_inline, g, s)
                        : launch_pw<CN, Arith, float, SD>(c, inline_planes, n_inline, g, s);
        };
        auto go_cn = [&](auto sd_tag) { return g.cn == 3 ? go(std::integral_constant<int, 3>{}, sd_tag) : go(std::integral_constant<int, 4>{}, sd_tag); };
        e = ten ? go_cn(std::integral_constant<int, SD_P010>{})
                : (planar_chroma ? go_cn(std::integral_constant<int, SD_I420>{}) : go_cn(std::integral_constant<int, SD_NV12>{}));
        return e == hipSuccess ? 1 : -(int)e - 1000;
    }
    if (info) {
        static const char* names[2][3] = {{"pointwise4_u8_cast_mul_sub_div", "pointwise4_u8_cast", "pointwise4_u8_interp"},
                                          {"pointwise4_u8_cast_mul_sub_div_f16", "pointwise4_u8_cast_f16", "pointwise4_u8_interp_f16"}};
        static const char* by_depth[6] = {"", "pointwise4_s8", "pointwise4_u16", "pointwise4_s16", "pointwise4_s32", "pointwise4_f32"};
        info->kernel = prog_id == 3 ? by_depth[c.read.depth] : names[f16][prog_id];
    }
    if (dry_run) return 1;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (prog_id == 3) {
        const ProgArgs& p = c.prog;
        const int o = c.read.depth == CVGS_DEPTH_32F ? 0 : 1; // ops after the leading cast
        const bool norm = p.n == o + 3 && p.opcode[o] == CVGS_OP_MUL && p.opcode[o + 1] == CVGS_OP_SUB && p.opcode[o + 2] == CVGS_OP_DIV &&
                          (o == 0 || (p.opcode[0] == CVGS_OP_CAST && p.aux[0] == CVGS_DEPTH_32F));
        switch (c.read.depth) {
        case CVGS_DEPTH_8S: e = launch_pw_depth<CVGS_DEPTH_8S>(norm, c, inline_planes, n_inline, g, s); break;
        case CVGS_DEPTH_16U: e = launch_pw_depth<CVGS_DEPTH_16U>(norm, c, inline_planes, n_inline, g, s); break;
        case CVGS_DEPTH_16S: e = launch_pw_depth<CVGS_DEPTH_16S>(norm, c, inline_planes, n_inline, g, s); break;
        case CVGS_DEPTH_32S: e = launch_pw_depth<CVGS_DEPTH_32S>(norm, c, inline_planes, n_inline, g, s); break;
        default: e = launch_pw_depth<CVGS_DEPTH_32F>(norm, c, inline_planes, n_inline, g, s); break;
        }
    } else {
        e = f16 ? launch_pw_cn<_Float16>(prog_id, c, inline_planes, n_inline, g, s)
                : launch_pw_cn<float>(prog_id, c, inline_planes, n_inline, g, s);
    }
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

} // namespace cvgs
