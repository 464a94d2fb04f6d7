// k_cvtcolor_u8.hip -- the standalone u8 -> u8 colour conversions of cvGS::cvtColor (reference include/cvGPUSpeedup.cuh:
// 151-161; tests/color/test_cvtColor.cu:105-123) as compile-time programs: RGB <-> BGR, +-alpha (with or without the
// swap) are pure BYTE PERMUTATIONS and never leave the integer registers; *2GRAY does its three multiplies per pixel.
// One thread owns SIXTEEN x-adjacent pixels: CN 16-byte loads, OCN 16-byte non-temporal stores (a 4K BGR -> RGB frame is
// 24.9 MB in + 24.9 MB out: a streaming copy with a shuffle in the middle).  Bit-identical to the interpreted
// pointwise4_u8_u8 kernel, which keeps every other u8 -> u8 program, ragged widths and unaligned images.
#include <cstring>

#include "k_pointwise_body.hpp"

#ifndef CVGS_CC_LOAD
#define CVGS_CC_LOAD(p) __builtin_nontemporal_load(p)
#endif
#ifndef CVGS_CC_STORE
#define CVGS_CC_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif

namespace cvgs {

typedef uint32_t u32x4a __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4a* gp_u32x4;
typedef __attribute__((address_space(1))) u32x4a* gp_u32x4_w;
typedef const __attribute__((address_space(1))) uint8_t* gp_u8c;

enum { PERM_ID = 0, PERM_SWAP = 1 };

// byte of the thread's 16-pixel input chunk that output byte e comes from; -1 - b: byte b of the added alpha.  ES = bytes
// per channel: 1 (CV_8U), or 2 (CV_16U: the reference's cvtColor test sweeps both, tests/color/test_cvtColor.cu:105-123) --
// a channel permutation of 16-bit pixels is a byte permutation of 2 CN-byte pixels.
template <int CN, int OCN, int PERM, int ES>
constexpr int src_byte(int e) {
    const int px = e / (OCN * ES), within = e % (OCN * ES), ch = within / ES, b = within % ES;
    if (ch == 3 && CN == 3) return -1 - b;
    const int sc = ch < 3 ? (PERM == PERM_SWAP ? 2 - ch : ch) : 3;
    return (px * CN + sc) * ES + b;
}

// The wave's 1024 pixels travel through a wave-private LDS region in both directions, so that EVERY global access is a
// fully coalesced 1 KB instruction (lane l touches bytes 16 l .. 16 l + 15 of the k-th KB) while every lane still owns 16
// whole pixels in between: 16-byte chunks at a 48-byte lane stride measured 15.6 us on a 4K BGR -> RGB frame, the coalesced
// form 11 us (profiles/r02_*; the LDS round trips are not what bounds it).  No barrier: a wave's LDS instructions execute in
// order and no other wave touches its region.  ds_read_b128 at a 48-byte lane stride is conflict-free (16-lane groups
// hit 16 distinct 16-byte slots), at 64 bytes (4-channel pixels) 4-way -- still far from the limiter.
template <int CN>
__device__ __forceinline__ void load_chunk16(const gp_u8c row, int tile_px0, int W, int lane, uint32_t* lds_wave, uint32_t* in) {
    const int row_bytes = W * CN, tile_b0 = tile_px0 * CN;
#pragma unroll
    for (int k = 0; k < CN; ++k) {
        const int off = tile_b0 + 1024 * k + 16 * lane;
        u32x4a v = {0, 0, 0, 0};
        if (off < row_bytes) v = CVGS_CC_LOAD((gp_u32x4)(row + (uint32_t)off));
        *(u32x4a*)(lds_wave + 256 * k + 4 * lane) = v;
    }
#pragma unroll
    for (int k = 0; k < CN; ++k) {
        const u32x4a v = *(const u32x4a*)(lds_wave + 4 * CN * lane + 4 * k);
        in[4 * k] = v.x; in[4 * k + 1] = v.y; in[4 * k + 2] = v.z; in[4 * k + 3] = v.w;
    }
}

template <int CN_, int OCN_, int PERM, int NPL, int ES = 1>
__global__ __launch_bounds__(256) void k_u8_permute16(const KernArgs<NPL> a, const PwGeom g, const uint32_t alpha) {
    constexpr int CN = CN_ * ES, OCN = OCN_ * ES; // BYTES per pixel in / out from here on
    constexpr int MAXC = CN > OCN ? CN : OCN;
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][256 * MAXC];
    const ChainArgs& c = a.c;
    const int z = (int)blockIdx.z;
    PlaneParams P;
    if constexpr (NPL == 0) P = c.read.table[z];
    else P = a.planes[z];
    const int W = g.w, H = g.h;
    asm volatile("" ::"s"(W), "s"(H), "s"(P.step), "s"(P.data), "s"(g.out), "s"(g.row_pitch), "s"(g.img_stride));
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int tile_px0 = (int)blockIdx.x * 1024;
    const int y = (int)blockIdx.y * 4 + wave;
    if (y >= H) return; // wave-uniform
    const gp_u8c row = (gp_u8c)P.data + (size_t)y * (size_t)P.step;
    uint32_t* const lw = lds[wave];
    uint32_t in[4 * CN];
    load_chunk16<CN>(row, tile_px0, W, lane, lw, in);
#pragma unroll
    for (int k = 0; k < OCN; ++k) {
        uint32_t q[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t w = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = src_byte<CN_, OCN_, PERM, ES>(16 * k + 4 * d + i);
                const uint32_t b = s < 0 ? ((alpha >> (8 * (-1 - s))) & 0xffu) : ((in[s < 0 ? 0 : (s >> 2)] >> (8 * (s & 3))) & 0xffu);
                w |= b << (8 * i);
            }
            q[d] = w;
        }
        const u32x4a v = {q[0], q[1], q[2], q[3]};
        *(u32x4a*)(lw + 4 * OCN * lane + 4 * k) = v;
    }
    __attribute__((address_space(1))) uint8_t* orow =
        (__attribute__((address_space(1))) uint8_t*)g.out + (size_t)z * (size_t)g.img_stride + (size_t)y * (size_t)g.row_pitch;
    const int out_row_bytes = W * OCN, out_b0 = tile_px0 * OCN;
#pragma unroll
    for (int k = 0; k < OCN; ++k) {
        const int off = out_b0 + 1024 * k + 16 * lane;
        const u32x4a v = *(const u32x4a*)(lw + 256 * k + 4 * lane);
        if (off < out_row_bytes) CVGS_CC_STORE(v, (gp_u32x4_w)(orow + (uint32_t)off));
    }
}

// *2GRAY: 16 pixels -> 16 bytes; the arithmetic is apply_op's (0.299 R + 0.587 G + 0.114 B in that order, round to nearest
// even), the channel order comes with `aux`.  Input through the LDS like the permutations; the 16 output bytes of a lane
// are already lane-contiguous.
// ORD: where R, G, B sit in the source pixel -- 1: channels 0,1,2 (RGB[A]2GRAY), 2: channels 2,1,0 (BGR[A]2GRAY), both resolved
// at compile time; 0: any other order, selected per pixel from `aux` (apply_op).  The luminance of integer pixels is rounded
// to nearest even; for the two compile-time orders the store-side conversion (sat_u8_insert / sat_u16_bits: RNE + clamp,
// k_common.hpp) IS that rounding, so a pixel costs 3 conversions, 3 multiplies, 2 adds and 1 convert-and-insert.
template <int CN, int NPL, int ES = 1, int ORD = 0>
__global__ __launch_bounds__(256) void k_u8_gray16(const KernArgs<NPL> a, const PwGeom g) {
    constexpr int CB = CN * ES; // bytes per source pixel
    constexpr int SD = ES == 1 ? CVGS_DEPTH_8U : CVGS_DEPTH_16U;
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][256 * CB];
    const ChainArgs& c = a.c;
    const int z = (int)blockIdx.z;
    PlaneParams P;
    if constexpr (NPL == 0) P = c.read.table[z];
    else P = a.planes[z];
    const int W = g.w, H = g.h, aux = c.prog.aux[0];
    asm volatile("" ::"s"(W), "s"(H), "s"(P.step), "s"(P.data), "s"(g.out), "s"(g.row_pitch), "s"(g.img_stride), "s"(aux));
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int tile_px0 = (int)blockIdx.x * 1024;
    const int x0 = tile_px0 + lane * 16;
    const int y = (int)blockIdx.y * 4 + wave;
    if (y >= H) return;
    const gp_u8c row = (gp_u8c)P.data + (size_t)y * (size_t)P.step;
    uint32_t in[4 * CB];
    load_chunk16<CB>(row, tile_px0, W, lane, lds[wave], in);
    if (x0 >= W) return;
    uint32_t q[4 * ES];
#pragma unroll
    for (int j = 0; j < 4 * ES; ++j) q[j] = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        Px p;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) p.v[ch] = ch < CN ? elem_value<SD>(in, i * CN + ch) : 0.f;
        if constexpr (ORD == 0) {
            int depth = SD, cn = CN;
            apply_op(CVGS_OP_GRAY, aux, c.prog.operand[0], p, depth, cn);
        } else {
            const float r = ORD == 1 ? p.v[0] : p.v[2], b = ORD == 1 ? p.v[2] : p.v[0];
            p.v[0] = (r * 0.299f + p.v[1] * 0.587f) + b * 0.114f; // apply_op's expression, rounded by the conversion below
        }
        if constexpr (ES == 1) q[i >> 2] = sat_u8_insert(p.v[0], (uint32_t)(i & 3), q[i >> 2]);
        else q[i >> 1] |= (ORD == 0 ? ((uint32_t)p.v[0] & 0xffffu) : sat_u16_bits(p.v[0])) << (16 * (i & 1));
    }
    __attribute__((address_space(1))) uint8_t* orow =
        (__attribute__((address_space(1))) uint8_t*)g.out + (size_t)z * (size_t)g.img_stride + (size_t)y * (size_t)g.row_pitch;
#pragma unroll
    for (int j = 0; j < ES; ++j) { // 16 pixels = ES lane-contiguous 16-byte chunks
        u32x4a v = {q[4 * j], q[4 * j + 1], q[4 * j + 2], q[4 * j + 3]};
        CVGS_CC_STORE(v, (gp_u32x4_w)(orow + (uint32_t)x0 * ES + 16 * j));
    }
}

// CV_32F -> CV_32F colour conversions (the reference's cvtColor test sweeps float types too, tests/color/test_cvtColor.cu:105-123):
// four pixels per thread -- the same 16 CN bytes per lane as sixteen u8 pixels, through the same wave-private LDS region, so every
// global access is a fully coalesced 1 KB instruction.  MODE 0 / 1: channel permutation (identity / R<->B swap, +- alpha);
// MODE 2 / 3: *2GRAY with R, G, B at channels 0,1,2 / 2,1,0 (apply_op's expression: (R 0.299 + G 0.587) + B 0.114, no rounding for
// float pixels); every other order stays on the interpreted kernel.
template <int CN, int OCN, int MODE, int NPL>
__global__ __launch_bounds__(256) void k_f32_colour4(const KernArgs<NPL> a, const PwGeom g, const uint32_t alpha_bits) {
    constexpr int MAXC = CN > OCN ? CN : OCN;
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][256 * MAXC];
    const ChainArgs& c = a.c;
    const int z = (int)blockIdx.z;
    PlaneParams P;
    if constexpr (NPL == 0) P = c.read.table[z];
    else P = a.planes[z];
    const int W = g.w, H = g.h;
    asm volatile("" ::"s"(W), "s"(H), "s"(P.step), "s"(P.data), "s"(g.out), "s"(g.row_pitch), "s"(g.img_stride));
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int tile_px0 = (int)blockIdx.x * 256; // 4 pixels per lane
    const int y = (int)blockIdx.y * 4 + wave;
    if (y >= H) return; // wave-uniform
    const gp_u8c row = (gp_u8c)P.data + (size_t)y * (size_t)P.step;
    uint32_t* const lw = lds[wave];
    uint32_t in[4 * CN];
    load_chunk16<CN>(row, tile_px0 * 4, W * 4, lane, lw, in); // (in units of 4-byte "pixels" of CN bytes: the same byte layout)
    uint32_t out[4 * OCN];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (MODE < 2) {
#pragma unroll
            for (int ch = 0; ch < OCN; ++ch) {
                if (ch == 3 && CN == 3) out[i * OCN + ch] = alpha_bits;
                else out[i * OCN + ch] = in[i * CN + (ch < 3 ? (MODE == 1 ? 2 - ch : ch) : 3)];
            }
        } else {
            const float r = __uint_as_float(in[i * CN + (MODE == 2 ? 0 : 2)]), gg = __uint_as_float(in[i * CN + 1]), b = __uint_as_float(in[i * CN + (MODE == 2 ? 2 : 0)]);
            out[i] = __float_as_uint((r * 0.299f + gg * 0.587f) + b * 0.114f);
        }
    }
    __attribute__((address_space(1))) uint8_t* orow =
        (__attribute__((address_space(1))) uint8_t*)g.out + (size_t)z * (size_t)g.img_stride + (size_t)y * (size_t)g.row_pitch;
    const int out_row_bytes = W * OCN * 4, out_b0 = tile_px0 * OCN * 4;
#pragma unroll
    for (int k = 0; k < OCN; ++k) {
        const u32x4a v = {out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]};
        *(u32x4a*)(lw + 4 * OCN * lane + 4 * k) = v;
    }
#pragma unroll
    for (int k = 0; k < OCN; ++k) {
        const int off = out_b0 + 1024 * k + 16 * lane;
        const u32x4a v = *(const u32x4a*)(lw + 256 * k + 4 * lane);
        if (off < out_row_bytes) CVGS_CC_STORE(v, (gp_u32x4_w)(orow + (uint32_t)off));
    }
}

template <int NPL>
static void fill_args(KernArgs<NPL>& a, const ChainArgs& c, const PlaneParams* ip, int ni) {
    a.c = c;
    if constexpr (NPL > 0) {
        for (int i = 0; i < NPL; ++i) a.planes[i] = i < ni ? ip[i] : PlaneParams{};
    } else {
        a.planes[0] = PlaneParams{};
    }
}

template <int CN, int OCN, int PERM, int ES = 1>
static hipError_t launch_perm(const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, uint32_t alpha, hipStream_t s) {
    const dim3 grid((g.w / 16 + 63) / 64, (g.h + 3) / 4, c.read.batch);
    if (c.read.table) {
        KernArgs<0> a;
        fill_args(a, c, ip, ni);
        hipLaunchKernelGGL((k_u8_permute16<CN, OCN, PERM, 0, ES>), grid, dim3(256), 0, s, a, g, alpha);
    } else {
        KernArgs<CVGS_KERNARG_PLANES> a;
        fill_args(a, c, ip, ni);
        hipLaunchKernelGGL((k_u8_permute16<CN, OCN, PERM, CVGS_KERNARG_PLANES, ES>), grid, dim3(256), 0, s, a, g, alpha);
    }
    return hipGetLastError();
}
template <int CN, int ES, int ORD>
static hipError_t launch_gray_ord(const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, hipStream_t s) {
    const dim3 grid((g.w / 16 + 63) / 64, (g.h + 3) / 4, c.read.batch);
    if (c.read.table) {
        KernArgs<0> a;
        fill_args(a, c, ip, ni);
        hipLaunchKernelGGL((k_u8_gray16<CN, 0, ES, ORD>), grid, dim3(256), 0, s, a, g);
    } else {
        KernArgs<CVGS_KERNARG_PLANES> a;
        fill_args(a, c, ip, ni);
        hipLaunchKernelGGL((k_u8_gray16<CN, CVGS_KERNARG_PLANES, ES, ORD>), grid, dim3(256), 0, s, a, g);
    }
    return hipGetLastError();
}
template <int CN, int ES = 1>
static hipError_t launch_gray(const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, hipStream_t s) {
    const int order = c.prog.aux[0] & 63; // (R, G, B) channel indices, 2 bits each
    if (order == (0 | (1 << 2) | (2 << 4))) return launch_gray_ord<CN, ES, 1>(c, ip, ni, g, s);
    if (order == (2 | (1 << 2) | (0 << 4))) return launch_gray_ord<CN, ES, 2>(c, ip, ni, g, s);
    return launch_gray_ord<CN, ES, 0>(c, ip, ni, g, s);
}

// Returns 1 if it took the chain, 0 if not eligible, <0 on error.  `g` is the packed-output geometry of pointwise4_u8_u8.
int launch_u8_colour16(const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, void* stream, bool dry_run, LaunchInfo* info) {
    const ReadArgs& r = c.read;
    const ProgArgs& p = c.prog;
    if (p.n != 1 || r.used != r.batch || (g.w & 15) || r.cn < 3) return 0;
    const int op = p.opcode[0], aux = p.aux[0];
    constexpr int kId3 = 0 | (1 << 2) | (2 << 4), kSwap3 = 2 | (1 << 2) | (0 << 4);
    int perm = -1;
    bool gray = false;
    if (op == CVGS_OP_REORDER) {
        if ((r.cn == 3 && aux == kSwap3) || (r.cn == 4 && aux == (kSwap3 | (3 << 6)))) perm = PERM_SWAP;
    } else if (op == CVGS_OP_ADD_ALPHA || op == CVGS_OP_DROP_ALPHA) {
        if ((aux & 63) == kId3) perm = PERM_ID;
        else if ((aux & 63) == kSwap3) perm = PERM_SWAP;
    } else if (op == CVGS_OP_GRAY) {
        gray = true;
    }
    if (perm < 0 && !gray) return 0;
    const bool wide = r.depth == CVGS_DEPTH_16U; // 16-bit channels (the caller passes 8U -> 8U or 16U -> 16U chains only)
    uint32_t alpha = 0;
    if (op == CVGS_OP_ADD_ALPHA) { // the interpreted kernel stores (uint8_t / uint16_t)operand[0]: take whole values in range only
        const float av = p.operand[0][0];
        if (!(av >= 0.f && av <= (wide ? 65535.f : 255.f)) || av != (float)(int)av) return 0;
        alpha = (uint32_t)(int)av;
    }
    // 16-byte accesses: every source row, the output base, its row pitch and image stride must be 16-byte aligned
    if (((uintptr_t)g.out & 15) || (g.row_pitch & 15) || (g.img_stride & 15)) return 0;
    if (r.table) return 0; // resident tables: alignment cannot be checked on the host
    for (int i = 0; i < ni; ++i)
        if (((uintptr_t)ip[i].data & 15) || (ip[i].step & 15)) return 0;
    if (info) info->kernel = wide ? (gray ? "pointwise16_u16_gray" : "pointwise16_u16_permute") : (gray ? "pointwise16_u8_gray" : "pointwise16_u8_permute");
    if (dry_run) return 1;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (wide) {
        if (gray) e = r.cn == 3 ? launch_gray<3, 2>(c, ip, ni, g, s) : launch_gray<4, 2>(c, ip, ni, g, s);
        else if (op == CVGS_OP_REORDER) e = r.cn == 3 ? launch_perm<3, 3, PERM_SWAP, 2>(c, ip, ni, g, 0, s) : launch_perm<4, 4, PERM_SWAP, 2>(c, ip, ni, g, 0, s);
        else if (op == CVGS_OP_ADD_ALPHA) e = perm == PERM_SWAP ? launch_perm<3, 4, PERM_SWAP, 2>(c, ip, ni, g, alpha, s) : launch_perm<3, 4, PERM_ID, 2>(c, ip, ni, g, alpha, s);
        else e = perm == PERM_SWAP ? launch_perm<4, 3, PERM_SWAP, 2>(c, ip, ni, g, 0, s) : launch_perm<4, 3, PERM_ID, 2>(c, ip, ni, g, 0, s);
        return e == hipSuccess ? 1 : -(int)e - 1000;
    }
    if (gray) e = r.cn == 3 ? launch_gray<3>(c, ip, ni, g, s) : launch_gray<4>(c, ip, ni, g, s);
    else if (op == CVGS_OP_REORDER) e = r.cn == 3 ? launch_perm<3, 3, PERM_SWAP>(c, ip, ni, g, 0, s) : launch_perm<4, 4, PERM_SWAP>(c, ip, ni, g, 0, s);
    else if (op == CVGS_OP_ADD_ALPHA) e = perm == PERM_SWAP ? launch_perm<3, 4, PERM_SWAP>(c, ip, ni, g, alpha, s) : launch_perm<3, 4, PERM_ID>(c, ip, ni, g, alpha, s);
    else e = perm == PERM_SWAP ? launch_perm<4, 3, PERM_SWAP>(c, ip, ni, g, 0, s) : launch_perm<4, 3, PERM_ID>(c, ip, ni, g, 0, s);
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

template <int CN, int OCN, int MODE>
static hipError_t launch_f32c(const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, uint32_t alpha_bits, hipStream_t s) {
    const dim3 grid((g.w / 4 + 63) / 64, (g.h + 3) / 4, c.read.batch);
    KernArgs<CVGS_KERNARG_PLANES> a;
    fill_args(a, c, ip, ni);
    hipLaunchKernelGGL((k_f32_colour4<CN, OCN, MODE, CVGS_KERNARG_PLANES>), grid, dim3(256), 0, s, a, g, alpha_bits);
    return hipGetLastError();
}

// CV_32F -> CV_32F chains that are ONE colour conversion.  Returns 1 if it took the chain, 0 if not eligible, <0 on error.
int launch_f32_colour4(const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, void* stream, bool dry_run, LaunchInfo* info) {
    const ReadArgs& r = c.read;
    const ProgArgs& p = c.prog;
    if (p.n != 1 || r.used != r.batch || (g.w & 3) || r.cn < 3 || r.table) return 0;
    const int op = p.opcode[0], aux = p.aux[0];
    constexpr int kId3 = 0 | (1 << 2) | (2 << 4), kSwap3 = 2 | (1 << 2) | (0 << 4);
    int mode = -1;
    if (op == CVGS_OP_REORDER) {
        if ((r.cn == 3 && aux == kSwap3) || (r.cn == 4 && aux == (kSwap3 | (3 << 6)))) mode = 1;
    } else if (op == CVGS_OP_ADD_ALPHA || op == CVGS_OP_DROP_ALPHA) {
        if ((aux & 63) == kId3) mode = 0;
        else if ((aux & 63) == kSwap3) mode = 1;
    } else if (op == CVGS_OP_GRAY) {
        if ((aux & 63) == kId3) mode = 2;
        else if ((aux & 63) == kSwap3) mode = 3;
    }
    if (mode < 0) return 0;
    uint32_t alpha_bits = 0;
    if (op == CVGS_OP_ADD_ALPHA) std::memcpy(&alpha_bits, &p.operand[0][0], 4); // the interpreted kernel stores operand[0] as it is
    if (((uintptr_t)g.out & 15) || (g.row_pitch & 15) || (g.img_stride & 15)) return 0;
    for (int i = 0; i < ni; ++i)
        if (((uintptr_t)ip[i].data & 15) || (ip[i].step & 15)) return 0;
    if (info) info->kernel = mode >= 2 ? "pointwise4_f32_gray" : "pointwise4_f32_permute";
    if (dry_run) return 1;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (mode >= 2) {
        if (r.cn == 3) e = mode == 2 ? launch_f32c<3, 1, 2>(c, ip, ni, g, 0, s) : launch_f32c<3, 1, 3>(c, ip, ni, g, 0, s);
        else e = mode == 2 ? launch_f32c<4, 1, 2>(c, ip, ni, g, 0, s) : launch_f32c<4, 1, 3>(c, ip, ni, g, 0, s);
    } else if (op == CVGS_OP_REORDER) {
        e = r.cn == 3 ? launch_f32c<3, 3, 1>(c, ip, ni, g, 0, s) : launch_f32c<4, 4, 1>(c, ip, ni, g, 0, s);
    } else if (op == CVGS_OP_ADD_ALPHA) {
        e = mode == 1 ? launch_f32c<3, 4, 1>(c, ip, ni, g, alpha_bits, s) : launch_f32c<3, 4, 0>(c, ip, ni, g, alpha_bits, s);
    } else {
        e = mode == 1 ? launch_f32c<4, 3, 1>(c, ip, ni, g, 0, s) : launch_f32c<4, 3, 0>(c, ip, ni, g, 0, s);
    }
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

} // namespace cvgs
