// k_k1_exp.hip -- EXPERIMENTAL variants of the K1 kernel for within-process A/B ablation (tools/k1_ab.py).
// Selected with bits 8..15 of cvgs_chain_desc.flags; never chosen by the normal dispatch.  u8c3 + the
// REORDER,MUL,SUB,DIV program only.  Results must stay bit-identical to the production kernel (tests check).
#include <initializer_list>
#include <type_traits>

#include "k_common.hpp"

namespace cvgs {

using ProgRMSD = StaticProg<CVGS_OP_REORDER, CVGS_OP_MUL, CVGS_OP_SUB, CVGS_OP_DIV>;

struct XGeom {
    uint32_t col_tiles, tiles_per_plane, total_tiles, padded_tiles;
    int64_t img_stride, ch_stride;
};

enum { MODE_FULL = 0, MODE_NOLOAD = 1, MODE_NOSTORE = 2, MODE_EMPTY = 3, MODE_COMPUTE = 4 };
enum { ST_PLAIN = 0, ST_NT = 1, ST_SC1 = 2 };

template <int STORE>
__device__ __forceinline__ void st(float* p, float v) {
    if constexpr (STORE == ST_NT) __builtin_nontemporal_store(v, p);
    else if constexpr (STORE == ST_SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

__device__ __forceinline__ uint64_t ld64(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

template <int THREADS, int RPW, int STORE, bool REMAP, int MODE, int NPL>
__global__ __launch_bounds__(THREADS) void k1_exp(const KernArgs<NPL> a, const XGeom g) {
    constexpr int CN = 3;
    constexpr int WAVES = THREADS / 64;
    const ChainArgs& c = a.c;
    const ReadArgs& r = c.read;
    const uint32_t bid = REMAP ? xcd_remap(blockIdx.x, g.padded_tiles) : blockIdx.x;
    if (bid >= g.total_tiles) return;
    const int z = (int)(bid / g.tiles_per_plane);
    const uint32_t t = bid - (uint32_t)z * g.tiles_per_plane;
    const int col_tile = (int)(t % g.col_tiles);
    const int row_tile = (int)(t / g.col_tiles);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int x = col_tile * 64 + lane;
    const int row0 = (row_tile * WAVES + wave) * RPW;
    if (row0 >= r.dst_h) return;
    const bool x_ok = x < r.dst_w;
    float* const out = (float*)c.write.data + (int64_t)z * g.img_stride + x;
    const int W = c.write.width;
    if constexpr (MODE == MODE_EMPTY) {
        if (x == 0x7fffffff) out[0] = 0.f;
        return;
    }

    Px bgp;
    int bdepth = CVGS_DEPTH_32F, bcn = CN;
#pragma unroll
    for (int k = 0; k < 4; ++k) bgp.v[k] = r.bg[k];
    ProgRMSD::run(c.prog, bgp, bdepth, bcn);

    if (z >= r.used) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int y = row0 + j;
            if (y < r.dst_h && x_ok)
#pragma unroll
                for (int k = 0; k < CN; ++k) st<STORE>(out + (int64_t)k * g.ch_stride + (int64_t)y * W, bgp.v[k]);
        }
        return;
    }
    PlaneParams P;
    if constexpr (NPL == 0) P = r.table[z];
    else P = a.planes[z];

    const bool in_x = x >= P.x1 && x <= P.x2;
    const int xr = in_x ? x - P.x1 : 0;
    const float sx = (float)xr * P.fx;
    const int x1 = (int)floorf(sx);
    const int x2 = x1 + 1;
    const float wxa = (float)x2 - sx;
    const float wxb = sx - (float)x1;
    const bool edge = x2 > P.w - 1;
    const int row_bytes = P.w * CN;
    const int o = x1 * CN;
    const int ol = min(o, max(row_bytes - 8, 0));
    const int sh = (o - ol) * 8;

    uint64_t va[RPW], vb[RPW];
    float wya[RPW], wyb[RPW];
    bool in_y[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = min(row0 + j, r.dst_h - 1);
        in_y[j] = y >= P.y1 && y <= P.y2;
        const int yr = in_y[j] ? y - P.y1 : 0;
        const float sy = (float)yr * P.fy;
        const int y1 = (int)floorf(sy);
        const int y2 = y1 + 1;
        const int y2r = min(y2, P.h - 1);
        wya[j] = (float)y2 - sy;
        wyb[j] = sy - (float)y1;
        if constexpr (MODE == MODE_NOLOAD || MODE == MODE_COMPUTE) {
            va[j] = 0x0102030405060708ull + (uint64_t)y1 + (uint64_t)ol;
            vb[j] = 0x1112131415161718ull + (uint64_t)y2r;
        } else {
            va[j] = ld64(P.data + (size_t)y1 * (size_t)P.step + ol);
            vb[j] = ld64(P.data + (size_t)y2r * (size_t)P.step + ol);
        }
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = row0 + j;
        const uint64_t A = va[j] >> sh, B = vb[j] >> sh;
        const uint32_t al = (uint32_t)A, ah = (uint32_t)(A >> 32), bl = (uint32_t)B, bh = (uint32_t)(B >> 32);
        float p00[3] = {(float)(al & 0xff), (float)((al >> 8) & 0xff), (float)((al >> 16) & 0xff)};
        float p10[3] = {(float)(al >> 24), (float)(ah & 0xff), (float)((ah >> 8) & 0xff)};
        float p01[3] = {(float)(bl & 0xff), (float)((bl >> 8) & 0xff), (float)((bl >> 16) & 0xff)};
        float p11[3] = {(float)(bl >> 24), (float)(bh & 0xff), (float)((bh >> 8) & 0xff)};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p10[k] = edge ? p00[k] : p10[k];
            p11[k] = edge ? p01[k] : p11[k];
        }
        const float w00 = wxa * wya[j], w10 = wxb * wya[j], w01 = wxa * wyb[j], w11 = wxb * wyb[j];
        Px p;
        p.v[3] = 0.f;
#pragma unroll
        for (int k = 0; k < CN; ++k) {
            float acc = p00[k] * w00;
            acc = acc + p10[k] * w10;
            acc = acc + p01[k] * w01;
            acc = acc + p11[k] * w11;
            p.v[k] = acc;
        }
        int depth = CVGS_DEPTH_32F, cn = CN;
        ProgRMSD::run(c.prog, p, depth, cn);
        const bool inside = in_x && in_y[j];
        if constexpr (MODE == MODE_NOSTORE || MODE == MODE_COMPUTE) {
#pragma unroll
            for (int k = 0; k < CN; ++k) asm volatile("" ::"v"(inside ? p.v[k] : bgp.v[k]));
        } else if (y < r.dst_h && x_ok) {
#pragma unroll
            for (int k = 0; k < CN; ++k)
                st<STORE>(out + (int64_t)k * g.ch_stride + (int64_t)y * W, inside ? p.v[k] : bgp.v[k]);
        }
    }
}

template <int THREADS, int RPW, int STORE, bool REMAP, int MODE>
static hipError_t xlaunch(const ChainArgs& c, const PlaneParams* ip, int ni, hipStream_t s) {
    XGeom g;
    const int rows_per_wg = (THREADS / 64) * RPW;
    g.col_tiles = (uint32_t)((c.read.dst_w + 63) / 64);
    const uint32_t row_tiles = (uint32_t)((c.read.dst_h + rows_per_wg - 1) / rows_per_wg);
    g.tiles_per_plane = g.col_tiles * row_tiles;
    g.total_tiles = g.tiles_per_plane * (uint32_t)c.read.batch;
    g.padded_tiles = REMAP ? (g.total_tiles + 7u) / 8u * 8u : g.total_tiles;
    g.img_stride = c.write.img_stride;
    g.ch_stride = c.write.ch_stride;
    if (c.read.table) {
        KernArgs<0> a;
        a.c = c;
        a.planes[0] = PlaneParams{};
        hipLaunchKernelGGL((k1_exp<THREADS, RPW, STORE, REMAP, MODE, 0>), dim3(g.padded_tiles), dim3(THREADS), 0, s, a, g);
    } else {
        KernArgs<CVGS_KERNARG_PLANES> a;
        a.c = c;
        for (int i = 0; i < ni; ++i) a.planes[i] = ip[i];
        for (int i = ni; i < CVGS_KERNARG_PLANES; ++i) a.planes[i] = PlaneParams{};
        hipLaunchKernelGGL((k1_exp<THREADS, RPW, STORE, REMAP, MODE, CVGS_KERNARG_PLANES>), dim3(g.padded_tiles),
                           dim3(THREADS), 0, s, a, g);
    }
    return hipGetLastError();
}


// ---- v2: global-address-space loads with uniform base + 32-bit lane offset, lanes beyond the target exit at
// once (no per-store divergence), background program evaluated only when a plane needs it, 2D grid (plane index =
// blockIdx.y: no integer division, no XCD remap).
typedef uint64_t u64_unaligned __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) u64_unaligned* gptr_u64;
typedef const __attribute__((address_space(1))) uint8_t* gptr_u8;

// crop rows narrower than 8 bytes: byte gather, never past the row (rare; kept out of line)
__device__ __forceinline__ uint64_t gather_pair(gptr_u8 row, int o, int row_bytes) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t b = row[min(o + k, row_bytes - 1)]; // always inside the row; bytes past it are never used
        if (k < 4) lo |= b << (8 * k);
        else hi |= b << (8 * (k - 4));
    }
    return ((uint64_t)hi << 32) | lo;
}

using ProgRMSM = StaticProg<CVGS_OP_REORDER, CVGS_OP_MUL, CVGS_OP_SUB, CVGS_OP_MUL>; // ablation: no division
using ProgR = StaticProg<CVGS_OP_REORDER>;                                           // ablation: no arithmetic

template <int THREADS, int RPW, int STORE, int NPL, class PROG = ProgRMSD>
__global__ __launch_bounds__(THREADS) void k1_v2(const KernArgs<NPL> a, const XGeom g) {
    constexpr int CN = 3;
    constexpr int WAVES = THREADS / 64;
    const ChainArgs& c = a.c;
    const ReadArgs& r = c.read;
    const int z = (int)blockIdx.y;
    int col_tile = 0, row_tile = (int)blockIdx.x;
    if (g.col_tiles > 1) {
        col_tile = (int)(blockIdx.x % g.col_tiles);
        row_tile = (int)(blockIdx.x / g.col_tiles);
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int x = col_tile * 64 + lane;
    const int row0 = (row_tile * WAVES + wave) * RPW;
    if (row0 >= r.dst_h || x >= r.dst_w) return;
    const int W = c.write.width;
    float* const out = (float*)c.write.data + (int64_t)z * g.img_stride;

    PlaneParams P;
    if constexpr (NPL == 0) P = r.table[z < r.used ? z : 0];
    else P = a.planes[z];
    const bool whole = z < r.used && P.x1 == 0 && P.y1 == 0 && P.x2 == r.dst_w - 1 && P.y2 == r.dst_h - 1; // uniform

    Px bgp;
    bgp.v[0] = bgp.v[1] = bgp.v[2] = bgp.v[3] = 0.f;
    if (!whole) {
        int bdepth = CVGS_DEPTH_32F, bcn = CN;
#pragma unroll
        for (int k = 0; k < 4; ++k) bgp.v[k] = r.bg[k];
        PROG::run(c.prog, bgp, bdepth, bcn);
        if (z >= r.used) {
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const int y = row0 + j;
                if (y < r.dst_h)
#pragma unroll
                    for (int k = 0; k < CN; ++k) st<STORE>(out + (int64_t)k * g.ch_stride + (int64_t)y * W + x, bgp.v[k]);
            }
            return;
        }
    }

    const bool in_x = x >= P.x1 && x <= P.x2;
    const int xr = in_x ? x - P.x1 : 0;
    const float sx = (float)xr * P.fx;
    const int x1 = (int)floorf(sx);
    const int x2 = x1 + 1;
    const float wxa = (float)x2 - sx;
    const float wxb = sx - (float)x1;
    const bool edge = x2 > P.w - 1;
    const int row_bytes = P.w * CN;
    const int o = x1 * CN;
    const bool tiny = row_bytes < 8;
    const uint32_t ol = (uint32_t)(tiny ? o : min(o, row_bytes - 8));
    const int sh = (o - (int)ol) * 8;
    const gptr_u8 src = (gptr_u8)P.data;

    uint64_t va[RPW], vb[RPW];
    float wya[RPW], wyb[RPW];
    bool in_y[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = min(row0 + j, r.dst_h - 1);
        in_y[j] = y >= P.y1 && y <= P.y2;
        const int yr = in_y[j] ? y - P.y1 : 0;
        const float sy = (float)yr * P.fy;
        const int y1 = (int)floorf(sy);
        const int y2 = y1 + 1;
        const int y2r = min(y2, P.h - 1);
        wya[j] = (float)y2 - sy;
        wyb[j] = sy - (float)y1;
        const gptr_u8 ra = src + (size_t)__builtin_amdgcn_readfirstlane(y1) * (size_t)P.step;
        const gptr_u8 rb = src + (size_t)__builtin_amdgcn_readfirstlane(y2r) * (size_t)P.step;
        if (!tiny) {
            va[j] = *(gptr_u64)(ra + ol);
            vb[j] = *(gptr_u64)(rb + ol);
        } else {
            va[j] = gather_pair(ra, o, row_bytes);
            vb[j] = gather_pair(rb, o, row_bytes);
        }
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = row0 + j;
        if (y < r.dst_h) { // uniform
        const uint64_t A = va[j] >> sh, B = vb[j] >> sh;
        const uint32_t al = (uint32_t)A, ah = (uint32_t)(A >> 32), bl = (uint32_t)B, bh = (uint32_t)(B >> 32);
        float p00[3] = {(float)(al & 0xff), (float)((al >> 8) & 0xff), (float)((al >> 16) & 0xff)};
        float p10[3] = {(float)(al >> 24), (float)(ah & 0xff), (float)((ah >> 8) & 0xff)};
        float p01[3] = {(float)(bl & 0xff), (float)((bl >> 8) & 0xff), (float)((bl >> 16) & 0xff)};
        float p11[3] = {(float)(bl >> 24), (float)(bh & 0xff), (float)((bh >> 8) & 0xff)};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p10[k] = edge ? p00[k] : p10[k];
            p11[k] = edge ? p01[k] : p11[k];
        }
        const float w00 = wxa * wya[j], w10 = wxb * wya[j], w01 = wxa * wyb[j], w11 = wxb * wyb[j];
        Px p;
        p.v[3] = 0.f;
#pragma unroll
        for (int k = 0; k < CN; ++k) {
            float acc = p00[k] * w00;
            acc = acc + p10[k] * w10;
            acc = acc + p01[k] * w01;
            acc = acc + p11[k] * w11;
            p.v[k] = acc;
        }
        int depth = CVGS_DEPTH_32F, cn = CN;
        PROG::run(c.prog, p, depth, cn);
        float* const orow = out + (int64_t)y * W; // uniform
        const bool take = whole || (in_x && in_y[j]);
#pragma unroll
        for (int k = 0; k < CN; ++k) st<STORE>(orow + (int64_t)k * g.ch_stride + x, take ? p.v[k] : bgp.v[k]);
        }
    }
}

template <int THREADS, int RPW, int STORE, class PROG = ProgRMSD>
static hipError_t v2launch(const ChainArgs& c, const PlaneParams* ip, int ni, hipStream_t s) {
    XGeom g;
    const int rows_per_wg = (THREADS / 64) * RPW;
    g.col_tiles = (uint32_t)((c.read.dst_w + 63) / 64);
    const uint32_t row_tiles = (uint32_t)((c.read.dst_h + rows_per_wg - 1) / rows_per_wg);
    g.tiles_per_plane = g.col_tiles * row_tiles;
    g.total_tiles = g.tiles_per_plane * (uint32_t)c.read.batch;
    g.padded_tiles = g.total_tiles;
    g.img_stride = c.write.img_stride;
    g.ch_stride = c.write.ch_stride;
    const dim3 grid(g.tiles_per_plane, c.read.batch);
    if (c.read.table) {
        KernArgs<0> a;
        a.c = c;
        a.planes[0] = PlaneParams{};
        hipLaunchKernelGGL((k1_v2<THREADS, RPW, STORE, 0, PROG>), grid, dim3(THREADS), 0, s, a, g);
    } else {
        KernArgs<CVGS_KERNARG_PLANES> a;
        a.c = c;
        for (int i = 0; i < ni; ++i) a.planes[i] = ip[i];
        for (int i = ni; i < CVGS_KERNARG_PLANES; ++i) a.planes[i] = PlaneParams{};
        hipLaunchKernelGGL((k1_v2<THREADS, RPW, STORE, CVGS_KERNARG_PLANES, PROG>), grid, dim3(THREADS), 0, s, a, g);
    }
    return hipGetLastError();
}

// ---- LDS-staged variant (the north star's suggestion), for the A/B record --------------------------------------
// Each wave stages the byte span of its source rows into its own LDS slots with coalesced dword loads
// (lane i loads dword i, i+64, ...), then every lane reads its 2x2 taps from LDS.  No cross-wave sharing, so no
// barrier; wave-local use of the LDS as a gather buffer.  Experimental limits: crops up to 512 px wide, and the
// staged span is rounded out to whole dwords (may touch <= 3 bytes beyond the crop row inside the frame).
constexpr int kLdsSlot = 1552; // 512 * 3 + 16

template <int RPW, int STORE, int NPL>
__global__ __launch_bounds__(256) void k1_lds(const KernArgs<NPL> a, const XGeom g) {
    constexpr int CN = 3;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const ChainArgs& c = a.c;
    const ReadArgs& r = c.read;
    const int z = (int)blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int x = lane;
    const int row0 = ((int)blockIdx.x * 4 + wave) * RPW;
    PlaneParams P;
    if constexpr (NPL == 0) P = r.table[z];
    else P = a.planes[z];
    if (row0 >= r.dst_h) return;
    const int W = c.write.width;
    float* const out = (float*)c.write.data + (int64_t)z * g.img_stride;

    const float sx = (float)x * P.fx;
    const int x1 = (int)floorf(sx);
    const int x2 = x1 + 1;
    const float wxa = (float)x2 - sx, wxb = sx - (float)x1;
    const bool edge = x2 > P.w - 1;
    const int x2r = edge ? x1 : x2;
    const gptr_u8 src = (gptr_u8)P.data;
    // span of the row this wave needs: columns [0, last tap] (lane 0 taps column 0)
    const int last_col = min((int)floorf((float)(min(r.dst_w, 64) - 1) * P.fx) + 1, P.w - 1);
    const int span_bytes = (last_col + 1) * CN;
    uint8_t* const my = smem + (size_t)wave * (2 * RPW) * kLdsSlot;

    float wya[RPW], wyb[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = min(row0 + j, r.dst_h - 1);
        const float sy = (float)y * P.fy;
        const int y1 = (int)floorf(sy);
        const int y2r = min(y1 + 1, P.h - 1);
        wya[j] = (float)(y1 + 1) - sy;
        wyb[j] = sy - (float)y1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const gptr_u8 row = src + (size_t)__builtin_amdgcn_readfirstlane(t ? y2r : y1) * (size_t)P.step;
            const uint32_t mis = (uint32_t)((uintptr_t)row & 3u);   // stage from the dword that contains byte 0
            const __attribute__((address_space(1))) uint32_t* g32 = (const __attribute__((address_space(1))) uint32_t*)(row - mis);
            uint32_t* l32 = (uint32_t*)(my + (size_t)(2 * j + t) * kLdsSlot);
            const int n_dw = (int)(mis + span_bytes + 3) >> 2;
            for (int i = lane; i < n_dw; i += 64) l32[i] = g32[i];
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = row0 + j;
        if (y < r.dst_h) {
            float p00[3], p10[3], p01[3], p11[3];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int yy = (int)floorf((float)min(y, r.dst_h - 1) * P.fy);
                const int ysel = t ? min(yy + 1, P.h - 1) : yy;
                const gptr_u8 row = src + (size_t)__builtin_amdgcn_readfirstlane(ysel) * (size_t)P.step;
                const uint32_t mis = (uint32_t)((uintptr_t)row & 3u);
                const uint8_t* l = my + (size_t)(2 * j + t) * kLdsSlot + mis;
                float* pa = t ? p01 : p00;
                float* pb = t ? p11 : p10;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    pa[k] = (float)l[x1 * CN + k];
                    pb[k] = (float)l[x2r * CN + k];
                }
            }
            const float w00 = wxa * wya[j], w10 = wxb * wya[j], w01 = wxa * wyb[j], w11 = wxb * wyb[j];
            Px p;
            p.v[3] = 0.f;
#pragma unroll
            for (int k = 0; k < CN; ++k) {
                float acc = p00[k] * w00;
                acc = acc + p10[k] * w10;
                acc = acc + p01[k] * w01;
                acc = acc + p11[k] * w11;
                p.v[k] = acc;
            }
            int depth = CVGS_DEPTH_32F, cn = CN;
            ProgRMSD::run(c.prog, p, depth, cn);
            float* const orow = out + (int64_t)y * W;
#pragma unroll
            for (int k = 0; k < CN; ++k) st<STORE>(orow + (int64_t)k * g.ch_stride + x, p.v[k]);
        }
    }
}

template <int RPW, int STORE>
static hipError_t ldslaunch(const ChainArgs& c, const PlaneParams* ip, int ni, hipStream_t s) {
    XGeom g;
    const int rows_per_wg = 4 * RPW;
    g.col_tiles = 1;
    const uint32_t row_tiles = (uint32_t)((c.read.dst_h + rows_per_wg - 1) / rows_per_wg);
    g.tiles_per_plane = row_tiles;
    g.total_tiles = row_tiles * (uint32_t)c.read.batch;
    g.padded_tiles = g.total_tiles;
    g.img_stride = c.write.img_stride;
    g.ch_stride = c.write.ch_stride;
    const dim3 grid(row_tiles, c.read.batch);
    const size_t shmem = (size_t)4 * 2 * RPW * kLdsSlot;
    if (c.read.table) {
        KernArgs<0> a;
        a.c = c;
        a.planes[0] = PlaneParams{};
        hipLaunchKernelGGL((k1_lds<RPW, STORE, 0>), grid, dim3(256), shmem, s, a, g);
    } else {
        KernArgs<CVGS_KERNARG_PLANES> a;
        a.c = c;
        for (int i = 0; i < CVGS_KERNARG_PLANES; ++i) a.planes[i] = i < ni ? ip[i] : PlaneParams{};
        hipLaunchKernelGGL((k1_lds<RPW, STORE, CVGS_KERNARG_PLANES>), grid, dim3(256), shmem, s, a, g);
    }
    return hipGetLastError();
}

const char* k1_exp_name(int v) {
    static const char* names[] = {"", "t256_r1", "t256_r2", "t256_r4", "t128_r1", "t64_r1", "t64_r2", "t64_r4",
                                  "t256_r1_nt", "t256_r1_sc1", "t256_r2_nt", "t256_r1_noremap", "t256_r1_NOLOAD",
                                  "t256_r1_NOSTORE", "t256_r1_EMPTY", "t128_r2", "t128_r4", "t256_r4_nt",
                                  "t64_r1_nt", "t128_r1_nt", "t256_r8", "t256_r8_nt",
                                  "v2_t256_r1", "v2_t256_r1_nt", "v2_t256_r2_nt", "v2_t256_r4_nt", "v2_t64_r1_nt", "v2_t256_r4",
                                  "v2_t128_r1_nt", "v2_t256_r8_nt", "v2_t512_r1_nt", "v2_t1024_r1_nt", "t64_r1_EMPTY", "t1024_r1_EMPTY",
                                  "v2_t512_r2_nt", "v2_t1024_r2_nt", "v2_t256_r1_nt_NODIV", "v2_t256_r4_nt_NODIV", "v2_t256_r1_nt_NOARITH",
                                  "v2_t256_r4_nt_NOARITH", "t256_r4_EMPTY", "t256_r4_COMPUTE", "t256_r4_NOLOAD", "t256_r4_NOSTORE", "lds_r1_nt", "lds_r2_nt", "lds_r4_nt"};
    return (v > 0 && v < (int)(sizeof(names) / sizeof(names[0]))) ? names[v] : nullptr;
}

int launch_k1_exp(int variant, const ChainArgs& c, const PlaneParams* ip, int ni, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    switch (variant) {
    case 1: e = xlaunch<256, 1, ST_PLAIN, true, MODE_FULL>(c, ip, ni, s); break;
    case 2: e = xlaunch<256, 2, ST_PLAIN, true, MODE_FULL>(c, ip, ni, s); break;
    case 3: e = xlaunch<256, 4, ST_PLAIN, true, MODE_FULL>(c, ip, ni, s); break;
    case 4: e = xlaunch<128, 1, ST_PLAIN, true, MODE_FULL>(c, ip, ni, s); break;
    case 5: e = xlaunch<64, 1, ST_PLAIN, true, MODE_FULL>(c, ip, ni, s); break;
    case 6: e = xlaunch<64, 2, ST_PLAIN, true, MODE_FULL>(c, ip, ni, s); break;
    case 7: e = xlaunch<64, 4, ST_PLAIN, true, MODE_FULL>(c, ip, ni, s); break;
    case 8: e = xlaunch<256, 1, ST_NT, true, MODE_FULL>(c, ip, ni, s); break;
    case 9: e = xlaunch<256, 1, ST_SC1, true, MODE_FULL>(c, ip, ni, s); break;
    case 10: e = xlaunch<256, 2, ST_NT, true, MODE_FULL>(c, ip, ni, s); break;
    case 11: e = xlaunch<256, 1, ST_PLAIN, false, MODE_FULL>(c, ip, ni, s); break;
    case 12: e = xlaunch<256, 1, ST_PLAIN, true, MODE_NOLOAD>(c, ip, ni, s); break;
    case 13: e = xlaunch<256, 1, ST_PLAIN, true, MODE_NOSTORE>(c, ip, ni, s); break;
    case 14: e = xlaunch<256, 1, ST_PLAIN, true, MODE_EMPTY>(c, ip, ni, s); break;
    case 15: e = xlaunch<128, 2, ST_PLAIN, true, MODE_FULL>(c, ip, ni, s); break;
    case 16: e = xlaunch<128, 4, ST_PLAIN, true, MODE_FULL>(c, ip, ni, s); break;
    case 17: e = xlaunch<256, 4, ST_NT, true, MODE_FULL>(c, ip, ni, s); break;
    case 18: e = xlaunch<64, 1, ST_NT, true, MODE_FULL>(c, ip, ni, s); break;
    case 19: e = xlaunch<128, 1, ST_NT, true, MODE_FULL>(c, ip, ni, s); break;
    case 20: e = xlaunch<256, 8, ST_PLAIN, true, MODE_FULL>(c, ip, ni, s); break;
    case 21: e = xlaunch<256, 8, ST_NT, true, MODE_FULL>(c, ip, ni, s); break;
    case 22: e = v2launch<256, 1, ST_PLAIN>(c, ip, ni, s); break;
    case 23: e = v2launch<256, 1, ST_NT>(c, ip, ni, s); break;
    case 24: e = v2launch<256, 2, ST_NT>(c, ip, ni, s); break;
    case 25: e = v2launch<256, 4, ST_NT>(c, ip, ni, s); break;
    case 26: e = v2launch<64, 1, ST_NT>(c, ip, ni, s); break;
    case 27: e = v2launch<256, 4, ST_PLAIN>(c, ip, ni, s); break;
    case 28: e = v2launch<128, 1, ST_NT>(c, ip, ni, s); break;
    case 29: e = v2launch<256, 8, ST_NT>(c, ip, ni, s); break;
    case 30: e = v2launch<512, 1, ST_NT>(c, ip, ni, s); break;
    case 31: e = v2launch<1024, 1, ST_NT>(c, ip, ni, s); break;
    case 32: e = xlaunch<64, 1, ST_PLAIN, true, MODE_EMPTY>(c, ip, ni, s); break;
    case 33: e = xlaunch<1024, 1, ST_PLAIN, true, MODE_EMPTY>(c, ip, ni, s); break;
    case 34: e = v2launch<512, 2, ST_NT>(c, ip, ni, s); break;
    case 35: e = v2launch<1024, 2, ST_NT>(c, ip, ni, s); break;
    case 36: e = v2launch<256, 1, ST_NT, ProgRMSM>(c, ip, ni, s); break;
    case 37: e = v2launch<256, 4, ST_NT, ProgRMSM>(c, ip, ni, s); break;
    case 38: e = v2launch<256, 1, ST_NT, ProgR>(c, ip, ni, s); break;
    case 39: e = v2launch<256, 4, ST_NT, ProgR>(c, ip, ni, s); break;
    case 40: e = xlaunch<256, 4, ST_PLAIN, true, MODE_EMPTY>(c, ip, ni, s); break;
    case 41: e = xlaunch<256, 4, ST_PLAIN, true, MODE_COMPUTE>(c, ip, ni, s); break;
    case 42: e = xlaunch<256, 4, ST_PLAIN, true, MODE_NOLOAD>(c, ip, ni, s); break;
    case 43: e = xlaunch<256, 4, ST_PLAIN, true, MODE_NOSTORE>(c, ip, ni, s); break;
    case 44: e = ldslaunch<1, ST_NT>(c, ip, ni, s); break;
    case 45: e = ldslaunch<2, ST_NT>(c, ip, ni, s); break;
    case 46: e = ldslaunch<4, ST_NT>(c, ip, ni, s); break;
    default: return -1;
    }
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

} // namespace cvgs
