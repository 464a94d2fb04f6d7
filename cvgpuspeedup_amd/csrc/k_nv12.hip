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
#include <cstdlib>
#include <memory>
#include <type_traits>

#include "k_taps.hpp"

namespace cvgs {

typedef uint16_t u16_unaligned __attribute__((aligned(1)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) u16_unaligned* gptr_u16;
typedef const __attribute__((address_space(1))) u32_unaligned* gptr_u32;
typedef uint64_t u64_unaligned __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) u64_unaligned* gptr_u64;

struct N12Geom {
    int32_t dst_w, dst_h, out_w, cn; // cn: 3, or 4 with alpha
    int64_t img_stride, ch_stride;
    uint8_t* out;
    int32_t out_step; // packed 2D writes: bytes per row
    int32_t packed;   // 0: planar fp32 tensor, 1: packed pixels through the generic write stage
    // optional second planar target with its own strides (CircularTensor push: history ring + ordered tensor)
    uint8_t* out2;
    int64_t img_stride2, ch_stride2;
    uint32_t col_tiles; // NPL == 0 (fused chains): blockIdx.x = row group * col_tiles + column tile
    uint32_t pad;
    // fused launches with host descriptors (NPL == 0; cvgs_api.cpp: ManyPool): the first work-item stores done_value into *done_word
    // (pinned host memory) when the kernel starts -- every earlier launch of the stream has finished by then (as K1: k_k1_impl.hpp)
    uint64_t* done_word;
    uint64_t done_value;
};

// NPL > 0: the planes travel in the kernel arguments, grid = (column tiles, row groups, planes).  NPL == 0: the chains of a
// cvgs_execute_many launch, planes in per-chain device tables, grid = (column tiles x row groups, planes, chains).
// NPL < 0: the segments with the planes of ALL chains inside the kernel arguments (KernArgsManyInline<-NPL>: fused chains described on the host, as K1).
template <int NPL> using K4Args = std::conditional_t<NPL == 0, KernArgsMany, std::conditional_t<(NPL < 0), KernArgsManyInline<(NPL < 0 ? -NPL : 1)>, KernArgs<(NPL > 0 ? NPL : 1)>>>;

// waves per workgroup (see k_k1.hip): an A/B build may override it
#ifndef CVGS_K4_WPB
#define CVGS_K4_WPB 4
#endif
constexpr int kK4Waves = CVGS_K4_WPB;
constexpr int kK4TileRow = 80; // floats between the rows of a wave's LDS tile (64 + padding: the 16-byte reads of a row group do not collide)

using N12SwapMulSubDiv = ProgSwapMulSubDiv; // the compile-time program of k_taps.hpp (incl. the division by the uniform divisor)

// RPW output rows per wave (the launcher uses 1, see launch_n12); CN output channels (3, or 4 with alpha).  One tap's conversion:
// k4_tap (k_common.hpp), shared with the descriptor queue's NV12 worker (k_queue.hip).
// S16: P010 -- the same geometry with 16-bit samples (10-bit code = sample >> 6): the two luma taps are ONE 4-byte load, the
// two chroma pairs ONE 8-byte load.
// WIN: the target may hold an aspect-ratio window (letterboxed detector inputs: PRESERVE_AR*) and default-value planes
// (usedPlanes < BATCH) -- K1's machinery: the background value runs through the program once, pixels outside the window take it.
// A separate instantiation: K4 is bound by its VALU work per row, and the window's selects cost the stretch-only launches 4-8 %
// when they are compiled in (tools/k4_ar_ab.sh: cfg #3 8.6 -> 9.0 us, 50 crops 4.92 -> 5.32 us).
// PL: planar chroma (I420 / YV12, software decoders' yuv420p): two (W/2) x (H/2) planes with rows of step/2 bytes behind the luma
// plane.  The two chroma taps of a plane are ONE unaligned 2-byte load per source row (6 loads per pixel instead of 4); the
// (U,V) pairs are then assembled in registers and everything downstream is the NV12 arithmetic, bit for bit.
template <int NPL, class Prog, typename OT = float, int RPW = 1, int CN = 3, bool S16 = false, bool WIN = false, bool PL = false>
__global__ __launch_bounds__(64 * kK4Waves) void k4_nv12_resize(const K4Args<NPL> a, const N12Geom g) {
    const ChainArgs& c = a.c;
    const int dst_w = g.dst_w, dst_h = g.dst_h, W = g.out_w;
    PlaneParams P;
    int z, col_tile, row_group, used;
    uint8_t* out_base;
    if constexpr (NPL <= 0) {
        z = (int)blockIdx.y;
        const ManySeg sg = a.seg[blockIdx.z];
        if (z >= sg.batch) return; // a shorter chain of the fused launch
        used = sg.used;
        if constexpr (NPL == 0) P = sg.table[z < used ? z : 0];
        else P = a.planes[(uint32_t)(uintptr_t)sg.table + (uint32_t)(z < used ? z : 0)]; // (sg.table: the chain's first index into a.planes)
        out_base = sg.out;
        col_tile = 0;
        row_group = (int)blockIdx.x;
        if (g.col_tiles > 1) { // the quotient comes out of the VALU: hand it back to the scalar side explicitly
            col_tile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % g.col_tiles));
            row_group = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / g.col_tiles));
        }
    } else {
        z = (int)blockIdx.z;
        used = c.read.used;
        P = a.planes[z];
        out_base = g.out;
        col_tile = (int)blockIdx.x;
        row_group = (int)blockIdx.y;
    }
    const int yuv_range = c.read.yuv_range, yuv_prim = c.read.yuv_primaries, packed = g.packed;
    const bool vu = c.read.yuv_layout == CVGS_YUV_NV21 || c.read.yuv_layout == CVGS_YUV_YV12; // wave-uniform: V comes first
    const int64_t img_stride = g.img_stride, ch_stride = g.ch_stride;
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    const f32x4s op0 = *(const f32x4s*)c.prog.operand[0], op1 = *(const f32x4s*)c.prog.operand[1],
                 op2 = *(const f32x4s*)c.prog.operand[2], op3 = *(const f32x4s*)c.prog.operand[3];
    // one batch of scalar loads, one wait (see k_k1.hip)
    if constexpr (WIN) asm volatile("" ::"s"(used), "s"(P.x1), "s"(P.y1), "s"(P.x2), "s"(P.y2));
    asm volatile("" ::"s"(dst_w), "s"(dst_h), "s"(W), "s"(CN), "s"(P.w), "s"(P.h), "s"(P.step), "s"(P.fx), "s"(P.fy), "s"(P.data), "s"(P.uv_off),
                 "s"(yuv_range), "s"(yuv_prim), "s"(packed), "s"(img_stride), "s"(ch_stride), "s"(out_base), "s"(op0), "s"(op1),
                 "s"(op2), "s"(op3));
    // (behind the scalar loads: a store in front of them would make the compiler fetch the descriptors with vector loads)
    if constexpr (NPL == 0) {
        if (g.done_word && (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && threadIdx.x == 0)
            __hip_atomic_store(g.done_word, g.done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const YuvK yk = yuv_matrix(yuv_range, yuv_prim, S16 ? CVGS_YUV_P010 : CVGS_YUV_NV12);

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int x = col_tile * 64 + lane;
    const int row0 = (row_group * kK4Waves + wave) * RPW;
    if (row0 >= dst_h || x >= dst_w) return;

    // one output pixel of row y (wave-uniform row pointers; planar: non-temporal rows, packed: the generic write stage)
    auto store_px = [&](const Px& p, int depth, int cn, int y) {
    if constexpr (std::is_same_v<OT, uint8_t>) {
        // packed u8 images (thumbnails, display surfaces): the chain's trailing SaturateCast is the store's conversion.  C3: a full
        // 64-column tile leaves as 48 dword stores (k_taps.hpp: store_u8c3_tile), ragged tiles as 3 bytes per lane; C4 (the
        // reference's own chain, tests/resize/test_fused_resize.cu:141-147: uchar4): one dword per lane, 256 bytes per wave
        const WriteArgs& w = c.write;
        auto put = [&](uint8_t* row) {
            if constexpr (CN == 4) {
                typedef uint32_t u32a1 __attribute__((aligned(1)));
                const uint32_t q = sat_u8_insert(p.v[3], 3, sat_u8_insert(p.v[2], 2, sat_u8_insert(p.v[1], 1, sat_u8_insert(p.v[0], 0, 0))));
                __builtin_nontemporal_store(q, (u32a1*)(row + (size_t)x * 4));
            } else {
                if (col_tile * 64 + 63 < dst_w) store_u8c3_tile(row + (size_t)(col_tile * 64) * 3, lane, p.v); // wave-uniform: every lane is alive
                else store_packed_px<3, uint8_t>(row + (size_t)x * 3, p.v, 3);
            }
        };
        put(w.kind == CVGS_WRITE_PIXEL_2D ? w.data + (size_t)y * (size_t)w.step : w.data + ((size_t)z * w.img_stride + (size_t)y * (size_t)W) * CN);
        if (w.kind == CVGS_WRITE_PIXEL_3D && w.data2) // wave-uniform (a second target with its own image stride)
            put(w.data2 + ((size_t)z * w.img_stride2 + (size_t)y * (size_t)W) * CN);
    } else if (packed) {
        const WriteArgs& w = c.write;
        if (depth == CVGS_DEPTH_32F && cn == CN) { // wave-uniform: packed float pixels leave as ONE dwordx3 / x4 store per lane
            float* const px = w.kind == CVGS_WRITE_PIXEL_2D ? (float*)(w.data + (size_t)y * (size_t)w.step) + (size_t)x * CN
                                                            : (float*)w.data + ((size_t)z * w.img_stride + (size_t)y * (size_t)W + x) * CN;
            store_packed_px<CN, float>(px, p.v, CN);
            if (w.kind == CVGS_WRITE_PIXEL_3D && w.data2)
                store_packed_px<CN, float>((float*)w.data2 + ((size_t)z * w.img_stride2 + (size_t)y * (size_t)W + x) * CN, p.v, CN);
        } else {
            write_px(c.write, c.dst_inline, x, y, z, p, depth, cn);
        }
    } else {
        // OT = _Float16: the chain's trailing CAST(CV_16F) is this round-to-nearest-even conversion
        const uint32_t xb = (uint32_t)x * (uint32_t)sizeof(OT);
        OT* const orow = (OT*)out_base + (int64_t)z * img_stride + (int64_t)y * W;
#pragma unroll
#ifdef CVGS_K4_PLAIN_PTR
        for (int k = 0; k < 4; ++k)
            if (k < cn) __builtin_nontemporal_store((OT)p.v[k], orow + (int64_t)k * ch_stride + x);
        if (g.out2) {
            OT* const orow2 = (OT*)g.out2 + (int64_t)z * g.img_stride2 + (int64_t)y * W;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < cn) __builtin_nontemporal_store((OT)p.v[k], orow2 + (int64_t)k * g.ch_stride2 + x);
        }
#else
        for (int k = 0; k < 4; ++k)
            if (k < cn) st_row(orow + (int64_t)k * ch_stride, xb, p.v[k]);
        if (g.out2) { // wave-uniform
            OT* const orow2 = (OT*)g.out2 + (int64_t)z * g.img_stride2 + (int64_t)y * W;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < cn) st_row(orow2 + (int64_t)k * g.ch_stride2, xb, p.v[k]);
        }
#endif
    }
    };
    // does the source cover the whole target?  (always, except aspect-ratio padding -- letterboxed detector inputs -- and planes
    // >= usedPlanes; wave-uniform.)  Otherwise the background value runs through the program once and replaces the pixels outside.
    const bool whole = !WIN || (z < used && ((P.x1 | P.y1 | (P.x2 ^ (dst_w - 1)) | (P.y2 ^ (dst_h - 1))) == 0));
    Px bgp;
    bgp.v[0] = bgp.v[1] = bgp.v[2] = bgp.v[3] = 0.f;
    bool in_x = true;
    int xr = x;
    if constexpr (WIN) {
        int bg_cn = CN, bg_depth = CVGS_DEPTH_32F;
        if (!whole) {
#pragma unroll
            for (int k = 0; k < 4; ++k) bgp.v[k] = k < CN ? c.read.bg[k] : 0.f;
            Prog::run(c.prog, bgp, bg_depth, bg_cn);
        }
        if (z >= used) { // a default-value plane: nothing is read
#pragma unroll
            for (int j = 0; j < RPW; ++j)
                if (row0 + j < dst_h) store_px(bgp, bg_depth, bg_cn, row0 + j);
            return;
        }
        in_x = x >= P.x1 && x <= P.x2;
        xr = in_x ? x - P.x1 : 0;
    }

    // column geometry (once per lane, reused for every row)
    const float sx = (float)xr * P.fx;
    const int x1 = (int)floorf(sx);
    const int x2 = x1 + 1;
    const float wxa = (float)x2 - sx, wxb = sx - (float)x1;
    const bool edge = x2 > P.w - 1;
    const int x2r = edge ? x1 : x2;
    constexpr int kSB = S16 ? 2 : 1; // bytes per sample
    const uint32_t yo = (uint32_t)min(x1, P.w - 2) * kSB;
    const int ysh = (x1 * kSB - (int)yo) * 8;
    const int c1 = x1 >> 1, c2 = x2r >> 1;
    // interleaved: a window of two pairs, clamped into the row; planar: a window of two samples of one chroma plane
    const uint32_t uo = PL ? (uint32_t)min(c1, ((P.w - 1) >> 1) - 1) : (uint32_t)min(2 * c1, P.w - 4) * kSB;
    const int ush = PL ? (c1 - (int)uo) * 8 : (2 * c1 * kSB - (int)uo) * 8;
    const bool same_pair = c2 == c1;
    // 8-bit interleaved chroma: v_perm_b32 selectors that pick the pixel's four tap samples {a0, a1, b0, b1} (row a / b, tap 0 / 1) out of its two
    // 2-byte luma windows / its two 4-byte chroma windows -- what was a shift, a mask and a select per sample: the luma window clamped back at
    // the right edge (then both taps are its second byte), the chroma window clamped back at the last pair, taps sharing a chroma pair, NV21's
    // byte order (k_nv12_x2.hip and the queue's k4q_rows pick their samples the same way)
    [[maybe_unused]] uint32_t sel_y = 0, sel_u = 0, sel_v = 0;
    if constexpr (!S16 && !PL) {
        sel_y = edge ? 0x05050101u : 0x05040100u;
        const uint32_t pr0 = 2 * c1 != (int)uo ? 2u : 0u; // byte of tap 0's pair inside the chroma window
        const uint32_t pr1 = same_pair ? pr0 : 2u;        // ... of tap 1's
        const uint32_t sel_c = pr0 | (pr1 << 8) | ((4u + pr0) << 16) | ((4u + pr1) << 24);
        const bool vu_first = c.read.yuv_layout == CVGS_YUV_NV21;
        sel_u = sel_c + (vu_first ? 0x01010101u : 0u);
        sel_v = sel_c + (vu_first ? 0u : 0x01010101u);
    }
    const gptr_u8 base = (gptr_u8)P.data;
    const size_t step = (size_t)P.step;
    const gptr_u8 uvp = base + (size_t)P.uv_off; // crops of a surface carry their own luma -> chroma offset
    const size_t cstep = PL ? step >> 1 : step;  // bytes per chroma row
    const size_t plane2 = (size_t)(P.h >> 1) * cstep; // planar: the second chroma plane follows the first

    using ChromaWin = std::conditional_t<S16, uint64_t, uint32_t>; // two (U,V) pairs
    uint32_t vya[RPW], vyb[RPW];
    ChromaWin vua[RPW], vub[RPW];
    uint32_t vva[RPW], vvb[RPW]; // planar chroma: the windows of the second plane
    float wya[RPW], wyb[RPW];
    bool in_y[RPW];
    // four rows per wave into a planar fp32 tensor: a full 64-column tile with all four rows inside the target leaves through the LDS transpose
    constexpr bool kRowsTile = RPW == 4 && std::is_same_v<OT, float> && !WIN;
    [[maybe_unused]] const bool tile_rows = kRowsTile && !packed && !g.out2 && col_tile * 64 + 63 < dst_w && row0 + RPW <= dst_h;
    [[maybe_unused]] float tv[RPW][4];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        // row geometry (wave-uniform)
        const int y = min(row0 + j, dst_h - 1);
        in_y[j] = !WIN || (y >= P.y1 && y <= P.y2);
        const int yr = WIN ? (in_y[j] ? y - P.y1 : 0) : y;
        const float sy = (float)yr * P.fy;
        const int y1 = (int)floorf(sy);
        const int y2 = y1 + 1;
        const int y2r = min(y2, P.h - 1);
        wya[j] = (float)y2 - sy;
        wyb[j] = sy - (float)y1;
        const int r1 = __builtin_amdgcn_readfirstlane(y1), r2 = __builtin_amdgcn_readfirstlane(y2r);
#ifdef CVGS_K4_PLAIN_PTR
#define K4_PIN(p) (p)
#else
#define K4_PIN(p) pin_uniform(p)
#endif
        const gptr_u8 ya = K4_PIN(base + (size_t)r1 * step);
        const gptr_u8 yb = K4_PIN(base + (size_t)r2 * step);
        const gptr_u8 ua = K4_PIN(uvp + (size_t)(r1 >> 1) * cstep);
        if constexpr (PL) {
            const gptr_u8 ub = K4_PIN(uvp + (size_t)(r2 >> 1) * cstep);
            vya[j] = *(gptr_u16)(ya + yo);
            vyb[j] = *(gptr_u16)(yb + yo);
            vua[j] = *(gptr_u16)(ua + uo);
            vub[j] = *(gptr_u16)(ub + uo);
            vva[j] = *(gptr_u16)(K4_PIN(ua + plane2) + uo);
            vvb[j] = *(gptr_u16)(K4_PIN(ub + plane2) + uo);
            continue;
        }
        if constexpr (S16) {
            vya[j] = *(gptr_u32)(ya + yo);
            vyb[j] = *(gptr_u32)(yb + yo);
            vua[j] = *(gptr_u64)(ua + uo);
        } else {
            vya[j] = *(gptr_u16)(ya + yo);
            vyb[j] = *(gptr_u16)(yb + yo);
            vua[j] = *(gptr_u32)(ua + uo);
        }
        // Every other row pair shares ONE chroma row; skipping its second load behind a wave-uniform branch was measured
        // and lost (tools/k4_ab.sh: cfg #3 9.15 vs 8.02 us, 50 NV12 crops 5.04 vs 4.52 us): the redundant load hits L1,
        // the branch delays the loads behind it.
#ifdef CVGS_K4_UVSKIP
        if ((r1 >> 1) == (r2 >> 1)) {
            vub[j] = vua[j];
        } else
#endif
        {
            const gptr_u8 ub = K4_PIN(uvp + (size_t)(r2 >> 1) * step);
            if constexpr (S16) vub[j] = *(gptr_u64)(ub + uo);
            else vub[j] = *(gptr_u32)(ub + uo);
        }
    }

#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = row0 + j;
        if (y >= dst_h) break; // wave-uniform
        float fy[4], fu[4], fv[4]; // taps 00, 10, 01, 11
        if constexpr (S16) {
            const uint32_t ya0 = (vya[j] >> ysh) & 0xffffu, ya1 = edge ? ya0 : vya[j] >> 16;
            const uint32_t yb0 = (vyb[j] >> ysh) & 0xffffu, yb1 = edge ? yb0 : vyb[j] >> 16;
            const uint32_t pa0 = (uint32_t)(vua[j] >> ush), pa1 = same_pair ? pa0 : (uint32_t)(vua[j] >> 32);
            const uint32_t pb0 = (uint32_t)(vub[j] >> ush), pb1 = same_pair ? pb0 : (uint32_t)(vub[j] >> 32);
            fy[0] = (float)(ya0 >> 6); fy[1] = (float)(ya1 >> 6); fy[2] = (float)(yb0 >> 6); fy[3] = (float)(yb1 >> 6);
            fu[0] = (float)((pa0 & 0xffffu) >> 6); fu[1] = (float)((pa1 & 0xffffu) >> 6);
            fu[2] = (float)((pb0 & 0xffffu) >> 6); fu[3] = (float)((pb1 & 0xffffu) >> 6);
            fv[0] = (float)(pa0 >> 22); fv[1] = (float)(pa1 >> 22); fv[2] = (float)(pb0 >> 22); fv[3] = (float)(pb1 >> 22);
        } else if constexpr (!PL) {
            const uint32_t ly = __builtin_amdgcn_perm(vyb[j], vya[j], sel_y);
            const uint32_t lu = __builtin_amdgcn_perm((uint32_t)vub[j], (uint32_t)vua[j], sel_u), lv = __builtin_amdgcn_perm((uint32_t)vub[j], (uint32_t)vua[j], sel_v);
            fy[0] = (float)(ly & 0xffu); fy[1] = (float)((ly >> 8) & 0xffu); fy[2] = (float)((ly >> 16) & 0xffu); fy[3] = (float)(ly >> 24);
            fu[0] = (float)(lu & 0xffu); fu[1] = (float)((lu >> 8) & 0xffu); fu[2] = (float)((lu >> 16) & 0xffu); fu[3] = (float)(lu >> 24);
            fv[0] = (float)(lv & 0xffu); fv[1] = (float)((lv >> 8) & 0xffu); fv[2] = (float)((lv >> 16) & 0xffu); fv[3] = (float)(lv >> 24);
        } else {
            const uint32_t ya0 = (vya[j] >> ysh) & 0xffu, ya1 = edge ? ya0 : (vya[j] >> 8) & 0xffu;
            const uint32_t yb0 = (vyb[j] >> ysh) & 0xffu, yb1 = edge ? yb0 : (vyb[j] >> 8) & 0xffu;
            uint32_t ca = vua[j], cb = vub[j];
            if constexpr (PL) { // spread the two samples of each plane into the (first, second) pairs of an interleaved window
                const uint32_t fa = (uint32_t)vua[j] >> ush, fb = (uint32_t)vub[j] >> ush, sa = vva[j] >> ush, sb = vvb[j] >> ush;
                ca = (fa & 0xffu) | ((sa & 0xffu) << 8) | ((fa & 0xff00u) << 8) | ((sa & 0xff00u) << 16);
                cb = (fb & 0xffu) | ((sb & 0xffu) << 8) | ((fb & 0xff00u) << 8) | ((sb & 0xff00u) << 16);
            }
            if (vu) { // NV21: swap the bytes of every pair once, then everything below is NV12
                ca = ((ca & 0x00ff00ffu) << 8) | ((ca >> 8) & 0x00ff00ffu);
                cb = ((cb & 0x00ff00ffu) << 8) | ((cb >> 8) & 0x00ff00ffu);
            }
            const int psh = PL ? 0 : ush; // planar: the windows were shifted before the spread
            const uint32_t pa0 = (ca >> psh) & 0xffffu, pa1 = same_pair ? pa0 : (ca >> 16) & 0xffffu;
            const uint32_t pb0 = (cb >> psh) & 0xffffu, pb1 = same_pair ? pb0 : (cb >> 16) & 0xffffu;
            fy[0] = (float)ya0; fy[1] = (float)ya1; fy[2] = (float)yb0; fy[3] = (float)yb1;
            fu[0] = (float)(pa0 & 0xffu); fu[1] = (float)(pa1 & 0xffu); fu[2] = (float)(pb0 & 0xffu); fu[3] = (float)(pb1 & 0xffu);
            fv[0] = (float)(pa0 >> 8); fv[1] = (float)(pa1 >> 8); fv[2] = (float)(pb0 >> 8); fv[3] = (float)(pb1 >> 8);
        }

        float t00[4], t10[4], t01[4], t11[4];
        if (yuv_range == CVGS_YUV_FULL) { // wave-uniform
            k4_tap<CN, true>(fy[0], fu[0], fv[0], yk, t00);
            k4_tap<CN, true>(fy[1], fu[1], fv[1], yk, t10);
            k4_tap<CN, true>(fy[2], fu[2], fv[2], yk, t01);
            k4_tap<CN, true>(fy[3], fu[3], fv[3], yk, t11);
        } else {
            k4_tap<CN, false>(fy[0], fu[0], fv[0], yk, t00);
            k4_tap<CN, false>(fy[1], fu[1], fv[1], yk, t10);
            k4_tap<CN, false>(fy[2], fu[2], fv[2], yk, t01);
            k4_tap<CN, false>(fy[3], fu[3], fv[3], yk, t11);
        }

        const float w00 = wxa * wya[j], w10 = wxb * wya[j], w01 = wxa * wyb[j], w11 = wxb * wyb[j];
        Px p;
        p.v[3] = 0.f;
#pragma unroll
        for (int k = 0; k < CN; ++k) {
            float acc = t00[k] * w00;
            acc = acc + t10[k] * w10;
            acc = acc + t01[k] * w01;
            acc = acc + t11[k] * w11;
            p.v[k] = acc;
        }
        int depth = CVGS_DEPTH_32F, cn = CN;
        Prog::run(c.prog, p, depth, cn);
        if constexpr (WIN) {
            if (!whole) { // wave-uniform: only padded planes pay the per-lane select
                const bool take = in_x && in_y[j];
#pragma unroll
                for (int k = 0; k < 4; ++k) p.v[k] = take ? p.v[k] : bgp.v[k];
            }
        }

        if constexpr (kRowsTile) {
            if (tile_rows) { // wave-uniform: the wave's four rows leave together below
#pragma unroll
                for (int k = 0; k < CN; ++k) tv[j][k] = p.v[k];
                continue;
            }
        }
        store_px(p, depth, cn, y);
    }
    if constexpr (kRowsTile) {
        if (tile_rows) {
            // the lane = column register layout transposed through a wave-private LDS tile: lane l then owns 4 consecutive columns of row
            // l / 16 -- 16 bytes per lane and store instruction, three stores for the wave's four rows instead of twelve (the descriptor
            // queue's row workers publish their rows this way, k_queue.hip: q_lds_put / q_lds_get; fused launches of 4:2:0 crops are bound by
            // the number of memory instructions: four tap loads per row and lane)
            __shared__ __attribute__((aligned(16))) float tiles[kK4Waves][CN * RPW * kK4TileRow];
            float* const tile = tiles[wave];
#pragma unroll
            for (int k = 0; k < CN; ++k)
#pragma unroll
                for (int j = 0; j < RPW; ++j) tile[(k * RPW + j) * kK4TileRow + lane] = tv[j][k];
            __builtin_amdgcn_wave_barrier(); // (compiler ordering only: one wave's LDS operations run in order)
            typedef float f32x4t __attribute__((ext_vector_type(4)));
            typedef f32x4t f32x4t_a4 __attribute__((aligned(4)));
            typedef __attribute__((address_space(1))) f32x4t_a4* gf4;
            const int i = lane >> 4, q = lane & 15;
            float* const orow = (float*)out_base + (int64_t)z * img_stride + (int64_t)(row0 + i) * W + col_tile * 64 + q * 4;
#pragma unroll
            for (int k = 0; k < CN; ++k) {
                const f32x4t o = *(const f32x4t*)(tile + (k * RPW + i) * kK4TileRow + q * 4);
                __builtin_nontemporal_store(o, (gf4)(orow + (int64_t)k * ch_stride));
            }
        }
    }
}

// what launch_nv12 hands to the instantiation it picks: the call's LaunchCtx and the chains of a cvgs_execute_many launch
struct N12Many {
    LaunchCtx* ctx;
    const ManySeg* segs;
    int n_segs;
    const PlaneParams* planes; // host-described fused chains whose planes travel in the kernel arguments (segs[i].table = first index), or null
    int n_planes;
};

template <class Prog, typename OT, int RPW, int CN, bool S16, bool WIN = false, bool PL = false>
static hipError_t launch_n12_r(const ChainArgs& c, const PlaneParams* ip, int ni, const N12Geom& g_in, const N12Many& many) {
    hipStream_t s = (hipStream_t)many.ctx->stream;
    N12Geom g = g_in;
    const uint32_t col_tiles = (uint32_t)((g.dst_w + 63) / 64), row_groups = (uint32_t)((g.dst_h + kK4Waves * RPW - 1) / (kK4Waves * RPW));
    g.col_tiles = col_tiles;
    g.pad = 0;
    g.done_word = nullptr;
    g.done_value = 0;
    constexpr bool kImage = std::is_same_v<OT, uint8_t>; // packed u8 images: never fused chains, never the 16 KB argument block
    if constexpr (!kImage && !PL && !WIN) if (many.segs && many.planes) {
        // host descriptors of at most kManyInlineLarge planes: segments + planes in the arguments (16 KB / 52 KB blocks), capturable
        const dim3 grid(col_tiles * row_groups, (unsigned)c.read.batch, (unsigned)many.n_segs);
        auto go = [&](auto cap_tag) {
            constexpr int CAP = decltype(cap_tag)::value;
            // the 16 KB / 52 KB argument block: staged in a per-thread heap buffer, handed over by address (as K1's: k_k1_impl.hpp launch_t)
            static thread_local std::unique_ptr<KernArgsManyInline<CAP>> staged;
            if (!staged) staged.reset(new KernArgsManyInline<CAP>());
            KernArgsManyInline<CAP>& a = *staged;
            a.c = c;
            for (int i = 0; i < CVGS_MAX_CHAINS; ++i) a.seg[i] = i < many.n_segs ? many.segs[i] : ManySeg{nullptr, nullptr, 0, 0};
            for (int i = 0; i < many.n_planes && i < CAP; ++i) a.planes[i] = many.planes[i];
            void* args[] = {(void*)&a, (void*)&g};
            (void)hipLaunchKernel((const void*)&k4_nv12_resize<-CAP, Prog, OT, RPW, CN, S16, WIN, PL>, grid, dim3(64 * kK4Waves), args, 0, s);
        };
        if (many.n_planes <= kManyInlineSmall) go(std::integral_constant<int, kManyInlineSmall>{});
        else go(std::integral_constant<int, kManyInlineLarge>{});
        return hipGetLastError();
    }
    if constexpr (!kImage) if (many.segs) {
        KernArgsMany a;
        a.c = c;
        for (int i = 0; i < CVGS_MAX_CHAINS; ++i) a.seg[i] = i < many.n_segs ? many.segs[i] : ManySeg{nullptr, nullptr, 0, 0};
        LaunchCtx& x = *many.ctx;
        if (x.done_word && !x.done_word_taken) {
            x.done_word_taken = true;
            g.done_word = x.done_word;
            g.done_value = x.done_value;
        }
        const dim3 grid(col_tiles * row_groups, (unsigned)c.read.batch, (unsigned)many.n_segs);
        hipLaunchKernelGGL((k4_nv12_resize<0, Prog, OT, RPW, CN, S16, WIN, PL>), grid, dim3(64 * kK4Waves), 0, s, a, g);
        return hipGetLastError();
    }
    const dim3 grid(col_tiles, row_groups, c.read.batch);
    if (ni <= 8) {
        KernArgs<8> a;
        a.c = c;
        for (int i = 0; i < 8; ++i) a.planes[i] = i < ni ? ip[i] : PlaneParams{};
        hipLaunchKernelGGL((k4_nv12_resize<8, Prog, OT, RPW, CN, S16, WIN, PL>), grid, dim3(64 * kK4Waves), 0, s, a, g);
    } else if (ni <= CVGS_KERNARG_PLANES) { // crop lists of a decoder surface: up to CVGS_KERNARG_PLANES descriptors in the kernel arguments
        KernArgs<CVGS_KERNARG_PLANES> a;
        a.c = c;
        for (int i = 0; i < CVGS_KERNARG_PLANES; ++i) a.planes[i] = i < ni ? ip[i] : PlaneParams{};
        hipLaunchKernelGGL((k4_nv12_resize<CVGS_KERNARG_PLANES, Prog, OT, RPW, CN, S16, WIN, PL>), grid, dim3(64 * kK4Waves), 0, s, a, g);
    } else if constexpr (!kImage) { // ... up to CVGS_KERNARG_PLANES_MAX in a 16 KB argument block (see cvgs_device.h: cheaper than a table for an eager call)
        KernArgs<kKernargPlanesBig> a;
        a.c = c;
        for (int i = 0; i < kKernargPlanesBig; ++i) a.planes[i] = i < ni ? ip[i] : PlaneParams{};
        hipLaunchKernelGGL((k4_nv12_resize<kKernargPlanesBig, Prog, OT, RPW, CN, S16, WIN, PL>), grid, dim3(64 * kK4Waves), 0, s, a, g);
    }
    return hipGetLastError();
}

// the channel count (3, or 4 with alpha) becomes a template argument
template <class Prog, typename OT, bool S16, bool WIN, bool PL>
static hipError_t launch_n12_cn(const ChainArgs& c, const PlaneParams* ip, int ni, const N12Geom& g, const N12Many& s) {
    if (g.cn == 4) return launch_n12_r<Prog, OT, 1, 4, S16, WIN, PL>(c, ip, ni, g, s);
    return launch_n12_r<Prog, OT, 1, 3, S16, WIN, PL>(c, ip, ni, g, s);
}

template <class Prog, typename OT = float>
static hipError_t launch_n12(const ChainArgs& c, const PlaneParams* ip, int ni, const N12Geom& g, const N12Many& s, bool win = false) {
    const bool pl = c.read.yuv_layout == CVGS_YUV_I420 || c.read.yuv_layout == CVGS_YUV_YV12;
    const bool s16 = c.read.yuv_layout == CVGS_YUV_P010;
    if (win) { // aspect-ratio windows / default-value planes: their own instantiations (see k4_nv12_resize)
        if (pl) return launch_n12_cn<Prog, OT, false, true, true>(c, ip, ni, g, s);
        if (s16) return launch_n12_cn<Prog, OT, true, true, false>(c, ip, ni, g, s);
        return launch_n12_cn<Prog, OT, false, true, false>(c, ip, ni, g, s);
    }
    // One output row per wave.  Two rows per wave were measured for whole-frame outputs (cfg #3: 14400 one-row waves need two
    // rounds of the chip's 8192 wave slots) and lost: 8.27 vs 8.08 us, and 5.29 vs 4.52 us on 50 crops -- the launch is
    // bound by the VALU work per row (~100 instructions x 14 waves per SIMD) plus the launch floor, not by residency.
    // launches in the throughput regime (cvgs_execute_many: the crops of several surfaces; one chain of hundreds of crops): four rows per wave, the rows leaving as
    // 16-byte stores through a wave-private LDS tile -- the launch is bound by its memory INSTRUCTIONS (four tap loads per row and lane):
    // 16 x 50 crops of NV12 surfaces 52 -> see profiles/r05_x_k4_tick_rows4.txt
    if constexpr (std::is_same_v<OT, float>) {
        const N12Many& many = s;
        if (g.cn == 3 && !pl) { // (a single chain of 256+ crops is in the same regime; P010 surfaces too)
            int64_t planes = many.segs ? 0 : c.read.batch;
            for (int i = 0; many.segs && i < many.n_segs; ++i) planes += many.segs[i].batch;
            if (planes * g.dst_h * ((g.dst_w + 63) / 64) >= 32768)
                return s16 ? launch_n12_r<Prog, OT, 4, 3, true>(c, ip, ni, g, s) : launch_n12_r<Prog, OT, 4, 3, false>(c, ip, ni, g, s);
        }
    }
    if (pl) return launch_n12_cn<Prog, OT, false, false, true>(c, ip, ni, g, s);
    if (s16) return launch_n12_cn<Prog, OT, true, false, false>(c, ip, ni, g, s);
    return launch_n12_cn<Prog, OT, false, false, false>(c, ip, ni, g, s);
}

// Returns 1 if it took the chain, 0 if not eligible, <0 on error.
// Can K4 serve these planes?  Rows wide enough for the 4-byte chroma window; stretch geometry for the callers that cannot pick
// the windowed instantiation (fused chains, staged tables).
bool k4_planes_eligible(const PlaneParams* planes, int n, int dst_w, int dst_h) { // stretch geometry: the WIN = false instantiations
    for (int i = 0; i < n; ++i) {
        const PlaneParams& P = planes[i];
        if (P.w < 4 || P.x1 != 0 || P.y1 != 0 || P.x2 != dst_w - 1 || P.y2 != dst_h - 1) return false;
    }
    return true;
}
static bool k4_planes_wide_enough(const PlaneParams* planes, int n) {
    for (int i = 0; i < n; ++i)
        if (planes[i].w < 4) return false;
    return true;
}

// ctx.segs (n_segs >= 1): the chains of a cvgs_execute_many launch -- their planes live in device tables that the caller
// has checked with k4_planes_eligible; c_in.read.batch is the largest batch.  nullptr: one chain (inline_planes).
int launch_nv12(const ChainArgs& c_in, const PlaneParams* inline_planes, int n_inline, int min_width, LaunchCtx& ctx, bool dry_run, LaunchInfo* info,
                uint32_t chain_flags) {
    const ManySeg* const segs = ctx.segs;
    const int n_segs = ctx.n_segs;
    void* const stream = ctx.stream;
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
    if (segs) {
        if (n_segs < 1 || n_segs > CVGS_MAX_CHAINS || c_in.write.data2) return 0;
        // segments without a table: the planes travel in the kernel arguments -- interleaved chroma only (launch_n12_r)
        if (!r.table && (!inline_planes || n_inline < 1 || n_inline > kManyInlineLarge || (r.yuv_layout != CVGS_YUV_NV12 && r.yuv_layout != CVGS_YUV_NV21 && r.yuv_layout != CVGS_YUV_P010))) return 0;
    } else {
        if (r.table || n_inline > kKernargPlanesBig || min_width < 4) return 0; // tiny frames / resident tables: generic kernel
        if (n_inline > CVGS_KERNARG_PLANES && !(planar_kind && (c_in.write.depth == CVGS_DEPTH_32F || f16))) return 0; // the large block: tensors only
        if (!k4_planes_wide_enough(inline_planes, r.used < n_inline ? r.used : n_inline)) return 0;
    }
    if (r.batch > 65535) return 0;
    const WriteArgs& w = c.write;
    const bool planar = planar_kind && (w.depth == CVGS_DEPTH_32F || f16);
    const bool packed = w.kind == CVGS_WRITE_PIXEL_2D || w.kind == CVGS_WRITE_PIXEL_3D;
    if (!planar && !packed) return 0;
    if (segs && !planar) return 0; // fused chains: planar tensors only

    // packed u8 C3 images (decoder surface -> thumbnail / display image): resize -> [REORDER / MUL / ADD / SUB / DIV in float] ->
    // CAST(CV_8U) -> write.  The trailing SaturateCast becomes the store's conversion and the store a coalesced tile.
    bool u8img = false;
    int u8_prog = 2; // 0: nothing in front of the cast, 1: the R<->B swap only, 2: interpreted
    ChainArgs c8 = c;
    // the reference's spelling casts first and reorders the bytes afterwards (SaturateCast<float4, uchar4> -> VectorReorder<uchar4, 2, 1, 0, 3>);
    // a pure permutation commutes with the per-channel cast, so the cast moves to the end (bit for bit the same image)
    if (c8.prog.n >= 2 && c8.prog.opcode[c8.prog.n - 1] == CVGS_OP_REORDER && c8.prog.opcode[c8.prog.n - 2] == CVGS_OP_CAST &&
        c8.prog.aux[c8.prog.n - 2] == CVGS_DEPTH_8U) {
        const int a = c8.prog.n - 2, b = c8.prog.n - 1;
        std::swap(c8.prog.opcode[a], c8.prog.opcode[b]);
        std::swap(c8.prog.aux[a], c8.prog.aux[b]);
        for (int k = 0; k < 4; ++k) std::swap(c8.prog.operand[a][k], c8.prog.operand[b][k]);
    }
    if (packed && !f16 && !segs && w.depth == CVGS_DEPTH_8U && n_inline <= CVGS_KERNARG_PLANES && c8.prog.n >= 1 &&
        c8.prog.opcode[c8.prog.n - 1] == CVGS_OP_CAST && c8.prog.aux[c8.prog.n - 1] == CVGS_DEPTH_8U) {
        u8img = true;
        for (int k = 0; k + 1 < c8.prog.n; ++k) {
            const int op = c8.prog.opcode[k];
            const bool arith = op == CVGS_OP_MUL || op == CVGS_OP_ADD || op == CVGS_OP_SUB || op == CVGS_OP_DIV || op == CVGS_OP_REORDER || op == CVGS_OP_NOP;
            if (!arith && !(op == CVGS_OP_CAST && c8.prog.aux[k] == CVGS_DEPTH_32F)) u8img = false; // the value stays out_cn floats up to the cast
        }
        const int swap_rb = r.out_cn == 3 ? (2 | (1 << 2) | (0 << 4)) : (2 | (1 << 2) | (0 << 4) | (3 << 6));
        if (c8.prog.n == 1) u8_prog = 0;
        else if (c8.prog.n == 2 && c8.prog.opcode[0] == CVGS_OP_REORDER && c8.prog.aux[0] == swap_rb) u8_prog = 1;
    }
    if (u8img) {
        c8.prog.n -= 1;
        c8.prog.fast_div = 0;
        bool canon8 = false; // brightness / contrast on the way to a u8 image, ...: the canonical arithmetic program (k_taps.hpp)
        if (u8_prog == 2) {
            ProgArgs canon;
            if (k1_canonicalise(c8.prog, r.out_cn, canon)) {
                c8.prog = canon;
                canon8 = true;
            }
        }
        N12Geom g8{};
        g8.dst_w = r.dst_w; g8.dst_h = r.dst_h; g8.out_w = w.width; g8.cn = r.out_cn;
        g8.out = w.data; g8.out_step = w.step; g8.packed = 1;
        if (info)
            info->kernel = r.out_cn == 3 ? (u8_prog == 0 ? "k4_nv12_resize_u8c3" : (u8_prog == 1 ? "k4_nv12_resize_swap_u8c3" : (canon8 ? "k4_nv12_resize_arith_u8c3" : "k4_nv12_resize_interp_u8c3")))
                                         : (u8_prog == 0 ? "k4_nv12_resize_u8c4" : (u8_prog == 1 ? "k4_nv12_resize_swap_u8c4" : (canon8 ? "k4_nv12_resize_arith_u8c4" : "k4_nv12_resize_interp_u8c4")));
        if (dry_run) return 1;
        const N12Many s8{&ctx, nullptr, 0, nullptr, 0};
        const bool win8 = r.used != r.batch || !k4_planes_eligible(inline_planes, n_inline, r.dst_w, r.dst_h);
        const hipError_t e8 = u8_prog == 0   ? launch_n12<ProgNone, uint8_t>(c8, inline_planes, n_inline, g8, s8, win8)
                              : u8_prog == 1 ? launch_n12<K1Prog<kOpSwapRB>, uint8_t>(c8, inline_planes, n_inline, g8, s8, win8)
                              : canon8       ? launch_n12<K1CanonProg, uint8_t>(c8, inline_planes, n_inline, g8, s8, win8)
                                             : launch_n12<InterpProg, uint8_t>(c8, inline_planes, n_inline, g8, s8, win8);
        return e8 == hipSuccess ? 1 : -(int)e8 - 1000;
    }

    N12Geom g;
    g.dst_w = r.dst_w; g.dst_h = r.dst_h; g.out_w = w.width; g.cn = r.out_cn;
    g.img_stride = w.img_stride; g.ch_stride = w.ch_stride;
    g.out = w.data; g.out_step = w.step; g.packed = packed ? 1 : 0;
    g.out2 = w.data2; g.img_stride2 = w.img_stride2; g.ch_stride2 = w.ch_stride2;

    const int swap = r.out_cn == 3 ? (2 | (1 << 2) | (0 << 4)) : (2 | (1 << 2) | (0 << 4) | (3 << 6));
    const ProgArgs& p = c.prog;
    const bool fast_prog = planar && p.n == 4 && p.opcode[0] == CVGS_OP_REORDER && p.aux[0] == swap &&
                           p.opcode[1] == CVGS_OP_MUL && p.opcode[2] == CVGS_OP_SUB && p.opcode[3] == CVGS_OP_DIV;
    // the same normalisation in the surface's own R, G, B order (cvtColorNV12<COLOR_YUV2RGB_NV12>: no swap)
    const bool fast_rgb = planar && !f16 && p.n == 3 && p.opcode[0] == CVGS_OP_MUL && p.opcode[1] == CVGS_OP_SUB && p.opcode[2] == CVGS_OP_DIV;
    ChainArgs c_fd = c;
    c_fd.prog.fast_div = 0;
    for (int k = 0; k < 4; ++k) c_fd.prog.rdiv[k] = 0.f;
    if (fast_prog) fast_div_setup(c_fd.prog, 3, 1, r.out_cn, r.bg);
    else if (fast_rgb) fast_div_setup(c_fd.prog, 2, 0, r.out_cn, r.bg);
    // whole surfaces stretched into large targets (cfg #3: 6K -> 1280 x 720): two output pixels per lane (k_nv12_x2.hip) once the
    // launch is paced by instruction issue rather than by its latency; CVGS_CHAIN_NO_THREAD_FUSION keeps the one-pixel kernel
    if ((fast_prog || fast_rgb) && !f16 && !segs && r.out_cn == 3 && !(chain_flags & CVGS_CHAIN_NO_THREAD_FUSION)) {
        const char* x2_env = getenv("CVGS_K4_X2"); // tuning / test hook: 0 = never, 1 = whenever eligible
        const int64_t wave_rows = (int64_t)r.batch * r.dst_h * ((r.dst_w + 63) / 64);
        if (x2_env ? x2_env[0] == '1' : wave_rows >= kK4X2MinWaveRows) {
            const int rc = launch_nv12_x2(c_fd, inline_planes, n_inline, fast_prog, stream, dry_run);
            if (rc != 0) {
                if (info) info->kernel = fast_prog ? "k4_nv12_x2_swap_mul_sub_div" : "k4_nv12_x2_mul_sub_div";
                return rc;
            }
        }
    }
    // any other chain of the canonical arithmetic shape ([swap] {mul|add|sub} x 0..2 [div] {mul|add|sub} x 0..2): the straight-line K1CanonProg
    // (k_taps.hpp; round 6 -- a tick of 16 surfaces x 50 crops with one more `add` ran 58 us interpreted against 37 for the compile-time program)
    bool canon_prog = false;
    if (!fast_prog && !(fast_rgb && !f16)) { // (planar tensors and packed fp32 / fp16 pixels alike)
        ProgArgs canon;
        if (k1_canonicalise(c_fd.prog, r.out_cn, canon)) {
            c_fd.prog = canon;
            canon_prog = true;
        }
    }
    if (info)
        info->kernel = f16 ? (fast_prog ? "k4_nv12_resize_swap_mul_sub_div_f16" : (canon_prog ? "k4_nv12_resize_arith_f16" : "k4_nv12_resize_interp_f16"))
                           : (fast_prog ? "k4_nv12_resize_swap_mul_sub_div" : (fast_rgb ? "k4_nv12_resize_mul_sub_div" : (canon_prog ? "k4_nv12_resize_arith" : "k4_nv12_resize_interp")));
    if (dry_run) return 1;
    const N12Many s{&ctx, segs, n_segs, segs && !r.table ? inline_planes : nullptr, segs && !r.table ? n_inline : 0};
    // the windowed instantiations: an aspect-ratio window or default-value planes (never for fused chains / staged tables, whose
    // callers admit stretch geometry only)
    const bool win = !segs && (r.used != r.batch || !k4_planes_eligible(inline_planes, n_inline, r.dst_w, r.dst_h));
    hipError_t e;
    if (f16) e = fast_prog ? launch_n12<N12SwapMulSubDiv, _Float16>(c_fd, inline_planes, n_inline, g, s, win)
                           : (canon_prog ? launch_n12<K1CanonProg, _Float16>(c_fd, inline_planes, n_inline, g, s, win)
                                         : launch_n12<InterpProg, _Float16>(c_fd, inline_planes, n_inline, g, s, win));
    else if (fast_prog) e = launch_n12<N12SwapMulSubDiv>(c_fd, inline_planes, n_inline, g, s, win);
    else if (fast_rgb) e = launch_n12<ProgMulSubDiv>(c_fd, inline_planes, n_inline, g, s, win);
    else e = canon_prog ? launch_n12<K1CanonProg>(c_fd, inline_planes, n_inline, g, s, win) : launch_n12<InterpProg>(c_fd, inline_planes, n_inline, g, s, win);
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

} // namespace cvgs
