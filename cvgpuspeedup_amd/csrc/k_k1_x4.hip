// k_k1_x4.hip -- whole-frame resize into PACKED pixels of the source's own type, several x-adjacent output pixels per lane.
//
// The chain: resize<INTER_LINEAR>(image of type I) -> convertTo<CV_32F, I> -> write<I>(packed image), I = CV_8UCn, CV_16UCn,
// CV_16SCn or CV_32FC1 (the types the reference sweeps) --
// the reference's tests/resize/test_resize_write.cu chain (cvGS::resize + cvGS::convertTo + cvGS::write, :110-123) at
// whole-frame sizes (1080p <-> 4K).  k1_resize_split (lane = ONE output column) is VALU + SALU issue bound there
// (369 VALU + 218 SALU per 4-row wave on 1080p -> 4K packed u8, 23 us, 0.17 of the HBM roofline): the
// per-pixel work is dominated by what is NOT the interpolation -- the 64-bit window shift, the edge selects, the wave
// shuffle that turns 3-byte pixels into dword stores, per-row address arithmetic.
//
// This kernel's mapping:
//  * lane = PX x-adjacent output pixels (4 of a u8 or fp32 image, 2 of a 16-bit one: 4*CN bytes of u8 / 16-bit pixels, 16 of
//    fp32), wave = 64*PX output columns x R consecutive output rows (R is a launch parameter).  The lane's pixels leave as
//    ONE store (rows may start on any byte): no shuffle.
//  * column geometry (x1, weights, window offset) once per lane and pixel, reused for all R rows; for u8 the window's byte
//    shift AND the right-edge duplication (x2 clamped onto x1) are ONE v_perm_b32 selector per dword, computed with the
//    geometry; 16-bit windows (16 bytes) take k_taps.hpp's shift + select only in the waves that hold a row's last pixels.
//  * row geometry for all R rows at once: lane j computes row row0 + j, the loop reads it back with v_readlane.
//  * the unpacked taps of a SOURCE row live in one of two register slots; consecutive source intervals alternate the slots'
//    roles (the interpolation is written once per order), so an output row whose source rows are already there (every
//    second row of a 2x up-scaling) loads and unpacks nothing and nothing is ever moved between registers.
//  * every source row of the wave's first PRE intervals (PRE = 1, 2, 4 by the rows the wave spans) is requested before
//    anything is computed: one memory wait per wave, then arithmetic and stores nobody waits for.
//  * the interpolation runs on pixel PAIRS (v_pk_mul_f32 / v_pk_add_f32): same IEEE operations in the same order as the
//    oracle (p00*w00 + p10*w10 + p01*w01 + p11*w11, left to right, no FMA), two pixels per instruction.
// Measured (tools/bench_upscale.py, profiles/r03_*): 1080p -> 4K packed u8c3 23.0 -> 10.8 us, u8c4 23.9 -> 11.0, u8c1 17.5 -> 8.3;
// 720p -> 1440p 13.8 -> 6.8; the reference's 4K -> 3870 x 2260: 8UC3 30.1 -> 19.8, 8UC1 23.6 -> 15.8, 8UC4 25.7 -> 19.4.
// Round 5: ONE shared 8-byte tap window per lane and source row for u8 images of 1-2 channels and one-channel 16-bit images (x4_load<SH>):
// the counters of the one-channel case (profiles/r05_h_c1_pmc_sq*.txt) showed 62 % of the wave cycles stalled on instruction issue with the
// pipes used one after the other (all resident waves are in the same phase) -- 20 window loads per wave were a third of the launch.  4K ->
// 3870 x 2260: 8UC1 15.7 -> 9.4 us, 16SC1 21.8 -> 14.4 us; 1080p -> 4K 8UC1 8.4 -> 6.0 us (profiles/r05_k_*, r05_l_*).
// What bounds it now: issue (tools/probes/pk_rate_probe.cpp: v_pk_mul_f32 / v_pk_add_f32 retire at 5.7 cycles per
// instruction per SIMD, conversions at 4.4 - 4.8; 3.4 M VALU + 1.4 M SALU instructions per 4K launch = ~7 us of issue at
// 4 waves per SIMD) -- a launch without its stores runs 10.3 us against 11.3 with them (profiles/r03_d_*).
// Round 6: the kernel's own skeletons at the reference's 4K -> 3870 x 2260 size (tools/probes/tick_ablation.py --workload resize_write;
// profiles/r06_f_resize_write_c*.txt) showed u8c3 bound by its tap-load INSTRUCTIONS (loads alone 13.3 us for 25 MB): u8c3 / u8c4 images without
// horizontal down-scaling now take ONE 16-byte (+ 4-byte) window per lane and source row (x4_load<W16>): 19.6 -> 17.2 us, u8c4 19.2 -> 18.9.
// Vertical DOWN-scaling stays with k1_resize_split: no source row is shared, and it ties or wins there (4K -> 1080p 10.3 us).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "k_taps.hpp"

namespace cvgs {

// Ablation hooks of tools/probes/tick_ablation.py --workload resize_write (round 6; never defined in the product build, see k_k1_impl.hpp):
//   1 no tap loads (windows synthesised from the lane id)   2 no arithmetic (the windows' bytes are stored as they are)   4 no stores
#ifndef CVGS_X4_ABLATE
#define CVGS_X4_ABLATE 0
#endif
constexpr int kX4Ablate = CVGS_X4_ABLATE;

constexpr int kX4Planes = 8;   // images per launch (blockIdx.y)
constexpr int kX4Waves = 4;    // waves per workgroup (independent)
#ifndef CVGS_X4_PRE
#define CVGS_X4_PRE 4
#endif
constexpr int kX4Pre = CVGS_X4_PRE; // most source intervals whose rows a wave requests up front (PRE + 1 source rows; PRE = 1, 2, 4)

struct X4Plane { // 32 bytes
    const uint8_t* data;
    int32_t w, h, step;
    float fx, fy;
    int32_t pad;
};
struct X4Args {
    X4Plane plane[kX4Planes];
    uint8_t* out;
    int64_t row_pitch, img_pitch; // bytes
    int32_t dst_w, dst_h;
    uint32_t col_tiles;           // ceil(dst_w / (64 * PX))
    int32_t rows_per_wave;        // <= 64
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// output pixels per lane and bytes per tap window (both horizontal taps of one pixel) by source kind
template <int SRC> constexpr int x4_px = (SRC == SRC_U16 || SRC == SRC_S16) ? 2 : 4;
template <int SRC> constexpr int x4_winb = (SRC == SRC_U16 || SRC == SRC_S16) ? 16 : 8;
template <int SRC> using x4_elem_t = std::conditional_t<SRC == SRC_U8, uint8_t, std::conditional_t<SRC == SRC_U16, uint16_t, std::conditional_t<SRC == SRC_S16, int16_t, float>>>;

// the unpacked taps of one source row for the lane's output pixels: [pixel pair][channel] x {pixel 2q, pixel 2q+1}
template <int CN, int PX>
struct X4Slot {
    f32x2 t0[PX / 2][CN]; // tap x1
    f32x2 t1[PX / 2][CN]; // tap x2 (= x1 at the right edge)
};

template <int CN, int SRC>
struct X4Col {
    static constexpr int PX = x4_px<SRC>;
    f32x2 wxa[PX / 2], wxb[PX / 2];
    uint32_t ol[PX]; // SH (shared window): ol[0] only
    // W16 (u8c3, no horizontal down-scaling): ONE unaligned 16-byte load per lane and source row at `ob` holds the taps of all four pixels
    // (they span at most 5 source pixels = 15 bytes); pixel i's window starts rel[i] = ol[i] - ob bytes in (0..9)
    uint32_t ob, rel[PX];
    // u8: the v_perm_b32 selectors of the window's two dwords; 16-bit: the window's shift in bits, fp32: "the window was
    // clamped back by one pixel" (fix_a), and the right-edge flag (fix_b)
    uint32_t fix_a[PX], fix_b[PX];
    bool any_fix; // wave-uniform: some pixel of this wave sits at its row's end (16-bit windows only)
};

// bytes [CN, 2CN) of a u8 window hold the second tap; at the right edge they are re-pointed at the first one
template <int CN> constexpr uint32_t x4_edge_sub(int dword) {
    uint32_t v = 0;
    for (int k = 0; k < 4; ++k) {
        const int b = dword * 4 + k;
        if (b >= CN && b < 2 * CN) v |= (uint32_t)CN << (8 * k);
    }
    return v;
}

// the lane's tap windows of one source row, as loaded
template <int SRC> struct X4Raw { uint64_t w[x4_px<SRC>]; };
template <> struct X4Raw<SRC_U16> { u32x4 w[2]; };
template <> struct X4Raw<SRC_S16> { u32x4 w[2]; };

// SH: ONE 8-byte window per lane and source row holds the taps of ALL the lane's pixels (u8 images of 1-2 channels when the lane's
// pixels lie close enough: launch_k1_packed_x4) -- a quarter of the load instructions of one window per pixel.
template <int CN, int SRC, bool SH = false, bool W16 = false>
__device__ __forceinline__ X4Raw<SRC> x4_load(const X4Col<CN, SRC>& col, gptr_u8 row) {
    X4Raw<SRC> r;
    if constexpr (W16 && CN == 4 && (kX4Ablate & 1) == 0) {
        // u8c4: the four pixels tap at most 5 source pixels = 20 bytes, dword aligned: one 16-byte + one 4-byte load, every window = two of the
        // five dwords, picked by the pixel's dword offset (0..3)
        static_assert(SRC == SRC_U8 && !SH, "wide windows: u8 images");
        typedef uint32_t u32_unaligned __attribute__((aligned(1)));
        typedef const __attribute__((address_space(1))) u32_unaligned* gptr_u32u;
        const u32x4 q = *(gptr_u32x4)(row + col.ob);
        const uint32_t e = *(gptr_u32u)(row + col.ob + 16);
#pragma unroll
        for (int i = 0; i < x4_px<SRC>; ++i) {
            const uint32_t k = col.rel[i] >> 2;
            const uint32_t lo = k == 0 ? q.x : (k == 1 ? q.y : (k == 2 ? q.z : q.w));
            const uint32_t hi = k == 0 ? q.y : (k == 1 ? q.z : (k == 2 ? q.w : e));
            r.w[i] = ((uint64_t)hi << 32) | lo;
        }
        return r;
    }
    if constexpr (W16 && CN == 3 && (kX4Ablate & 1) == 0) {
        static_assert(SRC == SRC_U8 && !SH, "wide windows: u8 images");
        // Round 6 (profiles/r06_f_resize_write_c3.txt): four overlapping unaligned 8-byte loads per lane and source row made the launch
        // LOAD-INSTRUCTION bound (tap loads alone 13.3 us for 25 MB; an unaligned vector load costs the address unit ~40 cycles whatever its
        // width).  One 16-byte load brings the same bytes; each pixel's 8-byte window is cut out of it with two v_alignbyte_b32 on the dwords
        // its offset selects (5 selects: the offset is 0..9 bytes, so the window starts in dword 0, 1 or 2).
        const u32x4 q = *(gptr_u32x4)(row + col.ob);
#pragma unroll
        for (int i = 0; i < x4_px<SRC>; ++i) {
            const uint32_t k = col.rel[i] >> 2, m = col.rel[i] & 3u;
            const uint32_t a = k == 0 ? q.x : (k == 1 ? q.y : q.z);
            const uint32_t b = k == 0 ? q.y : (k == 1 ? q.z : q.w);
            const uint32_t c = k == 0 ? q.z : q.w;
            const uint32_t lo = __builtin_amdgcn_alignbyte(b, a, m), hi = __builtin_amdgcn_alignbyte(c, b, m);
            r.w[i] = ((uint64_t)hi << 32) | lo;
        }
        return r;
    }
    if constexpr ((kX4Ablate & 1) != 0 && x4_winb<SRC> == 8) { // probe: no tap loads
#pragma unroll
        for (int i = 0; i < x4_px<SRC>; ++i) r.w[i] = (uint64_t)(threadIdx.x * 0x01010101u + (uint32_t)(uintptr_t)row + i) * 0x100000001ull;
        return r;
    }
    if constexpr (SH) {
        static_assert(SRC == SRC_U8 || ((SRC == SRC_U16 || SRC == SRC_S16) && CN == 1), "shared windows: u8 images, one-channel 16-bit images");
        const uint64_t w = *(gptr_u64)(row + col.ol[0]);
        if constexpr (SRC == SRC_U8) {
#pragma unroll
            for (int i = 0; i < x4_px<SRC>; ++i) r.w[i] = w;
        } else { // the 8 bytes (4 elements) in the first half of the first window slot
            r.w[0] = u32x4{(uint32_t)w, (uint32_t)(w >> 32), 0u, 0u};
            r.w[1] = r.w[0];
        }
        return r;
    }
#pragma unroll
    for (int i = 0; i < x4_px<SRC>; ++i) {
        if constexpr (x4_winb<SRC> == 16) r.w[i] = *(gptr_u32x4)(row + col.ol[i]);
        else r.w[i] = *(gptr_u64)(row + col.ol[i]);
    }
    return r;
}

template <int CN, int SRC, bool SH = false>
__device__ __forceinline__ void x4_unpack(X4Slot<CN, x4_px<SRC>>& s, const X4Col<CN, SRC>& col, const X4Raw<SRC>& r) {
#pragma unroll
    for (int i = 0; i < x4_px<SRC>; ++i) {
        if constexpr (SH && SRC != SRC_U8) { // one-channel 16-bit image, shared 8-byte window: the pixel's two elements sit fix_a[i] bits in
            const uint64_t w64 = ((uint64_t)r.w[0].y << 32) | r.w[0].x;
            const uint32_t v = (uint32_t)(w64 >> col.fix_a[i]);
            const float a = elem_to_float<SRC>(v & 0xffffu);
            s.t0[i >> 1][0][i & 1] = a;
            s.t1[i >> 1][0][i & 1] = col.fix_b[i] ? a : elem_to_float<SRC>(v >> 16);
        } else if constexpr (SRC == SRC_U8) {
            const uint32_t wl = (uint32_t)r.w[i], wh = (uint32_t)(r.w[i] >> 32);
            const uint32_t lo = __builtin_amdgcn_perm(wh, wl, col.fix_a[i]);
            [[maybe_unused]] uint32_t hi = 0;
            if constexpr (2 * CN > 4) hi = __builtin_amdgcn_perm(wh, wl, col.fix_b[i]);
#pragma unroll
            for (int c = 0; c < CN; ++c) {
                const int b0 = c, b1 = CN + c;
                const uint32_t e0 = ((b0 < 4 ? lo : hi) >> (8 * (b0 & 3))) & 0xffu;
                const uint32_t e1 = ((b1 < 4 ? lo : hi) >> (8 * (b1 & 3))) & 0xffu;
                s.t0[i >> 1][c][i & 1] = (float)e0;
                s.t1[i >> 1][c][i & 1] = (float)e1;
            }
        } else if constexpr (SRC == SRC_F32) {
            static_assert(SRC != SRC_F32 || CN == 1, "fp32 sources: one channel (the window is the pixel pair)");
            const float a0 = __builtin_bit_cast(float, (uint32_t)r.w[i]), a1 = __builtin_bit_cast(float, (uint32_t)(r.w[i] >> 32));
            const float first = col.fix_a[i] ? a1 : a0;
            s.t0[i >> 1][0][i & 1] = first;
            s.t1[i >> 1][0][i & 1] = col.fix_b[i] ? first : a1;
        } else {
            Win<2> w;
            w.lo = ((uint64_t)r.w[i].y << 32) | r.w[i].x;
            w.hi = ((uint64_t)r.w[i].w << 32) | r.w[i].z;
            float a[4], b[4];
            if (col.any_fix) unpack_pair<CN, SRC>(shift_win<2>(w, (int)col.fix_a[i]), col.fix_b[i] != 0, a, b); // a row's last pixels
            else unpack_pair<CN, SRC>(w, false, a, b);
#pragma unroll
            for (int c = 0; c < CN; ++c) {
                s.t0[i >> 1][c][i & 1] = a[c];
                s.t1[i >> 1][c][i & 1] = b[c];
            }
        }
    }
}

template <int CN, int SRC>
__device__ __forceinline__ void x4_row(const X4Slot<CN, x4_px<SRC>>& A, const X4Slot<CN, x4_px<SRC>>& B, const X4Col<CN, SRC>& col,
                                       float wya, float wyb, uint8_t* orow, uint32_t x0, int dst_w, bool wave_full) {
    constexpr int PX = x4_px<SRC>;
    using OT = x4_elem_t<SRC>;
    float v[PX * CN];
    if constexpr ((kX4Ablate & 2) != 0) { // probe: no arithmetic -- one unpacked tap of each source row, as it is
#pragma unroll
        for (int q = 0; q < PX / 2; ++q)
#pragma unroll
            for (int c = 0; c < CN; ++c) {
                v[(2 * q) * CN + c] = A.t0[q][c].x + B.t1[q][c].y;
                v[(2 * q + 1) * CN + c] = A.t1[q][c].y + B.t0[q][c].x;
            }
    } else
#pragma unroll
    for (int q = 0; q < PX / 2; ++q) {
        const f32x2 w00 = col.wxa[q] * wya;
        const f32x2 w10 = col.wxb[q] * wya;
        const f32x2 w01 = col.wxa[q] * wyb;
        const f32x2 w11 = col.wxb[q] * wyb;
#pragma unroll
        for (int c = 0; c < CN; ++c) {
            f32x2 acc = A.t0[q][c] * w00;
            acc = acc + A.t1[q][c] * w10;
            acc = acc + B.t0[q][c] * w01;
            acc = acc + B.t1[q][c] * w11;
            v[(2 * q) * CN + c] = acc.x;
            v[(2 * q + 1) * CN + c] = acc.y;
        }
    }
    typedef __attribute__((address_space(1))) uint8_t* gout;
    // the lane's PX pixels as dwords: the chain's trailing SaturateCast is the conversion (k_common.hpp: sat_u8_insert,
    // sat_u16_bits, sat_s16_bits -- the instructions tools/sat_probe.cpp checked over all 2^32 inputs); fp32 as it is
    constexpr int NW = PX * CN * (int)sizeof(OT) / 4;
    uint32_t word[NW];
    if constexpr (SRC == SRC_U8) {
#pragma unroll
        for (int k = 0; k < NW; ++k) word[k] = 0;
#pragma unroll
        for (int b = 0; b < PX * CN; ++b) word[b >> 2] = sat_u8_insert(v[b], (uint32_t)(b & 3), word[b >> 2]);
    } else if constexpr (SRC == SRC_F32) {
#pragma unroll
        for (int k = 0; k < NW; ++k) word[k] = __builtin_bit_cast(uint32_t, v[k]);
    } else {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const uint32_t e0 = SRC == SRC_U16 ? sat_u16_bits(v[2 * k]) : sat_s16_bits(v[2 * k]);
            const uint32_t e1 = SRC == SRC_U16 ? sat_u16_bits(v[2 * k + 1]) : sat_s16_bits(v[2 * k + 1]);
            word[k] = e0 | (e1 << 16);
        }
    }
    const gout p = (gout)pin_uniform(orow) + x0 * (uint32_t)(CN * sizeof(OT));
    if constexpr ((kX4Ablate & 4) != 0) { // probe: no stores (the test never holds on pixel data; the loads and the arithmetic stay alive)
        if (!(word[0] == 0x9e3779b9u && word[NW - 1] == 0x7f4a7c15u && v[0] == 1234.5f)) return;
    }
    auto store_all = [&]() {
        // byte-aligned types: rows of a 3870-pixel u8c3 image (the reference's tests/resize/test_resize_write.cu size) start on
        // any byte; the hardware takes the unaligned multi-dword store
        typedef uint32_t vw_a4 __attribute__((ext_vector_type(NW)));
        typedef vw_a4 vw __attribute__((aligned(1)));
        typedef __attribute__((address_space(1))) vw* gvw;
        typedef uint32_t u32a1 __attribute__((aligned(1)));
        if constexpr (NW == 1) {
            __builtin_nontemporal_store(word[0], (__attribute__((address_space(1))) u32a1*)p);
        } else {
            vw q;
#pragma unroll
            for (int k = 0; k < NW; ++k) q[k] = word[k];
            __builtin_nontemporal_store(q, (gvw)p);
        }
    };
    if (wave_full) { // scalar: no lane of this column tile hangs over the row's end
        store_all();
    } else if ((int)x0 + PX - 1 < dst_w) {
        store_all();
    } else { // the ragged lane: element by element
        typedef std::conditional_t<sizeof(OT) == 1, uint8_t, std::conditional_t<sizeof(OT) == 2, uint16_t, uint32_t>> ET;
        typedef ET ET_a1 __attribute__((aligned(1)));
        typedef __attribute__((address_space(1))) ET_a1* get;
        constexpr int EPW = 4 / (int)sizeof(OT); // elements per dword
#pragma unroll
        for (int e = 0; e < PX * CN; ++e)
            if ((int)x0 + e / CN < dst_w)
                __builtin_nontemporal_store((ET)(word[e / EPW] >> (8 * (int)sizeof(OT) * (e % EPW))), (get)p + e);
    }
}

template <int CN, int SRC, int PRE, bool SH = false, bool W16 = false>
__global__ __launch_bounds__(64 * kX4Waves) void k1_packed_x4(const X4Args a) {
    constexpr int PX = x4_px<SRC>;
    constexpr int EB = elem_bytes<SRC>;
    constexpr int WINB = x4_winb<SRC>;
    const int z = (int)blockIdx.y;
    const X4Plane P = a.plane[z];
    const int dst_w = a.dst_w, dst_h = a.dst_h, R = a.rows_per_wave;
    const uint32_t col_tiles = a.col_tiles;
    uint32_t col_tile = 0, row_blk = blockIdx.x;
    if (col_tiles > 1) {
        col_tile = blockIdx.x % col_tiles;
        row_blk = blockIdx.x / col_tiles;
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int row0 = ((int)row_blk * kX4Waves + wave) * R;
    if (row0 >= dst_h) return;
    const int nrows = min(R, dst_h - row0);
    const uint32_t x0 = (col_tile * 64u + (uint32_t)lane) * PX;

    // ---- column geometry: the oracle's expressions per pixel (sx = x * fx in fp32, x1 = floor, the two weights) ----
    X4Col<CN, SRC> col;
    const int row_bytes = P.w * CN * EB;
    bool fix = false;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int x = min((int)x0 + i, dst_w - 1); // lanes / pixels past the row compute a valid pixel and store nothing
        const float sx = (float)x * P.fx;
        const int x1 = (int)floorf(sx);
        const int x2 = x1 + 1;
        col.wxa[i >> 1][i & 1] = (float)x2 - sx;
        col.wxb[i >> 1][i & 1] = sx - (float)x1;
        const bool edge = x2 > P.w - 1;
        const int o = x1 * CN * EB;
        int ol = min(o, row_bytes - (SH ? 8 : WINB)); // (shared windows are 8 bytes for every type)
        if constexpr (SH) { // the lane's one window starts at its FIRST pixel's tap (clamped into the row): o >= that start for every pixel
            if (i == 0) col.ol[0] = (uint32_t)ol;
            ol = (int)col.ol[0];
        }
        const uint32_t sh = (uint32_t)(o - ol);
        if constexpr (!SH) col.ol[i] = (uint32_t)ol;
        if constexpr (W16) { // (row_bytes >= 16: the launcher checks)
            if (i == 0) col.ob = (uint32_t)min(o, row_bytes - (CN == 4 ? 20 : 16));
            col.rel[i] = (uint32_t)ol - col.ob;
        }
        if constexpr (SRC == SRC_U8) {
            col.fix_a[i] = 0x03020100u + sh * 0x01010101u - (edge ? x4_edge_sub<CN>(0) : 0u);
            col.fix_b[i] = 0x07060504u + sh * 0x01010101u - (edge ? x4_edge_sub<CN>(1) : 0u);
        } else {
            col.fix_a[i] = SRC == SRC_F32 ? (uint32_t)(sh != 0) : sh * 8u;
            col.fix_b[i] = (uint32_t)edge;
            fix = fix || sh != 0 || edge;
        }
    }
    col.any_fix = __builtin_amdgcn_ballot_w64(fix) != 0;

    // ---- row geometry: lane j holds output row row0 + j ----
    int y1v;
    float wyav, wybv;
    {
        const int y = min(row0 + lane, dst_h - 1);
        const float sy = (float)y * P.fy;
        y1v = (int)floorf(sy);
        wyav = (float)(y1v + 1) - sy;
        wybv = sy - (float)y1v;
    }

    const gptr_u8 src = (gptr_u8)P.data;
    uint8_t* const out = a.out + (int64_t)z * a.img_pitch + (int64_t)row0 * a.row_pitch;
    X4Slot<CN, PX> S0, S1;
    auto row_of = [&](int r) { return pin_uniform(src + (size_t)r * (size_t)P.step); };
    auto y1_of = [&](int j) { return __builtin_amdgcn_readlane(y1v, j); };
    const bool wave_full = (int)(col_tile + 1) * 64 * PX <= dst_w;
    auto emit = [&](const X4Slot<CN, PX>& A, const X4Slot<CN, PX>& B, int j) {
        const float wya = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wyav), j));
        const float wyb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wybv), j));
        x4_row<CN, SRC>(A, B, col, wya, wyb, out + (int64_t)j * a.row_pitch, x0, dst_w, wave_full);
    };
    int j = 0;
    const int h1 = P.h - 1;
    const bool mine = lane < nrows;
    // y1 never decreases with the output row, so the rows fed from source rows (s, s+1) are the lanes whose y1v == s: one ballot
    auto rows_on = [&](int s) { return (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(mine && y1v == s)); };

    // ---- the wave's first PRE source intervals: ALL their rows are requested before anything is computed ----
    // The wave then waits for memory ONCE; what follows is arithmetic and stores nobody waits for.  (Fetching each new source
    // row when its interval starts cost one load + store-acknowledge round trip per interval: 4K -> 3870 x 2260, one new source
    // row per output row, ran 8 such round trips per wave: 23.7 us.)  Source rows s, s+1 sit in (S0, S1) for the even intervals
    // and in (S1, S0) for the odd ones: one row is unpacked per interval and the slots never trade registers.
    {
        const int s0 = y1_of(0);
        X4Raw<SRC> raw[PRE + 1];
#pragma unroll
        for (int k = 0; k <= PRE; ++k) raw[k] = x4_load<CN, SRC, SH, W16>(col, row_of(min(s0 + k, h1)));
        x4_unpack<CN, SRC, SH>(S0, col, raw[0]);
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            if (j < nrows) { // wave-uniform
                const int jn = j + rows_on(s0 + k);
                if ((k & 1) == 0) {
                    x4_unpack<CN, SRC, SH>(S1, col, raw[k + 1]);
#pragma unroll 1
                    for (; j < jn; ++j) emit(S0, S1, j);
                } else {
                    x4_unpack<CN, SRC, SH>(S0, col, raw[k + 1]);
#pragma unroll 1
                    for (; j < jn; ++j) emit(S1, S0, j);
                }
            }
        }
    }

    // ---- whatever is left (the launcher sizes the wave's rows so that nothing is, for vertical up-scaling): interval by interval.
    // When the next output row starts one source row further down only ONE row is loaded and unpacked; any other step
    // (fy >= 2, or the next row is further away) restarts with both rows.
#pragma unroll 1
    while (j < nrows) {
        int s = y1_of(j);
        const X4Raw<SRC> first = x4_load<CN, SRC, SH, W16>(col, row_of(s));
        X4Raw<SRC> next = x4_load<CN, SRC, SH, W16>(col, row_of(min(s + 1, h1)));
        x4_unpack<CN, SRC, SH>(S0, col, first);
#pragma unroll 1
        for (;;) {
            x4_unpack<CN, SRC, SH>(S1, col, next);
            int jn = j + rows_on(s);
            bool more = jn < nrows && y1_of(jn) == s + 1;
            if (more) next = x4_load<CN, SRC, SH, W16>(col, row_of(min(s + 2, h1)));
#pragma unroll 1
            for (; j < jn; ++j) emit(S0, S1, j);
            if (!more) break;
            ++s;
            x4_unpack<CN, SRC, SH>(S0, col, next);
            jn = j + rows_on(s);
            more = jn < nrows && y1_of(jn) == s + 1;
            if (more) next = x4_load<CN, SRC, SH, W16>(col, row_of(min(s + 2, h1)));
#pragma unroll 1
            for (; j < jn; ++j) emit(S1, S0, j);
            if (!more) break;
            ++s;
        }
    }
}

// Host side.  Takes the chain when the target holds packed pixels of the SOURCE's type (8U / 16U / 16S with 1, 3 or 4 channels --
// 8U also 2 --, 32F with one), every image of the batch covers its whole target (no aspect-ratio padding, no unused planes),
// the source rows hold at least one tap window and -- unless `force`d (CVGS_K1_X4=1: tests) -- no image is scaled down vertically.
// Returns 1 launched / 0 not eligible / < 0 error (as launch_k1).
int launch_k1_packed_x4(const ChainArgs& c, const PlaneParams* planes, int n_planes, void* stream, bool dry_run, bool force) {
    const ReadArgs& r = c.read;
    const WriteArgs& w = c.write;
    if (r.cn < 1 || r.cn > 4 || r.table || w.data2 || w.table) return 0;
    if (w.kind != CVGS_WRITE_PIXEL_2D && w.kind != CVGS_WRITE_PIXEL_3D) return 0;
    // packed fp32 pixels of a u8 source stay with the one-pixel kernel: its per-pixel 12 / 16-byte stores already run the 4K
    // output at 4.6 TB/s, while a lane that owns 4 three-channel fp32 pixels writes 48-byte strides (measured: 1080p -> 4K 49 us against 22.8)
    if (w.depth != r.depth || w.cn != r.cn) return 0;
    int src;
    switch (r.depth) {
    case CVGS_DEPTH_8U: src = SRC_U8; break;
    case CVGS_DEPTH_16U: src = SRC_U16; break;
    case CVGS_DEPTH_16S: src = SRC_S16; break;
    case CVGS_DEPTH_32F: src = SRC_F32; break;
    default: return 0;
    }
    if (src != SRC_U8 && r.cn == 2) return 0;
    if (src == SRC_F32 && r.cn != 1) return 0;
    const int px = (src == SRC_U16 || src == SRC_S16) ? 2 : 4;
    const int eb = src == SRC_U8 ? 1 : (src == SRC_F32 ? 4 : 2), winb = px == 2 ? 16 : 8;
    if (r.batch < 1 || r.batch > kX4Planes || r.used != r.batch || n_planes != r.batch) return 0;
    if (r.dst_w < px || r.dst_h < 1) return 0;
    const int64_t px_bytes = (int64_t)w.cn * eb;
    const int64_t row_pitch = w.kind == CVGS_WRITE_PIXEL_2D ? w.step : w.width * px_bytes;
    const int64_t img_pitch = w.kind == CVGS_WRITE_PIXEL_2D ? 0 : w.img_stride * px_bytes;
    X4Args a;
    for (int i = 0; i < kX4Planes; ++i) {
        const PlaneParams& p = planes[i < n_planes ? i : 0];
        if (i < n_planes) {
            if (p.x1 != 0 || p.y1 != 0 || p.x2 != r.dst_w - 1 || p.y2 != r.dst_h - 1) return 0;
            if ((int64_t)p.w * r.cn * eb < winb || p.h < 1) return 0;
            // the kernel's gain is the source row shared by consecutive output rows: vertical up-scaling (or 1:1).  Down-scaling
            // launches tie with the one-pixel kernel or lose to it (4K -> 1080p 9.9 - 12 us against 10.3), so they stay there.
            if (!force && !(p.fy <= 1.0f)) return 0;
        }
        a.plane[i] = X4Plane{p.data, p.w, p.h, p.step, p.fx, p.fy, 0};
    }
    if (dry_run) return 1;
    a.out = w.data;
    a.row_pitch = row_pitch;
    a.img_pitch = img_pitch;
    a.dst_w = r.dst_w;
    a.dst_h = r.dst_h;
    a.col_tiles = (uint32_t)((r.dst_w + 64 * px - 1) / (64 * px));
    // rows per wave: enough waves to fill the chip (~4 per SIMD) before the column geometry is amortised over more rows
    // (tools/bench_upscale.py: 4K output 8 rows 10.7 us / 4 rows 14.2; 1440p 4 rows 7.0 / 8 rows 9.5; 720p 2 rows 4.0 / 8 rows 5.6),
    // and no more rows than the kX4Pre source intervals requested up front feed (4K -> 3870 x 2260, fy 0.956: 4 rows 20.4 us / 8 rows 22.8)
    const int64_t wave_rows = (int64_t)r.dst_h * a.col_tiles * r.batch;
    int rows_per_wave = (int)((wave_rows + 2048) / 4096);
    rows_per_wave = rows_per_wave < 2 ? 2 : (rows_per_wave > 8 ? 8 : rows_per_wave);
    float fy_max = 0.f;
    for (int i = 0; i < n_planes; ++i) fy_max = planes[i].fy > fy_max ? planes[i].fy : fy_max;
    const int rows_fed = fy_max > 0.f ? (int)((float)kX4Pre / fy_max) : 8;
    if (rows_fed < rows_per_wave) rows_per_wave = rows_fed < 1 ? 1 : rows_fed;
#ifdef CVGS_X4_ROWS // (tools/probes/build_ablate.sh: other rows per wave for A/B)
    rows_per_wave = CVGS_X4_ROWS;
#endif
    a.rows_per_wave = rows_per_wave > 64 ? 64 : rows_per_wave;
    const int rows_per_wg = kX4Waves * a.rows_per_wave;
    const uint32_t row_blks = (uint32_t)((r.dst_h + rows_per_wg - 1) / rows_per_wg);
    const dim3 grid(a.col_tiles * row_blks, (unsigned)r.batch), block(64 * kX4Waves);
    hipStream_t s = (hipStream_t)stream;
    // source intervals requested up front: what the wave's rows span (a wave that needs 2 source rows must not fetch 5)
    const float spanned = (float)a.rows_per_wave * (fy_max > 0.f ? fy_max : 1.f);
    const int pre = spanned <= 1.0f ? 1 : (spanned <= 2.0f ? 2 : kX4Pre);
    // shared tap windows (u8, 1-2 channels): the lane's 4 pixels tap at most (floor(3 fx) + 1) source pixels apart, + the second tap:
    // everything inside ONE 8-byte window
    const bool sh16 = (src == SRC_U16 || src == SRC_S16) && r.cn == 1; // 2 pixels per lane, 2-byte elements: 4 elements per window
    bool shared = ((src == SRC_U8 && r.cn <= 2) || sh16) && fy_max > 0.f;
    for (int i = 0; i < n_planes && shared; ++i)
        shared = ((int)std::floor((double)(px - 1) * (double)planes[i].fx * 1.0001) + 1) * r.cn * eb + 2 * r.cn * eb <= 8;
    // u8c3 without horizontal down-scaling: the lane's four pixels tap at most 5 source pixels (15 bytes) -- one 16-byte window per lane
    // (u8c4: 20 bytes = a 16-byte and a 4-byte load instead of four 8-byte ones)
    bool wide16 = src == SRC_U8 && r.cn >= 3;
    for (int i = 0; i < n_planes && wide16; ++i) wide16 = planes[i].fx <= 1.0f && (int64_t)planes[i].w * r.cn >= (r.cn == 4 ? 20 : 16);
#ifdef CVGS_X4_NO_W16 // (tools/probes/build_ablate.sh builds the four-window form for A/B)
    wide16 = false;
#endif
    auto go = [&](auto cn_tag, auto src_tag) {
        constexpr int CN = decltype(cn_tag)::value, SRC = decltype(src_tag)::value;
        if constexpr ((SRC == SRC_U8 && CN <= 2) || ((SRC == SRC_U16 || SRC == SRC_S16) && CN == 1)) {
            if (shared) {
                if (pre == 1) hipLaunchKernelGGL((k1_packed_x4<CN, SRC, 1, true>), grid, block, 0, s, a);
                else if (pre == 2) hipLaunchKernelGGL((k1_packed_x4<CN, SRC, 2, true>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((k1_packed_x4<CN, SRC, kX4Pre, true>), grid, block, 0, s, a);
                return;
            }
        }
        if constexpr (SRC == SRC_U8 && CN >= 3) {
            if (wide16) { // one 16-byte window per lane and source row (see x4_load)
                if (pre == 1) hipLaunchKernelGGL((k1_packed_x4<CN, SRC, 1, false, true>), grid, block, 0, s, a);
                else if (pre == 2) hipLaunchKernelGGL((k1_packed_x4<CN, SRC, 2, false, true>), grid, block, 0, s, a);
                else hipLaunchKernelGGL((k1_packed_x4<CN, SRC, kX4Pre, false, true>), grid, block, 0, s, a);
                return;
            }
        }
        if (pre == 1) hipLaunchKernelGGL((k1_packed_x4<CN, SRC, 1>), grid, block, 0, s, a);
        else if (pre == 2) hipLaunchKernelGGL((k1_packed_x4<CN, SRC, 2>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k1_packed_x4<CN, SRC, kX4Pre>), grid, block, 0, s, a);
    };
    auto by_cn = [&](auto src_tag) {
        constexpr int SRC = decltype(src_tag)::value;
        if constexpr (SRC == SRC_F32) {
            go(std::integral_constant<int, 1>{}, src_tag);
        } else {
            switch (r.cn) {
            case 1: go(std::integral_constant<int, 1>{}, src_tag); break;
            case 2:
                if constexpr (SRC == SRC_U8) go(std::integral_constant<int, 2>{}, src_tag);
                break;
            case 3: go(std::integral_constant<int, 3>{}, src_tag); break;
            default: go(std::integral_constant<int, 4>{}, src_tag); break;
            }
        }
    };
    if (src == SRC_U8) by_cn(std::integral_constant<int, SRC_U8>{});
    else if (src == SRC_U16) by_cn(std::integral_constant<int, SRC_U16>{});
    else if (src == SRC_S16) by_cn(std::integral_constant<int, SRC_S16>{});
    else by_cn(std::integral_constant<int, SRC_F32>{});
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

} // namespace cvgs
