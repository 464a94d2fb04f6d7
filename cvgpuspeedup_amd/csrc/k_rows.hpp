// k_rows.hpp -- the K1 ROW WORKER shared by the descriptor queue's server (k_queue.hip) and the tick kernel of cvgs_execute_many
// (k_tick.hip): one wave computes 4 output rows x 64 columns of one crop -- K1's arithmetic (k_k1_impl.hpp: same geometry, same tap
// windows, same fp32 expression order, the same program stages; bit-identical results) -- and hands the tile to memory as 16-byte
// stores through a wave-private LDS transpose.  Load / store flavours are template parameters: the server outlives kernel boundaries
// and publishes with sc1 write-through stores + sc1 loads; the tick kernel is an ordinary launch (plain cached loads, nt stores).
#pragma once

#include "k_taps.hpp"

namespace cvgs {

constexpr int kQRowsPerWave = 4, kQWaves = 4;

typedef __attribute__((address_space(1))) uint64_t* g_u64;
typedef __attribute__((address_space(1))) uint32_t* g_u32;
#define Q_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define Q_SYSTEM __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM
__device__ __forceinline__ uint64_t q_ld(const uint64_t* p) { return __hip_atomic_load((g_u64)p, Q_AGENT); }
__device__ __forceinline__ void q_st(uint64_t* p, uint64_t v) { __hip_atomic_store((g_u64)p, v, Q_AGENT); }
__device__ __forceinline__ uint64_t q_ld_sys(const uint64_t* p) { return __hip_atomic_load((g_u64)p, Q_SYSTEM); }
__device__ __forceinline__ void q_st_sys(uint64_t* p, uint64_t v) { __hip_atomic_store((g_u64)p, v, Q_SYSTEM); }
// a wave-uniform value the compiler cannot prove uniform (it came through a vector load): pin it into SGPRs
__device__ __forceinline__ uint32_t q_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t q_uni(uint64_t v) { return (uint64_t)q_uni((uint32_t)v) | ((uint64_t)q_uni((uint32_t)(v >> 32)) << 32); }
__device__ __forceinline__ uint64_t q_ldu(const uint64_t* p) { return q_uni(__hip_atomic_load((g_u64)p, Q_AGENT)); }
__device__ __forceinline__ uint64_t q_ldu_sys(const uint64_t* p) { return q_uni(__hip_atomic_load((g_u64)p, Q_SYSTEM)); }
typedef uint64_t q_u64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) q_u64x2* g_u64x2;
__device__ __forceinline__ void q_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t q_lane_u32(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ uint64_t q_lane_u64(uint32_t v, int lane) { return (uint64_t)q_lane_u32(v, lane) | ((uint64_t)q_lane_u32(v, lane + 1) << 32); }
__device__ __forceinline__ float q_lane_f32(uint32_t v, int lane) { return __uint_as_float(q_lane_u32(v, lane)); }

// tap window load flavours: LD 0 plain (cached; A/B upper bound only: may serve a stale line of a rewritten source), 1 sc1
// 16-bit pixels: the 16-byte window (k_taps.hpp: Win<2>)
template <int LD>
__device__ __forceinline__ Win<2> q_load_win16(gptr_u8 p) {
    Win<2> w;
    if constexpr (LD == 0) {
        const u32x4 v = *(gptr_u32x4)p;
        w.lo = ((uint64_t)v.y << 32) | v.x;
        w.hi = ((uint64_t)v.w << 32) | v.z;
    } else { // two sc1 8-byte loads (the atomic builtin has no 16-byte form)
        w.lo = __hip_atomic_load((g_u64)(__attribute__((address_space(1))) uint8_t*)p, Q_AGENT);
        w.hi = __hip_atomic_load((g_u64)(__attribute__((address_space(1))) uint8_t*)(p + 8), Q_AGENT);
    }
    return w;
}
template <int LD>
__device__ __forceinline__ Win<1> q_load_win(gptr_u8 p) {
    Win<1> w;
    if constexpr (LD == 0) w.lo = *(gptr_u64)p;
    else w.lo = __hip_atomic_load((g_u64)(__attribute__((address_space(1))) uint8_t*)p, Q_AGENT); // global_load_dwordx2 ... sc1 (unaligned is fine for the hardware)
    return w;
}

// The 4 x 64 tile a wave has computed (lane = column, register = row) leaves as 16-byte stores: lane (i = lane >> 4, q = lane & 15)
// stores columns 4q .. 4q+3 of row i, so that 16 consecutive lanes cover one row's 256 contiguous bytes and the memory pipeline merges
// four lanes into one 64-byte request (a first version gave consecutive lanes consecutive ROWS -- the layout of its DPP quad
// transposes --: 64 partial-line requests per store instead of 16 whole ones, the L2's request rate then bounded write-heavy
// batches).  The transpose goes through a wave-private LDS tile: four conflict-free ds_write_b32 and one ds_read_b128 per channel
// instead of two DPP rounds of selects (48 VALU instructions per 4 rows of 3 channels).  A 16-lane phase of the read covers 64
// consecutive floats of one tile row: all 64 banks once (the row stride only has to keep 16-byte alignment).  One wave's LDS
// operations execute in order, so the tile is reused without a barrier.
constexpr int kQLdsRow = 80, kQLdsChan = kQRowsPerWave * kQLdsRow, kQLdsWave = 4 * kQLdsChan; // floats
typedef float q_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void q_lds_put(float* tile, int k, const float (&r)[4], int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[k * kQLdsChan + j * kQLdsRow + lane] = r[j];
}
__device__ __forceinline__ q_f32x4 q_lds_get(const float* tile, int k, int lane) {
    return *(const q_f32x4*)(tile + k * kQLdsChan + (lane >> 4) * kQLdsRow + (lane & 15) * 4);
}

struct QTask { // everything a wave needs for its 4 rows, wave-uniform
    PlaneParams P;
    float mul[4], sub[4], div[4], rdiv[4], bg[4];
    int32_t used, dst_w, dst_h, out_w, swap, fast_div;
    int64_t img_stride, ch_stride;
    uint8_t* out;
    uint32_t out_bytes;
    int32_t yuv_range, yuv_primaries, yuv_vu; // QK_NV12 only
    int32_t out_half;                         // CV_16F tensor: 2-byte elements, round-to-nearest-even in the store
    float* tile;                              // this wave's LDS tile (q_lds_put / q_lds_get)
};

// The vertical geometry of 64 consecutive output rows, one row per lane (computed once per 64 rows of a task; the row loop reads its
// rows' values with v_readlane: 4 instructions per row instead of the 16 of computing wave-uniform values on the vector pipe)
struct QRowGeo {
    uint32_t oa, ob;   // byte offsets of the two tap rows: y1 * step, min(y1 + 1, h - 1) * step (sources stay below 4 GB: queue_submit)
    uint32_t ca, cb;   // NV12: byte offsets of their chroma rows inside the UV plane, (y >> 1) * step
    float wya, wyb;    // the taps' weights
    uint64_t in_y;     // bit i: row first + i lies inside the destination window (aspect-ratio modes)
};
__device__ __forceinline__ QRowGeo q_row_geo(const PlaneParams& P, int dst_h, int first, int lane) {
    QRowGeo g;
    const int y = min(first + lane, dst_h - 1);
    const bool in = y >= P.y1 && y <= P.y2;
    const int yr = in ? y - P.y1 : 0;
    const float sy = (float)yr * P.fy;
    const int y1 = (int)floorf(sy);
    const int y2 = y1 + 1;
    const int y2r = min(y2, P.h - 1);
    g.oa = (uint32_t)y1 * (uint32_t)P.step;
    g.ob = (uint32_t)y2r * (uint32_t)P.step;
    g.ca = (uint32_t)(y1 >> 1) * (uint32_t)P.step;
    g.cb = (uint32_t)(y2r >> 1) * (uint32_t)P.step;
    g.wya = (float)y2 - sy;
    g.wyb = sy - (float)y1;
    g.in_y = __builtin_amdgcn_ballot_w64(in);
    return g;
}

// One wave's share of a task: rows row0..row0+3 of plane z, columns col_tile*64 + lane.  K1's arithmetic (k_k1_impl.hpp:
// same geometry, same tap windows, same fp32 expression order, the same program stages) -- bit-identical results.
// ST: 0 = nt dword stores, NOT published safely (A/B upper bound only), 1 = sc1 dword stores, 2 = sc1 16-byte transposed stores,
//     3 = nt 16-byte transposed stores (the tick kernel: an ordinary launch, the kernel boundary publishes)
// TINY: rows narrower than the tap window are gathered byte by byte (wave-uniform branch; the server's submit refuses such crops on the
//       host, the tick kernel cannot see the planes of a device table)
template <int CN, int LD, int ST, int SRC = SRC_U8, bool TINY = false>
__device__ __forceinline__ void k1q_rows(const QTask& t, int z, int col_tile, int row0, int lane, const QRowGeo& geo, int gi) { // geo lane gi + j <-> row0 + j
    constexpr int EB = elem_bytes<SRC>, WINB = 8 * EB;
    constexpr int kAux = (ST == 0 || ST == 3) ? 2 /* nt */ : 16 /* sc1 */;
    const PlaneParams& P = t.P;
    const int dst_w = t.dst_w, dst_h = t.dst_h, W = t.out_w;
    const int x = col_tile * 64 + lane;
    if (row0 >= dst_h) return; // wave-uniform
    const bool live = x < dst_w;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(t.out, 0, (int)t.out_bytes, 0x00020000);
    const uint32_t esh = t.out_half ? 1u : 2u; // log2 of the element size (wave-uniform)
    const uint32_t plane_off = (uint32_t)(((int64_t)z * t.img_stride) << esh); // byte offsets fit 32 bits (checked at submit)
    const uint32_t ch_bytes = (uint32_t)(t.ch_stride << esh);
    // channel k of the value goes to plane k -- or, with the chain's R <-> B swap, 2 - k for k = 0, 2 (wave-uniform)
    const uint32_t ch_off[4] = {t.swap ? 2u * ch_bytes : 0u, ch_bytes, t.swap ? 0u : 2u * ch_bytes, 3u * ch_bytes};
    ProgArgs prog; // registers: only the static program's operands are ever read
    prog.fast_div = t.fast_div;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        prog.operand[0][c] = t.mul[c];
        prog.operand[1][c] = t.sub[c];
        prog.operand[2][c] = t.div[c];
        prog.rdiv[c] = t.rdiv[c];
    }
    auto run_prog = [&](Px& p) {
        int depth = CVGS_DEPTH_32F, cn = CN;
        ProgMulSubDiv::run(prog, p, depth, cn); // (the R <-> B swap is not executed: QTask carries its operands and planes exchanged)
    };
    auto store_rows = [&](const float (&v)[kQRowsPerWave][4]) { // v[j][k]: row j, channel k at this lane's column
        const bool full = col_tile * 64 + 63 < dst_w; // wave-uniform: every lane of the tile is alive
        if (ST >= 2 && full) {
            const int i = lane >> 4, q = lane & 15; // 16 consecutive lanes = one row's 256 contiguous bytes: four lanes per 64-byte request
            const bool row_ok = row0 + i < dst_h;
            const uint32_t off = plane_off + ((uint32_t)((row0 + i) * W + col_tile * 64 + q * 4) << esh);
#pragma unroll
            for (int k = 0; k < CN; ++k) {
                const float r[4] = {v[0][k], v[1][k], v[2][k], v[3][k]};
                q_lds_put(t.tile, k, r, lane);
            }
            __builtin_amdgcn_wave_barrier(); // (compiler ordering only: the hardware runs one wave's LDS operations in order)
#pragma unroll
            for (int k = 0; k < CN; ++k) {
                const q_f32x4 o = q_lds_get(t.tile, k, lane);
                if (t.out_half) { // wave-uniform: four halves, 8 bytes per lane
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    typedef uint32_t u32x2q __attribute__((ext_vector_type(2)));
                    const h2 lo = {(_Float16)o[0], (_Float16)o[1]}, hi = {(_Float16)o[2], (_Float16)o[3]};
                    const u32x2q d = {__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
                    if (row_ok) __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, off + ch_off[k], 0, kAux);
                } else {
                    typedef uint32_t u32x4q __attribute__((ext_vector_type(4)));
                    const u32x4q d = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
                    if (row_ok) __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, off + ch_off[k], 0, kAux);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kQRowsPerWave; ++j) {
                if (row0 + j < dst_h && live) {
                    const uint32_t off = plane_off + ((uint32_t)((row0 + j) * W + x) << esh);
#pragma unroll
                    for (int k = 0; k < CN; ++k) {
                        if (t.out_half) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (_Float16)v[j][k]), rsrc, off + ch_off[k], 0, kAux);
                        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[j][k]), rsrc, off + ch_off[k], 0, kAux);
                    }
                }
            }
        }
    };

    const bool whole = z < t.used && ((P.x1 | P.y1 | (P.x2 ^ (dst_w - 1)) | (P.y2 ^ (dst_h - 1))) == 0);
    Px bgp;
    bgp.v[0] = bgp.v[1] = bgp.v[2] = bgp.v[3] = 0.f;
    if (!whole) { // the background value through the whole chain: planes >= usedPlanes and aspect-ratio padding
#pragma unroll
        for (int k = 0; k < 4; ++k) bgp.v[k] = t.bg[k];
        run_prog(bgp);
        if (z >= t.used) {
            float v[kQRowsPerWave][4];
#pragma unroll
            for (int j = 0; j < kQRowsPerWave; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[j][k] = bgp.v[k];
            store_rows(v);
            return;
        }
    }
    // ---- per-lane column geometry ----
    const int xc = live ? x : dst_w - 1; // dead lanes of a ragged tile compute the last column (never stored)
    const bool in_x = xc >= P.x1 && xc <= P.x2;
    const int xr = in_x ? xc - P.x1 : 0;
    const float sx = (float)xr * P.fx;
    const int x1 = (int)floorf(sx);
    const int x2 = x1 + 1;
    const float wxa = (float)x2 - sx;
    const float wxb = sx - (float)x1;
    const bool edge = x2 > P.w - 1;
    const int row_bytes = P.w * CN * EB;
    const int o = x1 * CN * EB;
    const bool tiny = TINY && row_bytes < WINB; // wave-uniform
    const uint32_t ol = (uint32_t)(tiny ? o : min(o, row_bytes - WINB));
    const int sh = (o - (int)ol) * 8;
    const gptr_u8 src = (gptr_u8)P.data;

    Win<EB> va[kQRowsPerWave], vb[kQRowsPerWave];
    float wya[kQRowsPerWave], wyb[kQRowsPerWave];
    bool in_y[kQRowsPerWave];
#pragma unroll
    for (int j = 0; j < kQRowsPerWave; ++j) {
        in_y[j] = (geo.in_y >> (gi + j)) & 1;
        wya[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geo.wya), gi + j));
        wyb[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geo.wyb), gi + j));
        // (a uniform 64-bit base + a 32-bit lane offset: one v_readlane and one v_add per load instead of a 64-bit scalar multiply-add
        //  and a 64-bit vector add -- issuing a group's loads took a worker 0.75 us of dependent scalar arithmetic)
        const uint32_t oa = (uint32_t)__builtin_amdgcn_readlane((int)geo.oa, gi + j) + ol, ob = (uint32_t)__builtin_amdgcn_readlane((int)geo.ob, gi + j) + ol;
        const gptr_u8 ra = pin_uniform(src) + oa, rb = pin_uniform(src) + ob;
        if (tiny) { // (ra / rb = the row's first byte + o: gather_win clamps o + k into the row)
            va[j] = gather_win<CN, EB>(ra - o, o, row_bytes);
            vb[j] = gather_win<CN, EB>(rb - o, o, row_bytes);
        } else if constexpr (EB == 1) { // (rows narrower than the tap window never reach the server: queue_submit refuses them)
            va[j] = q_load_win<LD>(ra);
            vb[j] = q_load_win<LD>(rb);
        } else {
            va[j] = q_load_win16<LD>(ra);
            vb[j] = q_load_win16<LD>(rb);
        }
    }
    float outv[kQRowsPerWave][4];
#pragma unroll
    for (int j = 0; j < kQRowsPerWave; ++j) {
        float p00[4], p10[4], p01[4], p11[4];
        unpack_pair<CN, SRC>(shift_win<EB>(va[j], sh), edge, p00, p10);
        unpack_pair<CN, SRC>(shift_win<EB>(vb[j], sh), edge, p01, p11);
        const float w00 = wxa * wya[j];
        const float w10 = wxb * wya[j];
        const float w01 = wxa * wyb[j];
        const float w11 = wxb * wyb[j];
        Px p;
        p.v[0] = p.v[1] = p.v[2] = p.v[3] = 0.f;
#pragma unroll
        for (int k = 0; k < CN; ++k) {
            float acc = p00[k] * w00;
            acc = acc + p10[k] * w10;
            acc = acc + p01[k] * w01;
            acc = acc + p11[k] * w11;
            p.v[k] = acc;
        }
        run_prog(p);
        const bool take = whole || (in_x && in_y[j]);
#pragma unroll
        for (int k = 0; k < 4; ++k) outv[j][k] = take ? p.v[k] : bgp.v[k];
    }
    store_rows(outv);
}

} // namespace cvgs
