// k_queue.hip -- the device-side descriptor queue: K1 batches WITHOUT one serialised launch per batch.
//
// Why (DESIGN.md 4 "the 50-crop floor"): a 50-crop K1 launch moves 9 MB behind a ~1.8 us launch/drain
// boundary, and all 6400 waves of one launch are in the same phase at the same time -- the load burst, then the store burst.
// 16 frames fused into one launch run at 2.4 us per frame, one launch per frame at 4.4 us.  The reference's call shape is one
// executeOperations per frame (include/cvGPUSpeedup.cuh:464-473); keeping that shape while removing the boundary needs a
// different submission path: a resident "server" grid that takes batch descriptors from a ring and lets batch k+1's loads
// overlap batch k's stores, because its waves move from batch to batch without any grid-wide barrier.
//
// Protocol (hand-offs follow the guide's recipe R1: write-through payload, drained, then ONE flag word; pollers use relaxed
// loads; nothing depends on dispatch order or XCD placement):
//   host    cvgs_queue_submit lowers the chain (cvgs_execute's validation and double-precision geometry) and writes the batch's
//           slot + its index entry STRAIGHT INTO DEVICE MEMORY through the PCIe BAR (write-combined stores, ~0.2 us per slot;
//           tools/probes/bar_probe.cpp -- probed at create, staged through a copy kernel where the BAR is not mapped), then
//           the queue's tail word.  The control block / ring / index live in UNCACHED device memory, so no device cache can
//           hold a stale copy of what the host rewrites.
//   worker  4 independent waves per workgroup, 4 G in all.  A task = R output rows x 64 columns of one crop in a cumulative task
//           numbering (R = 4 / 16 / 64 / 128 by queue depth); a worker HOLDS one task number (worker w starts with T = w) and draws its
//           next one from the ticket counter of its residue class (16 counters, T = w mod 16) when it has finished a task: work
//           goes to whoever is free, oldest first.  Batches are consecutive task ranges, so they interleave over the whole chip
//           and batch k+1 starts while batch k's stores drain.  A worker
//           finds the batch that holds T with ONE wave-wide load of a 64-entry window of the batch index (read together with the
//           tail: a worker only goes to sleep on a tail whose newest batch it has SEEN in the window).  After a task it
//           drains its write-through stores and bumps the batch's arrival counter; the LAST arrival publishes the batch's
//           completion flag (device word for hipStreamWaitValue64, host word for cvgs_queue_wait).
//   janitor (workgroup 0, one wave) retires the grid after `idle_us` without work (Dekker hand-shake with the host on words in
//           HOST memory: PCIe ordering makes the device's state store visible before its re-read of the host's tail), and
//           trips a watchdog when a batch makes no progress -- the server never outlives its work and cannot hang a box.
// Stores are sc1 WRITE-THROUGH 16-byte vectors: the lane = column register layout is transposed through a wave-private LDS tile
// (q_lds_put / q_lds_get) so that a lane owns 4 consecutive columns of ONE row -- a dword-per-lane sc1 store is one fabric write per lane
// (MI355X_MICROARCH.md "stores of each flavour").  Tap loads are sc1 too: the server outlives kernel boundaries, so its
// L1 / L2 never see the invalidate a kernel start performs; agent-coherent loads never serve a stale copy of a source buffer
// the caller has rewritten between two submits (tests/test_gpu_queue.py rewrites one).
//
// Four kinds of batches, one server instantiation each (QK_*; a queue's first submit decides): crops of 8UC3 / 8UC4 frames
// (k1q_rows: K1's arithmetic -- the headline), of 16UC3 / 16UC4 / 16SC3 / 16SC4 frames (the same worker on 16-byte windows), of
// NV12 / NV21 decoder surfaces (k4q_rows: K4's arithmetic -- BASELINE cfg #3 and the decode-side 50-crop batch: 8.0 -> 4.9 us and
// 4.7 -> 2.2 us per frame, tools/bench_more.py) and of P010 decoder surfaces (k4q_rows<S16>: cfg #3's 10-bit sibling 10.7 -> 6.2 us).
#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define CVGS_HOST_X86 1
#else
#define CVGS_HOST_X86 0
#endif

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "k_taps.hpp"

namespace cvgs {

// an idle worker's pause between two polls of the tail word, in units of 64 clocks (A/B: tools/probes/queue_idle_sleep_ab.sh)
#ifndef CVGS_QUEUE_IDLE_SLEEP
#define CVGS_QUEUE_IDLE_SLEEP 32
#endif
constexpr int kQSlotBytes = 4096;   // one batch: 256 B of parameters, planes from byte 512
constexpr int kQPlanesOff = 512;
constexpr int kQMaxPlanes = (kQSlotBytes - kQPlanesOff) / (int)sizeof(PlaneParams); // 74
constexpr int kQMaxRing = 256;
constexpr int kQRowsPerWave = 4, kQWaves = 4, kQRowsPerTask = 16, kQRowsPerTaskMid = 32, kQRowsPerTaskDeep = 128;
constexpr int kQSubOff = 256;      // 16 cumulative sub-counter targets (8 bytes each) behind the parameters
constexpr int kQSubs = 16;         // arrival sub-counters per slot: task T arrives at sub-counter T % 16
constexpr int kQCtrStride = 16;    // counters are 128 bytes (16 words) apart

// dwords 0..63 of a slot; a worker wave reads them with ONE wave-wide dword load and picks fields with v_readlane
struct QParams {
    uint64_t stamp;     // batch number + 1
    uint64_t task_base; // cumulative index of the batch's first task
    uint32_t n_tasks, tiles_per_plane, col_tiles, n_planes;
    uint32_t used, dst_w, dst_h, out_w;
    uint32_t cn, swap, fast_div, rows_per_task; // rows_per_task: 4 (shallow queue: latency) or 16 (deep queue: throughput)
    float mul[4], sub[4], div[4], rdiv[4], bg[4];
    int64_t img_stride, ch_stride; // output elements
    uint64_t out, out_bytes;
    uint64_t arrive_target;        // value of the slot's TOP arrival counter when the batch's last sub-counter has filled
    uint32_t kind;                 // QK_*: what the planes are
    uint32_t yuv_range, yuv_primaries, yuv_vu; // QK_NV12: cvgs_yuv_range / cvgs_yuv_primaries, V-before-U (NV21)
    uint32_t src_signed;           // QK_PIXELS16: CV_16S pixels (else CV_16U)
    uint32_t out_half;             // the tensor holds CV_16F elements (the chain's trailing convertTo<CV_32F, CV_16F> is the store's conversion)
    uint32_t gated;                // stream-ordered submit: no tap of this batch is loaded before the slot's gate word has reached `stamp`
    uint32_t pad[11];
};
static_assert(sizeof(QParams) == 256, "QParams is one wave-wide dword load");
enum { QD_STAMP = 0, QD_TASK_BASE = 2, QD_N_TASKS = 4, QD_TPP = 5, QD_COL_TILES = 6, QD_N_PLANES = 7, QD_USED = 8, QD_DST_W = 9, QD_DST_H = 10,
       QD_OUT_W = 11, QD_CN = 12, QD_SWAP = 13, QD_FAST_DIV = 14, QD_ROWS_PER_TASK = 15, QD_MUL = 16, QD_SUB = 20, QD_DIV = 24, QD_RDIV = 28, QD_BG = 32,
       QD_IMG_STRIDE = 36, QD_CH_STRIDE = 38, QD_OUT = 40, QD_OUT_BYTES = 42, QD_ARRIVE_TARGET = 44, QD_KIND = 46, QD_YUV_RANGE = 47,
       QD_YUV_PRIM = 48, QD_YUV_VU = 49, QD_SRC_SIGNED = 50, QD_OUT_HALF = 51, QD_GATED = 52 };
// What a queue serves -- latched by its first submit; each kind has its own server instantiation (the 8-bit-pixel worker is the
// tuned headline path and carries nothing of the other's code or registers).
enum { QK_PIXELS = 0 /* 8UC3 / 8UC4 crops (K1's shape) */, QK_NV12 = 1 /* crops of NV12 / NV21 decoder surfaces (K4's shape) */,
       QK_PIXELS16 = 2 /* 16UC3 / 16UC4 / 16SC3 / 16SC4 crops: the other source types of the reference's K1 sweep (test_batchresize_x_split3D.cu:427-432) */,
       QK_P010 = 3 /* crops of P010 decoder surfaces (10-bit, 16-bit samples): K4's S16 shape */ };
constexpr bool q_kind_yuv(int kind) { return kind == QK_NV12 || kind == QK_P010; }

// The batch index: one 32-byte entry per ring slot, rewritten by the host while workers may be looking: every 8-byte word is
// written atomically, and `check` ties the four words together (a torn entry is simply not a candidate).
struct QIndex {
    uint64_t end_task;  // first task index BEYOND the batch (tasks are numbered cumulatively over batches)
    uint64_t stamp;     // batch number + 1 (0 = never written)
    uint32_t n_tasks, tiles_per_plane;
    uint32_t n_planes, check;
};
static_assert(sizeof(QIndex) == 32, "QIndex layout");
__host__ __device__ inline uint32_t q_index_check(uint64_t end_task, uint64_t stamp, uint32_t n_tasks, uint32_t tpp, uint32_t n_planes) {
    uint64_t h = end_task * 0x9E3779B97F4A7C15ull ^ (stamp + 0x632BE59BD9B4E019ull) * 0xD6E8FEB86659FD93ull;
    h ^= ((uint64_t)n_tasks << 32 | tpp) * 0xA24BAED4963EE407ull ^ n_planes;
    return (uint32_t)(h >> 32) ^ (uint32_t)h;
}

struct alignas(128) QLine { // one word per 128-byte line
    uint64_t v;
    uint64_t pad[15];
};
struct QDevCtl {   // UNCACHED device memory; `tail` is written by the host through the BAR (or by the staging kernel)
    QLine tail;     // batches published
    QLine stop_gen; // == the launch's generation: every workgroup of that launch returns
};
enum { QS_IDLE = 0, QS_RUNNING = 1, QS_EXITING = 2, QS_EXITED = 3 };
struct QHostCtl {  // pinned host memory: the words of the retirement hand-shake, errors, instrumentation
    QLine tail;     // host -> janitor: batches submitted (the copy the Dekker hand-shake reads)
    QLine state;    // QS_*
    QLine stop_req; // host -> janitor: destroy
    QLine yield_req; // host -> janitor: another queue of this device wants to launch its server: retire as soon as nothing is in flight
    QLine error;    // device -> host: 1 = stalled (watchdog), 2 = protocol violation
    QLine stat_rounds, stat_launch_ticks;
    uint64_t prof[32]; // instrumentation of the last server (100 MHz ticks / counts), see queue_prof
};

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
// ST: 0 = nt dword stores, NOT published safely (A/B upper bound only), 1 = sc1 dword stores, 2 = sc1 16-byte transposed stores
template <int CN, int LD, int ST, int SRC = SRC_U8>
__device__ __forceinline__ void k1q_rows(const QTask& t, int z, int col_tile, int row0, int lane, const QRowGeo& geo, int gi) { // geo lane gi + j <-> row0 + j
    constexpr int EB = elem_bytes<SRC>, WINB = 8 * EB;
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
        if (ST == 2 && full) {
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
                    if (row_ok) __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, off + ch_off[k], 0, 16 /* sc1 */);
                } else {
                    typedef uint32_t u32x4q __attribute__((ext_vector_type(4)));
                    const u32x4q d = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
                    if (row_ok) __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, off + ch_off[k], 0, 16 /* sc1 */);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kQRowsPerWave; ++j) {
                if (row0 + j < dst_h && live) {
                    const uint32_t off = plane_off + ((uint32_t)((row0 + j) * W + x) << esh);
#pragma unroll
                    for (int k = 0; k < CN; ++k) {
                        if (t.out_half) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (_Float16)v[j][k]), rsrc, off + ch_off[k], 0, ST == 0 ? 2 : 16);
                        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[j][k]), rsrc, off + ch_off[k], 0, ST == 0 ? 2 /* nt */ : 16 /* sc1 */);
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
    const uint32_t ol = (uint32_t)min(o, row_bytes - WINB);
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
        if constexpr (EB == 1) { // (rows narrower than the tap window never reach the server: queue_submit refuses them)
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

// The NV12 worker's share of a task: rows row0..row0+3 of plane z, columns col_tile*64 + lane of a crop of an NV12 / NV21 decoder
// surface.  K4's arithmetic (k_nv12.hip: same geometry, the two luma taps of a row in ONE 2-byte load, the two chroma pairs in ONE
// 4-byte load, the per-tap YCbCr -> RGB conversion k4_tap, the same fp32 expression order and program stages) -- bit-identical results.
typedef uint16_t q_u16_unaligned __attribute__((aligned(1)));
typedef uint32_t q_u32_unaligned __attribute__((aligned(1)));
template <int LD>
__device__ __forceinline__ uint32_t q_load_u16(gptr_u8 p) {
    typedef __attribute__((address_space(1))) uint16_t* g_u16;
    if constexpr (LD == 0) return *(const __attribute__((address_space(1))) q_u16_unaligned*)p;
    else return __hip_atomic_load((g_u16)(__attribute__((address_space(1))) uint8_t*)p, Q_AGENT);
}
template <int LD>
__device__ __forceinline__ uint32_t q_load_u32(gptr_u8 p) {
    if constexpr (LD == 0) return *(const __attribute__((address_space(1))) q_u32_unaligned*)p;
    else return __hip_atomic_load((g_u32)(__attribute__((address_space(1))) uint8_t*)p, Q_AGENT);
}

// S16: P010 surfaces -- NV12's geometry with 16-bit samples (10-bit code = sample >> 6, k_nv12.hip's S16 instantiation): the two luma taps
// of a row are ONE 4-byte load, its two chroma pairs ONE 8-byte load; the samples are picked with shifts (no byte selectors).
template <int LD, int ST, bool S16 = false>
__device__ __forceinline__ void k4q_rows(const QTask& t, int z, int col_tile, int row0, int lane, const QRowGeo& geo, int gi) { // geo lane gi + j <-> row0 + j
    constexpr int CN = 3;
    const PlaneParams& P = t.P;
    const int dst_w = t.dst_w, dst_h = t.dst_h, W = t.out_w;
    const int x = col_tile * 64 + lane;
    if (row0 >= dst_h) return; // wave-uniform
    const bool live = x < dst_w;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(t.out, 0, (int)t.out_bytes, 0x00020000);
    const uint32_t esh = t.out_half ? 1u : 2u; // log2 of the element size (wave-uniform)
    const uint32_t plane_off = (uint32_t)(((int64_t)z * t.img_stride) << esh);
    const uint32_t ch_bytes = (uint32_t)(t.ch_stride << esh);
    // channel k of the value goes to plane k -- or, with the chain's R <-> B swap, 2 - k for k = 0, 2 (wave-uniform)
    const uint32_t ch_off[4] = {t.swap ? 2u * ch_bytes : 0u, ch_bytes, t.swap ? 0u : 2u * ch_bytes, 3u * ch_bytes};
    ProgArgs prog;
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
    auto store_rows = [&](const float (&v)[kQRowsPerWave][4]) { // as k1q_rows
        const bool full = col_tile * 64 + 63 < dst_w;
        if (ST == 2 && full) {
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
                    if (row_ok) __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, off + ch_off[k], 0, 16 /* sc1 */);
                } else {
                    typedef uint32_t u32x4q __attribute__((ext_vector_type(4)));
                    const u32x4q d = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
                    if (row_ok) __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, off + ch_off[k], 0, 16 /* sc1 */);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kQRowsPerWave; ++j) {
                if (row0 + j < dst_h && live) {
                    const uint32_t off = plane_off + ((uint32_t)((row0 + j) * W + x) << esh);
#pragma unroll
                    for (int k = 0; k < CN; ++k) {
                        if (t.out_half) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (_Float16)v[j][k]), rsrc, off + ch_off[k], 0, ST == 0 ? 2 : 16);
                        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[j][k]), rsrc, off + ch_off[k], 0, ST == 0 ? 2 /* nt */ : 16 /* sc1 */);
                    }
                }
            }
        }
    };

    const bool whole = z < t.used && ((P.x1 | P.y1 | (P.x2 ^ (dst_w - 1)) | (P.y2 ^ (dst_h - 1))) == 0);
    Px bgp;
    bgp.v[0] = bgp.v[1] = bgp.v[2] = bgp.v[3] = 0.f;
    if (!whole) { // the background value through the whole chain: planes >= usedPlanes and letterbox padding
#pragma unroll
        for (int k = 0; k < 4; ++k) bgp.v[k] = k < CN ? t.bg[k] : 0.f;
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
    const YuvK yk = yuv_matrix(t.yuv_range, t.yuv_primaries, S16 ? CVGS_YUV_P010 : CVGS_YUV_NV12);
    // ---- per-lane column geometry (k4_nv12_resize's) ----
    const int xc = live ? x : dst_w - 1;
    const bool in_x = xc >= P.x1 && xc <= P.x2;
    const int xr = in_x ? xc - P.x1 : 0;
    const float sx = (float)xr * P.fx;
    const int x1 = (int)floorf(sx);
    const int x2 = x1 + 1;
    const float wxa = (float)x2 - sx, wxb = sx - (float)x1;
    const bool edge = x2 > P.w - 1;
    const int x2r = edge ? x1 : x2;
    constexpr uint32_t kSB = S16 ? 2 : 1; // bytes per sample
    const uint32_t yo = (uint32_t)min(x1, P.w - 2) * kSB;
    const int c1 = x1 >> 1, c2 = x2r >> 1;
    const uint32_t uo = (uint32_t)min(2 * c1, P.w - 4) * kSB;
    const bool same_pair = c2 == c1;
    [[maybe_unused]] const uint32_t ysh = ((uint32_t)x1 * kSB - yo) * 8, ush = ((uint32_t)(2 * c1) * kSB - uo) * 8; // S16: 0 / 16 and 0 / 32
    // The four taps' samples as bytes of three dwords -- luma {a0, a1, b0, b1} (row a / b, tap 0 / 1), U and V likewise -- picked by
    // v_perm_b32 out of the two 2-byte luma loads and the two 4-byte chroma loads; the selectors hold what was a shift, a mask and
    // a select per sample: the window clamped back at the right edge (edge <=> the 2-byte luma window starts one pixel early and both
    // taps are its second byte; the last chroma pair <=> the 4-byte window starts one pair early), taps that share a chroma pair,
    // and NV21's byte order (wave-uniform).  Then one v_cvt_f32_ubyteN per sample.
    const uint32_t sel_y = edge ? 0x05050101u : 0x05040100u;
    const uint32_t pr0 = (uint32_t)(2 * c1) * kSB != uo ? 2u : 0u;   // byte of tap 0's pair inside the chroma window
    const uint32_t pr1 = same_pair ? pr0 : 2u;          // ... of tap 1's
    const uint32_t sel_c = pr0 | (pr1 << 8) | ((4u + pr0) << 16) | ((4u + pr1) << 24);
    const uint32_t sel_u = sel_c + (t.yuv_vu ? 0x01010101u : 0u), sel_v = sel_c + (t.yuv_vu ? 0u : 0x01010101u);
    const gptr_u8 base = (gptr_u8)P.data;
    const gptr_u8 uvp = base + (size_t)P.uv_off;

    using ChromaWin = std::conditional_t<S16, uint64_t, uint32_t>; // two (U,V) pairs
    uint32_t vya[kQRowsPerWave], vyb[kQRowsPerWave];
    ChromaWin vua[kQRowsPerWave], vub[kQRowsPerWave];
    float wya[kQRowsPerWave], wyb[kQRowsPerWave];
    bool in_y[kQRowsPerWave];
#pragma unroll
    for (int j = 0; j < kQRowsPerWave; ++j) {
        in_y[j] = (geo.in_y >> (gi + j)) & 1;
        wya[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geo.wya), gi + j));
        wyb[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(geo.wyb), gi + j));
        const uint32_t oa = (uint32_t)__builtin_amdgcn_readlane((int)geo.oa, gi + j) + yo, ob = (uint32_t)__builtin_amdgcn_readlane((int)geo.ob, gi + j) + yo;
        const uint32_t ca = (uint32_t)__builtin_amdgcn_readlane((int)geo.ca, gi + j) + uo, cb = (uint32_t)__builtin_amdgcn_readlane((int)geo.cb, gi + j) + uo;
        if constexpr (S16) {
            vya[j] = q_load_u32<LD>(pin_uniform(base) + oa);
            vyb[j] = q_load_u32<LD>(pin_uniform(base) + ob);
            vua[j] = q_load_win<LD>(pin_uniform(uvp) + ca).lo;
            vub[j] = q_load_win<LD>(pin_uniform(uvp) + cb).lo;
        } else {
            vya[j] = q_load_u16<LD>(pin_uniform(base) + oa); // (uniform base + 32-bit lane offset, as k1q_rows)
            vyb[j] = q_load_u16<LD>(pin_uniform(base) + ob);
            vua[j] = q_load_u32<LD>(pin_uniform(uvp) + ca);
            vub[j] = q_load_u32<LD>(pin_uniform(uvp) + cb);
        }
    }
    float outv[kQRowsPerWave][4];
#pragma unroll
    for (int j = 0; j < kQRowsPerWave; ++j) {
        float fy[4], fu[4], fv[4]; // taps 00, 10, 01, 11
        if constexpr (S16) { // k4_nv12_resize's S16 picks: the window shifted to tap 0, tap 1 = the next sample / pair unless clamped
            const uint32_t ya0 = (vya[j] >> ysh) & 0xffffu, ya1 = edge ? ya0 : vya[j] >> 16;
            const uint32_t yb0 = (vyb[j] >> ysh) & 0xffffu, yb1 = edge ? yb0 : vyb[j] >> 16;
            const uint32_t pa0 = (uint32_t)(vua[j] >> ush), pa1 = same_pair ? pa0 : (uint32_t)(vua[j] >> 32);
            const uint32_t pb0 = (uint32_t)(vub[j] >> ush), pb1 = same_pair ? pb0 : (uint32_t)(vub[j] >> 32);
            fy[0] = (float)(ya0 >> 6); fy[1] = (float)(ya1 >> 6); fy[2] = (float)(yb0 >> 6); fy[3] = (float)(yb1 >> 6);
            fu[0] = (float)((pa0 & 0xffffu) >> 6); fu[1] = (float)((pa1 & 0xffffu) >> 6);
            fu[2] = (float)((pb0 & 0xffffu) >> 6); fu[3] = (float)((pb1 & 0xffffu) >> 6);
            fv[0] = (float)(pa0 >> 22); fv[1] = (float)(pa1 >> 22); fv[2] = (float)(pb0 >> 22); fv[3] = (float)(pb1 >> 22);
        } else {
            const uint32_t ly = __builtin_amdgcn_perm(vyb[j], vya[j], sel_y);
            const uint32_t lu = __builtin_amdgcn_perm(vub[j], vua[j], sel_u), lv = __builtin_amdgcn_perm(vub[j], vua[j], sel_v);
            fy[0] = (float)(ly & 0xffu); fy[1] = (float)((ly >> 8) & 0xffu); fy[2] = (float)((ly >> 16) & 0xffu); fy[3] = (float)(ly >> 24);
            fu[0] = (float)(lu & 0xffu); fu[1] = (float)((lu >> 8) & 0xffu); fu[2] = (float)((lu >> 16) & 0xffu); fu[3] = (float)(lu >> 24);
            fv[0] = (float)(lv & 0xffu); fv[1] = (float)((lv >> 8) & 0xffu); fv[2] = (float)((lv >> 16) & 0xffu); fv[3] = (float)(lv >> 24);
        }
        float t00[4], t10[4], t01[4], t11[4];
        if (t.yuv_range == CVGS_YUV_FULL) { // wave-uniform
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
        run_prog(p);
        const bool take = whole || (in_x && in_y[j]);
#pragma unroll
        for (int k = 0; k < 4; ++k) outv[j][k] = take ? p.v[k] : bgp.v[k];
    }
    store_rows(outv);
}

__device__ __forceinline__ uint64_t q_wave_min(uint64_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ uint64_t q_wave_max(uint64_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ uint64_t q_bcast_u64(uint64_t v, int src_lane) {
    return (uint64_t)(uint32_t)__shfl((int)(uint32_t)v, src_lane) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src_lane) << 32);
}

// worker-side instrumentation costs ~10 SGPRs in a kernel that has none to spare: compiled in with -DCVGS_QUEUE_PROFILE only
#ifdef CVGS_QUEUE_PROFILE
#define QPROF(...) __VA_ARGS__
#else
#define QPROF(...)
#endif


// ---- workgroup 0's janitor wave: in-order completion count (for retirement / the watchdog only), retirement --------------
__device__ void k1q_janitor(QDevCtl* dc, QHostCtl* hc, const uint64_t* dflags, const uint8_t* ring, const uint64_t* gates, uint32_t R, uint64_t gen,
                            uint64_t idle_ticks, uint64_t stall_ticks, uint64_t gate_ticks, uint64_t done) {
    const int lane = (int)threadIdx.x;
    const uint64_t t_launch = wall_clock64();
    uint64_t t_last = t_launch, rounds = 0, gate_since = 0;
    int why = -1; // 0 = idle retirement, 1 = stall, 3 = destroy, 4 = a stream-ordered batch whose gate never opened
    while (why < 0) {
        ++rounds;
        const uint64_t tail = q_ldu_sys(&dc->tail.v);
        if (q_ldu_sys(&hc->stop_req.v)) {
            why = 3;
            break;
        }
        if (q_ldu_sys(&hc->error.v)) {
            why = 2;
            break;
        }
        if (done < tail) {
            // lane i looks at batch done + i: how many of the oldest in-flight batches have their flag up?
            const uint64_t bi = done + (uint64_t)lane;
            const uint64_t f = bi < tail && (uint32_t)lane < R ? q_ld_sys(dflags + kQCtrStride * (bi % R)) : 0;
            const uint64_t up = __builtin_amdgcn_ballot_w64(f >= bi + 1);
            const int n = up == ~0ull ? 64 : __builtin_ctzll(~up);
            if (n > 0) {
                done += (uint64_t)n;
                t_last = wall_clock64();
                gate_since = 0;
            } else if (wall_clock64() - t_last > stall_ticks) {
                // No progress -- but the oldest open batch may be a stream-ordered one whose producer (the work in front of it on the
                // caller's stream) has not finished: its workers are WAITING at the gate, nothing has stalled.  That wait has its own,
                // much longer limit (a producer that never finishes -- a destroyed stream, a failed launch -- must not keep the
                // server alive for ever: hipDeviceSynchronize waits for it).
                const uint8_t* slot = ring + (size_t)(done % R) * kQSlotBytes;
                const uint64_t stamp = q_ldu_sys((const uint64_t*)slot);
                const uint32_t gated = q_uni((uint32_t)__hip_atomic_load((g_u32)((const uint32_t*)slot + QD_GATED), Q_SYSTEM));
                const bool closed = stamp == done + 1 && gated != 0 && q_ldu_sys(gates + kQCtrStride * (done % R)) < done + 1;
                const uint64_t now = wall_clock64();
                if (!closed) why = 1;
                else if (gate_since == 0) gate_since = t_last, t_last = now;
                else if (now - gate_since > gate_ticks) why = 4;
                else t_last = now;
            }
            __builtin_amdgcn_s_sleep(16);
        } else if (wall_clock64() - t_last > idle_ticks || q_ldu_sys(&hc->yield_req.v)) {
            // retire (idle, or asked to make room for another queue's server): announce, then look at the host's tail once more (the host publishes its tail, fences, then reads state;
            // the read below is a PCIe read and cannot pass the posted state write)
            if (lane == 0) q_st_sys(&hc->state.v, QS_EXITING);
            q_drain();
            if (q_ldu_sys(&hc->tail.v) != tail) {
                if (lane == 0) q_st_sys(&hc->state.v, QS_RUNNING);
                q_drain();
                t_last = wall_clock64();
            } else {
                why = 0;
            }
        } else {
            __builtin_amdgcn_s_sleep(32);
        }
    }
    if (lane == 0) {
        q_st_sys(&dc->stop_gen.v, gen);
        if (why == 1) q_st_sys(&hc->error.v, 1);
        if (why == 4) q_st_sys(&hc->error.v, 3);
        q_st_sys(&hc->stat_rounds.v, rounds);
        q_st_sys(&hc->stat_launch_ticks.v, wall_clock64() - t_launch);
        q_drain();
        q_st_sys(&hc->state.v, QS_EXITED);
    }
}

// ---- the server grid -----------------------------------------------------------------------------------------------
// Workgroup 0: the janitor.  Workgroups 1..G: four INDEPENDENT worker waves each.  No barrier; a wave's shared writes are its
// write-through output rows, one returning atomic on its batch's arrival counter, its resume word, and -- the last arrival
// of a batch -- the batch's two completion flags.
struct QDevMem { // the server's device-side state, one uncached allocation (host-writable through the BAR)
    QDevCtl* dc;
    uint8_t* ring;      // R slots
    QIndex* index;      // R entries
    uint64_t* arrive;   // ordinary (cached) device memory: per slot 1 top + 16 sub arrival counters, 128 bytes apart; device atomics only
    uint64_t* dflags;   // R completion flags, 128 bytes apart (hipStreamWaitValue64 targets)
    uint64_t* gates;    // R gate words, 128 bytes apart (uncached): batch b of a stream-ordered submit may be read from once gates[b % R] >= b + 1
    uint64_t* prog;     // 4 G resume words: the ticket (task number) each worker holds
    uint64_t* ticket;   // 16 ticket counters (one per residue class of the task numbering), 128 bytes apart
};

// (register budget: 4 waves per SIMD -- 128 VGPRs -- for the 8-bit pixel worker (103) and, since its tap samples are picked by v_perm_b32 and
// its row geometry lives in lanes, for the NV12 worker too (125, no scratch: tests/test_kernel_resources.py); the 16-bit worker holds
// 16-byte windows and gets 3 waves per SIMD -- 168 VGPRs -- which is what the default 3 workgroups per CU use anyway)
template <int LD, int ST, int KIND = QK_PIXELS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KIND == QK_PIXELS16 || KIND == QK_P010 ? 3 : 4, 8))) void k1q_server(QDevMem m, QHostCtl* hc, uint64_t* hflags, uint32_t R, uint32_t G,
                                                                                             uint64_t gen, uint64_t done0, uint64_t idle_ticks,
                                                                                             uint64_t stall_ticks, uint64_t gate_ticks) {
    __shared__ __attribute__((aligned(16))) float q_tiles[kQWaves * kQLdsWave]; // one transpose tile per wave (20 KB per workgroup)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t n_workers = G * kQWaves;
    if (blockIdx.x == 0) {
        if (wave == 0) k1q_janitor(m.dc, hc, m.dflags, m.ring, m.gates, R, gen, idle_ticks, stall_ticks, gate_ticks, done0);
        return;
    }
    const uint32_t wid = (blockIdx.x - 1) * kQWaves + (uint32_t)wave;
    uint64_t T = q_ldu(m.prog + wid); // the ticket this worker holds (a resumed server: the one its predecessor held)
    const uint32_t cls = wid & (kQSubs - 1);                                   // this worker's tickets are T = cls (mod 16)
    const uint64_t cls_first = (n_workers - cls + kQSubs - 1) / kQSubs;        // tickets cls, cls + 16, ... < 4G were handed out at create
    uint64_t b = done0;                   // every batch below was complete when this server was launched
    // the NEXT task's index window, requested before the current task's rows and consumed after them: with a deep queue the
    // dispatch round trip disappears behind the arithmetic (loads the compiler counts: it waits where they are first used)
    bool pre_valid = false;
    uint64_t pre_wb = 0, pre_tail = 0, pre_e[4] = {0, 0, 0, 0};
    QPROF(uint64_t p_slot = 0, p_iters = 0; uint64_t p_tasks = 0, p_find = 0, p_rows = 0, p_drain = 0, p_idle = 0, p_arrive = 0, p_t4 = 0, p_first = 0, p_last = 0;)
    for (;;) {
        QPROF(const uint64_t p_t0 = wall_clock64();)
        // ---- find the batch that holds task T: the tail + a 64-entry window of the batch index, ONE round trip; then the
        // batch's parameters and the crop's, ONE more.  A worker only reads the slot of the batch that holds its own unprocessed
        // task -- that batch cannot complete, so its slot cannot be recycled under the read; index entries say which batch they
        // describe and carry a check word, so a recycled / half-written / never-written entry is simply not a candidate.
        uint32_t v, pv;   // the batch's 64 parameter dwords / the crop's 12, one per lane (fields by v_readlane)
        uint64_t tb, te;  // the batch's task range
        uint64_t sub_target; // value of my arrival sub-counter once the batch's last task of my residue class has arrived
        uint64_t tail_seen = ~0ull; // the tail at the last window read that found nothing
        for (;;) {
            if (tail_seen != ~0ull) { // idle: poll the 8-byte tail only; the 2 KB window is read again once it has moved
                if (q_ldu_sys(&m.dc->tail.v) == tail_seen) {
                    if (q_ldu_sys(&m.dc->stop_gen.v) == gen) {
                        QPROF(if (wid == 0 && lane == 0) {
                            q_st_sys(&hc->prof[0], p_slot);
                            q_st_sys(&hc->prof[1], p_iters);
                            q_st_sys(&hc->prof[8], p_tasks);
                            q_st_sys(&hc->prof[9], p_find);
                            q_st_sys(&hc->prof[10], p_rows);
                            q_st_sys(&hc->prof[11], p_drain);
                            q_st_sys(&hc->prof[12], p_idle);
                            q_st_sys(&hc->prof[5], p_arrive);
                            q_st_sys(&hc->prof[4], p_last - p_first);
                        })
                        return;
                    }
                    __builtin_amdgcn_s_sleep(CVGS_QUEUE_IDLE_SLEEP);
                    QPROF(p_idle += 1;)
                    continue;
                }
            }
            QPROF(++p_iters;)
            uint64_t wb, tail_c; // window base, the tail that came with the window
            q_u64x2 e0, e1;
            if (pre_valid) { // the window this wave asked for BEFORE its previous task's rows: it has long landed
                pre_valid = false;
                wb = pre_wb;
                tail_c = pre_tail;
                e0.x = pre_e[0];
                e0.y = pre_e[1];
                e1.x = pre_e[2];
                e1.y = pre_e[3];
            } else {
                wb = b;
                const uint64_t* ep = (const uint64_t*)(m.index + ((wb + (uint64_t)lane) % R));
                tail_c = q_ld_sys(&m.dc->tail.v);
                e0.x = q_ld_sys(ep);
                e0.y = q_ld_sys(ep + 1);
                e1.x = q_ld_sys(ep + 2);
                e1.y = q_ld_sys(ep + 3);
            }
            const uint64_t tail_u = q_uni(tail_c);
            const uint64_t cand = e0.y - 1; // the batch this entry describes
            const bool valid = e0.y != 0 && cand >= wb && cand < tail_u && cand < wb + R &&
                               (uint32_t)(e1.y >> 32) == q_index_check(e0.x, e0.y, (uint32_t)e1.x, (uint32_t)(e1.x >> 32), (uint32_t)e1.y);
            const bool hit = valid && e0.x > T;
            // (the tail and the window come back in one round trip, in no particular order: the window may predate a batch the
            // tail already counts.  Only entries actually SEEN move b: a batch that ends at or below T, and all before it, are behind.)
            const uint64_t passed = q_uni(q_wave_max(valid && e0.x <= T ? cand + 1 : 0));
            if (passed > b) b = passed;
            if (__builtin_amdgcn_ballot_w64(hit) == 0) {
                // Nothing for this worker in the window.  Before it goes to sleep on the tail it has just read: was the window as NEW as
                // that tail?  The two came back from one round trip in no particular order, so the tail may already count a batch whose
                // index entry the window read predates (the host writes the entry, fences, then the tail, ~100 ns apart).  A worker that
                // then waits for the tail to move waits forever if that batch is the last one for a while: ONE task of 96 missing, the
                // watchdog's 250 ms (tools/probes/queue_relaunch_stress.py: every run).  The newest batch the tail counts must have been
                // seen in the window, or the pair is read again.
                const bool newest_seen = tail_u <= wb || tail_u > wb + 64 || __builtin_amdgcn_ballot_w64(valid && cand == tail_u - 1) != 0;
                tail_seen = (passed > wb || !newest_seen) ? ~0ull : tail_u; // progress / a stale window: look again at once; else wait for the tail to move
                if (tail_seen == ~0ull && q_ldu_sys(&m.dc->stop_gen.v) == gen) return;
                continue;
            }
            // the EARLIEST candidate (lanes are not sorted by batch once the window wraps the ring)
            const uint64_t best = q_uni(q_wave_min(hit ? cand : ~0ull));
            const int src = __builtin_ctzll(__builtin_amdgcn_ballot_w64(hit && cand == best));
            te = q_bcast_u64(e0.x, src);
            const uint64_t e1x = q_bcast_u64(e1.x, src), e1y = q_bcast_u64(e1.y, src);
            tb = te - (uint32_t)e1x;
            if (T < tb) { // T belongs to an earlier batch whose entry this window read predates: read again
                tail_seen = ~0ull;
                continue;
            }
            const uint32_t tpp = (uint32_t)(e1x >> 32), n_planes = (uint32_t)e1y;
            const uint32_t z = (uint32_t)(T - tb) / tpp;
            const uint32_t pz = z < n_planes ? z : n_planes - 1;
            const uint8_t* slot = m.ring + (size_t)(best % R) * kQSlotBytes;
            QPROF(const uint64_t p_s0 = wall_clock64();)
            v = __hip_atomic_load((g_u32)((const uint32_t*)slot + lane), Q_SYSTEM);
            pv = __hip_atomic_load((g_u32)((const uint32_t*)(slot + kQPlanesOff + (size_t)pz * sizeof(PlaneParams)) + (lane < 12 ? lane : 0)), Q_SYSTEM);
            sub_target = q_ld_sys((const uint64_t*)(slot + kQSubOff) + (T & (kQSubs - 1)));
            uint64_t gate = q_ld_sys(m.gates + kQCtrStride * (best % R)); // (the same round trip as the parameters)
            q_drain();
            sub_target = q_uni(sub_target);
            QPROF(p_slot += wall_clock64() - p_s0;)
            if (q_lane_u64(v, QD_STAMP) != best + 1 || q_lane_u64(v, QD_TASK_BASE) != tb) { // a check-word collision: read again
                tail_seen = ~0ull;
                continue;
            }
            b = best;
            // Stream-ordered batch (cvgs_queue_submit_on): its sources belong to work that is still in front of the gate kernel on the
            // caller's stream.  No tap is loaded before the gate word says that stream has got there; the gate kernel's store follows a
            // kernel boundary on that stream (the producer's release), the tap loads below are agent-coherent (sc1).
            if (q_lane_u32(v, QD_GATED) != 0) {
                gate = q_uni(gate);
                // (back-off: a closed gate can have a thousand workers in front of it, all reading ONE uncached word; A/B'd against a fixed
                //  0.25 us interval: no measurable difference either way -- kept short of a microsecond)
                int nap = 0;
                while (gate < best + 1) {
                    if (q_ldu_sys(&m.dc->stop_gen.v) == gen) return; // (the ticket stays in this worker's resume word)
                    if (nap == 0) __builtin_amdgcn_s_sleep(8);
                    else if (nap == 1) __builtin_amdgcn_s_sleep(16);
                    else __builtin_amdgcn_s_sleep(32); // ~1 us: what a gate that opens late adds to its batch's latency at most
                    ++nap;
                    gate = q_ldu_sys(m.gates + kQCtrStride * (best % R));
                }
            }
            break;
        }
        QPROF(const uint64_t p_t1 = wall_clock64();)
        {
            const uint32_t local = (uint32_t)(T - tb);
            const uint32_t tpp = q_lane_u32(v, QD_TPP), col_tiles = q_lane_u32(v, QD_COL_TILES);
            const uint32_t z = local / tpp, rt = local - z * tpp;
            const uint32_t row_tile = rt / col_tiles, col_tile = rt - row_tile * col_tiles;
            QTask t;
            t.tile = q_tiles + wave * kQLdsWave;
            t.P.data = (const uint8_t*)q_lane_u64(pv, 0);
            t.P.w = (int)q_lane_u32(pv, 2);
            t.P.h = (int)q_lane_u32(pv, 3);
            t.P.step = (int)q_lane_u32(pv, 4);
            t.P.fx = q_lane_f32(pv, 5);
            t.P.fy = q_lane_f32(pv, 6);
            t.P.x1 = (int)q_lane_u32(pv, 7);
            t.P.y1 = (int)q_lane_u32(pv, 8);
            t.P.x2 = (int)q_lane_u32(pv, 9);
            t.P.y2 = (int)q_lane_u32(pv, 10);
            t.P.uv_off = q_kind_yuv(KIND) ? (int)q_lane_u32(pv, 11) : 0;
            if constexpr (q_kind_yuv(KIND)) {
                t.yuv_range = (int)q_lane_u32(v, QD_YUV_RANGE);
                t.yuv_primaries = (int)q_lane_u32(v, QD_YUV_PRIM);
                t.yuv_vu = (int)q_lane_u32(v, QD_YUV_VU);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t.mul[c] = q_lane_f32(v, QD_MUL + c);
                t.sub[c] = q_lane_f32(v, QD_SUB + c);
                t.div[c] = q_lane_f32(v, QD_DIV + c);
                t.rdiv[c] = q_lane_f32(v, QD_RDIV + c);
                t.bg[c] = q_lane_f32(v, QD_BG + c);
            }
            t.used = (int)q_lane_u32(v, QD_USED);
            t.dst_w = (int)q_lane_u32(v, QD_DST_W);
            t.dst_h = (int)q_lane_u32(v, QD_DST_H);
            t.out_w = (int)q_lane_u32(v, QD_OUT_W);
            t.swap = (int)q_lane_u32(v, QD_SWAP);
            if (t.swap) { // swap, then per-channel mul / sub / div == per-channel stages with operands 0 <-> 2 exchanged, channel k stored in plane 2 - k
                float x;
                x = t.mul[0]; t.mul[0] = t.mul[2]; t.mul[2] = x;
                x = t.sub[0]; t.sub[0] = t.sub[2]; t.sub[2] = x;
                x = t.div[0]; t.div[0] = t.div[2]; t.div[2] = x;
                x = t.rdiv[0]; t.rdiv[0] = t.rdiv[2]; t.rdiv[2] = x;
            }
            t.fast_div = (int)q_lane_u32(v, QD_FAST_DIV);
            t.img_stride = (int64_t)q_lane_u64(v, QD_IMG_STRIDE);
            t.ch_stride = (int64_t)q_lane_u64(v, QD_CH_STRIDE);
            t.out = (uint8_t*)q_lane_u64(v, QD_OUT);
            t.out_bytes = (uint32_t)q_lane_u64(v, QD_OUT_BYTES);
            t.out_half = (int)q_lane_u32(v, QD_OUT_HALF);
            [[maybe_unused]] const bool c3 = q_lane_u32(v, QD_CN) == 3;
            {
                pre_wb = b;
                const uint64_t* ep = (const uint64_t*)(m.index + ((pre_wb + (uint64_t)lane) % R));
                pre_tail = q_ld_sys(&m.dc->tail.v);
                pre_e[0] = q_ld_sys(ep);
                pre_e[1] = q_ld_sys(ep + 1);
                pre_e[2] = q_ld_sys(ep + 2);
                pre_e[3] = q_ld_sys(ep + 3);
                pre_valid = true;
            }
            const int rows_per_task = (int)q_lane_u32(v, QD_ROWS_PER_TASK);
            QRowGeo geo{};
#pragma nounroll
            for (int grp = 0; grp * kQRowsPerWave < rows_per_task; ++grp) {
                const int row0 = (int)row_tile * rows_per_task + grp * kQRowsPerWave;
                if (row0 >= t.dst_h) break;
                const int gi = (grp & 15) * kQRowsPerWave;
                if (gi == 0) geo = q_row_geo(t.P, t.dst_h, row0, lane); // the next 64 rows' vertical geometry, one row per lane
                if constexpr (q_kind_yuv(KIND)) {
                    k4q_rows<LD, ST, KIND == QK_P010>(t, (int)z, (int)col_tile, row0, lane, geo, gi);
                } else if constexpr (KIND == QK_PIXELS16) {
                    const bool sgn = q_lane_u32(v, QD_SRC_SIGNED) != 0; // wave-uniform
                    if (c3) {
                        if (sgn) k1q_rows<3, LD, ST, SRC_S16>(t, (int)z, (int)col_tile, row0, lane, geo, gi);
                        else k1q_rows<3, LD, ST, SRC_U16>(t, (int)z, (int)col_tile, row0, lane, geo, gi);
                    } else {
                        if (sgn) k1q_rows<4, LD, ST, SRC_S16>(t, (int)z, (int)col_tile, row0, lane, geo, gi);
                        else k1q_rows<4, LD, ST, SRC_U16>(t, (int)z, (int)col_tile, row0, lane, geo, gi);
                    }
                } else if (c3) k1q_rows<3, LD, ST>(t, (int)z, (int)col_tile, row0, lane, geo, gi);
                else k1q_rows<4, LD, ST>(t, (int)z, (int)col_tile, row0, lane, geo, gi);
            }
        }
        // ---- arrive: the write-through stores are visible device-wide once drained; the batch's last arrival raises its flags ----
        QPROF(const uint64_t p_t2 = wall_clock64();)
        q_drain();
        QPROF(const uint64_t p_t3 = wall_clock64();)
        QPROF(++p_tasks; p_find += p_t1 - p_t0; p_rows += p_t2 - p_t1; p_drain += p_t3 - p_t2; const uint64_t p_t3x = p_t3;)
        // Two-level arrival (one word takes ~88 returning atomics per microsecond: 400-1600 arrivals on ONE counter would cost
        // a batch 5-18 us of latency): task T bumps sub-counter T % 16 of its slot; whoever fills a sub-counter bumps the top one.
        uint64_t* ctr = m.arrive + (size_t)(b % R) * (1 + kQSubs) * kQCtrStride;
        const uint32_t sub = (uint32_t)T & (kQSubs - 1);
        // The next ticket.  Tasks were statically owned at first (worker w: T = w mod 4G): no atomics, but the grid then runs at the pace
        // of its SLOWEST worker -- the fast ones run ahead to the end of the ring and poll there (30 % of a worker's time at the
        // headline with a full 128-slot ring, 43 batches complete behind an incomplete oldest one).  Now a worker draws its next task
        // number from the counter of its residue class (T = w mod 16, the arrival sub-counters' classes: 16 words share the ~88
        // atomics per microsecond one word takes): work goes to whoever is free, oldest first.  The class's first 4G / 16 tickets are
        // the workers' initial ones (worker w starts with T = w).
        uint64_t before = 0, drawn = 0;
        if (lane == 0) {
            drawn = __hip_atomic_fetch_add((g_u64)(m.ticket + cls * kQCtrStride), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            before = __hip_atomic_fetch_add((g_u64)(ctr + (1 + sub) * kQCtrStride), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        T = (uint64_t)cls + (uint64_t)kQSubs * (cls_first + q_uni(drawn));
        if (lane == 0) q_st(m.prog + wid, T);
        if (q_uni(before) + 1 == sub_target) {
            if (lane == 0) before = __hip_atomic_fetch_add((g_u64)ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (q_uni(before) + 1 == q_lane_u64(v, QD_ARRIVE_TARGET)) {
                if (lane == 0) {
                    q_st_sys(m.dflags + kQCtrStride * (b % R), b + 1);
                    q_st_sys(hflags + (b % R), b + 1);
                }
            }
        }
        QPROF(p_t4 = wall_clock64(); p_arrive += p_t4 - p_t3x; if (!p_first) p_first = p_t0; p_last = p_t4;)
        if (T >= te) ++b;
    }
}

// staging path (no host-visible BAR): one wave copies n slots + index entries from the pinned host copies, then the tail
__global__ void k1q_stage(QDevMem m, const uint8_t* host_ring, const QIndex* host_index, uint32_t R, uint64_t first, uint32_t n, uint64_t new_tail) {
    const int lane = (int)threadIdx.x;
    for (uint32_t s = 0; s < n; ++s) {
        const uint64_t k = (first + s) % R;
        const uint64_t* hp = (const uint64_t*)(host_ring + k * kQSlotBytes) + lane * 8;
        uint64_t* dp = (uint64_t*)(m.ring + k * kQSlotBytes) + lane * 8;
        for (int i = 0; i < 8; ++i) q_st_sys(dp + i, q_ld_sys(hp + i));
        if (lane < 4) q_st_sys((uint64_t*)(m.index + k) + lane, q_ld_sys((const uint64_t*)(host_index + k) + lane));
    }
    q_drain();
    if (lane == 0) q_st_sys(&m.dc->tail.v, new_tail);
}

// The gate of a stream-ordered batch, enqueued on the CALLER's stream by cvgs_queue_submit_on: everything in front of it on that
// stream (the decoder / producer kernel that writes the frame) has completed when it runs, and the kernel boundary in front of it
// has released those writes; it opens the batch's gate and -- unless the caller defers the wait -- holds the stream until the
// batch's completion word is up, so that whatever follows on the stream sees the tensor.  One launch per submit: the reference's
// contract "asynchronous on the given stream" (include/cvGPUSpeedup.cuh:464-473) without a host synchronisation anywhere.
// A server that has reported an error (host word) releases the stream: the failure is reported by the next call, never waited for.
struct QGateTickets { // the batches behind ONE gate kernel (cvgs_queue_submit_many_on: a tick's frames; tickets need not be consecutive)
    uint64_t t[64];
    uint32_t n;
};
// `trace` (CVGS_QUEUE_DEBUG=2, tools/probes only; else null): per ticket {gate kernel start, completion seen} in 100 MHz ticks
__global__ void k1q_gate(uint64_t* gates, uint64_t* hgates, const uint64_t* dflags, uint32_t R, QGateTickets tk, uint32_t wait, const uint64_t* host_error,
                         uint64_t timeout_ticks, uint64_t* trace) {
    const uint32_t i = threadIdx.x;
    if (i >= tk.n) return;
    const uint64_t t = tk.t[i];
    const uint64_t t0 = wall_clock64();
    q_st_sys(gates + kQCtrStride * (t % R), t + 1);
    q_st_sys(hgates + (t % R), t + 1); // the host's copy: the admission budget of closed batches (queue_submit_on) is kept from it
    q_drain();
    if (trace) q_st_sys(trace + 2 * (t & 4095), t0);
    if (!wait) return;
    unsigned polls = 0;
    while (q_ld_sys(dflags + kQCtrStride * (t % R)) < t + 1) {
        __builtin_amdgcn_s_sleep(4);
        if ((++polls & 63) == 0) {
            if (q_ld_sys(host_error) != 0) return;
            if (wall_clock64() - t0 > timeout_ticks) return;
        }
    }
    if (trace) q_st_sys(trace + 2 * (t & 4095) + 1, wall_clock64());
}
// cvgs_queue_stream_wait: the stream waits until batches [first, last] are complete (one wave; lane i watches batch first + i, in
// rounds of 64).  hipStreamWaitValue64 on ordinary device memory proved unusable on a hot path: a wait that is not already satisfied
// when the stream reaches it costs ~1.6 ms on this runtime (tools/probes/stream_ordered_rate.py, deferred waits trailing by 4 batches).
__global__ void k1q_wait(const uint64_t* dflags, uint32_t R, uint64_t first, uint64_t last, const uint64_t* host_error, uint64_t timeout_ticks, uint64_t* trace) {
    const uint64_t t0 = wall_clock64();
    if (trace && threadIdx.x == 0) q_st_sys(trace + 8192 + 2 * (last & 4095), t0); // (probes: words 8192.. = {wait kernel start, end} per last ticket)
    for (uint64_t base = first; base <= last; base += 64) {
        const uint64_t b = base + threadIdx.x;
        unsigned polls = 0;
        if (b <= last)
            while (q_ld_sys(dflags + kQCtrStride * (b % R)) < b + 1) {
                __builtin_amdgcn_s_sleep(4);
                if ((++polls & 63) == 0 && (q_ld_sys(host_error) != 0 || wall_clock64() - t0 > timeout_ticks)) return;
            }
    }
    if (trace) {
        __builtin_amdgcn_wave_barrier();
        if (threadIdx.x == 0) q_st_sys(trace + 8192 + 2 * (last & 4095) + 1, wall_clock64());
    }
}
// host-side gate opening without a writable BAR (a failed gate launch must not leave workers waiting)
__global__ void k1q_gate_open(uint64_t* gate, uint64_t value) {
    if (threadIdx.x == 0) q_st_sys(gate, value);
}

// ====================================================================================================================
// host side
// ====================================================================================================================
struct Queue {
    int device = 0;
    hipStream_t stream = nullptr;       // the server's stream
    hipStream_t stage_stream = nullptr; // staging path only
    uint32_t R = 0, G = 0;
    int ld = 1, st = 2;
    int kind = -1;                      // QK_*: latched by the first submit
    int cus = 0;                        // compute units of the device
    bool direct = false;                // the host writes device memory through the BAR
    bool g_explicit = false;            // G came from the flags / CVGS_QUEUE_G: the per-kind default below does not apply
    uint8_t* dev_block = nullptr;       // ONE uncached device allocation (host-written): ctl | ring | index | dflags
    uint8_t* dev_counters = nullptr;    // ordinary device memory (device-only): arrival counters | resume words
    QDevMem m{};
    uint8_t* host_ring = nullptr;       // staging path: pinned copies the staging kernel reads
    QIndex* host_index = nullptr;
    QHostCtl* hc = nullptr;             // pinned
    uint64_t* hflags = nullptr;         // pinned: R completion flags
    uint64_t* gate_trace = nullptr;     // pinned, CVGS_QUEUE_DEBUG=2 only: 4096 x {gate kernel start, completion seen}
    uint64_t* hgates = nullptr;         // pinned: the gate kernels' host copies of the gate words (R)
    struct Closed { uint64_t ticket; uint32_t tasks; };
    std::vector<Closed> closed;         // stream-ordered batches whose gate the host has not yet seen open
    uint64_t closed_tasks = 0;
    std::vector<uint64_t> arrive_cum;   // per slot, 1 + 16 words: each arrival counter's value once every batch that used the slot has arrived
    uint64_t next_seq = 0, next_task = 0, done_inorder = 0, gen = 0, launches = 0;
    std::atomic<uint64_t> done_hint{0}; // every batch below is complete: what waiters (which do not take the mutex) have seen so far
    uint64_t idle_ticks = 0, stall_ticks = 0, gate_ticks = 0;
    uint64_t failed_upto = 0;              // every batch below was submitted before the last recovery
    std::vector<uint64_t> lost_tickets;    // ... and these had not completed then: their tensors may be incomplete (the newest 4096 are remembered)
    uint64_t n_gated = 0, n_direct = 0; // stream-ordered submits taken by the server / by a direct launch (hybrid policy)
    int server_prio = 0;
    bool server_prio_known = false, prio_range_nonempty = false;
    struct StreamTail { void* stream; uint64_t ticket; };
    std::vector<StreamTail> stream_tail; // the newest ticket of every stream that has submitted with an immediate wait (hybrid policy)
    std::mutex mu;
    std::mutex gate_mu;                  // held across "publish a group closed" + "enqueue its gate kernel" (taken before mu, never inside it)
    uint64_t ns_ring_wait = 0, n_sub = 0; // host side of submit: time spent waiting for a ring slot
    uint64_t n_ring_waits = 0, sum_done_behind_head = 0;               // head-of-line blocking: batches already complete behind an incomplete oldest one
};

static inline volatile uint64_t& hv(QLine& l) { return *(volatile uint64_t*)&l.v; }
static inline uint64_t hflag(Queue* q, uint64_t slot) { return *(volatile uint64_t*)(q->hflags + slot); }

// ---- host-side primitives: x86-64 has the real ones, anything else gets portable stand-ins (and the staged path by default) -------
static inline void cpu_pause() {
#if CVGS_HOST_X86
    _mm_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}
// orders the write-combined stores to device memory in front of it before the ones behind it
static inline void wc_fence() {
#if CVGS_HOST_X86
    _mm_sfence();
#else
    std::atomic_thread_fence(std::memory_order_seq_cst);
#endif
}

// Is device memory writable from the host (large BAR, the allocation mapped into this process)?  Decided WITHOUT ever faulting: round 3
// probed with a store guarded by process-wide SIGSEGV / SIGBUS handlers and siglongjmp -- inside a library that races every other
// thread's faults and every other user of sigaction.  Now: (1) flag bit 0 forces the staged path; (2) the device must report a large BAR (hipDeviceAttributeIsLargeBar); (3) the WHOLE block must lie
// inside read-write mappings of this process (/proc/self/maps: with a large BAR the runtime maps VRAM allocations through the render
// node at their device address; without one the range is a PROT_NONE reservation) -- only then is the test store issued, and (4) it must
// read back from the device.  Anything unknown (no /proc, a foreign OS, a non-x86 host) means "staged".
static bool range_is_mapped_rw(const void* p, size_t bytes) {
    FILE* f = std::fopen("/proc/self/maps", "r");
    if (!f) return false;
    uintptr_t need = (uintptr_t)p;
    const uintptr_t end = need + bytes;
    char line[512];
    bool ok = false;
    while (std::fgets(line, sizeof(line), f)) { // (the file is sorted by address)
        unsigned long long lo = 0, hi = 0;
        char perms[8] = {0};
        if (std::sscanf(line, "%llx-%llx %7s", &lo, &hi, perms) != 3) continue;
        if ((uintptr_t)hi <= need) continue;
        if ((uintptr_t)lo > need) break;                   // a hole at `need`
        if (perms[0] != 'r' || perms[1] != 'w') break;     // reserved / read-only
        need = (uintptr_t)hi;
        if (need >= end) { ok = true; break; }
    }
    std::fclose(f);
    return ok;
}
static bool decide_direct(int device, uint8_t* block, size_t block_bytes, uint64_t* probe_word, uint32_t flags) {
    if (flags & 1u) return false; // the caller asked for the staged ring
    if (!CVGS_HOST_X86) return false; // the write-combining path below is tuned and tested on x86-64 only
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess || !large_bar) return false;
    if (!range_is_mapped_rw(block, block_bytes)) return false;
    *(volatile uint64_t*)probe_word = 0x5157455545ull;
    wc_fence();
    uint64_t back = 0;
    if (hipMemcpy(&back, probe_word, 8, hipMemcpyDeviceToHost) != hipSuccess) return false;
    return back == 0x5157455545ull;
}

static void advance_done(Queue* q) { // (under q->mu)
    const uint64_t seen = q->done_hint.load(std::memory_order_relaxed);
    if (seen > q->done_inorder) q->done_inorder = seen;
    while (q->done_inorder < q->next_seq && hflag(q, q->done_inorder % q->R) >= q->done_inorder + 1) ++q->done_inorder;
}
static void raise_done_hint(Queue* q, uint64_t d) {
    uint64_t cur = q->done_hint.load(std::memory_order_relaxed);
    while (cur < d && !q->done_hint.compare_exchange_weak(cur, d, std::memory_order_relaxed)) {}
}

// ONE server grid per device at a time.  Every worker of a server holds a task number (its ticket), so every one of its workgroups must be resident;
// two queues' grids do not fit the chip together (3 of 4 wave slots per SIMD each): a second server stays partly resident for as
// long as the first one is fed, and its batches -- whose tasks are spread over ALL its workers -- cannot complete (the watchdog
// reports them after 250 ms).  So a queue that needs to launch asks the device's current server to retire as soon as it has
// nothing in flight, and waits for it; the other queue's submits wait for that retirement too (cvgs_queue_submit's yield check).  Alternating submits to two queues
// therefore cost a server switch each (~20 us); batches submitted in bursts per queue do not.
static std::mutex g_server_mu;
static Queue* g_server_owner[64]; // by device: the queue whose server may be alive

static hipError_t queue_launch(Queue* q) {
    std::lock_guard<std::mutex> owner_lock(g_server_mu);
    Queue* const o = g_server_owner[q->device & 63];
    if (o && o != q) {
        const uint64_t st = hv(o->hc->state);
        if (st == QS_RUNNING || st == QS_EXITING) {
            hv(o->hc->yield_req) = 1;
            std::atomic_thread_fence(std::memory_order_seq_cst);
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            for (;;) {
                const uint64_t s2 = hv(o->hc->state);
                if (s2 != QS_RUNNING && s2 != QS_EXITING) break;
                cpu_pause();
                if ((++spins & 0xffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                    hv(o->hc->yield_req) = 0;
                    return hipErrorLaunchTimeOut; // the other queue's server never drained
                }
            }
            hv(o->hc->yield_req) = 0;
        }
    }
    g_server_owner[q->device & 63] = q;
    ++q->gen;
    ++q->launches;
    advance_done(q);
    hv(q->hc->state) = QS_RUNNING;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    const dim3 grid(q->G + 1), block(256);
#define Q_LAUNCH(LD_, ST_, K_) hipLaunchKernelGGL((k1q_server<LD_, ST_, K_>), grid, block, 0, q->stream, q->m, q->hc, q->hflags, q->R, q->G, q->gen, q->done_inorder, q->idle_ticks, q->stall_ticks, q->gate_ticks)
    if (q->kind == QK_NV12) { // the product flavour, and the unsafe upper bound of the A/B tool
        if (q->ld == 0 && q->st == 0) Q_LAUNCH(0, 0, QK_NV12);
        else Q_LAUNCH(1, 2, QK_NV12);
    } else if (q->kind == QK_PIXELS16) {
        Q_LAUNCH(1, 2, QK_PIXELS16);
    } else if (q->kind == QK_P010) {
        Q_LAUNCH(1, 2, QK_P010);
    } else if (q->ld == 0 && q->st == 0) Q_LAUNCH(0, 0, QK_PIXELS);
    else if (q->ld == 0 && q->st == 1) Q_LAUNCH(0, 1, QK_PIXELS);
    else if (q->ld == 0) Q_LAUNCH(0, 2, QK_PIXELS);
    else if (q->st == 0) Q_LAUNCH(1, 0, QK_PIXELS);
    else if (q->st == 1) Q_LAUNCH(1, 1, QK_PIXELS);
    else Q_LAUNCH(1, 2, QK_PIXELS);
#undef Q_LAUNCH
    return hipGetLastError();
}

// after a batch has been published: make sure a server is (still) there to take it
static int queue_ensure_running(Queue* q) {
    std::atomic_thread_fence(std::memory_order_seq_cst); // host.tail store, then state load (the janitor does the mirror image)
    for (;;) {
        const uint64_t s = hv(q->hc->state);
        if (s == QS_RUNNING) return 0;
        if (s == QS_EXITING) { // the janitor is deciding: it re-reads host.tail and either resumes or exits, within microseconds
            cpu_pause();
            continue;
        }
        if (hv(q->hc->error)) return -2;
        return queue_launch(q) == hipSuccess ? 0 : -1;
    }
}

int queue_create(Queue** out, int device, int depth, uint32_t flags, double idle_us, std::string& err) {
    Queue* q = new (std::nothrow) Queue();
    if (!q) return -1;
    q->device = device;
    q->R = depth <= 0 ? 128 : (depth > kQMaxRing ? kQMaxRing : depth);
    // experiments (tools/queue_ab.py): bits 8..9 = store flavour + 1, bits 12..13 = load flavour + 1, bits 16..27 = workgroups;
    // bit 0: never write device memory from the host (force the staging path)
    if ((flags >> 8) & 3) q->st = (int)((flags >> 8) & 3) - 1;
    if ((flags >> 12) & 3) q->ld = (int)((flags >> 12) & 3) - 1;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { err = "hipGetDeviceProperties"; delete q; return -1; }
    int per_cu = 0;
    e = q->st == 2 && q->ld == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k1q_server<1, 2>, 256, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k1q_server<0, 0>, 256, 0);
    if (e != hipSuccess || per_cu < 1) { err = "occupancy query failed"; delete q; return -1; }
    // Every workgroup must be resident (each worker holds a ticket).  The occupancy API can overstate by one where the SGPR
    // file is the binding limit (MI355X_MICROARCH.md "Residency"); at <= 5 workgroups per CU the VGPR file binds and it is exact.
    int use = per_cu > 8 ? 8 : per_cu;
    if (use > 5) use -= 1;
    // Three workgroups per CU already run the headline at the memory system's rate for its access pattern (2.4-2.6 us per 50-crop
    // batch, the rate of 16-64 frames fused into one launch; tools/queue_ab.py: 767 vs 1023 workgroups tie) and leave more than
    // half of every SIMD's wave slots to the caller's other kernels while the server is alive.
    if (use > 3) use = 3;
    uint32_t G = (uint32_t)(prop.multiProcessorCount * use) - 1;
    q->cus = prop.multiProcessorCount;
    if ((flags >> 16) & 0xfff) {
        G = (flags >> 16) & 0xfff;
        q->g_explicit = true;
    } else if (const char* ge = getenv("CVGS_QUEUE_G")) { // tuning hook: worker workgroups (the flags' bits 16..27 say the same per queue)
        const int v = atoi(ge);
        if (v >= 1 && v <= 4095) {
            G = (uint32_t)v;
            q->g_explicit = true;
        }
    }
    const uint32_t g_cap = (uint32_t)prop.multiProcessorCount * 4u - 1u; // every workgroup must be resident (each worker holds a ticket): 4 per CU at most
    if (G > g_cap) G = g_cap;
    // A task of residue class T % 16 is only ever drawn by the workers of that class (w % 16): fewer than 16 workers -- G < 4 -- would
    // leave classes nobody serves, and every batch of 5 or more tasks would sit until the watchdog.
    if (G * kQWaves < (uint32_t)kQSubs) G = (uint32_t)(kQSubs + kQWaves - 1) / kQWaves;
    q->G = G;
    const double tick_hz = 100e6; // s_memrealtime: constant 100 MHz
    q->idle_ticks = (uint64_t)((idle_us <= 0 ? 200.0 : idle_us) * 1e-6 * tick_hz);
    // a batch that makes no progress for 250 ms is reported, not waited for.  CVGS_QUEUE_STALL_MS moves the limit (a process whose OTHER
    // kernels can hold the whole chip for longer than that -- the server's workgroups must all be resident -- wants a larger one)
    double stall_s = 0.25;
    if (const char* sm = getenv("CVGS_QUEUE_STALL_MS")) {
        const double v = atof(sm);
        if (v >= 1.0 && v <= 600000.0) stall_s = v * 1e-3;
    }
    q->stall_ticks = (uint64_t)(stall_s * tick_hz);
    // a stream-ordered batch may wait this long at its gate for the work in front of it on the caller's stream (then: error 3)
    const double gate_s = 10.0;
    q->gate_ticks = (uint64_t)(gate_s * tick_hz);
    const size_t R = q->R, NW = (size_t)q->G * kQWaves;
    const size_t off_ring = 4096, off_index = off_ring + R * kQSlotBytes, off_dflags = off_index + R * sizeof(QIndex), off_gates = off_dflags + R * 128, total = off_gates + R * 128;
    const size_t ctr_bytes = R * (1 + kQSubs) * kQCtrStride * 8, prog_bytes = (NW * 8 + 127) & ~(size_t)127, total_ctr = ctr_bytes + prog_bytes + kQSubs * kQCtrStride * 8;
    // The server is a LONG-LIVED kernel: whatever shares its hardware queue waits until it retires.  The runtime multiplexes streams
    // onto a few hardware queues PER PRIORITY LEVEL, so the server's stream is created at the highest priority and the staging stream
    // at the lowest: neither shares a hardware queue with the caller's (default-priority) streams -- on which the gate kernels of
    // stream-ordered submits and hipStreamWaitValue64 consumers must be able to run while the server is alive (found by
    // tools/probes/stream_ordered_rate.py: with four caller streams one of them shared the server's queue, its gate kernel never
    // started, the server waited at that gate for the 10 s limit) -- nor with each other.  Callers' streams of the HIGHEST priority would
    // meet the server in that level's pool: do not attach those.  (A CU-masked stream -- a hardware queue of its own -- was tried for the
    // server instead: tests/test_gpu_queue.py::test_queue_nv12_many_batches_in_flight then failed, and the gate trace did not change.)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if ((e = hipStreamCreateWithPriority(&q->stream, hipStreamNonBlocking, prio_greatest)) != hipSuccess ||
        (e = hipStreamCreateWithPriority(&q->stage_stream, hipStreamNonBlocking, prio_least)) != hipSuccess ||
        (e = hipExtMallocWithFlags((void**)&q->dev_block, total, hipDeviceMallocUncached)) != hipSuccess ||
        (e = hipMalloc((void**)&q->dev_counters, total_ctr)) != hipSuccess ||
        (e = hipHostMalloc((void**)&q->hc, sizeof(QHostCtl), hipHostMallocDefault)) != hipSuccess ||
        (e = hipHostMalloc((void**)&q->hflags, R * 8, hipHostMallocDefault)) != hipSuccess ||
        (e = hipHostMalloc((void**)&q->hgates, R * 8, hipHostMallocDefault)) != hipSuccess) {
        err = std::string("queue allocation: ") + hipGetErrorString(e);
        return -1; // (leaks on this cold path are reclaimed at process exit)
    }
    q->m.dc = (QDevCtl*)q->dev_block;
    q->m.ring = q->dev_block + off_ring;
    q->m.index = (QIndex*)(q->dev_block + off_index);
    q->m.dflags = (uint64_t*)(q->dev_block + off_dflags);
    q->m.gates = (uint64_t*)(q->dev_block + off_gates);
    q->m.arrive = (uint64_t*)q->dev_counters;
    q->m.prog = (uint64_t*)(q->dev_counters + ctr_bytes);
    q->m.ticket = (uint64_t*)(q->dev_counters + ctr_bytes + prog_bytes);
    std::memset((void*)q->hc, 0, sizeof(QHostCtl));
    std::memset((void*)q->hflags, 0, R * 8);
    std::memset((void*)q->hgates, 0, R * 8);
    q->arrive_cum.assign(R * (1 + kQSubs), 0);
    std::vector<uint64_t> p0(NW);
    for (size_t i = 0; i < NW; ++i) p0[i] = i; // worker w's first task is w
    if ((e = hipMemset(q->dev_block, 0, total)) != hipSuccess || (e = hipMemset(q->dev_counters, 0, total_ctr)) != hipSuccess ||
        (e = hipMemcpy(q->m.prog, p0.data(), NW * 8, hipMemcpyHostToDevice)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) {
        err = std::string("queue init: ") + hipGetErrorString(e);
        return -1;
    }
    // (the server's priority, for queue_submit_on's guard against caller streams that may share its hardware queue)
    q->prio_range_nonempty = prio_least != prio_greatest;
    q->server_prio_known = hipStreamGetPriority(q->stream, &q->server_prio) == hipSuccess;
    (void)hipGetLastError();
    if (const char* dbg = getenv("CVGS_QUEUE_DEBUG"); dbg && dbg[0] == '2' && hipHostMalloc((void**)&q->gate_trace, 4096 * 32, hipHostMallocDefault) == hipSuccess) std::memset(q->gate_trace, 0, 4096 * 32);
    q->direct = decide_direct(device, q->dev_block, total, &q->m.dc->stop_gen.pad[0], flags);
    if (!q->direct) {
        if ((e = hipHostMalloc((void**)&q->host_ring, R * kQSlotBytes, hipHostMallocDefault)) != hipSuccess ||
            (e = hipHostMalloc((void**)&q->host_index, R * sizeof(QIndex), hipHostMallocDefault)) != hipSuccess) {
            err = std::string("queue staging buffers: ") + hipGetErrorString(e);
            return -1;
        }
        std::memset(q->host_ring, 0, R * kQSlotBytes);
        std::memset((void*)q->host_index, 0, R * sizeof(QIndex));
    }
    *out = q;
    return 0;
}

// `bytes` (a multiple of 16) to a 16-byte aligned destination with non-temporal stores
static inline void wc_copy16(void* dst, const void* src, size_t bytes) {
#if CVGS_HOST_X86
    __m128i* d = (__m128i*)dst;
    const __m128i* s = (const __m128i*)src;
    for (size_t i = 0; i < bytes / 16; ++i) _mm_stream_si128(d + i, _mm_loadu_si128(s + i));
#else
    volatile uint64_t* d = (volatile uint64_t*)dst;
    const uint64_t* s = (const uint64_t*)src;
    for (size_t i = 0; i < bytes / 8; ++i) d[i] = s[i];
#endif
}

// Can the server take this chain?  K1's hot shape: 8U C3 / C4 crops -> bilinear resize -> [swap R,B] mul sub div -> fp32 planar
// tensor (NCHW / CNHW), descriptors inline, one target -- or K4's: the same behind crops of an NV12 / NV21 decoder surface
// (cvtColorNV12 in front of the resize, 3 channels).  A queue serves ONE of the two (its first submit decides); everything else
// belongs to cvgs_execute.
// a stream-ordered batch (published with its gate closed)
struct GateInfo {
    void* stream;        // the caller's stream (a key, never dereferenced)
    bool defer;          // the caller's stream is not held on the batch
    uint32_t rows;       // rows per task, decided by queue_submit_on (the closed-batch budget is kept in tasks)
    uint32_t n_tasks;    // out
};
static int queue_submit_slot(Queue* q, const ChainArgs& c_in, const PlaneParams* planes, int n_planes, uint64_t* ticket, std::string& err, GateInfo* gate = nullptr) {
    const bool gated = gate != nullptr;
    const ReadArgs& r = c_in.read;
    const WriteArgs& w = c_in.write;
    const bool planar = w.kind == CVGS_WRITE_TENSOR_SPLIT || w.kind == CVGS_WRITE_TENSOR_T_SPLIT;
    const bool nv12 = r.kind == CVGS_READ_NV12_RESIZE_LINEAR;
    const bool wide = !nv12 && (r.depth == CVGS_DEPTH_16U || r.depth == CVGS_DEPTH_16S);
    const bool p010 = nv12 && r.yuv_layout == CVGS_YUV_P010;
    const int kind = p010 ? QK_P010 : (nv12 ? QK_NV12 : (wide ? QK_PIXELS16 : QK_PIXELS));
    const int vcn = nv12 ? r.out_cn : r.cn; // channels of the value the program sees
    const bool half = w.depth == CVGS_DEPTH_16F; // the half-precision hand-off: the chain ends with CAST(CV_16F), which the store performs
    if (!planar || (w.depth != CVGS_DEPTH_32F && !half) || w.data2 || r.table || n_planes < 1 || n_planes > kQMaxPlanes || n_planes != r.batch ||
        (!nv12 && (r.kind != CVGS_READ_RESIZE_LINEAR || (r.depth != CVGS_DEPTH_8U && !wide) || (r.cn != 3 && r.cn != 4))) ||
        (nv12 && ((r.yuv_layout != CVGS_YUV_NV12 && r.yuv_layout != CVGS_YUV_NV21 && !p010) || r.out_cn != 3))) {
        err = "queue: chain is not a batched 8U / 16U / 16S C3 / C4 (or NV12 / NV21 / P010 -> 3 channels) resize into an fp32 / fp16 planar tensor with host plane descriptors";
        return 1;
    }
    for (int i = 0; i < n_planes && i < r.used; ++i) {
        if (nv12 ? planes[i].w < 4 : planes[i].w * r.cn < 8) { // (8 ELEMENTS: the window is 8 bytes of 8-bit, 16 bytes of 16-bit pixels)
            err = nv12 ? "queue: a surface crop narrower than 4 pixels" : "queue: a crop narrower than the tap window (1-2 pixels)";
            return 1;
        }
        if ((uint64_t)planes[i].h * (uint64_t)planes[i].step >= (1ull << 32)) { // the workers address a crop's rows with 32-bit byte offsets
            err = "queue: a source crop spanning 4 GB or more";
            return 1;
        }
    }
    ChainArgs c = c_in;
    c.prog.fast_div = 0;
    for (int k = 0; k < 4; ++k) c.prog.rdiv[k] = 0.f;
    if (half) {
        if (c.prog.n < 1 || c.prog.opcode[c.prog.n - 1] != CVGS_OP_CAST || c.prog.aux[c.prog.n - 1] != CVGS_DEPTH_16F) {
            err = "queue: an fp16 tensor needs a chain that ends with convertTo<CV_32F, CV_16F>";
            return 1;
        }
        c.prog.n -= 1;
    }
    const int prog_id = k1_classify_program(c.prog, vcn);
    if (prog_id > 1) {
        err = "queue: the pointwise program must be [RGB<->BGR swap,] mul, sub, div";
        return 1;
    }
    fast_div_setup(c.prog, prog_id == 0 ? 3 : 2, prog_id == 0 ? 1 : 0, vcn, r.bg);
    const int o = prog_id == 0 ? 1 : 0; // index of the MUL stage
    // output bytes addressed through ONE 32-bit-offset buffer descriptor
    const int64_t last = (int64_t)(r.batch - 1) * w.img_stride + (int64_t)(vcn - 1) * w.ch_stride + (int64_t)r.dst_h * w.width;
    if (last <= 0 || last * (half ? 2 : 4) >= (int64_t)1 << 31) {
        err = "queue: output tensor beyond 2 GB";
        return 1;
    }
    std::lock_guard<std::mutex> lock(q->mu);
    if (hv(q->hc->error)) {
        err = "queue: the server reported a stall / protocol error earlier (cvgs_queue_recover resets the queue)";
        return -2;
    }
    if (hv(q->hc->yield_req)) { // another queue of this device waits to launch its server: this one drains and retires first
        const auto ty = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (hv(q->hc->yield_req)) {
            const uint64_t st = hv(q->hc->state);
            if (st != QS_RUNNING && st != QS_EXITING) break;
            cpu_pause();
            if ((++spins & 0xffff) == 0 && std::chrono::steady_clock::now() - ty > std::chrono::seconds(5)) break;
        }
    }
    if (q->kind < 0) {
        q->kind = kind;
        // The 8-bit pixel worker is bound by the memory system, and TWO workgroups per CU feed it slightly better than three (round 4, A/B
        // on one box: 2.147 against 2.191 us per 50-crop batch, 0.532 against 0.522; 639: 2.161) -- while the surface kinds, bound by their
        // arithmetic, lose 5-9 % with two (cfg #3 5.27 against 5.02 us, P010 6.89 against 6.29).  The create call sized everything for
        // three; nothing has been launched yet, so the kind's default is applied here.
        if (kind == QK_PIXELS && !q->g_explicit && q->G > (uint32_t)q->cus * 2u - 1u) q->G = (uint32_t)q->cus * 2u - 1u;
        if (kind != QK_PIXELS) { // every workgroup must be resident (each worker holds a ticket): these workers' register budget allows 3 per CU
            int per_cu = 0;
            const hipError_t oe = kind == QK_NV12   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k1q_server<1, 2, QK_NV12>, 256, 0)
                                  : kind == QK_P010 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k1q_server<1, 2, QK_P010>, 256, 0)
                                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k1q_server<1, 2, QK_PIXELS16>, 256, 0);
            if (oe != hipSuccess || per_cu < 1) {
                q->kind = -1;
                err = "queue: occupancy query failed";
                return -1;
            }
            const uint32_t g_max = (uint32_t)(q->cus * (per_cu > 4 ? 4 : per_cu)) - 1;
            if (q->G > g_max) q->G = g_max; // (nothing has been launched yet: the first submit decides the kind)
        }
    }
    if (q->kind != kind) {
        err = q->kind == QK_P010 ? "queue: this queue serves P010 surface crops (its first submit decided); use another queue for the other kinds"
              : q->kind == QK_NV12 ? "queue: this queue serves NV12 / NV21 surface crops (its first submit decided); use another queue for pixel crops"
              : (q->kind == QK_PIXELS16 ? "queue: this queue serves 16-bit pixel crops (its first submit decided); use another queue for the other kinds"
                                        : "queue: this queue serves 8UC3 / 8UC4 crops (its first submit decided); use another queue for the other kinds");
        return 1;
    }
    auto ns_since = [](std::chrono::steady_clock::time_point a) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - a).count(); };
    // ring space: batch next_seq reuses the slot of batch next_seq - R, which must be complete
    // (the host side of a submit is ~1.2 us, a third of it the write-combined stores below: no clock is read on the way unless the
    //  ring is full -- tools/probes/submit_host_cost.cpp, submit_cost_probe.cpp)
    const uint64_t k = q->next_seq % q->R;
    if (q->next_seq >= q->R && hflag(q, k) < q->next_seq - q->R + 1) {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        ++q->n_ring_waits;
        for (uint64_t bb = q->next_seq - q->R + 1; bb < q->next_seq; ++bb) q->sum_done_behind_head += hflag(q, bb % q->R) >= bb + 1;
        while (hflag(q, k) < q->next_seq - q->R + 1) {
            if ((spins & 255) == 0) {
                const int rc = queue_ensure_running(q);
                if (rc) { err = "queue: server launch failed / stalled"; return rc; }
            }
            cpu_pause();
            if ((++spins & 0xffff) == 0 && ns_since(t0) > 2000000000ull) { err = "queue: ring full for 2 s"; return -2; }
        }
        q->ns_ring_wait += ns_since(t0);
    }
    QParams p;
    std::memset(&p, 0, sizeof(p));
    p.stamp = q->next_seq + 1;
    p.task_base = q->next_task;
    p.col_tiles = (uint32_t)((r.dst_w + 63) / 64);
    // task size: a shallow queue is a latency problem (4 rows per task: four times the waves on each batch), a deep one a throughput
    // problem -- every worker is busy there, a task costs its worker ~4 us of dispatch (ticket, index window, slot, drain, arrival)
    // beside ~0.8 us per row, and larger tasks keep a wave on one crop's rows (headline with the deep size pinned to 16 / 32 / 64 / 128
    // rows: 2.49 / 2.38 / 2.25 / 2.21 us per batch).  The price is the tail of a burst -- the last batches' tasks run 60 - 110 us: a burst
    // of 16 batches takes 71 us with 64-row tasks from 8 batches in flight, 53 us with 16-row tasks (tools/probes/queue_burst_tail.py).
    // So: 16 rows with 2..7 batches in flight, 32 from 8, and the large size (128 rows; 64 on rings shallower than 128 slots, whose
    // batches would not hold two tasks per worker) only when the ring is three quarters full, i.e. the stream is sustained.
    advance_done(q);
    const uint64_t in_flight = q->next_seq - q->done_inorder;
    const bool sustained = q->R >= 32 && in_flight * 4 >= q->R * 3;
    const uint32_t deep_rows = sustained ? (q->R >= 128 ? (uint32_t)kQRowsPerTaskDeep : 64u) : (uint32_t)kQRowsPerTaskMid;
    p.rows_per_task = gated ? gate->rows : (in_flight >= 8 ? deep_rows : (in_flight >= 2 ? (uint32_t)kQRowsPerTask : (uint32_t)kQRowsPerWave));
    p.tiles_per_plane = p.col_tiles * (uint32_t)((r.dst_h + (int)p.rows_per_task - 1) / (int)p.rows_per_task);
    p.n_tasks = p.tiles_per_plane * (uint32_t)r.batch;
    p.n_planes = (uint32_t)n_planes;
    p.used = (uint32_t)r.used;
    p.dst_w = (uint32_t)r.dst_w;
    p.dst_h = (uint32_t)r.dst_h;
    p.out_w = (uint32_t)w.width;
    p.cn = (uint32_t)vcn;
    p.kind = (uint32_t)kind;
    p.yuv_range = (uint32_t)r.yuv_range;
    p.yuv_primaries = (uint32_t)r.yuv_primaries;
    p.yuv_vu = r.yuv_layout == CVGS_YUV_NV21;
    p.src_signed = r.depth == CVGS_DEPTH_16S;
    p.swap = prog_id == 0;
    p.fast_div = (uint32_t)c.prog.fast_div;
    for (int i = 0; i < 4; ++i) {
        p.mul[i] = c.prog.operand[o][i];
        p.sub[i] = c.prog.operand[o + 1][i];
        p.div[i] = c.prog.operand[o + 2][i];
        p.rdiv[i] = c.prog.rdiv[i];
        p.bg[i] = r.bg[i];
    }
    p.img_stride = w.img_stride;
    p.ch_stride = w.ch_stride;
    p.out = (uint64_t)w.data;
    p.out_bytes = (uint64_t)last * (half ? 2 : 4);
    p.out_half = half ? 1u : 0u;
    p.gated = gated ? 1u : 0u;
    // cumulative arrival targets of the slot's counters (they are never reset): sub-counter s takes the tasks T = s (mod 16)
    uint64_t sub_targets[kQSubs];
    uint64_t* cum = &q->arrive_cum[k * (1 + kQSubs)];
    uint32_t filled = 0;
    for (int sres = 0; sres < kQSubs; ++sres) {
        const uint64_t lo = p.task_base, hi = p.task_base + p.n_tasks; // tasks in [lo, hi) with T % 16 == sres
        const uint64_t cnt = (hi + (kQSubs - 1 - sres)) / kQSubs - (lo + (kQSubs - 1 - sres)) / kQSubs;
        cum[1 + sres] += cnt;
        sub_targets[sres] = cnt ? cum[1 + sres] : 0; // 0: never matched (counters start at 0 and the comparison is against old + 1 >= 1)
        filled += cnt != 0;
    }
    cum[0] += filled;
    p.arrive_target = cum[0];
    QIndex ix;
    ix.end_task = p.task_base + p.n_tasks;
    ix.stamp = p.stamp;
    ix.n_tasks = p.n_tasks;
    ix.tiles_per_plane = p.tiles_per_plane;
    ix.n_planes = p.n_planes;
    ix.check = q_index_check(ix.end_task, ix.stamp, ix.n_tasks, ix.tiles_per_plane, ix.n_planes);
    // 1. the retirement hand-shake, on host memory only (no write-combined store is pending, so the fence is cheap): once
    //    host.tail is ahead of the device's tail the janitor cannot retire, whatever it sees next
    if (ticket) *ticket = q->next_seq;
    q->next_seq += 1;
    q->next_task += p.n_tasks;
    hv(q->hc->tail) = q->next_seq;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    bool need_launch = false;
    for (;;) {
        const uint64_t st = hv(q->hc->state);
        if (st == QS_RUNNING) break;
        if (st == QS_EXITING) { // the janitor is deciding: it re-reads host.tail and resumes or exits within microseconds
            cpu_pause();
            continue;
        }
        if (hv(q->hc->error)) { err = "queue: the server reported a stall / protocol error"; return -2; }
        need_launch = true;
        break;
    }
    // A retired grid's workers leave within a microsecond of their janitor -- but one that sees the tail move before it looks at the
    // stop word would take a task of the batch published below, and so would the worker of the NEXT grid that resumes at the same
    // task number: two arrivals for one task, an arrival counter that skips its target, a batch that never completes (the watchdog's
    // 250 ms).  Nothing new is published until the old grid is gone.
    if (need_launch) (void)hipStreamSynchronize(q->stream);
    // 2. the slot, its index entry, then the tail
    uint8_t* slot = (q->direct ? q->m.ring : q->host_ring) + k * kQSlotBytes;
    QIndex* ixp = (q->direct ? q->m.index : q->host_index) + k;
    if (q->direct) { // write-combined device memory: every 64-byte line exactly once, front to back (memcpy's overlapping head / tail
                     // stores flush half-filled combining buffers: 550 vs 380 ns for a 50-crop slot)
        wc_copy16(slot, &p, sizeof(p));
        wc_copy16(slot + kQSubOff, sub_targets, sizeof(sub_targets));
        wc_copy16(slot + kQPlanesOff, planes, (size_t)n_planes * sizeof(PlaneParams));
    } else {
        std::memcpy(slot, &p, sizeof(p));
        std::memcpy(slot + kQSubOff, sub_targets, sizeof(sub_targets));
        std::memcpy(slot + kQPlanesOff, planes, (size_t)n_planes * sizeof(PlaneParams));
    }
    for (int i = 0; i < 4; ++i) ((volatile uint64_t*)ixp)[i] = ((const uint64_t*)&ix)[i]; // four atomic 8-byte stores
    if (q->direct) {
        wc_fence(); // the write-combined slot / index stores leave before the tail does
        *(volatile uint64_t*)&q->m.dc->tail.v = q->next_seq;
        wc_fence();
    } else {
        hipLaunchKernelGGL(k1q_stage, dim3(1), dim3(64), 0, q->stage_stream, q->m, q->host_ring, q->host_index, q->R, q->next_seq - 1, 1u, q->next_seq);
        if (hipGetLastError() != hipSuccess) { err = "queue: staging kernel launch failed"; return -1; }
    }
    q->n_sub += 1;
    if (gated) {
        gate->n_tasks = p.n_tasks;
        q->closed.push_back({q->next_seq - 1, p.n_tasks});
        q->closed_tasks += p.n_tasks;
    }
    // 3. a retired server is replaced AFTER the batch is in place
    if (need_launch && queue_launch(q) != hipSuccess) { err = "queue: server launch failed"; return -1; }
    return 0;
}

// A ring slot holds 74 planes.  A larger batch (the reference's own sweep of this chain goes to 300 crops, tests/batchresize/
// test_batchresize_x_split3D.cu:384-392) is submitted as consecutive slots, each a batch of its own over its planes' slice of the
// tensor; the ticket is the last slot's, and a wait for it covers the earlier ones (queue_wait).
int queue_submit(Queue* q, const ChainArgs& c, const PlaneParams* planes, int n_planes, uint64_t* ticket, std::string& err) {
    if (n_planes <= kQMaxPlanes || n_planes != c.read.batch || c.read.table) return queue_submit_slot(q, c, planes, n_planes, ticket, err);
    const bool nv12 = c.read.kind == CVGS_READ_NV12_RESIZE_LINEAR;
    for (int i = 0; i < n_planes && i < c.read.used; ++i) // what a later slot would be refused for is refused before the first one is published
        if (nv12 ? planes[i].w < 4 : planes[i].w * c.read.cn < 8) {
            err = nv12 ? "queue: a surface crop narrower than 4 pixels" : "queue: a crop narrower than the tap window (1-2 pixels)";
            return 1;
        }
    const int64_t elem = c.write.depth == CVGS_DEPTH_16F ? 2 : 4;
    for (int base = 0; base < n_planes; base += kQMaxPlanes) {
        const int cnt = n_planes - base < kQMaxPlanes ? n_planes - base : kQMaxPlanes;
        ChainArgs part = c;
        part.read.batch = cnt;
        part.read.used = c.read.used <= base ? 0 : (c.read.used - base < cnt ? c.read.used - base : cnt);
        part.write.data = c.write.data + (int64_t)base * c.write.img_stride * elem;
        if (const int rc = queue_submit_slot(q, part, planes + base, cnt, ticket, err)) return rc;
    }
    return 0;
}

// debugging aid (CVGS_QUEUE_DEBUG=1; =2 also records the gate kernels' timestamps for tools/probes/gate_trace.py): what the device-side state of the oldest incomplete batch looks like when the server reports a stall
static void queue_debug_dump(Queue* q) {
    static const bool on = getenv("CVGS_QUEUE_DEBUG") != nullptr;
    if (!on) return;
    advance_done(q);
    const uint64_t b = q->done_inorder, k = b % q->R;
    fprintf(stderr, "[cvgs queue] stall: next_seq %llu done_inorder %llu gen %llu launches %llu state %llu dev tail %llu host tail %llu\n",
            (unsigned long long)q->next_seq, (unsigned long long)b, (unsigned long long)q->gen, (unsigned long long)q->launches,
            (unsigned long long)hv(q->hc->state), (unsigned long long)*(volatile uint64_t*)&q->m.dc->tail.v, (unsigned long long)hv(q->hc->tail));
    const QParams* p = (const QParams*)(q->m.ring + k * kQSlotBytes);
    const QIndex* ix = q->m.index + k;
    fprintf(stderr, "  slot %llu: stamp %llu task_base %llu n_tasks %u arrive_target %llu | index end_task %llu stamp %llu\n", (unsigned long long)k,
            (unsigned long long)p->stamp, (unsigned long long)p->task_base, p->n_tasks, (unsigned long long)p->arrive_target,
            (unsigned long long)ix->end_task, (unsigned long long)ix->stamp);
    uint64_t ctr[1 + kQSubs] = {0};
    for (int i = 0; i <= kQSubs; ++i) (void)hipMemcpyAsync(&ctr[i], q->m.arrive + (k * (1 + kQSubs) + i) * kQCtrStride, 8, hipMemcpyDeviceToHost, q->stage_stream);
    (void)hipStreamSynchronize(q->stage_stream);
    const uint64_t* sub = (const uint64_t*)((const uint8_t*)p + kQSubOff);
    fprintf(stderr, "  top counter %llu (target %llu); sub counters / targets:", (unsigned long long)ctr[0], (unsigned long long)p->arrive_target);
    for (int i = 0; i < kQSubs; ++i) fprintf(stderr, " %llu/%llu", (unsigned long long)ctr[1 + i], (unsigned long long)sub[i]);
    fprintf(stderr, "\n");
    const size_t NW = (size_t)q->G * kQWaves;
    std::vector<uint64_t> prog(NW);
    (void)hipMemcpyAsync(prog.data(), q->m.prog, NW * 8, hipMemcpyDeviceToHost, q->stage_stream);
    (void)hipStreamSynchronize(q->stage_stream);
    size_t behind = 0;
    uint64_t first = ~0ull;
    for (size_t i = 0; i < NW; ++i)
        if (prog[i] < p->task_base + p->n_tasks) { ++behind; if (prog[i] < first) first = prog[i]; }
    fprintf(stderr, "  workers whose next task lies inside / before the batch: %zu of %zu (smallest next task %llu; batch tasks [%llu, %llu))\n", behind, NW,
            (unsigned long long)first, (unsigned long long)p->task_base, (unsigned long long)(p->task_base + p->n_tasks));
}

// Tickets are handed out in submit order, but batches do NOT complete in that order (a batch's tasks are spread over the workers,
// 400 tasks over 3068 of them: with a full ring ~20 batches are complete behind an incomplete older one).  A wait is for the
// ticket AND every batch submitted before it -- "wait(last) means all done" is what callers assume.  A slot's flag only ever grows
// (a later batch in the slot carries a larger stamp and was submitted after the earlier one had completed), so flag >= b + 1
// says "batch b is complete" however stale b is.
int queue_wait(Queue* q, uint64_t ticket, double timeout_s, std::string& err) {
    if (ticket >= q->next_seq) { err = "queue: ticket was never issued"; return 1; }
    if (ticket < q->failed_upto) { // submitted before the last recovery: complete unless it is one of the batches declared lost then
        std::lock_guard<std::mutex> lock(q->mu);
        for (const uint64_t t : q->lost_tickets)
            if (t <= ticket && (ticket < q->R || t + q->R > ticket)) { // (a wait covers the ticket and the batches before it)
                err = "queue: a batch up to this ticket was in flight when the server stalled; the queue was recovered, its tensor may be incomplete";
                return -2;
            }
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    auto wait_flag = [&](uint64_t b) {
        while (hflag(q, b % q->R) < b + 1) {
            if (hv(q->hc->error)) { queue_debug_dump(q); err = "queue: the server reported a stall / protocol error"; return -2; }
            cpu_pause();
            if ((++spins & 1023) == 0 && timeout_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
                err = "queue: wait timed out";
                return -3;
            }
        }
        return 0;
    };
    // the ticket's own flag first (the usual last one to rise), then whatever older batch is still open
    if (const int rc = wait_flag(ticket)) return rc;
    uint64_t b = q->done_hint.load(std::memory_order_relaxed);
    if (ticket >= q->R && b + q->R <= ticket) b = ticket - q->R + 1; // batch ticket took the slot of batch ticket - R: that one and all before it were complete
    for (; b < ticket; ++b)
        if (const int rc = wait_flag(b)) return rc;
    std::atomic_thread_fence(std::memory_order_acquire);
    raise_done_hint(q, ticket + 1);
    return 0;
}

// The stream form: one hipStreamWaitValue64 per batch up to the ticket that the host has not yet seen complete (usually one or
// two: the consumer of frame k is enqueued right after frame k's submit).
int queue_stream_wait(Queue* q, uint64_t ticket, void* stream, std::string& err) {
    if (ticket >= q->next_seq) { err = "queue: ticket was never issued"; return 1; }
    uint64_t d = q->done_hint.load(std::memory_order_relaxed);
    if (ticket >= q->R && d + q->R <= ticket) d = ticket - q->R + 1;
    while (d <= ticket && hflag(q, d % q->R) >= d + 1) ++d;
    raise_done_hint(q, d < ticket + 1 ? d : ticket + 1);
    if (d > ticket) return 0; // the host has already seen every batch up to the ticket complete: nothing to wait for
    // (ONE one-wave polling kernel: an unsatisfied hipStreamWaitValue64 on device memory costs ~1.6 ms on this runtime -- round 3's spelling)
    hipLaunchKernelGGL(k1q_wait, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint64_t*)q->m.dflags, q->R, d, ticket, (const uint64_t*)&q->hc->error.v, q->gate_ticks, q->gate_trace);
    if (hipGetLastError() != hipSuccess) { err = "queue: the wait kernel could not be launched on the consumer's stream"; return -1; }
    return 0;
}

// ---- stream-ordered submit -------------------------------------------------------------------------------------
// cvgs_queue_submit's batch is ordered behind NOTHING: a decoder or a previous kernel writing the frame on stream S had to be
// host-synchronised before the call.  This form keeps the reference's contract -- "asynchronous on the given stream"
// (include/cvGPUSpeedup.cuh:464-473) -- on the queue: the batch is published with its gate closed, ONE one-wave kernel (k1q_gate) goes
// onto S behind the producer, opens the gate when S gets there and -- unless QSUB_DEFER_WAIT -- holds S until the batch's completion
// word is up.  Workers that draw a task of the batch wait at the gate (they poll an uncached word); every other batch in the ring
// proceeds.  QSUB_HYBRID: when nothing the batch could overlap with is in flight, the caller is told to take the direct launch (2):
// a lone batch on the server costs a round-trip chain of ~14 us against ~7 us for one launch.
enum { QSUB_DEFER_WAIT = 1, QSUB_HYBRID = 2 };
// n chains (1..64) behind ONE gate kernel: the frames of one tick (several cameras' pictures written by the work in front of the call,
// several crop lists of one picture).  chains[i] / planes[i] / n_planes[i]; tickets[i] out; *n_queued = how many the server took (the
// rest -- return 2 -- is the caller's to launch directly, in order, behind what was queued).
//
// (A budget on the tasks that may sit behind closed gates was implemented in round 4 and removed in round 5: it bounded a batch's p90
// latency with four strict streams but halved what ticks absorb -- 2.5 -> 8.9 us per batch -- and was never on by default: HISTORY.md.)
// DEAD-LOCK FREEDOM needs no such bound: a group is published and its gate kernel enqueued under ONE lock (gate_mu), so ring order
// == the order of the gate kernels inside every hardware queue.  The earliest incomplete batch of the ring is then either open -- its
// tasks were all drawn before any later batch's, by workers that are executing them -- or closed with nothing but completed gate
// kernels and the caller's own producers in front of its gate kernel: it always makes progress.
static void prune_closed(Queue* q) { // (under q->mu)
    size_t w = 0;
    for (size_t i = 0; i < q->closed.size(); ++i) {
        const Queue::Closed c = q->closed[i];
        if (*(volatile uint64_t*)(q->hgates + c.ticket % q->R) >= c.ticket + 1 || c.ticket < q->failed_upto) q->closed_tasks -= c.tasks;
        else q->closed[w++] = c;
    }
    q->closed.resize(w);
}
int queue_submit_on(Queue* q, const ChainArgs* const* chains, const PlaneParams* const* planes, const int* n_planes, int n, void* stream, uint32_t flags,
                    uint64_t* tickets, int* n_queued, std::string& err) {
    if (n_queued) *n_queued = 0;
    if (n < 1 || n > 64) { err = "queue: 1..64 chains behind one gate"; return 1; }
    for (int i = 0; i < n; ++i)
        if (n_planes[i] > kQMaxPlanes) { err = "queue: a stream-ordered batch holds at most 74 planes"; return 1; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
        err = "queue: the stream is capturing (the ring is written at submit time; capture cvgs_execute instead)";
        return 1;
    }
    const bool defer = (flags & QSUB_DEFER_WAIT) != 0, hybrid = (flags & QSUB_HYBRID) != 0;
    void* const key = stream ? stream : (void*)q; // (the null stream is a stream too)
    // A caller stream at the SERVER's own priority (the greatest: torch.cuda.Stream(priority=-1) is exactly that on ROCm) may be multiplexed
    // onto the server's hardware queue, where its gate kernel would sit behind the long-lived server grid and never start -- the batch
    // would wait the 10 s gate limit and end in error 3.  Not a stream the server takes: the hybrid policy launches directly.
    // One hipStreamGetPriority per stream key, cached.
    if (stream && q->server_prio_known) {
        // asked anew on every call (a cache keyed by the stream handle could answer for a destroyed stream whose handle the runtime
        // has handed out again at another priority; the query is cheap beside the gate kernel's launch)
        int prio = 0;
        const bool same = hipStreamGetPriority((hipStream_t)stream, &prio) == hipSuccess && prio == q->server_prio && q->prio_range_nonempty;
        if (same) {
            err = "queue: the stream has the server's own (highest) priority and may share its hardware queue -- its gate kernel could never start; "
                  "use a default-priority stream, or the hybrid policy's direct launches";
            return 1;
        }
    }
    int done = 0;
    while (done < n) {
        // ---- how many of the remaining chains go behind the next gate kernel, and with which task size ----
        uint32_t rows = kQRowsPerWave;
        int take = 0;
        {
            std::unique_lock<std::mutex> lock(q->mu);
            advance_done(q);
            prune_closed(q);
            // what can overlap with these batches: a strictly ordered stream has ONE gate open at a time however far its host has run
            // ahead, so "batches in flight" says nothing (a lone strict stream first got the sustained stream's 128-row tasks: 44 us
            // per batch); it is the open batches of OTHER strict streams and the rest of this group.  Deferred waits: the caller's
            // trailing distance, which this call cannot see -- 16-row tasks.
            uint64_t parallel = (uint64_t)(n - done - 1);
            bool overlap = parallel > 0;
            if (defer) {
                overlap = overlap || q->next_seq > q->done_inorder;
                parallel += q->next_seq - q->done_inorder > 7 ? 7 : q->next_seq - q->done_inorder;
            } else {
                for (const auto& st : q->stream_tail)
                    if (st.stream != key && st.ticket >= q->done_inorder && hflag(q, st.ticket % q->R) < st.ticket + 1) { ++parallel; overlap = true; }
            }
            // The latency policy, from the measurements (tools/probes/stream_ordered_rate.py; DESIGN 4 "Round 4"): stream order costs one
            // launch per gate, so a gate in front of FEWER than `min_group` chains (default 8) never beats launching them directly --
            // lone strict stream 14-16 us on the server against 8-9 us as launches, single submits on 4-16 streams 10-16 against 8-9,
            // ticks of 4 frames 3.8-10 (runtime-dependent), ticks of 8 and more 3.1 -> 2.4 us per frame.  And a batch nothing in flight
            // could overlap with is a launch whatever its size.
            const int min_group = (int)((flags >> 8) & 0xffu) ? (int)((flags >> 8) & 0xffu) : 8;
            if (hybrid && (!overlap || n - done < min_group)) { ++q->n_direct; if (n_queued) *n_queued = done; return 2; }
            rows = parallel >= 8 ? (uint32_t)kQRowsPerTaskMid : (parallel >= 1 ? (uint32_t)kQRowsPerTask : (uint32_t)kQRowsPerWave);
            // (at most half the ring behind ONE gate kernel: the chains are published before their gate kernel is enqueued, so a group
            //  larger than the ring would wait for its own first slot to complete -- behind a gate nobody has launched yet)
            const int max_take = q->R >= 4 ? (int)(q->R / 2) : 1;
            take = n - done;
            if (take > 64) take = 64;
            if (take > max_take) take = max_take;
        }
        // ---- publish them closed, then ONE gate kernel on the caller's stream -- under one lock: ring order == gate-kernel order ----
        std::lock_guard<std::mutex> gate_lock(q->gate_mu);
        QGateTickets tk;
        tk.n = 0;
        int rc = 0;
        for (int i = 0; i < take; ++i) {
            uint64_t t = 0;
            GateInfo g{key, defer, rows, 0};
            rc = queue_submit_slot(q, *chains[done + i], planes[done + i], n_planes[done + i], &t, err, &g);
            if (rc) break; // (what has been published still gets its gate opened below)
            tk.t[tk.n++] = t;
            if (tickets) tickets[done + i] = t;
        }
        if (tk.n > 0) {
            hipLaunchKernelGGL(k1q_gate, dim3(1), dim3(64), 0, (hipStream_t)stream, q->m.gates, q->hgates, (const uint64_t*)q->m.dflags, q->R, tk, defer ? 0u : 1u,
                               (const uint64_t*)&q->hc->error.v, q->gate_ticks, q->gate_trace);
            if (hipGetLastError() != hipSuccess) { // never leave workers at a gate nobody will open
                for (uint32_t i = 0; i < tk.n; ++i) {
                    uint64_t* gate = q->m.gates + kQCtrStride * (tk.t[i] % q->R);
                    if (q->direct) *(volatile uint64_t*)gate = tk.t[i] + 1;
                    else hipLaunchKernelGGL(k1q_gate_open, dim3(1), dim3(64), 0, q->stage_stream, gate, tk.t[i] + 1);
                    *(volatile uint64_t*)(q->hgates + tk.t[i] % q->R) = tk.t[i] + 1;
                }
                if (q->direct) wc_fence();
                err = "queue: the gate kernel could not be launched on the caller's stream";
                if (n_queued) *n_queued = done;
                return -1;
            }
            std::lock_guard<std::mutex> lock(q->mu);
            q->n_gated += tk.n;
            if (!defer) {
                bool found = false;
                for (auto& st : q->stream_tail)
                    if (st.stream == key) { st.ticket = tk.t[tk.n - 1]; found = true; break; }
                if (!found) {
                    if (q->stream_tail.size() >= 256) q->stream_tail.erase(q->stream_tail.begin());
                    q->stream_tail.push_back({key, tk.t[tk.n - 1]});
                }
            }
        }
        done += (int)tk.n;
        if (n_queued) *n_queued = done;
        if (rc) return rc;
    }
    return 0;
}

// After the watchdog has fired (error word set: another kernel held the chip beyond the stall limit, a gate that never opened, ...) the
// queue used to be dead for good: every later submit failed, nothing reset the error.  Recovery: wait for the failed
// server to leave, declare the batches that were in flight LOST (waits on their tickets report it; stream waiters are released),
// renumber from a clean task base -- every counter of the protocol back to its initial state -- and clear the error.  The next
// submit launches a fresh server.
int queue_recover(Queue* q, uint64_t* lost, std::string& err) {
    std::lock_guard<std::mutex> lock(q->mu);
    if (lost) *lost = 0;
    if (!hv(q->hc->error)) return 0;
    (void)hipStreamSynchronize(q->stage_stream);
    if (hipStreamSynchronize(q->stream) != hipSuccess) { err = "queue: the failed server did not leave"; return -1; }
    advance_done(q);
    uint64_t n_lost = 0;
    for (uint64_t b = q->done_inorder; b < q->next_seq; ++b)
        if (hflag(q, b % q->R) < b + 1) { // (a batch the late workgroups still finished is complete, not lost)
            ++n_lost;
            if (q->lost_tickets.size() >= 4096) q->lost_tickets.erase(q->lost_tickets.begin());
            q->lost_tickets.push_back(b);
        }
    q->failed_upto = q->next_seq;
    const size_t R = q->R, NW = (size_t)q->G * kQWaves;
    const uint64_t base = (q->next_task + (kQSubs - 1)) & ~(uint64_t)(kQSubs - 1);
    q->next_task = base;
    std::vector<uint64_t> p0(NW), tk((size_t)kQSubs * kQCtrStride, 0), fl(R * kQCtrStride, 0);
    for (size_t i = 0; i < NW; ++i) p0[i] = base + i;
    for (int c = 0; c < kQSubs; ++c) tk[(size_t)c * kQCtrStride] = base / kQSubs;
    // completion words: every batch ever submitted reads as complete (the lost ones are reported through their tickets)
    for (uint64_t b = q->next_seq >= R ? q->next_seq - R : 0; b < q->next_seq; ++b) {
        fl[(b % R) * kQCtrStride] = b + 1;
        *(volatile uint64_t*)(q->hflags + (b % R)) = b + 1;
    }
    const size_t ctr_bytes = R * (1 + kQSubs) * kQCtrStride * 8;
    hipError_t e;
    if ((e = hipMemset(q->m.arrive, 0, ctr_bytes)) != hipSuccess || (e = hipMemcpy(q->m.prog, p0.data(), NW * 8, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(q->m.ticket, tk.data(), tk.size() * 8, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(q->m.dflags, fl.data(), fl.size() * 8, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(q->m.gates, fl.data(), fl.size() * 8, hipMemcpyHostToDevice)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) {
        err = std::string("queue recovery: ") + hipGetErrorString(e);
        return -1;
    }
    std::fill(q->arrive_cum.begin(), q->arrive_cum.end(), 0);
    q->done_inorder = q->next_seq;
    raise_done_hint(q, q->next_seq);
    q->stream_tail.clear();
    q->closed.clear();
    q->closed_tasks = 0;
    hv(q->hc->yield_req) = 0;
    hv(q->hc->error) = 0;
    hv(q->hc->state) = QS_IDLE;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (lost) *lost = n_lost;
    return 0;
}

const uint64_t* queue_gate_trace(Queue* q) { return q->gate_trace; }

void queue_prof(Queue* q, uint64_t* out16) {
    for (int i = 0; i < 16; ++i) out16[i] = *(volatile uint64_t*)&q->hc->prof[i];
    out16[6] = q->n_ring_waits;
    out16[7] = q->n_ring_waits ? q->sum_done_behind_head / q->n_ring_waits : 0;
    out16[13] = q->n_sub ? q->ns_ring_wait / q->n_sub : 0; // host side of submit, ns per call (since create)
    out16[14] = q->n_gated | (q->n_direct << 32); // stream-ordered submits: taken by the server | launched directly by the latency policy
    out16[15] = 0;
    out16[2] = out16[3] = 0; // (reserved: the closed-batch budget these two slots reported was removed in round 5 -- ADVICE r5)
}

void queue_stats(Queue* q, uint64_t* out8) {
    std::lock_guard<std::mutex> lock(q->mu);
    advance_done(q);
    out8[0] = q->next_seq;
    out8[1] = q->done_inorder;
    out8[2] = q->launches;
    out8[3] = hv(q->hc->stat_rounds);
    out8[4] = hv(q->hc->stat_launch_ticks); // 100 MHz ticks the last retired server lived
    out8[5] = q->G;
    out8[6] = q->R | (q->direct ? 1ull << 32 : 0);
    out8[7] = hv(q->hc->error);
}

void* queue_stream(Queue* q) { return (void*)q->stream; }

int queue_destroy(Queue* q) {
    if (!q) return 0;
    // the batches in flight complete first (the janitor retires the grid as soon as it sees the stop word, whatever is in the ring);
    // bounded: a stalled server or a gate that never opens must not hang the caller
    if (q->next_seq > q->failed_upto && !hv(q->hc->error)) {
        std::string ignored;
        (void)queue_wait(q, q->next_seq - 1, 2.0, ignored);
    }
    {
        std::lock_guard<std::mutex> lock(q->mu);
        hv(q->hc->stop_req) = 1;
        std::atomic_thread_fence(std::memory_order_seq_cst);
    }
    (void)hipStreamSynchronize(q->stage_stream);
    (void)hipStreamSynchronize(q->stream); // the janitor sees stop_req within one round and retires the grid
    {
        std::lock_guard<std::mutex> owner_lock(g_server_mu);
        if (g_server_owner[q->device & 63] == q) g_server_owner[q->device & 63] = nullptr;
    }
    // Gate / wait kernels of stream-ordered submits may still sit on the CALLERS' streams (behind a long producer, or holding a stream on
    // a batch that will never complete now): they read the gate / completion words and the host error word freed below.  The error word
    // releases the ones that are waiting (they look at it every 64 polls), and the device is drained before anything is freed.
    bool gates_pending;
    {
        std::lock_guard<std::mutex> lock(q->mu);
        prune_closed(q);
        gates_pending = !q->closed.empty() || q->n_gated > 0;
    }
    if (gates_pending) {
        if (!hv(q->hc->error)) hv(q->hc->error) = 4; // "destroyed"
        std::atomic_thread_fence(std::memory_order_seq_cst);
        (void)hipDeviceSynchronize();
    }
    (void)hipStreamDestroy(q->stream);
    (void)hipStreamDestroy(q->stage_stream);
    (void)hipFree(q->dev_block);
    (void)hipFree(q->dev_counters);
    (void)hipHostFree((void*)q->hc);
    (void)hipHostFree((void*)q->hflags);
    (void)hipHostFree((void*)q->hgates);
    if (q->gate_trace) (void)hipHostFree(q->gate_trace);
    if (q->host_ring) (void)hipHostFree(q->host_ring);
    if (q->host_index) (void)hipHostFree((void*)q->host_index);
    delete q;
    return 0;
}

} // namespace cvgs
