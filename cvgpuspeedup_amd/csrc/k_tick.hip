// k_tick.hip -- K1 for TICKS: the chains of one cvgs_execute_many call (a tick: the frames of several cameras, each with its crop list
// and its output tensor) served by ONE ordinary launch whose waves WALK the tick's tasks the way the descriptor queue's server walks its
// ring -- without anything resident between two ticks.
//
// Why (DESIGN.md "submission"): the reference's call shape is one executeOperations per frame
// (include/cvGPUSpeedup.cuh:464-473); a serving loop with several cameras has a tick's worth of such calls whose chains share one shape.
// cvgs_execute_many's first kernel gave every 4 x 64 tile its own workgroup (grid z = chain: 51,200 two-wave workgroups for 16 x 50
// crops): all of them start in the same phase, each pays its kernel-argument and descriptor round trips, rows leave as one dword per
// lane, and the descriptor table's slot was recycled through a HIP event that kept the stream's next kernel ~4 us behind.  Here:
//   * a grid of G workgroups x 4 INDEPENDENT worker waves (no barrier), G = a few per CU; a task = R output rows x 64 columns of one
//     crop in ONE numbering over all chains of the tick; a worker holds a task number and draws the next one from one of 16 ticket
//     counters (its residue class: one word takes ~88 returning atomics per microsecond) BEFORE it processes the one it holds, so the
//     draw's round trip hides behind the rows; work goes to whoever is free: chain k+1's loads overlap chain k's stores;
//   * the row worker is the server's (k_rows.hpp: k1q_rows -- lane = output column for taps and arithmetic, the 4 x 64 tile transposed
//     through a wave-private LDS tile so that 16 consecutive lanes store one row's 256 contiguous bytes as 16-byte stores), with the
//     flavours of an ordinary launch: plain cached tap loads, non-temporal stores -- the kernel boundary publishes;
//   * the LAST worker to leave resets the counters (the block is clean for the stream's next tick: kernels of one stream never overlap)
//     and stores the launch's sequence number into a pinned host word: the host recycles the descriptor table's slot by reading that
//     word -- no HIP event, no marker packet behind the launch, nothing touched that a destroyed stream could invalidate.
// Bit-identical to k1_resize_split (tests/test_gpu_tick.py compares both with the oracle and with each other).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "k_rows.hpp"

namespace cvgs {

constexpr int kTickClasses = 16;   // ticket counters per launch
constexpr int kTickCtrStride = 16; // ... 128 bytes apart (uint64 words)

template <int NS>
struct TickArgs {
    float mul[4], sub[4], div[4], rdiv[4], bg[4]; // (with the chain's R <-> B swap: channels 0 and 2 already exchanged)
    int32_t dst_w, dst_h, out_w, cn, swap, fast_div, out_half, n_segs;
    uint32_t tpp, col_tiles, rows_per_task, n_tasks, n_workers, pad;
    int64_t img_stride, ch_stride; // output elements
    uint64_t* counters;            // device: [0..15] ticket counters, [16..31] exit counters, [32] top -- kTickCtrStride words apart, all 0 between launches
    uint64_t* done_word;           // pinned host word (or null): receives `seq` when the launch's last worker has left
    uint64_t seq;
    TickSeg seg[NS];
};
static_assert(sizeof(TickArgs<16>) <= 1024 && sizeof(TickArgs<128>) <= 4096 + 512, "tick kernel arguments");

template <int NS, int ST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k1_tick(const TickArgs<NS> a) {
    __shared__ __attribute__((aligned(16))) float q_tiles[kQWaves * kQLdsWave]; // one transpose tile per wave (20 KB per workgroup)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t wid = blockIdx.x * kQWaves + (uint32_t)wave;
    const uint32_t cls = wid & (kTickClasses - 1);
    const uint32_t n_workers = a.n_workers, n_tasks = a.n_tasks;
    const uint64_t cls_first = (n_workers - cls + kTickClasses - 1) / kTickClasses; // tickets cls, cls + 16, ... < n_workers are the workers' initial ones
    uint64_t* const ticket = a.counters + cls * kTickCtrStride;
    uint64_t T = wid;
#pragma nounroll
    while (T < n_tasks) {
        // ---- the NEXT ticket, requested before this task's rows and consumed after them ----
        uint64_t drawn = 0;
        if (lane == 0) drawn = __hip_atomic_fetch_add((g_u64)ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- which chain, which crop, which tile ----
        const uint32_t t32 = (uint32_t)T;
        const uint32_t gp = t32 / a.tpp, rt = t32 - gp * a.tpp; // plane in the tick's numbering, tile inside the plane
        int s = 0;
        for (int base = 0; base < a.n_segs; base += 64) { // segments are ordered by their first plane: count the ones that start at or below gp
            const int i = base + lane;
            const bool le = i < a.n_segs && a.seg[i < NS ? i : NS - 1].plane0 <= gp;
            s += __builtin_popcountll(__builtin_amdgcn_ballot_w64(le));
        }
        s = __builtin_amdgcn_readfirstlane(s - 1);
        const TickSeg sg = a.seg[s];
        const uint32_t z = gp - sg.plane0;
        const uint32_t row_tile = rt / a.col_tiles, col_tile = rt - row_tile * a.col_tiles;
        QTask t;
        t.tile = q_tiles + wave * kQLdsWave;
        {   // the crop's 12 descriptor dwords, one per lane (one vector load; fields by v_readlane -- wave-uniform values in SGPRs)
            const uint32_t* pp = (const uint32_t*)(sg.table + ((int)z < sg.used ? z : 0));
            const uint32_t pv = pp[lane < 12 ? lane : 0];
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
            t.P.uv_off = 0;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t.mul[c] = a.mul[c];
            t.sub[c] = a.sub[c];
            t.div[c] = a.div[c];
            t.rdiv[c] = a.rdiv[c];
            t.bg[c] = a.bg[c];
        }
        t.used = sg.used;
        t.dst_w = a.dst_w;
        t.dst_h = a.dst_h;
        t.out_w = a.out_w;
        t.swap = a.swap;
        t.fast_div = a.fast_div;
        t.img_stride = a.img_stride;
        t.ch_stride = a.ch_stride;
        t.out = sg.out;
        t.out_bytes = sg.out_bytes;
        t.out_half = a.out_half;
        t.yuv_range = t.yuv_primaries = t.yuv_vu = 0;
        const bool c3 = a.cn == 3;
        const int rows_per_task = (int)a.rows_per_task;
        QRowGeo geo{};
#pragma nounroll
        for (int grp = 0; grp * kQRowsPerWave < rows_per_task; ++grp) {
            const int row0 = (int)row_tile * rows_per_task + grp * kQRowsPerWave;
            if (row0 >= t.dst_h) break;
            const int gi = (grp & 15) * kQRowsPerWave;
            if (gi == 0) geo = q_row_geo(t.P, t.dst_h, row0, lane); // the next 64 rows' vertical geometry, one row per lane
            if (c3) k1q_rows<3, 0, ST, SRC_U8, true>(t, (int)z, (int)col_tile, row0, lane, geo, gi);
            else k1q_rows<4, 0, ST, SRC_U8, true>(t, (int)z, (int)col_tile, row0, lane, geo, gi);
        }
        T = (uint64_t)cls + (uint64_t)kTickClasses * (cls_first + q_uni(drawn));
    }
    // ---- leave: the last worker of the launch resets the counters and tells the host ----
    uint64_t before = 0;
    if (lane == 0) before = __hip_atomic_fetch_add((g_u64)(a.counters + (kTickClasses + cls) * kTickCtrStride), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (q_uni(before) + 1 != cls_first) return; // (cls_first == the number of workers of this class)
    if (lane == 0) before = __hip_atomic_fetch_add((g_u64)(a.counters + 2 * kTickClasses * kTickCtrStride), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t classes_alive = n_workers < (uint32_t)kTickClasses ? n_workers : (uint32_t)kTickClasses;
    if (q_uni(before) + 1 != classes_alive) return;
    // every worker has left: nobody draws or reads a table any more
    if (lane <= 2 * kTickClasses) q_st(a.counters + lane * kTickCtrStride, 0);
    q_drain();
    if (a.done_word && lane == 0) q_st_sys(a.done_word, a.seq);
}

static int tick_env(const char* name, int fallback, int lo, int hi) {
    const char* e = getenv(name);
    if (!e || !*e) return fallback;
    const int v = atoi(e);
    return v < lo ? lo : (v > hi ? hi : v);
}

template <int NS>
static hipError_t tick_launch_ns(const TickArgs<128>& full, int st, unsigned grid, hipStream_t s) {
    TickArgs<NS> a;
    std::memcpy((void*)&a, (const void*)&full, offsetof(TickArgs<NS>, seg));
    for (int i = 0; i < NS; ++i) a.seg[i] = i < full.n_segs ? full.seg[i] : TickSeg{nullptr, nullptr, 0, 0, 0xffffffffu, 0};
    if (st == 0) hipLaunchKernelGGL((k1_tick<NS, 0>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k1_tick<NS, 3>), dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}

int tick_counter_words() { return (2 * kTickClasses + 1) * kTickCtrStride; }
static std::atomic<uint64_t> g_tick_launches{0};
uint64_t tick_launches() { return g_tick_launches.load(std::memory_order_relaxed); }

int launch_k1_tick(const ChainArgs& c_in, const TickSeg* segs, int n_segs, const TickLaunch& tl, void* stream, bool dry_run, LaunchInfo* info) {
    static const int enabled = tick_env("CVGS_TICK", 1, 0, 1);
    if (!enabled) return 0;
    const ReadArgs& r = c_in.read;
    const WriteArgs& w = c_in.write;
    if (r.kind != CVGS_READ_RESIZE_LINEAR || r.depth != CVGS_DEPTH_8U || (r.cn != 3 && r.cn != 4)) return 0;
    if (w.kind != CVGS_WRITE_TENSOR_SPLIT && w.kind != CVGS_WRITE_TENSOR_T_SPLIT) return 0;
    if (w.data2 || n_segs < 1 || n_segs > CVGS_MAX_CHAINS || !segs) return 0;
    const bool f16 = w.depth == CVGS_DEPTH_16F;
    if (!f16 && w.depth != CVGS_DEPTH_32F) return 0;
    ProgArgs prog = c_in.prog;
    if (f16) {
        if (prog.n < 1 || prog.opcode[prog.n - 1] != CVGS_OP_CAST) return 0;
        --prog.n; // the trailing CAST(CV_16F) happens in the store
    }
    for (int k = 0; k < prog.n; ++k)
        if (prog.opcode[k] == CVGS_OP_CAST || prog.opcode[k] == CVGS_OP_CAST_TRUNC) return 0;
    const int prog_id = k1_classify_program(prog, r.cn);
    if (prog_id > 1) return 0; // [swap] mul sub div only: anything else keeps the grid kernel and its interpreter
    prog.fast_div = 0;
    for (int k = 0; k < 4; ++k) prog.rdiv[k] = 0.f;
    fast_div_setup(prog, prog_id == 0 ? 3 : 2, prog_id == 0 ? 1 : 0, r.cn, r.bg);
    const int at = prog_id == 0 ? 1 : 0; // index of MUL
    TickArgs<128> a;
    std::memset((void*)&a, 0, sizeof(a));
    for (int ch = 0; ch < 4; ++ch) {
        a.mul[ch] = prog.operand[at][ch];
        a.sub[ch] = prog.operand[at + 1][ch];
        a.div[ch] = prog.operand[at + 2][ch];
        a.rdiv[ch] = prog.rdiv[ch];
        a.bg[ch] = r.bg[ch];
    }
    a.swap = prog_id == 0;
    if (a.swap) {
        // The swap is not executed per pixel: the stages after it are per-channel, so "swap, then stage k with operand[c] on channel c"
        // == "stage k with operand[2 - c] on the UN-swapped channel c, stored into plane 2 - c" (k1q_rows stores channel k to plane
        // ch_off[k]).  The background is a value of the READ stage -- in front of the swap -- and runs through the same program: untouched.
        std::swap(a.mul[0], a.mul[2]);
        std::swap(a.sub[0], a.sub[2]);
        std::swap(a.div[0], a.div[2]);
        std::swap(a.rdiv[0], a.rdiv[2]);
    }
    a.fast_div = prog.fast_div;
    a.dst_w = r.dst_w;
    a.dst_h = r.dst_h;
    a.out_w = w.width;
    a.cn = r.cn;
    a.out_half = f16;
    a.n_segs = n_segs;
    a.img_stride = w.img_stride;
    a.ch_stride = w.ch_stride;
    const int esz = f16 ? 2 : 4;
    uint64_t planes = 0;
    for (int i = 0; i < n_segs; ++i) {
        if (segs[i].batch < 1 || !segs[i].table || !segs[i].out) return 0;
        // the chain's extent in bytes (CNHW: channel cn-1 of image batch-1 lies (cn-1) * ch_stride + (batch-1) * img_stride + one plane in)
        const uint64_t plane = (uint64_t)w.width * (uint64_t)w.height;
        const uint64_t bytes = ((uint64_t)(r.cn - 1) * (uint64_t)w.ch_stride + (uint64_t)(segs[i].batch - 1) * (uint64_t)w.img_stride + plane) * (uint64_t)esz;
        if (bytes >= (1ull << 32)) return 0; // the row worker addresses a tensor through a buffer descriptor with 32-bit offsets
        a.seg[i] = segs[i];
        a.seg[i].plane0 = (uint32_t)planes;
        a.seg[i].out_bytes = (uint32_t)bytes;
        planes += (uint64_t)segs[i].batch;
    }
    if (info) {
        static const char* names[2][2][2] = {{{"k1_tick_u8c3_swap_mul_sub_div", "k1_tick_u8c3_mul_sub_div"}, {"k1_tick_u8c4_swap_mul_sub_div", "k1_tick_u8c4_mul_sub_div"}},
                                            {{"k1_tick_u8c3_swap_mul_sub_div_f16", "k1_tick_u8c3_mul_sub_div_f16"}, {"k1_tick_u8c4_swap_mul_sub_div_f16", "k1_tick_u8c4_mul_sub_div_f16"}}};
        info->kernel = names[f16][r.cn == 4][prog_id];
    }
    // task size: rows per task.  Small enough that every worker gets several tasks (the tail of the launch is one task long), large
    // enough that the per-task dispatch (ticket, segment, the crop's descriptor: two dependent round trips) stays a small part of it.
    static const int rows_env = tick_env("CVGS_TICK_ROWS", 0, 0, 4096) & ~3;
    static const int wgs_env = tick_env("CVGS_TICK_WGS_PER_CU", 0, 0, 8);
    static const int st_env = tick_env("CVGS_TICK_ST", 3, 0, 3);
    int cus = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        static int cached_cus[64] = {0};
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            if (!cached_cus[dev] && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached_cus[dev] = prop.multiProcessorCount;
            if (cached_cus[dev]) cus = cached_cus[dev];
        }
    }
    const int wgs_per_cu = wgs_env ? wgs_env : 4;
    const uint64_t col_tiles = (uint64_t)((r.dst_w + 63) / 64);
    uint64_t max_workers = (uint64_t)cus * wgs_per_cu * kQWaves;
    int R = rows_env;
    if (!R) { // >= 6 tasks per worker where the tick is large enough, tasks of 4 .. 32 rows
        R = 32;
        while (R > 4 && planes * col_tiles * (uint64_t)((r.dst_h + R - 1) / R) < 6 * max_workers) R >>= 1;
    }
    const uint64_t tpp = col_tiles * (uint64_t)((r.dst_h + R - 1) / R);
    const uint64_t n_tasks = planes * tpp;
    if (n_tasks >= (1ull << 31)) return 0;
    a.tpp = (uint32_t)tpp;
    a.col_tiles = (uint32_t)col_tiles;
    a.rows_per_task = (uint32_t)R;
    a.n_tasks = (uint32_t)n_tasks;
    uint64_t grid = (n_tasks + kQWaves - 1) / kQWaves; // never more workers than tasks
    if (grid > (uint64_t)cus * wgs_per_cu) grid = (uint64_t)cus * wgs_per_cu;
    a.n_workers = (uint32_t)(grid * kQWaves);
    if (dry_run) return 1;
    if (!tl.counters) return 0;
    a.counters = tl.counters;
    a.done_word = tl.done_word;
    a.seq = tl.seq;
    hipStream_t s = (hipStream_t)stream;
    const hipError_t e = n_segs <= 16 ? tick_launch_ns<16>(a, st_env, (unsigned)grid, s)
                         : (n_segs <= 64 ? tick_launch_ns<64>(a, st_env, (unsigned)grid, s) : tick_launch_ns<128>(a, st_env, (unsigned)grid, s));
    if (e == hipSuccess) g_tick_launches.fetch_add(1, std::memory_order_relaxed);
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

} // namespace cvgs
