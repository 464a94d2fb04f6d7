// cvgs_device.h -- structs shared by the host-side lowering (cvgs_api.cpp) and the HIP kernels.
// Everything here crosses the host->device boundary BY VALUE inside the kernel-argument block,
// exactly like the reference passes its IOp parameters, so one fused chain is one
// launch with no side uploads.  Large crop lists use a device-resident plane table instead.
#pragma once

#include <stdint.h>

#include <string>

#include "../../include/cvgs_hip_ext.h" // (includes cvgs_hip.h; the engine also implements the queue / exchange extensions)

namespace cvgs {

// Per-plane read parameters, precomputed on the host in double precision so that the device never
// re-derives a scale factor (bit-exactness of fx/fy is part of the parity contract).
struct PlaneParams {        // 48 bytes
    const uint8_t* data;    // pixel (0,0) of the crop / image
    int32_t w, h;           // source extent in pixels (NV12: luma extent)
    int32_t step;           // bytes between source rows
    float fx, fy;           // source step per destination pixel (resize kinds)
    int32_t x1, y1, x2, y2; // inclusive destination window fed from the source (AR modes)
    int32_t uv_off;         // NV12: bytes from `data` to the UV row of luma row 0 (h * step for a whole surface)
};
static_assert(sizeof(PlaneParams) == 48, "PlaneParams layout");

// Device plane table = PlaneParams[batch]  (cvgs_plane_table_build).

struct DstPlane {           // SPLIT_2D / PIXEL_2D_BATCH targets, 16 bytes
    uint8_t* data;
    int32_t step;
    int32_t pad;
};

struct ReadArgs {
    int32_t kind;
    int32_t depth, cn;      // source pixel type
    int32_t batch, used;
    int32_t dst_w, dst_h;   // extent of one output plane
    int32_t is_resize;
    float bg[4];
    int32_t yuv_range, yuv_primaries, yuv_alpha;
    int32_t out_cn;         // channels produced by the read stage
    int32_t yuv_layout;     // cvgs_yuv_layout
    int32_t pad;
    const PlaneParams* table; // device table, or nullptr -> planes inline in the kernel args
};

struct ProgArgs {
    int32_t n;
    int32_t opcode[CVGS_MAX_OPS];
    int32_t aux[CVGS_MAX_OPS];
    float operand[CVGS_MAX_OPS][4];
    // K1's compile-time programs only (set by launch_k1, 0 elsewhere): the DIV stage's divisors are wave-uniform, so
    // their correctly rounded reciprocals are computed ONCE on the host (rdiv[c] = 1.0f / operand[c], IEEE) and the
    // kernel divides with two FMA correction steps (k_taps.hpp: div_by_uniform) -- same bits as the IEEE division.
    int32_t fast_div;
    float rdiv[4];
};

struct WriteArgs {
    int32_t kind;
    int32_t depth, cn;      // type of the value written
    int32_t width, height;  // plane extent
    int32_t step;           // PIXEL_2D pitch (bytes)
    int32_t planes;         // tensor N (TensorTSplit stride)
    int32_t pad;
    uint8_t* data;
    const DstPlane* table;  // device table for SPLIT_2D / PIXEL_2D_BATCH beyond the inline ones
    // Tensor kinds: element strides of the primary target, and an optional SECOND target that receives the same
    // values (CircularTensor: the new frame goes to the history ring and to the ordered tensor in one pass).
    int64_t img_stride, ch_stride;
    uint8_t* data2;
    int64_t img_stride2, ch_stride2;
};

static constexpr int kInlineDst = 16; // inline destination planes (e.g. batch 4 x 4 channels)

struct ChainArgs {
    ReadArgs read;
    ProgArgs prog;
    WriteArgs write;
    DstPlane dst_inline[kInlineDst];
};

template <int NPLANES>
struct KernArgs {
    ChainArgs c;
    PlaneParams planes[NPLANES];
};
// NPLANES == 0: the planes come from the device table; one dummy slot keeps the struct non-empty.
template <>
struct KernArgs<0> {
    ChainArgs c;
    PlaneParams planes[1];
};
static_assert(sizeof(KernArgs<CVGS_KERNARG_PLANES>) <= 4096, "kernel-argument block must fit 4 KB");
// K1's planar-tensor kernels also exist with a LARGE kernel-argument block: the reference's own benchmark sweeps the batch to
// 300 crops per call (tests/batchresize/test_batchresize_x_split3D.cu:384-392), and a 16 KB argument block costs ~2 us more
// host enqueue time than a 4 KB one, while staging the descriptors in device memory costs a copy, an event and a stream wait
// (~18 us per eager call) and cannot be captured into a HIP graph.
static constexpr int kKernargPlanesBig = CVGS_KERNARG_PLANES_MAX;
static_assert(sizeof(KernArgs<kKernargPlanesBig>) <= 16384, "large kernel-argument block: 16 KB");

// Extra write targets (cvgs_write_desc.mirrors): the same values at the same element offsets in up to 7 more tensors
// (the peers' copies of the sharded [N,C,H,W] tensor).  Travels as its own kernel argument to the kernels that
// implement it (K1 planar inside K1Geom, the interpreted kernel).
struct MirrorArgs {          // 64 bytes
    uint8_t* p[CVGS_MAX_MIRRORS];
    int32_t n;
    int32_t pad;
};

struct ManySeg;
// What a K1 / K4 launch takes beside its chain: ONE explicit argument from the C-ABI call down to the launch site (round 6; rounds 4-5 passed
// these through three thread-local slots, and a launch site that forgot to consume one fell back to a stream synchronise).
struct LaunchCtx {
    void* stream = nullptr;        // hipStream_t
    MirrorArgs mirrors{};          // extra write targets (K1's planar kernels)
    const ManySeg* segs = nullptr; // the chains of a cvgs_execute_many launch (nullptr: one chain)
    int n_segs = 0;
    // The launch that reads a pooled descriptor table signals the slot's event ITSELF (hipExtLaunchKernelGGL's stopEvent: the kernel packet's own
    // completion signal) instead of a hipEventRecord behind it -- a marker packet that kept the NEXT kernel of the stream ~4 us behind.  Set by
    // the call that uploaded the table; a launch site that attaches it says so in stop_event_taken (the others get an event recorded behind them).
    void* stop_event = nullptr;    // hipEvent_t
    bool stop_event_taken = false;
    // The fused launch of cvgs_execute_many reports its stream's progress itself: its first work-item stores done_value (the sequence number
    // of the stream's PREVIOUS fused launch, which has finished once this kernel runs) into the pinned word done_word (cvgs_api.cpp: ManyPool).
    uint64_t* done_word = nullptr;
    uint64_t done_value = 0;
    bool done_word_taken = false;
    LaunchCtx() = default;
    explicit LaunchCtx(void* s) : stream(s) {}
};

// cvgs_execute_many: one segment per fused chain.  The K1 kernel's table variants ALWAYS read their planes through a
// segment (a single chain with a device table is one segment), so fusing chains adds no kernel variant.
struct ManySeg {             // 24 bytes
    const PlaneParams* table; // device: PlaneParams[batch] of this chain
    uint8_t* out;             // this chain's output tensor
    int32_t batch, used;
};
struct KernArgsMany {
    ChainArgs c;
    ManySeg seg[CVGS_MAX_CHAINS];
};
static_assert(sizeof(KernArgsMany) <= 4096 - 256, "segment block + K1Geom must fit the kernel-argument block");
// Fused chains described on the HOST (fresh crop lists every tick): the planes of ALL chains travel in the kernel arguments behind the
// segment block (ManySeg::table then holds the chain's first index into `planes`, not an address).  The runtime places kernel arguments
// in device memory, so the kernel reads its descriptors as it reads a device table -- a pinned host table costs every XCD a PCIe round
// trip per crop and 1.75 us per 16 x 50-crop tick (tools/bench_tick.py) -- and the launch can be captured.  Two block sizes: the host
// pays for the bytes of the DECLARED block (tools/probes/big_kernarg_probe.cpp: 16 KB + 1.1 us, 52 KB + 5 us per launch; a launch may
// not supply less than the kernel declares: tools/probes/partial_kernarg_probe.cpp).
static constexpr int kManyInlineSmall = 256, kManyInlineLarge = 1024;
template <int N>
struct KernArgsManyInline {
    ChainArgs c;
    ManySeg seg[CVGS_MAX_CHAINS];
    PlaneParams planes[N];
};
static_assert(sizeof(KernArgsManyInline<kManyInlineSmall>) <= 16384 && sizeof(KernArgsManyInline<kManyInlineLarge>) <= 53248, "inline tick blocks: 16 KB / 52 KB");

// CV_64F chains: the double operands travel next to the float ones, with a small inline plane block.
struct Prog64Args {
    double operand[CVGS_MAX_OPS][4];
};
template <int NPLANES>
struct KernArgs64 {
    ChainArgs c;
    Prog64Args p64;
    PlaneParams planes[NPLANES];
};
template <>
struct KernArgs64<0> {
    ChainArgs c;
    Prog64Args p64;
    PlaneParams planes[1];
};
static constexpr int kInline64 = 8;

// ---- launch entry points implemented in the .hip files -----------------------------------------
struct LaunchInfo {
    const char* kernel; // name of the kernel variant chosen
};

// generic interpreted kernel: any valid chain
int launch_generic(const ChainArgs& c, const PlaneParams* inline_planes, int n_inline, const MirrorArgs& mirrors, void* stream,
                   bool dry_run, LaunchInfo* info);

// K1 fast path: u8 C3/C4 -> resize linear -> program -> fp32 TensorSplit / TensorTSplit.
// Returns 1 if it took the chain, 0 if not eligible, <0 on error.
// ctx.segs (n_segs >= 1): the chains of a cvgs_execute_many launch (their planes live in device tables; c.read.batch is
// the largest batch); nullptr: one chain described by c / inline_planes.
int launch_k1(const ChainArgs& c, const PlaneParams* inline_planes, int n_inline, LaunchCtx& ctx, bool dry_run, LaunchInfo* info, uint32_t chain_flags = 0);
// K1's whole-frame form for packed targets, four output pixels per lane (k_k1_x4.hip); 1 launched / 0 not eligible / < 0 error
int launch_k1_packed_x4(const ChainArgs& c, const PlaneParams* planes, int n_planes, void* stream, bool dry_run, bool force);
static constexpr int64_t kX4MinWaveRows = 12288; // output rows x 64-column tiles x images from which launch_k1 prefers it

// interpreted kernel for chains that touch CV_64F
int launch_generic64(const ChainArgs& c, const Prog64Args& p64, const PlaneParams* inline_planes, int n_inline, void* stream,
                     bool dry_run, LaunchInfo* info);

// K4 fast path: NV12 read-back fused into the bilinear resize -> program -> planar fp32 tensor or packed pixels.
int launch_nv12(const ChainArgs& c, const PlaneParams* inline_planes, int n_inline, int min_width, LaunchCtx& ctx, bool dry_run, LaunchInfo* info,
                uint32_t chain_flags = 0);
// K4 at frame size, two output pixels per lane (k_nv12_x2.hip); 1 launched / 0 not eligible / < 0 error
int launch_nv12_x2(const ChainArgs& c, const PlaneParams* planes, int n_planes, bool prog_swap, void* stream, bool dry_run);
static constexpr int64_t kK4X2MinWaveRows = 4096; // output rows x 64-column tiles x surfaces from which launch_nv12 prefers it
bool k4_planes_eligible(const PlaneParams* planes, int n, int dst_w, int dst_h);

// Thread-fused pointwise chains on u8 sources (4 pixels per thread) -> fp32 planar / packed.
// n_segs chains of ONE thread-fused pointwise shape (per-pixel reads of u8 planes -> fp32 tensor / packed fp32 pixels) in one launch: the
// chains' planes travel in the kernel arguments (segs[i].table = the chain's first index into `planes`), grid z = chain x max_batch + plane.
// 1 launched / 0 not this shape (the caller runs the chains one by one) / < 0 error.
int launch_pointwise_many(const ChainArgs& c, const PlaneParams* planes, int n_planes, const ManySeg* segs, int n_segs, int max_batch,
                          uint32_t chain_flags, void* stream, bool dry_run);
int launch_pointwise(const ChainArgs& c, const PlaneParams* inline_planes, int n_inline, uint32_t chain_flags, void* stream,
                     bool dry_run, LaunchInfo* info);

// Warp reads (cvGS::warp): per-plane source view + the inverse transform + the plane's destination size, 64 bytes.
struct WarpPlane {
    const uint8_t* data;
    int32_t w, h, step;
    float m[9];            // row-major 3x3, destination -> source; affine kinds ignore m[6..8]
    int32_t dw, dh;        // destination extent of THIS plane (cvgs_read_desc.warp_dst_sizes; else the launch's dst size)
};
static_assert(sizeof(WarpPlane) == 64, "WarpPlane layout");
static constexpr int kInlineWarp = 52; // planes whose WarpPlane travels in the kernel arguments (4 KB block)
// interpreted warp kernel: `planes` = host array of n (inline when n <= kInlineWarp), else `dev_table` (device copy)
int launch_warp(const ChainArgs& c, const WarpPlane* planes, int n, const WarpPlane* dev_table, uint32_t chain_flags, void* stream,
                bool dry_run, LaunchInfo* info);

// warp reads in front of a CV_64F program (k_generic64.hip)
static constexpr int kInlineWarp64 = 8;
int launch_warp64(const ChainArgs& c, const Prog64Args& p64, const WarpPlane* planes, int n, const WarpPlane* dev_table, void* stream,
                  bool dry_run, LaunchInfo* info);

// plane-to-plane copies of the CircularTensor update (K9): dst[i] <- src[i], `bytes` each
struct CopyJob {
    const uint8_t* src;
    uint8_t* dst;
};
static constexpr int kMaxCopyJobs = 64;
int launch_plane_copies(const CopyJob* jobs, int n_jobs, size_t bytes, void* stream);
// device-indexed CircularTensor update (capturable): every plane job derived from *count on the device
struct CircDev {
    uint8_t* out;         // ordered tensor (unused when mirrored)
    uint8_t* ring;        // history ring (2 * batch slots when mirrored)
    const uint8_t* stage; // the new frame, standard [c][y][x] order (or packed pixels, color_planes == 1)
    const uint64_t* count;
    size_t plane_bytes;
    int32_t batch, color_planes, order, transposed, mirrored, pad;
};
int launch_circular_dev(const CircDev& a, void* stream);
// single-launch CircularTensor update (new frame through the thread-fused pointwise chain + all plane copies):
// returns 1 if it took the update, 0 if the chain / layout is not eligible, <0 on error
int launch_circular_push(const ChainArgs& c, const PlaneParams& plane, const CopyJob* jobs, int n_jobs, size_t plane_bytes,
                         uint32_t chain_flags, void* stream, const CircDev* dev = nullptr);
int launch_circular_bump(const uint64_t* count, void* stream);


// ---- device-side arrival flags of the P2P fused-write exchange (k_exchange.hip) ----------------------------------------------
#define CVGS_MAX_EXCHANGE_PEERS 16
int launch_exchange_signal(void* const* peer_flags, int n, uint64_t value, uint64_t* counter, void* stream);
int launch_exchange_step(void* const* peer_flags, const void* const* own_flags, int n, uint64_t* counter, uint64_t lag, double timeout_ms, void* err_words,
                         void* stream);
int launch_exchange_wait(const void* const* flags, int n, uint64_t value, const uint64_t* counter, uint64_t lag, double timeout_ms, void* err_words,
                         void* stream);

// ---- device-side descriptor queue (k_queue.hip): K1 batches served by a resident grid, no launch per batch ----------------
struct Queue;
int queue_create(Queue** out, int device, int depth, uint32_t flags, double idle_us, std::string& err);
// 0 = queued (ticket = batch number), 1 = this chain is not one the server takes (use cvgs_execute), < 0 = error
int queue_submit(Queue* q, const ChainArgs& c, const PlaneParams* planes, int n_planes, uint64_t* ticket, std::string& err);
// stream-ordered: 0 = queued behind `stream` (gate kernel enqueued), 1 = not a chain / situation the server takes, 2 = hybrid policy:
// take the direct launch (nothing in flight to overlap with), < 0 = error.  flags: 1 = defer the completion wait, 2 = hybrid policy
int queue_submit_on(Queue* q, const ChainArgs* const* chains, const PlaneParams* const* planes, const int* n_planes, int n, void* stream, uint32_t flags,
                    uint64_t* tickets, int* n_queued, std::string& err);
int queue_recover(Queue* q, uint64_t* lost, std::string& err);
const uint64_t* queue_gate_trace(Queue* q); // null unless CVGS_QUEUE_DEBUG=2 was set at create
int queue_wait(Queue* q, uint64_t ticket, double timeout_s, std::string& err);
int queue_stream_wait(Queue* q, uint64_t ticket, void* stream, std::string& err);
void queue_stats(Queue* q, uint64_t* out8);
void queue_prof(Queue* q, uint64_t* out16); // instrumentation of the last retired server
void* queue_stream(Queue* q);
int queue_destroy(Queue* q);

} // namespace cvgs
