// k_k1_impl.hpp -- K1's kernel template and its launch selectors, shared by the translation units that instantiate it
// (k_k1.hip: the dispatcher, packed / separate-plane / 1-2 channel / mirrored variants; k_k1_c3.hip, k_k1_c4.hip: the planar-tensor
// variants of 3- and 4-channel sources -- split only so that the ~300 instantiations compile in parallel).  See k_k1.hip.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <memory>
#include <type_traits>

#include <hip/hip_ext.h>

#include "k_taps.hpp"

namespace cvgs {

// Write stages of the fast path.  Planar is the hot one (TensorSplit / TensorTSplit); the other two serve the
// single-image chains of the reference's resize tests (tests/resize/test_resize_write.cu: resize -> convertTo<32F,8U> ->
// write; tests/resize/test_resize_x_split.cu: resize -> mul -> sub -> div -> split(vector<GpuMat>)).
enum { WM_PLANAR = 0, WM_PACKED = 1, WM_SPLIT2D = 2 };

// waves per workgroup (each wave owns RPW output rows of 64 columns; waves of a workgroup share nothing, so this only sets
// the dispatch granularity).  Round-2 A/B (-DCVGS_K1_WPB=1/2/4/8): the 50-crop launch 4.48 / 4.34 / 4.40 / 4.55 us,
// 16 x 50 crops 38.4 us for 2 and 4, whole-frame resizes within noise -> 2 (128-thread workgroups).
#ifndef CVGS_K1_WPB
#define CVGS_K1_WPB 2
#endif
constexpr int kK1Waves = CVGS_K1_WPB;

// Ablation hooks of tools/probes/tick_ablation.py (round 6): the product build never defines CVGS_K1_ABLATE -- every hook below is an
// `if constexpr` on 0 there, the generated code is unchanged (tests/test_kernel_resources.py holds the kernels' register / code-size
// census).  The probe builds k_k1_c3.hip again with -DCVGS_K1_ABLATE=<bits> into build/ablate/ (NOT into libcvgs_hip.so), so that the
// skeletons run the production kernel's own grid, scalar loads and tap addresses:
//   1 no tap loads (the windows are synthesised from the lane id)     2 no arithmetic (the windows' dwords are stored as they are)
//   4 no stores (kept behind a data-dependent test that never holds, so the loads stay)
//   8 the linear workgroup index is re-read chain-fastest (consecutive workgroups belong to different chains of the tick)
//  16 the wave ends behind its batch of scalar loads (descriptor fetch only)
//  32 XCD-banded work lists (tools/probes/xcd_worklist_probe.py): workgroup (slot, chain) looks its (crop, row tile) up in a per-chain list in
//     which slot s belongs to XCD s % 8 and every XCD's items tap one band of source rows -- overlapping crops of a frame then share an L2
#ifndef CVGS_K1_ABLATE
#define CVGS_K1_ABLATE 0
#endif
constexpr int kAblate = CVGS_K1_ABLATE;

struct K1Geom {
    uint32_t col_tiles;  // ceil(dst_w / 64)
    int32_t dst_w, dst_h;
    int32_t used;        // planes >= used carry the background value
    int32_t out_w;       // output row length in elements
    int32_t pad;
    int64_t img_stride;  // output elements between images
    int64_t ch_stride;   // output elements between channel planes
    void* out;           // float* or _Float16* (template parameter OT)
    void* out2;          // optional second target (CircularTensor ring + tensor), own strides
    int64_t img_stride2, ch_stride2;
    // WM_PACKED: bytes between output rows / images (both targets dense or pitched alike); WM_SPLIT2D: plane table
    int64_t row_pitch, img_pitch, row_pitch2, img_pitch2;
    const DstPlane* planes2d;
    // planar only: further tensors that receive the same values at the same offsets (cvgs_write_desc.mirrors: the
    // peers' copies of a sharded tensor, written through P2P-mapped pointers)
    uint8_t* mirror[CVGS_MAX_MIRRORS];
    int32_t n_mirror;
    int32_t pad2;
    // fused launches with host descriptors (NPL == 0; cvgs_api.cpp: ManyPool): the first work-item stores done_value into *done_word
    // (pinned host memory) when the kernel starts -- every earlier launch of the stream has finished by then
    uint64_t* done_word;
    uint64_t done_value;
};

// NPL > 0: the planes travel in the kernel arguments.  NPL == 0: they live in device tables, one segment per fused
// chain (cvgs_execute_many; a single chain with a resident table is one segment), blockIdx.z = segment.
// NPL < 0: the same segments with the planes of all chains INSIDE the kernel arguments (KernArgsManyInline<-NPL>: fused chains described on the host).
template <int NPL> using K1Args = std::conditional_t<NPL == 0, KernArgsMany, std::conditional_t<(NPL < 0), KernArgsManyInline<(NPL < 0 ? -NPL : 1)>, KernArgs<(NPL > 0 ? NPL : 1)>>>;

// packed pixels / separate pitched planes: one output pixel of row y, column x, plane z
// WIDE: the throughput regime (4 rows per wave, whole-frame outputs), where the store instruction count matters; small
// launches are latency bound and keep the shuffle off their critical path (measured: 4K->1080p 10.8 vs 12.1 us with it,
// 1080p->4K 29 vs 24 us).
template <int WM, typename OT, int CN, bool WIDE>
__device__ __forceinline__ void k1_store_other(const K1Geom& g, const ChainArgs& c, int z, int y, int x, const float* v, int cn) {
    if constexpr (WM == WM_PACKED) {
        uint8_t* row = (uint8_t*)g.out + (int64_t)z * g.img_pitch + (int64_t)y * g.row_pitch;
        if constexpr (std::is_same_v<OT, uint8_t> && CN == 3 && WIDE) {
            const int lane = (int)(threadIdx.x & 63), x0 = x - lane;
            if (cn == 3 && x0 + 63 < g.dst_w) { // wave-uniform: every lane of the tile is alive
                store_u8c3_tile(row + (int64_t)x0 * 3, lane, v);
                if (g.out2) store_u8c3_tile((uint8_t*)g.out2 + (int64_t)z * g.img_pitch2 + (int64_t)y * g.row_pitch2 + (int64_t)x0 * 3, lane, v);
                return;
            }
        }
        store_packed_px<CN, OT>((OT*)row + (int64_t)x * cn, v, cn);
        if (g.out2) {
            uint8_t* row2 = (uint8_t*)g.out2 + (int64_t)z * g.img_pitch2 + (int64_t)y * g.row_pitch2;
            store_packed_px<CN, OT>((OT*)row2 + (int64_t)x * cn, v, cn);
        }
    } else {
        const DstPlane* planes = g.planes2d ? g.planes2d : c.dst_inline;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < cn) {
                const DstPlane d = planes[z * cn + k];
                static_assert(std::is_same_v<OT, float>, "separate planes are written as fp32");
                typedef __attribute__((address_space(1))) float* gptr_f32; // global, not flat, stores
                __builtin_nontemporal_store(v[k], (gptr_f32)(float*)(d.data + (int64_t)y * d.step) + x);
            }
    }
}

// The kernel's leading SCALAR parameters: 14 dwords, the most the hardware preloads into user SGPRs with the dispatch
// (-mllvm -amdgpu-kernarg-preload-count=14 in the Makefile; aggregates are never preloaded).  They repeat what the wave needs FIRST --
// segment 0 (a single chain IS segment 0) and the target's extent -- so that the crop's descriptor can be requested without waiting for
// a load of the argument block: one memory round trip less in front of the taps.  Firmware without the feature runs the compiler's
// compatibility prologue (plain loads of the same values).
#define K1_PRELOADED_PARAMS                                                                                                               \
    const PlaneParams *pre_table, uint8_t *pre_out, int64_t pre_img_stride, int64_t pre_ch_stride, int32_t pre_batch, int32_t pre_used,  \
        uint32_t pre_col_tiles, int32_t pre_dst_w, int32_t pre_dst_h, int32_t pre_out_w
// MIR: the instantiations that also write cvgs_write_desc.mirrors (kept out of the others' code)
template <int CN, int NPL, int RPW, class Prog, int SRC = SRC_U8, typename OT = float, int WM = WM_PLANAR, bool MIR = false>
__global__ __launch_bounds__(64 * kK1Waves) void k1_resize_split(K1_PRELOADED_PARAMS, const K1Args<NPL> a, const K1Geom g) {
    constexpr int EB = elem_bytes<SRC>;
    constexpr int WINB = SRC == SRC_F32 ? 2 * CN * 4 : 8 * EB; // bytes per tap window (fp32: exactly the pixel pair)
    const ChainArgs& c = a.c;
    uint32_t bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if constexpr ((kAblate & 8) != 0) { // chain-fastest order of the same grid (probe only)
        const uint32_t lin = bx + gridDim.x * (by + gridDim.y * bz);
        bz = lin % gridDim.z;
        const uint32_t rest = lin / gridDim.z;
        bx = rest % gridDim.x;
        by = rest / gridDim.x;
    }
    if constexpr ((kAblate & 32) != 0) { // probe: (crop, row tile) from the chain's XCD-banded work list (g.planes2d / g.row_pitch carry base / slots)
        const uint32_t item = ((const uint32_t*)g.planes2d)[(size_t)bz * (size_t)g.row_pitch + bx];
        if (item == 0xffffffffu) return;
        by = item >> 8;
        bx = item & 0xffu;
    }
    const int z = (int)by;
    // ---- one batch of scalar loads: the crop's parameters, the program operands come in together; what the wave needs to ask for its
    // crop's descriptor (segment 0's table, the target's extent) arrives in SGPRs with the dispatch (kernel-argument preload) ----
    const int dst_w = pre_dst_w, dst_h = pre_dst_h, W = pre_out_w;
    const uint32_t col_tiles = pre_col_tiles;
    const int64_t img_stride = pre_img_stride, ch_stride = pre_ch_stride;
    int used;
    OT* out_base;
    PlaneParams P;
    if constexpr (NPL <= 0) {
        const PlaneParams* table = pre_table; // NPL < 0: the chain's first index into a.planes
        int batch = pre_batch;
        used = pre_used;
        out_base = (OT*)pre_out;
        if (bz != 0) { // the other chains of a fused launch: their segment comes from the argument block
            const ManySeg sg = a.seg[bz];
            table = sg.table;
            batch = sg.batch;
            used = sg.used;
            out_base = (OT*)sg.out;
        }
        if (z >= batch) return; // a shorter chain of the fused launch
        if constexpr (NPL == 0) P = table[z < used ? z : 0];
        else P = a.planes[(uint32_t)(uintptr_t)table + (uint32_t)(z < used ? z : 0)];
    } else {
        used = pre_used;
        out_base = (OT*)pre_out;
        P = a.planes[z];
    }
    OT* const out2_base = (OT*)g.out2;
    const int n_mirror = MIR ? g.n_mirror : 0;
    const int64_t img_stride2 = g.img_stride2, ch_stride2 = g.ch_stride2;
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    const f32x4s op0 = *(const f32x4s*)c.prog.operand[0], op1 = *(const f32x4s*)c.prog.operand[1],
                 op2 = *(const f32x4s*)c.prog.operand[2], op3 = *(const f32x4s*)c.prog.operand[3];
    // naming every value in one asm statement makes the compiler issue ALL these scalar loads back to back and wait
    // once; otherwise each early-exit test gets its own load + wait (3-4 serial scalar-memory round trips per wave)
    asm volatile("" ::"s"(dst_w), "s"(dst_h), "s"(used), "s"(W), "s"(col_tiles), "s"(P.w), "s"(P.h), "s"(P.step), "s"(P.x1),
                 "s"(P.y1), "s"(P.x2), "s"(P.y2), "s"(P.fx), "s"(P.fy), "s"(P.data), "s"(img_stride), "s"(ch_stride),
                 "s"(out_base), "s"(out2_base), "s"(img_stride2), "s"(ch_stride2), "s"(op0), "s"(op1), "s"(op2), "s"(op3), "s"(n_mirror));
    // (behind the scalar loads: a store in front of them would make the compiler fetch the descriptors with vector loads)
    if constexpr (NPL == 0) {
        if (g.done_word && (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && threadIdx.x == 0)
            __hip_atomic_store(g.done_word, g.done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }

    if constexpr ((kAblate & 16) != 0) return;
    int col_tile = 0, row_tile = (int)bx;
    if (col_tiles > 1) {
        col_tile = (int)(bx % col_tiles);
        row_tile = (int)(bx / col_tiles);
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int x = col_tile * 64 + lane;
    const int row0 = (row_tile * kK1Waves + wave) * RPW;
    if (row0 >= dst_h || x >= dst_w) return;
    OT* const out = out_base + (int64_t)z * img_stride;
    OT* const out2 = out2_base ? out2_base + (int64_t)z * img_stride2 : nullptr; // wave-uniform

    // does the source cover the whole target?  (always, except AR padding and planes >= usedPlanes)
    const bool whole = z < used && ((P.x1 | P.y1 | (P.x2 ^ (dst_w - 1)) | (P.y2 ^ (dst_h - 1))) == 0);

    Px bgp;
    bgp.v[0] = bgp.v[1] = bgp.v[2] = bgp.v[3] = 0.f;
    int out_cn = CN;
    if (!whole) {
        // background value pushed through the whole chain: planes >= usedPlanes and AR padding
        int bdepth = CVGS_DEPTH_32F, bcn = CN;
#pragma unroll
        for (int k = 0; k < 4; ++k) bgp.v[k] = c.read.bg[k];
        Prog::run(c.prog, bgp, bdepth, bcn);
        out_cn = bcn;
        if (z >= used) {
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const int y = row0 + j;
                if (y < dst_h) {
                    if constexpr (WM == WM_PLANAR) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (k < bcn) {
                                st_nt(out + (int64_t)k * ch_stride + (int64_t)y * W + x, bgp.v[k]);
                                if (out2) st_nt(out2 + (int64_t)k * ch_stride2 + (int64_t)y * W + x, bgp.v[k]);
                                if constexpr (MIR)
                                    for (int m = 0; m < n_mirror; ++m)
                                        st_sys((OT*)g.mirror[m] + (int64_t)z * img_stride + (int64_t)k * ch_stride + (int64_t)y * W + x, bgp.v[k]);
                            }
                    } else {
                        k1_store_other<WM, OT, CN, (RPW >= 4)>(g, c, z, y, x, bgp.v, bcn);
                    }
                }
            }
            return;
        }
    }

    // ---- per-lane column geometry (reused for every row) ----
    const bool in_x = x >= P.x1 && x <= P.x2;
    const int xr = in_x ? x - P.x1 : 0;
    const float sx = (float)xr * P.fx;
    const int x1 = (int)floorf(sx);
    const int x2 = x1 + 1;
    const float wxa = (float)x2 - sx;
    const float wxb = sx - (float)x1;
    const bool edge = x2 > P.w - 1;
    const int row_bytes = P.w * CN * EB;
    const int o = x1 * CN * EB;
    const bool tiny = row_bytes < WINB; // wave-uniform
    const uint32_t ol = (uint32_t)(tiny ? o : min(o, row_bytes - WINB));
    const int sh = (o - (int)ol) * 8;
    const gptr_u8 src = (gptr_u8)P.data;

    typename Prog::State pst = Prog::prefetch(c.prog); // (an interpreted arithmetic program: its words are requested before the taps)
    Win<EB> va[RPW], vb[RPW];
    float wya[RPW], wyb[RPW];
    bool in_y[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = min(row0 + j, dst_h - 1);
        in_y[j] = y >= P.y1 && y <= P.y2;
        const int yr = in_y[j] ? y - P.y1 : 0;
        const float sy = (float)yr * P.fy;
        const int y1 = (int)floorf(sy);
        const int y2 = y1 + 1;
        const int y2r = min(y2, P.h - 1);
        wya[j] = (float)y2 - sy;
        wyb[j] = sy - (float)y1;
        const gptr_u8 ra = pin_uniform(src + (size_t)__builtin_amdgcn_readfirstlane(y1) * (size_t)P.step);
        const gptr_u8 rb = pin_uniform(src + (size_t)__builtin_amdgcn_readfirstlane(y2r) * (size_t)P.step);
        if constexpr ((kAblate & 1) != 0 && SRC == SRC_U8) { // probe: no tap loads
            va[j].lo = (uint64_t)(uint32_t)(lane * 0x01010101 + y1) * 0x100000001ull;
            vb[j].lo = (uint64_t)(uint32_t)(lane * 0x01010101 + y2r) * 0x100000001ull;
            (void)ra;
            (void)rb;
        } else if constexpr (SRC == SRC_F32) {
            if (!tiny) {
                va[j] = load_win_f32<CN>(ra + ol);
                vb[j] = load_win_f32<CN>(rb + ol);
            } else {
                va[j] = gather_win_f32<CN>(ra);
                vb[j] = gather_win_f32<CN>(rb);
            }
        } else if (!tiny) {
            va[j] = load_win<EB>(ra + ol);
            vb[j] = load_win<EB>(rb + ol);
        } else {
            va[j] = gather_win<CN, EB>(ra, o, row_bytes);
            vb[j] = gather_win<CN, EB>(rb, o, row_bytes);
        }
    }

    Prog::settle(pst);
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = row0 + j;
        if (y < dst_h) { // wave-uniform
            float p00[4], p10[4], p01[4], p11[4];
            if constexpr ((kAblate & 2) != 0 && SRC == SRC_U8) { // probe: no arithmetic -- the four dwords of the two windows, as they are
                p00[0] = __uint_as_float((uint32_t)va[j].lo), p00[1] = __uint_as_float((uint32_t)(va[j].lo >> 32) ^ (uint32_t)vb[j].lo);
                p00[2] = __uint_as_float((uint32_t)(vb[j].lo >> 32)), p00[3] = 0.f;
            } else if constexpr (SRC == SRC_F32) {
                unpack_pair_f32<CN>(va[j], sh != 0, edge, p00, p10);
                unpack_pair_f32<CN>(vb[j], sh != 0, edge, p01, p11);
            } else {
                unpack_pair<CN, SRC>(shift_win<EB>(va[j], sh), edge, p00, p10);
                unpack_pair<CN, SRC>(shift_win<EB>(vb[j], sh), edge, p01, p11);
            }
            const float w00 = wxa * wya[j];
            const float w10 = wxb * wya[j];
            const float w01 = wxa * wyb[j];
            const float w11 = wxb * wyb[j];
            Px p;
            p.v[0] = p.v[1] = p.v[2] = p.v[3] = 0.f;
            int depth = CVGS_DEPTH_32F, cn = CN;
            if constexpr ((kAblate & 2) != 0 && SRC == SRC_U8) {
#pragma unroll
                for (int k = 0; k < CN; ++k) p.v[k] = p00[k];
            } else {
#pragma unroll
                for (int k = 0; k < CN; ++k) {
                    float acc = p00[k] * w00;
                    acc = acc + p10[k] * w10;
                    acc = acc + p01[k] * w01;
                    acc = acc + p11[k] * w11;
                    p.v[k] = acc;
                }
                Prog::run(c.prog, pst, p, depth, cn);
            }
            out_cn = cn;
            if constexpr ((kAblate & 4) != 0) { // probe: no stores (the test never holds on pixel data; the loads stay alive)
                if (!(__float_as_uint(p.v[0]) == 0x7fc12345u && __float_as_uint(p.v[1]) == 0x7fc54321u && __float_as_uint(p.v[2]) == 0x7fc00001u)) continue;
            }
            const bool take = whole || (in_x && in_y[j]);
            if constexpr (WM == WM_PLANAR) {
                OT* const orow = out + (int64_t)y * W; // wave-uniform
                const uint32_t xb = (uint32_t)x * (uint32_t)sizeof(OT);
                if (!whole) { // wave-uniform: only aspect-ratio padded planes pay the per-lane select
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < cn) p.v[k] = take ? p.v[k] : bgp.v[k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < cn) {
                        const float v = p.v[k];
                        st_row(orow + (int64_t)k * ch_stride, xb, v);
                        if (out2) st_row(out2 + (int64_t)y * W + (int64_t)k * ch_stride2, xb, v);
                        if constexpr (MIR)
                            for (int m = 0; m < n_mirror; ++m) // wave-uniform trip count; peers' tensors share the strides
                                st_row_sys((OT*)g.mirror[m] + (int64_t)z * img_stride + (int64_t)y * W + (int64_t)k * ch_stride, xb, v);
                    }
            } else {
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = p.v[k];
                if (!whole) { // wave-uniform, as in the planar mode
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = take ? p.v[k] : bgp.v[k];
                }
                k1_store_other<WM, OT, CN, (RPW >= 4)>(g, c, z, y, x, v, cn);
            }
        }
    }
    (void)out_cn;
}

// ------------------------------------------------------------------------------------------------
#if (CVGS_K1_ABLATE & 32)
struct ProbeWorklist { const uint32_t* base = nullptr; uint32_t slots = 0; };
inline ProbeWorklist& probe_worklist() { static ProbeWorklist w; return w; }
#endif

template <int CN, int NPL, int RPW, class Prog, int SRC, typename OT, int WM = WM_PLANAR, bool MIR = false>
static hipError_t launch_t(const ChainArgs& c, const PlaneParams* inline_planes, int n_inline, int out_cn, LaunchCtx& x) {
    hipStream_t stream = (hipStream_t)x.stream;
    // Argument blocks beyond 8 KB (the 16 KB / 52 KB blocks of large batches and host-described ticks) are staged in a per-thread heap buffer and
    // handed to the launch BY ADDRESS (hipLaunchKernel's argument-pointer form): nothing that size lives on the caller's stack or is copied
    // through it by value (a tick call used 100-150 KB of stack -- more than some worker threads have).  The buffer starts
    // zeroed and keeps the previous call's descriptors beyond this call's: no uninitialised byte reaches the argument block.
    constexpr bool kBig = sizeof(K1Args<NPL>) > 8192;
    K1Args<NPL> small_block;
    K1Args<NPL>* block = &small_block;
    if constexpr (kBig) {
        static thread_local std::unique_ptr<K1Args<NPL>> staged;
        if (!staged) staged.reset(new K1Args<NPL>());
        block = staged.get();
    }
    K1Args<NPL>& a = *block;
    a.c = c;
    unsigned grid_z = 1;
    if constexpr (NPL > 0) {
        for (int i = 0; i < n_inline; ++i) a.planes[i] = inline_planes[i];
        for (int i = n_inline; i < NPL; ++i) a.planes[i] = PlaneParams{};
    } else {
        if constexpr (NPL < 0) { // fused chains with host descriptors: every chain's planes behind the segments (segs[i].table = first index)
            for (int i = 0; i < n_inline && i < -NPL; ++i) a.planes[i] = inline_planes[i];
        }
        if (x.segs) {
            grid_z = (unsigned)x.n_segs;
            for (int i = 0; i < CVGS_MAX_CHAINS; ++i) a.seg[i] = i < x.n_segs ? x.segs[i] : ManySeg{nullptr, nullptr, 0, 0};
        } else {
            a.seg[0] = ManySeg{c.read.table, c.write.data, c.read.batch, c.read.used};
            for (int i = 1; i < CVGS_MAX_CHAINS; ++i) a.seg[i] = ManySeg{nullptr, nullptr, 0, 0};
        }
    }
    K1Geom g;
    const int rows_per_wg = kK1Waves * RPW;
    g.col_tiles = (uint32_t)((c.read.dst_w + 63) / 64);
    const uint32_t row_tiles = (uint32_t)((c.read.dst_h + rows_per_wg - 1) / rows_per_wg);
    g.dst_w = c.read.dst_w;
    g.dst_h = c.read.dst_h;
    g.used = c.read.used;
    g.out_w = c.write.width;
    g.pad = 0;
    (void)out_cn;
    g.img_stride = c.write.img_stride;
    g.ch_stride = c.write.ch_stride;
    g.out = c.write.data;
    g.out2 = c.write.data2;
    g.img_stride2 = c.write.img_stride2;
    g.ch_stride2 = c.write.ch_stride2;
    // packed targets: byte pitches (PIXEL_2D: the image's step, one image; PIXEL_3D: dense planes)
    const int64_t px_bytes = (int64_t)sizeof(OT) * c.write.cn;
    g.row_pitch = c.write.kind == CVGS_WRITE_PIXEL_2D ? c.write.step : c.write.width * px_bytes;
    g.img_pitch = c.write.kind == CVGS_WRITE_PIXEL_2D ? 0 : c.write.img_stride * px_bytes;
    g.row_pitch2 = c.write.width * px_bytes;
    g.img_pitch2 = c.write.img_stride2 * px_bytes;
    g.planes2d = c.write.table;
    g.n_mirror = MIR ? x.mirrors.n : 0;
    g.pad2 = 0;
    for (int i = 0; i < CVGS_MAX_MIRRORS; ++i) g.mirror[i] = i < g.n_mirror ? x.mirrors.p[i] : nullptr;
    dim3 grid(g.col_tiles * row_tiles, (unsigned)c.read.batch, grid_z);
#if (CVGS_K1_ABLATE & 32)
    if (probe_worklist().base) {
        grid = dim3(probe_worklist().slots, 1, grid_z);
        g.planes2d = (const DstPlane*)probe_worklist().base;
        g.row_pitch = (int64_t)probe_worklist().slots;
    }
#endif
    g.done_word = nullptr;
    g.done_value = 0;
    if constexpr (NPL == 0) {
        if (x.done_word && !x.done_word_taken) {
            x.done_word_taken = true;
            g.done_word = x.done_word;
            g.done_value = x.done_value;
        }
    }
    // segment 0 / the single chain, repeated in front of the argument block (K1_PRELOADED_PARAMS)
    const PlaneParams* pre_table = nullptr;
    uint8_t* pre_out = (uint8_t*)g.out;
    int32_t pre_batch = c.read.batch, pre_used = g.used;
    if constexpr (NPL <= 0) {
        pre_table = a.seg[0].table;
        pre_out = a.seg[0].out;
        pre_batch = a.seg[0].batch;
        pre_used = a.seg[0].used;
    }
    // every argument by address, in the kernel's parameter order (K1_PRELOADED_PARAMS, the argument block, the geometry)
    void* args[] = {(void*)&pre_table, (void*)&pre_out, (void*)&g.img_stride, (void*)&g.ch_stride, (void*)&pre_batch, (void*)&pre_used, (void*)&g.col_tiles,
                    (void*)&g.dst_w, (void*)&g.dst_h, (void*)&g.out_w, (void*)&a, (void*)&g};
    const void* fn = (const void*)&k1_resize_split<CN, NPL, RPW, Prog, SRC, OT, WM, MIR>;
    hipError_t e;
    if (x.stop_event && !x.stop_event_taken) {
        x.stop_event_taken = true;
        e = hipExtLaunchKernel(fn, grid, dim3(64 * kK1Waves), args, 0, stream, nullptr, (hipEvent_t)x.stop_event, 0);
    } else {
        e = hipLaunchKernel(fn, grid, dim3(64 * kK1Waves), args, 0, stream);
    }
    return e != hipSuccess ? e : hipGetLastError();
}

// packed / separate-plane targets; one row per wave, four for whole-frame sizes
template <int CN, typename OT, int WM, class Prog = InterpProg, int SRC = SRC_U8>
static hipError_t launch_other(bool table, int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, LaunchCtx& s) {
    if (rpw >= 4) {
        if (table) return launch_t<CN, 0, 4, Prog, SRC, OT, WM>(c, ip, ni, c.write.cn, s);
        return launch_t<CN, CVGS_KERNARG_PLANES, 4, Prog, SRC, OT, WM>(c, ip, ni, c.write.cn, s);
    }
    if (table) return launch_t<CN, 0, 1, Prog, SRC, OT, WM>(c, ip, ni, c.write.cn, s);
    return launch_t<CN, CVGS_KERNARG_PLANES, 1, Prog, SRC, OT, WM>(c, ip, ni, c.write.cn, s);
}
// the same with the program picked at run time: empty (nothing between the resize and the folded cast / the write) or interpreted
template <int CN, typename OT, int WM, int SRC = SRC_U8>
static hipError_t launch_other_np(bool none, bool table, int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, LaunchCtx& s, bool canon = false) {
    if (none) return launch_other<CN, OT, WM, ProgNone, SRC>(table, rpw, c, ip, ni, s);
    if constexpr (SRC == SRC_U8 && WM == WM_PACKED) { // (k_taps.hpp: the chain was rewritten into the canonical arithmetic pipeline)
        if (canon) return launch_other<CN, OT, WM, K1CanonProg, SRC>(table, rpw, c, ip, ni, s);
    }
    return launch_other<CN, OT, WM, InterpProg, SRC>(table, rpw, c, ip, ni, s);
}
// 16-bit and CV_32F sources into packed pixels of the SOURCE's own type (the reference's single-image resize tests sweep
// CV_16U / CV_16S C1, C3, C4 and CV_32FC1: resize -> convertTo<CV_32F, I> -> write<I>, tests/resize/test_resize_write.cu:55-56,
// 110-123), and 16-bit sources into separate fp32 planes (tests/resize/test_resize_x_split.cu)
template <int CN>
static hipError_t launch_same_type_packed(int src, bool table, int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, LaunchCtx& s) {
    const bool none = c.prog.n == 0;
    if (src == SRC_U16) return launch_other_np<CN, uint16_t, WM_PACKED, SRC_U16>(none, table, rpw, c, ip, ni, s);
    if (src == SRC_S16) return launch_other_np<CN, int16_t, WM_PACKED, SRC_S16>(none, table, rpw, c, ip, ni, s);
    return launch_other_np<CN, float, WM_PACKED, SRC_F32>(none, table, rpw, c, ip, ni, s);
}
template <int CN>
static hipError_t launch_split2d_16(int src, bool table, int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, LaunchCtx& s) {
    if (src == SRC_U16) return launch_other<CN, float, WM_SPLIT2D, InterpProg, SRC_U16>(table, rpw, c, ip, ni, s);
    return launch_other<CN, float, WM_SPLIT2D, InterpProg, SRC_S16>(table, rpw, c, ip, ni, s);
}
// separate planes: the reference's K2 chain (mul, sub, div; with or without the R<->B swap) gets its compile-time program
template <int CN>
static hipError_t launch_split2d(int prog_id, bool table, int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, LaunchCtx& s) {
    if (prog_id == 0) return launch_other<CN, float, WM_SPLIT2D, ProgSwapMulSubDiv>(table, rpw, c, ip, ni, s);
    if (prog_id == 1) return launch_other<CN, float, WM_SPLIT2D, ProgMulSubDiv>(table, rpw, c, ip, ni, s);
    return launch_other<CN, float, WM_SPLIT2D>(table, rpw, c, ip, ni, s);
}

template <int CN, int NPL, class Prog, int SRC, typename OT>
static hipError_t launch_rpw(int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, int out_cn, LaunchCtx& s) {
    // the interpreted program keeps its opcode loop rolled; more than one row per wave only bloats it
    if constexpr (std::is_same_v<Prog, InterpProg>) return launch_t<CN, NPL, 1, Prog, SRC, OT>(c, ip, ni, out_cn, s);
    else if constexpr (std::is_same_v<Prog, InterpProgArith>) return launch_t<CN, NPL, 1, Prog, SRC, OT>(c, ip, ni, out_cn, s); // (2 rows per wave: 58.7 vs 50.7 us per tick)
    else if constexpr (SRC != SRC_U8) { // 16-bit sources: two row counts are enough
        if (rpw == 1) return launch_t<CN, NPL, 1, Prog, SRC, OT>(c, ip, ni, out_cn, s);
        return launch_t<CN, NPL, 4, Prog, SRC, OT>(c, ip, ni, out_cn, s);
    } else
    switch (rpw) {
    case 1: return launch_t<CN, NPL, 1, Prog, SRC, OT>(c, ip, ni, out_cn, s);
    case 2: return launch_t<CN, NPL, 2, Prog, SRC, OT>(c, ip, ni, out_cn, s);
    default: return launch_t<CN, NPL, 4, Prog, SRC, OT>(c, ip, ni, out_cn, s);
    }
}

template <int CN, class Prog, int SRC, typename OT>
static hipError_t launch_npl(bool table, int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, int out_cn,
                             LaunchCtx& s) {
    if (table) return launch_rpw<CN, 0, Prog, SRC, OT>(rpw, c, ip, ni, out_cn, s);
    if constexpr (SRC == SRC_U8) { // cvgs_execute_many on host descriptors (u8 sources, 3 / 4 channels): segments + ALL planes in the arguments
        if (s.segs) {
            if (ni <= kManyInlineSmall) return launch_rpw<CN, -kManyInlineSmall, Prog, SRC, OT>(rpw, c, ip, ni, out_cn, s);
            return launch_rpw<CN, -kManyInlineLarge, Prog, SRC, OT>(rpw, c, ip, ni, out_cn, s);
        }
    }
    if (ni > CVGS_KERNARG_PLANES) return launch_rpw<CN, kKernargPlanesBig, Prog, SRC, OT>(rpw, c, ip, ni, out_cn, s); // 16 KB argument block
    return launch_rpw<CN, CVGS_KERNARG_PLANES, Prog, SRC, OT>(rpw, c, ip, ni, out_cn, s);
}

template <int CN, int SRC, typename OT = float>
static hipError_t launch_prog(int prog_id, bool table, int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, int out_cn,
                              LaunchCtx& s) {
    if (prog_id == 0) return launch_npl<CN, ProgSwapMulSubDiv, SRC, OT>(table, rpw, c, ip, ni, out_cn, s);
    if (prog_id == 1) return launch_npl<CN, ProgMulSubDiv, SRC, OT>(table, rpw, c, ip, ni, out_cn, s);
    if (prog_id == 3) return launch_npl<CN, K1CanonProg, SRC, OT>(table, rpw, c, ip, ni, out_cn, s); // (k_taps.hpp: the canonical arithmetic pipeline)
    return launch_npl<CN, InterpProgArith, SRC, OT>(table, rpw, c, ip, ni, out_cn, s);
}

// 1- and 2-channel sources (grayscale / two-plane images; the reference's single-image resize tests sweep C1 types,
// tests/resize/test_resize_write.cu:120-123): planar fp32 for every source kind, packed fp32 / u8 for 8U sources
template <int CN, int SRC>
static hipError_t launch_few_planar(int prog_id, bool table, int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, LaunchCtx& s) {
    const int r = rpw >= 4 ? 4 : 1;
    if constexpr (SRC == SRC_U8) {
        if (prog_id == 3) { // the canonical arithmetic pipeline (k_taps.hpp)
            if (table) return r == 4 ? launch_t<CN, 0, 4, K1CanonProg, SRC, float>(c, ip, ni, CN, s) : launch_t<CN, 0, 1, K1CanonProg, SRC, float>(c, ip, ni, CN, s);
            return r == 4 ? launch_t<CN, CVGS_KERNARG_PLANES, 4, K1CanonProg, SRC, float>(c, ip, ni, CN, s)
                          : launch_t<CN, CVGS_KERNARG_PLANES, 1, K1CanonProg, SRC, float>(c, ip, ni, CN, s);
        }
    }
    if (prog_id == 1) {
        if (table) return r == 4 ? launch_t<CN, 0, 4, ProgMulSubDiv, SRC, float>(c, ip, ni, CN, s) : launch_t<CN, 0, 1, ProgMulSubDiv, SRC, float>(c, ip, ni, CN, s);
        return r == 4 ? launch_t<CN, CVGS_KERNARG_PLANES, 4, ProgMulSubDiv, SRC, float>(c, ip, ni, CN, s)
                      : launch_t<CN, CVGS_KERNARG_PLANES, 1, ProgMulSubDiv, SRC, float>(c, ip, ni, CN, s);
    }
    if (table) return launch_t<CN, 0, 1, InterpProg, SRC, float>(c, ip, ni, CN, s);
    return launch_t<CN, CVGS_KERNARG_PLANES, 1, InterpProg, SRC, float>(c, ip, ni, CN, s);
}
template <int CN>
static hipError_t launch_few(int src, bool planar, bool u8out, int prog_id, bool table, int rpw, const ChainArgs& c, const PlaneParams* ip,
                             int ni, LaunchCtx& s, bool canon = false) {
    if (planar) {
        return src == SRC_U8    ? launch_few_planar<CN, SRC_U8>(prog_id, table, rpw, c, ip, ni, s)
               : src == SRC_U16 ? launch_few_planar<CN, SRC_U16>(prog_id, table, rpw, c, ip, ni, s)
               : src == SRC_S16 ? launch_few_planar<CN, SRC_S16>(prog_id, table, rpw, c, ip, ni, s)
                                : launch_few_planar<CN, SRC_F32>(prog_id, table, rpw, c, ip, ni, s);
    }
    if (u8out) return launch_other_np<CN, uint8_t, WM_PACKED>(c.prog.n == 0, table, rpw, c, ip, ni, s, canon);
    return launch_other_np<CN, float, WM_PACKED>(c.prog.n == 0, table, rpw, c, ip, ni, s, canon);
}


} // namespace cvgs
