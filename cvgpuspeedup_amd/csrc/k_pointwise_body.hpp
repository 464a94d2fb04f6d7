// k_pointwise_body.hpp -- the thread-fused pointwise body (4 x-adjacent pixels per thread) shared by k_pointwise4
// (k_pointwise.hip) and by the single-launch CircularTensor push (k_circular.hip).  See k_pointwise.hip.
#pragma once

#include <type_traits>

#include "k_common.hpp"

namespace cvgs {

typedef uint32_t u32u __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) u32u* gp_u32;
typedef const __attribute__((address_space(1))) uint8_t* gp_u8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4u __attribute__((aligned(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16x4 f16x4u __attribute__((aligned(2)));

// four consecutive output elements in one non-temporal vector store (fp16: the chain's trailing CAST(CV_16F) is this
// round-to-nearest-even conversion)
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    f32x4 q = {a, b, c, d};
    __builtin_nontemporal_store(q, (f32x4u*)p);
}
__device__ __forceinline__ void store4(_Float16* p, float a, float b, float c, float d) {
    f16x4 q = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
    __builtin_nontemporal_store(q, (f16x4u*)p);
}
__device__ __forceinline__ void store1(float* p, float v) { *p = v; }
__device__ __forceinline__ void store1(_Float16* p, float v) { *p = (_Float16)v; }
// packed u8 pixels (the 4:2:0 read mode's `-> convertTo<CV_32FCn, CV_8UCn> -> write` chains): the chain's trailing
// SaturateCast is the store's conversion (k_common.hpp: sat_u8_insert)
__device__ __forceinline__ void store4(uint8_t* p, float a, float b, float c, float d) {
    typedef uint32_t u32a1 __attribute__((aligned(1)));
    __builtin_nontemporal_store(sat_u8_insert(d, 3, sat_u8_insert(c, 2, sat_u8_insert(b, 1, sat_u8_insert(a, 0, 0)))), (u32a1*)p);
}
__device__ __forceinline__ void store1(uint8_t* p, float v) { *p = (uint8_t)sat_u8_insert(v, 0, 0); }
template <typename OT> __device__ __forceinline__ OT cvt_out(float v) {
    if constexpr (std::is_same_v<OT, uint8_t>) return (uint8_t)sat_u8_insert(v, 0, 0);
    else return (OT)v;
}

using ProgCastMulSubDiv = StaticProg<CVGS_OP_CAST, CVGS_OP_MUL, CVGS_OP_SUB, CVGS_OP_DIV>;
using ProgCast = StaticProg<CVGS_OP_CAST>;

// The run-time programs k_pointwise4 accepts (pointwise4_plan checks it on the host): channel permutations, ONE cast to
// CV_32F (none for CV_32F sources), then MUL / ADD / SUB / DIV / REORDER on fp32 values, CN channels throughout.  The
// general interpreter (InterpProg) re-derives the value's depth and channel count at every stage of every pixel; here both
// are compile-time facts, so a stage is one wave-uniform branch and 4 x CN arithmetic instructions -- same operations, same
// order, same bits (the reference's tests/read/test_read_x_write.cu chain, convertTo -> sub -> mul -> div -> add, on a 4K
// frame: 45 -> 21 us).
template <int CN, int SD>
struct ArithProg {
    // The first kUnrolled stages are unrolled AND their opcodes, operands and the division's reciprocals are read before the first stage: scalar
    // loads at fixed kernel-argument offsets, requested together, one wait.  A loop over k fetched each stage's opcode and operands when it got
    // there -- one scalar-memory round trip per stage and wave (a one-channel thread has 4 values to spend it on: 4K 8UC1 -> 32FC1 ran 8 of its
    // 12 us with its loads and stores removed).
    static constexpr int kUnrolled = 6;
    static __device__ __forceinline__ void run4(const ProgArgs& prog, Px (&px)[4], int& depth, int& cn) {
        int op[kUnrolled], aux[kUnrolled];
        float o[kUnrolled][4];
        const int n = prog.n, fast_div = prog.fast_div;
        float r[4] = {prog.rdiv[0], prog.rdiv[1], prog.rdiv[2], prog.rdiv[3]};
#pragma unroll
        for (int k = 0; k < kUnrolled; ++k) {
            op[k] = prog.opcode[k];
            aux[k] = prog.aux[k];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) o[k][ch] = ch < CN ? prog.operand[k][ch] : 0.f;
        }
        // (the values are pinned in scalar registers HERE: left alone, the compiler sinks each load into the stage that uses it)
#pragma unroll
        for (int k = 0; k < kUnrolled; ++k) {
            asm volatile("" : "+s"(op[k]), "+s"(aux[k]));
#pragma unroll
            for (int ch = 0; ch < CN; ++ch) asm volatile("" : "+s"(o[k][ch]));
        }
#pragma unroll
        for (int ch = 0; ch < CN; ++ch) asm volatile("" : "+s"(r[ch]));
#pragma unroll
        for (int k = 0; k < kUnrolled; ++k)
            if (k < n) stage4(op[k], aux[k], o[k], fast_div == 2 + k, r, px);
        for (int k = kUnrolled; k < n; ++k) {
            const float ok[4] = {prog.operand[k][0], prog.operand[k][1], prog.operand[k][2], prog.operand[k][3]};
            stage4(prog.opcode[k], prog.aux[k], ok, fast_div == 2 + k, r, px);
        }
        depth = CVGS_DEPTH_32F;
        cn = CN;
    }
    static __device__ __forceinline__ void stage4(const int op, const int aux, const float (&o)[4], const bool div_fits, const float (&r)[4], Px (&px)[4]) {
        {
            if (op == CVGS_OP_MUL) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ch = 0; ch < CN; ++ch) px[i].v[ch] = px[i].v[ch] * o[ch];
            } else if (op == CVGS_OP_ADD) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ch = 0; ch < CN; ++ch) px[i].v[ch] = px[i].v[ch] + o[ch];
            } else if (op == CVGS_OP_SUB) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ch = 0; ch < CN; ++ch) px[i].v[ch] = px[i].v[ch] - o[ch];
            } else if (op == CVGS_OP_DIV) {
                if (div4_guarded(div_fits, o, r, px, CN)) return; // (k_common.hpp: the divisors fit and every dividend of the wave does)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ch = 0; ch < CN; ++ch) px[i].v[ch] = px[i].v[ch] / o[ch];
            } else if (op == CVGS_OP_CAST) { // -> CV_32F: 8- / 16-bit integers are exact floats already, CV_32S travels as raw bits
                if constexpr (SD == CVGS_DEPTH_32S) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int ch = 0; ch < CN; ++ch) px[i].v[ch] = (float)as_int(px[i].v[ch]);
                }
            } else { // CVGS_OP_REORDER
#pragma unroll
                for (int i = 0; i < 4; ++i) reorder_px(px[i], aux, CN);
            }
        }
    }
};

struct PwGeom {
    int32_t w, h, used, cn;
    int32_t packed;    // 1: packed pixels (PIXEL_2D / PIXEL_3D), 0: planar tensor, 2: separate pitched planes (SPLIT_2D)
    int32_t row_pitch; // packed: bytes between output rows
    int32_t row_pitch2;
    int32_t narrow;    // log2(rows per wave): 0 = a wave is 64 lanes x 4 pixels of ONE row (-1: and of the row 4 below, pw4_body_rows2); 1 / 2 = 32 / 16 lanes per row, 2 / 4 rows per
                       // wave, for planes at most 128 / 64 pixels wide (a 60-pixel crop would leave 49 of 64 lanes idle)
    int64_t img_stride, ch_stride, img_stride2, ch_stride2; // planar: elements; packed: img_stride in BYTES
    uint8_t* out;
    uint8_t* out2;
};

// u8 -> u8 colour conversions as compile-time programs (k_cvtcolor_u8.hip); 1 = took the chain, 0 = not eligible
int launch_u8_colour16(const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, void* stream, bool dry_run, LaunchInfo* info);
int launch_f32_colour4(const ChainArgs& c, const PlaneParams* ip, int ni, const PwGeom& g, void* stream, bool dry_run, LaunchInfo* info);

// source element -> the work pixel's float (8/16-bit integers exact, CV_32S as raw bits, CV_32F as is: load_px's rule)
template <int SD>
__device__ __forceinline__ float elem_value(const uint32_t* raw, int e) {
    if constexpr (SD == CVGS_DEPTH_8U) return (float)((raw[e >> 2] >> (8 * (e & 3))) & 0xffu);
    else if constexpr (SD == CVGS_DEPTH_8S) return (float)(int8_t)((raw[e >> 2] >> (8 * (e & 3))) & 0xffu);
    else if constexpr (SD == CVGS_DEPTH_16U) return (float)((raw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
    else if constexpr (SD == CVGS_DEPTH_16S) return (float)(int16_t)((raw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
    else return __uint_as_float(raw[e]);
}
// pseudo source depths of pw4_body: 4:2:0 decoder surfaces with interleaved chroma read WITHOUT a resize (CVGS_READ_NV12; NV12 /
// NV21 8-bit samples, P010 16-bit samples) -- the pixel arrives as CV_32F R, G, B[, A] through k_common.hpp's yuv_to_rgb
constexpr int SD_NV12 = 64, SD_P010 = 65, SD_I420 = 66; // SD_I420: planar chroma (I420 / YV12: two quarter-size planes behind the luma)
template <int SD> constexpr bool is_yuv_sd = SD == SD_NV12 || SD == SD_P010 || SD == SD_I420;
template <int SD> constexpr int src_elem_bytes = (SD == CVGS_DEPTH_8U || SD == CVGS_DEPTH_8S) ? 1 : ((SD == CVGS_DEPTH_16U || SD == CVGS_DEPTH_16S) ? 2 : 4);

// The thread's 4 work pixels travel to the write stage BY VALUE.  Round 2 passed `const Px (&)[4]`: after inlining, LLVM kept
// part of the array in memory (SROA gave up on a <4 x float> load that overlapped two pixels), AMDGPUPromoteAlloca moved
// those 24 bytes per thread into LDS for CN = 3 (+6 KB per workgroup, and the kernel started reading the dispatch packet
// for its linear thread id) and into SCRATCH for CN = 4 -- +20..25 us on every launch of these kernels
// (profiles/r02_k vs r02_o).  tests/test_kernel_resources.py now fails the CPU suite on scratch / promoted allocas.
struct Px4 {
    Px p[4];
};
template <int CN, typename OT>
__device__ __forceinline__ void pw4_write(const ChainArgs& c, const PwGeom& g, const Px4 px4, int cn, int bx, int x0, int y, int z, int npx,
                                          int wave, int lane, int sh);
__device__ __forceinline__ Px4 px4_of(const Px (&px)[4]) {
    Px4 q;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) q.p[i].v[ch] = px[i].v[ch];
    return q;
}

// ---- the read stage of a per-pixel source: 4 pixels = 4*CN elements = NDW dwords, one wide load ----
template <int CN, int SD> struct PwRaw { uint32_t w[CN * src_elem_bytes<SD>]; };
template <int CN, int SD>
__device__ __forceinline__ void pw4_load(PwRaw<CN, SD>& raw, const PlaneParams& P, int x0, int y, int npx, bool live) {
    constexpr int EB = src_elem_bytes<SD>;
    constexpr int NDW = CN * EB;
#ifndef CVGS_PW_ABLATE
#define CVGS_PW_ABLATE 0
#endif
    if constexpr ((CVGS_PW_ABLATE & 1) != 0) {
#pragma unroll
        for (int k = 0; k < NDW; ++k) raw.w[k] = (uint32_t)(x0 + y + k);
        return;
    }
    if (live) {
        const gp_u8 row = (gp_u8)P.data + (size_t)y * (size_t)P.step + (size_t)x0 * CN * EB;
        if (npx == 4) {
#pragma unroll
            for (int k = 0; k < NDW; ++k) raw.w[k] = *(gp_u32)(row + 4 * k);
        } else {
#pragma unroll
            for (int k = 0; k < NDW; ++k) raw.w[k] = 0;
#pragma unroll
            for (int b = 0; b < 4 * CN * EB; ++b)
                if (b < npx * CN * EB) raw.w[b >> 2] |= (uint32_t)row[b] << (8 * (b & 3));
        }
    }
}
// ---- the rest of the thread's work on those 4 pixels: source elements -> work pixels, the program, the write stage ----
template <int CN, class Prog, typename OT, int SD>
__device__ __forceinline__ void pw4_finish(const ChainArgs& c, const PwGeom& g, const PwRaw<CN, SD>& raw, int bx, int x0, int y, int z, int npx, int wave,
                                           int lane, int sh) {
    const int used = g.used;
    Px px[4];
    int depth = SD, cn = CN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if (ch < CN) {
                // default-value planes carry the background in the source type (CV_32S: as an integer)
                const float bgv = SD == CVGS_DEPTH_32S ? from_int((int)c.read.bg[ch]) : c.read.bg[ch];
                px[i].v[ch] = z < used ? elem_value<SD>(raw.w, i * CN + ch) : bgv;
            } else {
                px[i].v[ch] = 0.f;
            }
        }
    }
    Prog::run4(c.prog, px, depth, cn);
    if constexpr ((CVGS_PW_ABLATE & 4) != 0) {
        if (!(px[0].v[0] == 1234.5f && px[3].v[0] == 77.25f)) return;
    }
    pw4_write<CN, OT>(c, g, px4_of(px), cn, bx, x0, y, z, npx, wave, lane, sh);
}

// One thread's work: pixels x0..x0+3 of row y of plane z.  (bx, by) = the 256-pixel column group and the 4-row group.
// SD = source depth (8U is the hot one; the reference sweeps its pointwise chains over 8S/16U/16S/32S/32F as well,
// tests/batchread/test_batchread_x_write3D.cu:202-227).
template <int CN, class Prog, typename OT, int SD = CVGS_DEPTH_8U>
__device__ __forceinline__ void pw4_body(const ChainArgs& c, const PlaneParams& P, const PwGeom& g, int bx, int by, int z) {
    const int W = g.w, H = g.h, used = g.used;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int sh = g.narrow; // wave-uniform
    const int lpr = 64 >> sh; // lanes per row
    const int x0 = (bx * lpr + (lane & (lpr - 1))) * 4;
    const int y = ((by * 4 + wave) << sh) + (lane >> (6 - sh));
    if (y >= H || x0 >= W) return;
    const int npx = min(4, W - x0);

    if constexpr (is_yuv_sd<SD>) {
        // ---- 4 pixels of a 4:2:0 surface: 4 luma samples + the 2 chroma pairs they share (x0 is a multiple of 4) ----
        constexpr int SB = SD == SD_P010 ? 2 : 1; // bytes per sample
        const YuvK yk = yuv_matrix(c.read.yuv_range, c.read.yuv_primaries, SD == SD_P010 ? CVGS_YUV_P010 : CVGS_YUV_NV12);
        const bool vu = c.read.yuv_layout == CVGS_YUV_NV21 || c.read.yuv_layout == CVGS_YUV_YV12; // wave-uniform: V comes first
        Px px[4];
        int depth = CVGS_DEPTH_32F, cn = CN;
        if (z < used) {
            const gp_u8 yrow = (gp_u8)P.data + (size_t)y * (size_t)P.step + (size_t)x0 * SB;
            uint32_t yw[SB], cw[SB];
            if constexpr (SD == SD_I420) {
                // planar chroma: (W/2) x (H/2) planes with rows of step/2 bytes, one after the other (nv12_px's addressing); the two
                // samples of each plane are re-interleaved into NV12's pair format
                typedef uint16_t u16u __attribute__((aligned(1)));
                typedef const __attribute__((address_space(1))) u16u* gp_u16;
                const size_t cstep = (size_t)(P.step >> 1);
                const gp_u8 first = (gp_u8)P.data + (size_t)P.uv_off + (size_t)(y >> 1) * cstep + (size_t)(x0 >> 1);
                const gp_u8 second = first + (size_t)(P.h >> 1) * cstep;
                uint32_t a, b;
                if (npx == 4) {
                    yw[0] = *(gp_u32)yrow;
                    a = *(gp_u16)first;
                    b = *(gp_u16)second;
                } else {
                    yw[0] = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < npx) yw[0] |= (uint32_t)yrow[k] << (8 * k);
                    a = first[0];
                    b = second[0];
                }
                cw[0] = (a & 0xffu) | ((b & 0xffu) << 8) | ((a & 0xff00u) << 8) | ((b & 0xff00u) << 16);
            } else {
            const gp_u8 crow = (gp_u8)P.data + (size_t)P.uv_off + (size_t)(y >> 1) * (size_t)P.step + (size_t)x0 * SB;
            if (npx == 4) {
#pragma unroll
                for (int k = 0; k < SB; ++k) {
                    yw[k] = *(gp_u32)(yrow + 4 * k);
                    cw[k] = *(gp_u32)(crow + 4 * k);
                }
            } else { // ragged tail (planes are even-sized: 2 pixels = one chroma pair)
#pragma unroll
                for (int k = 0; k < SB; ++k) yw[k] = cw[k] = 0;
#pragma unroll
                for (int b = 0; b < 4 * SB; ++b)
                    if (b < npx * SB) {
                        yw[b >> 2] |= (uint32_t)yrow[b] << (8 * (b & 3));
                        cw[b >> 2] |= (uint32_t)crow[b] << (8 * (b & 3));
                    }
            }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int iu = 2 * (i >> 1); // the pair's first sample
                float Y, A, B;
                if constexpr (SD == SD_P010) {
                    Y = (float)(((yw[i >> 1] >> (16 * (i & 1))) & 0xffffu) >> 6);
                    A = (float)((cw[iu >> 1] & 0xffffu) >> 6);
                    B = (float)(cw[iu >> 1] >> 22);
                } else {
                    Y = (float)((yw[0] >> (8 * i)) & 0xffu);
                    A = (float)((cw[0] >> (8 * iu)) & 0xffu);
                    B = (float)((cw[0] >> (8 * iu + 8)) & 0xffu);
                }
                yuv_to_rgb(Y, vu ? B : A, vu ? A : B, yk, px[i]);
#pragma unroll
                for (int ch = CN; ch < 4; ++ch) px[i].v[ch] = 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) px[i].v[ch] = ch < CN ? c.read.bg[ch] : 0.f;
        }
        Prog::run4(c.prog, px, depth, cn);
        pw4_write<CN, OT>(c, g, px4_of(px), cn, bx, x0, y, z, npx, wave, lane, sh);
        return;
    }
    PwRaw<CN, SD> raw;
    pw4_load<CN, SD>(raw, P, x0, y, npx, z < used);
    pw4_finish<CN, Prog, OT, SD>(c, g, raw, bx, x0, y, z, npx, wave, lane, sh);
}

// Two rows per thread (round 6; whole ONE-CHANNEL frames: g.narrow == -1, launch_pw): rows y and y + 4 of an 8-row block, both rows' loads in flight
// before either is converted and stored (a one-channel thread moves 20 bytes per row).
template <int CN, class Prog, typename OT, int SD>
__device__ __forceinline__ void pw4_body_rows2(const ChainArgs& c, const PlaneParams& P, const PwGeom& g, int bx, int by, int z) {
    static_assert(!is_yuv_sd<SD>, "per-pixel reads only");
    const int W = g.w, H = g.h, used = g.used;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int x0 = (bx * 64 + lane) * 4;
    const int ya = by * 8 + wave, yb = ya + 4; // wave-uniform
    if (ya >= H || x0 >= W) return;
    const int npx = min(4, W - x0);
    PwRaw<CN, SD> ra, rb;
    pw4_load<CN, SD>(ra, P, x0, ya, npx, z < used);
    if (yb < H) pw4_load<CN, SD>(rb, P, x0, yb, npx, z < used);
    pw4_finish<CN, Prog, OT, SD>(c, g, ra, bx, x0, ya, z, npx, wave, lane, 0);
    if (yb < H) pw4_finish<CN, Prog, OT, SD>(c, g, rb, bx, x0, yb, z, npx, wave, lane, 0);
}

// ---- the write stage of pw4_body: 4 pixels of row y, plane z ----
template <int CN, typename OT>
__device__ __forceinline__ void pw4_write(const ChainArgs& c, const PwGeom& g, const Px4 px4, int cn, int bx, int x0, int y, int z, int npx,
                                          int wave, int lane, int sh) {
    const int W = g.w;
    const Px (&px)[4] = px4.p;
    if (g.packed == 2) {
        // cvGS::split(std::vector<GpuMat>) / SplitWrite<_2D>: cn pitched planes per batch element (the reference's
        // tests/read/test_read_x_split.cu chain): the planar stores below, each plane with its own base and pitch
        const DstPlane* planes = c.write.table ? c.write.table : c.dst_inline; // wave-uniform
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if (ch < cn) {
                const DstPlane d = planes[(size_t)z * cn + ch];
                OT* o = (OT*)(d.data + (size_t)y * (size_t)d.step) + x0;
                if (npx == 4) {
                    store4(o, px[0].v[ch], px[1].v[ch], px[2].v[ch], px[3].v[ch]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < npx) store1(o + i, px[i].v[ch]);
                }
            }
        }
    } else if (g.packed) {
        // cn floats per pixel, contiguous: 4 pixels = cn float4
        uint8_t* rows[2] = {g.out + (size_t)z * g.img_stride + (size_t)y * g.row_pitch,
                            g.out2 ? g.out2 + (size_t)z * g.img_stride2 + (size_t)y * g.row_pitch2 : nullptr};
        constexpr int EPW = 256 * CN; // elements per wave
        __shared__ __attribute__((aligned(16))) OT stage[4][EPW]; // wave-private quarters, wave-synchronous use (no workgroup barrier)
        if (sh == 0 && bx * 256 + 255 < W) { // wave-uniform: the whole 256-pixel group exists, every lane is alive
            // Each lane owns 4*CN consecutive output elements (48 / 64 bytes for fp32): stored directly, every 16-byte
            // store instruction would scatter the wave over a 3-4 KB span.  Transpose through LDS instead: lanes write
            // their elements, then lane l stores the wave's l-th, (64+l)-th, ... 16-byte chunk -> 1 KB contiguous per
            // instruction.  Wave-private LDS region, wave-synchronous (no workgroup barrier).
            constexpr int EPC = 16 / (int)sizeof(OT);      // elements per 16-byte chunk
            constexpr int CHUNKS = EPW / EPC;
            OT* mine = &stage[wave][lane * 4 * CN];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
                    if (ch < CN) mine[i * CN + ch] = cvt_out<OT>(px[i].v[ch]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            typedef u32x4 u32x4_a2 __attribute__((aligned(2))); // fp16 rows may start on any even address,
            typedef u32x4 u32x4_a1 __attribute__((aligned(1))); // u8 rows anywhere
            using u32x4u = std::conditional_t<(sizeof(OT) < 2), u32x4_a1, u32x4_a2>;
            const size_t row_off = (size_t)bx * EPW * sizeof(OT);
#pragma unroll
            for (int k = 0; k < (CHUNKS + 63) / 64; ++k) {
                const int cidx = k * 64 + lane;
                if (cidx < CHUNKS) {
                    const u32x4 q = *(const u32x4*)((const uint8_t*)&stage[wave][0] + (size_t)cidx * 16);
                    __builtin_nontemporal_store(q, (u32x4u*)(rows[0] + row_off + (size_t)cidx * 16));
                    if (rows[1]) __builtin_nontemporal_store(q, (u32x4u*)(rows[1] + row_off + (size_t)cidx * 16));
                }
            }
            return;
        }
        // NARROW planes (round 6; the reference's batched 60 x 120 crops: 16 lanes x 4 rows per wave): a lane's direct stores are 16 bytes at a
        // stride of 16 * CN -- every store instruction touches every line of the wave's rows, CN times over (ticks of 16 x 50 such planes ran at
        // 1.5 TB/s with 4 channels).  When the plane's rows are DENSE the wave's 2 / 4 rows are one contiguous span: the same LDS transpose, the
        // span cut into 16-byte chunks that the wave's ALIVE lanes store in order (lanes past the row's end have left: chunk = round x alive + rank).
        if constexpr (sizeof(OT) == 4) {
            const int nrows = 1 << sh;
            const int y0 = y - (lane >> (6 - sh)); // the wave's first row (wave-uniform)
            const bool dense = sh > 0 && !rows[1] && g.row_pitch == W * CN * (int)sizeof(OT) && (W & 3) == 0 && y0 + nrows <= g.h &&
                               ((nrows * W * CN * (int)sizeof(OT)) & 15) == 0;
            if (dense) { // wave-uniform
                OT* mine = &stage[wave][(((lane >> (6 - sh)) * W) + x0) * CN];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch)
                        if (ch < CN) mine[i * CN + ch] = cvt_out<OT>(px[i].v[ch]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                typedef uint32_t u32x4n __attribute__((ext_vector_type(4)));
                typedef u32x4n u32x4n_a4 __attribute__((aligned(4)));
                const uint64_t alive = __builtin_amdgcn_ballot_w64(true);
                const int n_alive = (int)__builtin_popcountll(alive);
                const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(alive >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)alive, 0u));
                const int chunks = nrows * W * CN * (int)sizeof(OT) / 16;
                uint8_t* const span = g.out + (size_t)z * g.img_stride + (size_t)y0 * g.row_pitch;
                for (int cidx = rank; cidx < chunks; cidx += n_alive) {
                    const u32x4n q = *(const u32x4n*)((const uint8_t*)&stage[wave][0] + (size_t)cidx * 16);
                    __builtin_nontemporal_store(q, (u32x4n_a4*)(span + (size_t)cidx * 16));
                }
                return;
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (!rows[t]) continue;
            OT* o = (OT*)rows[t] + (size_t)x0 * cn;
            if (npx == 4) {
                float flat[16];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch)
                        if (ch < CN) flat[i * CN + ch] = px[i].v[ch];
#pragma unroll
                for (int v = 0; v < CN; ++v) store4(o + 4 * v, flat[4 * v], flat[4 * v + 1], flat[4 * v + 2], flat[4 * v + 3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch)
                        if (i < npx && ch < CN) store1(o + i * CN + ch, px[i].v[ch]);
            }
        }
    } else {
        OT* bases[2] = {(OT*)g.out + (int64_t)z * g.img_stride, g.out2 ? (OT*)g.out2 + (int64_t)z * g.img_stride2 : nullptr};
        const int64_t chs[2] = {g.ch_stride, g.ch_stride2};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (!bases[t]) continue;
            OT* o = bases[t] + (int64_t)y * W + x0;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                if (ch < cn) {
                    if (npx == 4) {
                        store4(o + (int64_t)ch * chs[t], px[0].v[ch], px[1].v[ch], px[2].v[ch], px[3].v[ch]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (i < npx) store1(o + (int64_t)ch * chs[t] + i, px[i].v[ch]);
                    }
                }
            }
        }
    }
}

// host side (k_pointwise.hip): eligibility + geometry of the thread-fused path
// (u8out: packed u8 pixels behind a 4:2:0 read, accepted only when the caller passes the flag)
bool pointwise4_plan(const ChainArgs& c_in, int n_inline, uint32_t chain_flags, ChainArgs& c, PwGeom& g, int& prog_id, bool& f16, bool* u8out = nullptr);

} // namespace cvgs
