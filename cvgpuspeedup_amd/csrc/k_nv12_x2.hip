// k_nv12_x2.hip -- K4 at frame size: two x-adjacent output pixels per lane.
//
// The chain: Resize<INTER_LINEAR>(ReadYUV<NV12> + ConvertYUVToRGB) -> [RGB<->BGR] mul sub div -> planar fp32 tensor -- BASELINE cfg #3
// (a 6K decoder surface -> 1280 x 720 normalized NCHW; reference tests/resize/test_fused_resize.cu:141-147 fused with the K1 tail).
// k4_nv12_resize (k_nv12.hip: lane = one output column, wave = one output row) runs cfg #3 in 8.4 us: 14,400 waves of 140 VALU +
// 90 SALU instructions (profiles/r03_f_cfg3_pmc_sq1.txt), and no memory-side change moves it (cached surfaces: 7.3-7.7 us; the
// descriptor queue: 6.9-7.3).  This kernel halves the waves and runs the arithmetic that dominates -- four YCbCr -> RGB conversions
// and the bilinear blend per pixel -- on PIXEL PAIRS (v_pk_mul_f32 / v_pk_add_f32: the same IEEE operations in the same order, two
// pixels per instruction; the multi-pixel resize kernel k_k1_x4.hip does the same for packed images):
//  * lane = 2 x-adjacent output pixels, wave = 128 output columns; each pixel keeps K4's loads (one unaligned 2-byte load for both
//    luma taps of a source row, one 4-byte load for both chroma pairs): 8 loads per lane and row;
//  * large launches walk kN2Rows rows per wave with the NEXT row's tap words requested before the current row is computed, the
//    stores through one buffer descriptor (a store past the target is dropped by the hardware: no control flow around it, so the
//    compiler's s_waitcnt counts stay exact and no row ever waits for the previous row's stores); small ones keep one row per wave;
//  * the scalar side (row geometry, row pointers, program operands) is paid once per 128 columns instead of once per 64;
//  * planar stores are 8 bytes per lane and channel: 512-byte rows per wave, non-temporal.
// Measured (tools/bench_more.py, tools/bench_nv12_letterbox.py): cfg #3 8.4 -> 8.0 us (round 3), 1080p / 4K surface -> 640 x 640 detector
// input 4.3 / 4.6 -> 3.9 / 4.4 us.  What bounds one cfg #3 launch, decomposed on the final tree (round 6: the kernel's own skeletons,
// tools/probes/tick_ablation.py --workload cfg3, profiles/r06_c_cfg3_rows2_m1.txt): an empty launch of the grid 1.82 us; the arithmetic alone
// (no loads, no stores) 4.27; the tap loads alone 5.54; the stores alone 3.02; loads + stores without arithmetic 7.36; the product kernel
// 7.36-7.50 -- it runs AT its memory skeleton (round 3's "7.2 us with loads and stores disabled" described the kernel before its VALU
// diet; the arithmetic now hides under the load phase).  The skeleton itself is a launch floor + 26 MB that every wave reads, then
// writes, in the same phase: a single frame-size launch has no second launch to overlap with (ticks of several surfaces do: 5.2 us per frame).
// Stretch geometry only (every surface covers its whole target), NV12 / NV21 8-bit, three channels, the two compile-time
// programs; everything else stays on k4_nv12_resize.  Bit-identical to it and to the oracle (tests/test_gpu_k4_x2.py).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "k_taps.hpp"

namespace cvgs {

constexpr int kN2Planes = 8; // surfaces per launch (blockIdx.z)
// Launch shapes (round 6, tools/probes/tick_ablation.py --workload cfg3, round-robin on one box; profiles/r06_c_cfg3_*): the pipelined form walks
// kN2Rows rows per wave in workgroups of kN2PipeWaves waves -- 2 rows x 1 wave: 7.36 us at cfg #3 and 5.24 us per frame in a tick of 4 surfaces, against
// 7.97 / 5.95 for round 5's 4 rows x 4 waves (8 rows: 9.7; every row's loads up front: 7.85; an XCD-banded tile order: 8.16).  Both are macros only so
// that tools/probes/build_ablate.sh can build the other shapes.
#ifndef CVGS_K4_WAVES
#define CVGS_K4_WAVES 1
#endif
#ifndef CVGS_K4_ROWS
#define CVGS_K4_ROWS 2
#endif
constexpr int kN2Waves = 4;                 // waves per workgroup of the one-row form (independent)
constexpr int kN2PipeWaves = CVGS_K4_WAVES; // ... of the pipelined form
constexpr int kN2Rows = CVGS_K4_ROWS;       // rows per wave of the pipelined form (even)

// Ablation hooks of tools/probes/tick_ablation.py --workload cfg3 (round 6; never defined in the product build, see k_k1_impl.hpp):
//   1 no tap loads (words synthesised from the lane id)   2 no arithmetic (the tap words are stored as they are)
//   4 no stores (behind a test that never holds)          16 the wave ends behind its geometry
#ifndef CVGS_K4_ABLATE
#define CVGS_K4_ABLATE 0
#endif
constexpr int kN2Ablate = CVGS_K4_ABLATE;

typedef uint16_t n2_u16_unaligned __attribute__((aligned(1)));
typedef uint32_t n2_u32_unaligned __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) n2_u16_unaligned* n2_gptr_u16;
typedef const __attribute__((address_space(1))) n2_u32_unaligned* n2_gptr_u32;
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct N2Plane { // 32 bytes
    const uint8_t* data;
    int32_t w, h, step, uv_off;
    float fx, fy;
};
struct N2Args {
    N2Plane plane[kN2Planes];
    float* out;
    int64_t img_stride, ch_stride; // elements
    int32_t dst_w, dst_h, out_w;
    int32_t yuv_range, yuv_primaries, yuv_vu;
    uint32_t out_bytes;            // RPW > 1: the whole tensor behind one buffer descriptor
    ProgArgs prog; // [swap,] mul, sub, div with the host's reciprocals (fast_div)
};

// one tap of the pair: k4_tap's expressions on two pixels at once
struct N2Rgb { f32x2 r, g, b; };
template <bool FULL>
__device__ __forceinline__ N2Rgb n2_tap(f32x2 Y, f32x2 U, f32x2 V, const YuvK& k) {
    const f32x2 cb = U - k.csub, cr = V - k.csub;
    const f32x2 yv = FULL ? Y : (Y - k.ysub) * k.yscale;
    N2Rgb t;
    t.r = yv + k.rv * cr;
    t.g = (yv + k.gu * cb) + k.gv * cr;
    t.b = yv + k.bu * cb;
    return t;
}

// The kernel's leading SCALAR parameters (14 dwords: what the hardware preloads into user SGPRs with the dispatch; Makefile: PRELOAD):
// surface 0 and the target's extent -- a one-surface launch (cfg #3) asks for its taps without waiting for any load of the argument block.
#define N2_PRELOADED_PARAMS                                                                                                          \
    const uint8_t *p0_data, int32_t p0_w, int32_t p0_h, int32_t p0_step, int32_t p0_uv_off, float p0_fx, float p0_fy, int32_t pre_dst_w, \
        int32_t pre_dst_h, int32_t pre_out_w, int32_t pre_yuv_range, int32_t pre_yuv_primaries, int32_t pre_yuv_vu
template <class Prog, int RPW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k4_nv12_x2(N2_PRELOADED_PARAMS, const N2Args a) {
    const int z = (int)blockIdx.z;
    N2Plane P = N2Plane{p0_data, p0_w, p0_h, p0_step, p0_uv_off, p0_fx, p0_fy};
    if (z != 0) P = a.plane[z];
    const int dst_w = pre_dst_w, dst_h = pre_dst_h, W = pre_out_w;
    const int yuv_range = pre_yuv_range, vu = pre_yuv_vu;
    const YuvK yk = yuv_matrix(yuv_range, pre_yuv_primaries, CVGS_YUV_NV12);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    int col_tile = (int)blockIdx.x, row_group = (int)blockIdx.y;
#if defined(CVGS_K4_XCD) && CVGS_K4_XCD // probe: workgroup i runs on XCD i % 8 -- give every XCD one contiguous band of the target's row groups
    {
        const uint32_t total = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
        const uint32_t k = lin % 8u, base = total / 8u, rem = total % 8u;
        const uint32_t tile = k * base + (k < rem ? k : rem) + lin / 8u; // XCD k: tiles [start_k, start_k + base + (k < rem))
        col_tile = (int)(tile % gridDim.x);
        row_group = (int)(tile / gridDim.x);
    }
#endif
    const int row0 = (row_group * WAVES + wave) * RPW;
    const int x0 = (col_tile * 64 + lane) * 2;
    if (row0 >= dst_h || x0 >= dst_w) return;
    if constexpr ((kN2Ablate & 16) != 0) return;

    // ---- column geometry of the lane's two pixels (k4_nv12_resize's, per pixel) ----
    f32x2 wxa, wxb;
    uint32_t yo[2], uo[2];
    // v_perm_b32 selectors that pick a pixel's four tap samples {a0, a1, b0, b1} (row a / b, tap 0 / 1) out of its two 2-byte luma
    // loads / its two 4-byte chroma loads: what used to be a shift, a mask and a select per sample -- the luma window clamped back at
    // the right edge (then both taps are its second byte), the chroma window clamped back at the last pair, taps sharing a chroma pair,
    // NV21's byte order (csrc/k_queue.hip: k4q_rows does the same)
    uint32_t sel_y[2], sel_u[2], sel_v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int x = min(x0 + i, dst_w - 1); // an odd target's last lane computes its last pixel twice and stores it once
        const float sx = (float)x * P.fx;
        const int x1 = (int)floorf(sx);
        const int x2 = x1 + 1;
        wxa[i] = (float)x2 - sx;
        wxb[i] = sx - (float)x1;
        const bool edge = x2 > P.w - 1;
        const int x2r = edge ? x1 : x2;
        yo[i] = (uint32_t)min(x1, P.w - 2);
        const int c1 = x1 >> 1, c2 = x2r >> 1;
        uo[i] = (uint32_t)min(2 * c1, P.w - 4);
        sel_y[i] = edge ? 0x05050101u : 0x05040100u;
        const uint32_t pr0 = 2 * c1 != (int)uo[i] ? 2u : 0u; // byte of tap 0's pair inside the chroma window
        const uint32_t pr1 = c2 == c1 ? pr0 : 2u;            // ... of tap 1's
        const uint32_t sel_c = pr0 | (pr1 << 8) | ((4u + pr0) << 16) | ((4u + pr1) << 24);
        sel_u[i] = sel_c + (vu ? 0x01010101u : 0u);
        sel_v[i] = sel_c + (vu ? 0u : 0x01010101u);
    }
    const gptr_u8 base = (gptr_u8)P.data;
    const size_t step = (size_t)P.step;
    const gptr_u8 uvp = base + (size_t)P.uv_off;

    struct Raw {
        uint32_t vya[2], vyb[2], vua[2], vub[2];
        float wya, wyb;
    };
    auto load_row = [&](int y_in) { // the tap words of one output row: requests only
        Raw r;
        const int y = min(y_in, dst_h - 1);
        const float sy = (float)y * P.fy;
        const int y1 = (int)floorf(sy);
        const int y2 = y1 + 1;
        const int y2r = min(y2, P.h - 1);
        r.wya = (float)y2 - sy;
        r.wyb = sy - (float)y1;
        const int r1 = __builtin_amdgcn_readfirstlane(y1), r2 = __builtin_amdgcn_readfirstlane(y2r);
        const gptr_u8 ya = pin_uniform(base + (size_t)r1 * step), yb = pin_uniform(base + (size_t)r2 * step);
        const gptr_u8 ua = pin_uniform(uvp + (size_t)(r1 >> 1) * step), ub = pin_uniform(uvp + (size_t)(r2 >> 1) * step);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr ((kN2Ablate & 1) != 0) { // probe: no tap loads
                r.vya[i] = (uint32_t)(lane * 3 + r1 + i) & 0xffffu;
                r.vyb[i] = (uint32_t)(lane * 5 + r2 + i) & 0xffffu;
                r.vua[i] = (uint32_t)(lane * 0x01010101 + r1);
                r.vub[i] = (uint32_t)(lane * 0x01010101 + r2);
            } else {
                r.vya[i] = *(n2_gptr_u16)(ya + yo[i]);
                r.vyb[i] = *(n2_gptr_u16)(yb + yo[i]);
                r.vua[i] = *(n2_gptr_u32)(ua + uo[i]);
                r.vub[i] = *(n2_gptr_u32)(ub + uo[i]);
            }
        }
        return r;
    };

    float* const out = a.out + (int64_t)z * a.img_stride;
    const bool both = x0 + 1 < dst_w;
    // RPW > 1: the output tensor behind ONE buffer descriptor -- a store whose offset lies beyond it is dropped by the hardware,
    // so rows past the target and the control flow around them disappear and every row issues the same 8 loads + 3 stores
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);
    auto finish_row = [&](const Raw& raw, int y) {
        Px p0, p1;
        p0.v[3] = p1.v[3] = 0.f;
        if constexpr ((kN2Ablate & 2) != 0) { // probe: no arithmetic -- the tap words as they are
            p0.v[0] = __uint_as_float(raw.vya[0] | (raw.vyb[0] << 16)); p0.v[1] = __uint_as_float(raw.vua[0]); p0.v[2] = __uint_as_float(raw.vub[0]);
            p1.v[0] = __uint_as_float(raw.vya[1] | (raw.vyb[1] << 16)); p1.v[1] = __uint_as_float(raw.vua[1]); p1.v[2] = __uint_as_float(raw.vub[1]);
        } else {
        f32x2 fy[4], fu[4], fv[4]; // taps 00, 10, 01, 11 of the pixel pair
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t ly = __builtin_amdgcn_perm(raw.vyb[i], raw.vya[i], sel_y[i]);
            const uint32_t lu = __builtin_amdgcn_perm(raw.vub[i], raw.vua[i], sel_u[i]), lv = __builtin_amdgcn_perm(raw.vub[i], raw.vua[i], sel_v[i]);
            fy[0][i] = (float)(ly & 0xffu); fy[1][i] = (float)((ly >> 8) & 0xffu); fy[2][i] = (float)((ly >> 16) & 0xffu); fy[3][i] = (float)(ly >> 24);
            fu[0][i] = (float)(lu & 0xffu); fu[1][i] = (float)((lu >> 8) & 0xffu); fu[2][i] = (float)((lu >> 16) & 0xffu); fu[3][i] = (float)(lu >> 24);
            fv[0][i] = (float)(lv & 0xffu); fv[1][i] = (float)((lv >> 8) & 0xffu); fv[2][i] = (float)((lv >> 16) & 0xffu); fv[3][i] = (float)(lv >> 24);
        }
        N2Rgb t[4];
        if (yuv_range == CVGS_YUV_FULL) { // wave-uniform
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = n2_tap<true>(fy[k], fu[k], fv[k], yk);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = n2_tap<false>(fy[k], fu[k], fv[k], yk);
        }
        const f32x2 w00 = wxa * raw.wya, w10 = wxb * raw.wya, w01 = wxa * raw.wyb, w11 = wxb * raw.wyb;
        f32x2 acc[3];
        {
            f32x2 v = t[0].r * w00; v = v + t[1].r * w10; v = v + t[2].r * w01; v = v + t[3].r * w11; acc[0] = v;
            v = t[0].g * w00; v = v + t[1].g * w10; v = v + t[2].g * w01; v = v + t[3].g * w11; acc[1] = v;
            v = t[0].b * w00; v = v + t[1].b * w10; v = v + t[2].b * w01; v = v + t[3].b * w11; acc[2] = v;
        }
        // the program, per pixel: the compile-time stages of k_taps.hpp (incl. the division by the uniform divisor)
        p0.v[0] = acc[0].x; p0.v[1] = acc[1].x; p0.v[2] = acc[2].x;
        p1.v[0] = acc[0].y; p1.v[1] = acc[1].y; p1.v[2] = acc[2].y;
        int depth = CVGS_DEPTH_32F, cn = 3;
        Prog::run(a.prog, p0, depth, cn);
        depth = CVGS_DEPTH_32F; cn = 3;
        Prog::run(a.prog, p1, depth, cn);
        }
        if constexpr ((kN2Ablate & 4) != 0) { // probe: no stores (the test never holds on pixel data)
            if (!(__float_as_uint(p0.v[0]) == 0x7fc12345u && __float_as_uint(p0.v[1]) == 0x7fc54321u && __float_as_uint(p1.v[2]) == 0x7fc00001u)) return;
        }
        if constexpr (RPW > 1) {
            // (even target widths only: the launcher checks) offsets in bytes from the tensor's start fit 32 bits (checked on the host)
            const uint32_t row_off = (uint32_t)(((int64_t)z * a.img_stride + (int64_t)y * W) * 4);
            const uint32_t off = y < dst_h ? row_off + (uint32_t)x0 * 4u : 0xfffffff0u;
            typedef uint32_t u32x2q __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const u32x2q q = {__float_as_uint(p0.v[k]), __float_as_uint(p1.v[k])};
                __builtin_amdgcn_raw_buffer_store_b64(q, rsrc, off, (uint32_t)((int64_t)k * a.ch_stride * 4), 2 /* nt */);
            }
        } else {
            float* const orow = out + (int64_t)y * W; // wave-uniform
            typedef __attribute__((address_space(1))) char* gchar;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const gchar r = (gchar)(__attribute__((address_space(1))) float*)pin_uniform(orow + (int64_t)k * a.ch_stride);
                const uint32_t xb = (uint32_t)x0 * 4u;
                if (both) {
                    typedef float f2u __attribute__((ext_vector_type(2)));
                    typedef f2u f2u_a4 __attribute__((aligned(4)));
                    const f2u q = {p0.v[k], p1.v[k]};
                    __builtin_nontemporal_store(q, (__attribute__((address_space(1))) f2u_a4*)(r + xb));
                } else {
                    __builtin_nontemporal_store(p0.v[k], (__attribute__((address_space(1))) float*)(r + xb));
                }
            }
        }
    };

    if constexpr (RPW == 1) {
        finish_row(load_row(row0), row0);
    } else {
        // the wave's rows one after the other, the NEXT row's tap words requested before this row is computed: across the chip the
        // rows' loads and stores interleave in time (a launch of one-row waves reads everything, then writes everything)
#if defined(CVGS_K4_UPFRONT) && CVGS_K4_UPFRONT // probe (tools/probes/build_ablate.sh): every row's tap words requested before the first row is computed
        Raw r[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) r[j] = load_row(row0 + j);
#pragma unroll
        for (int j = 0; j < RPW; ++j) finish_row(r[j], row0 + j);
#else
        Raw r0 = load_row(row0), r1;
#pragma unroll
        for (int j = 0; j < RPW; j += 2) {
            r1 = load_row(row0 + j + 1);
            finish_row(r0, row0 + j);
            if (j + 2 < RPW) r0 = load_row(row0 + j + 2);
            finish_row(r1, row0 + j + 1);
        }
#endif
    }
}

// Host side: 1 launched / 0 not eligible / < 0 error (as launch_nv12).  `c.prog` holds the chain's program with its trailing
// stages as launch_nv12 prepared them (fast_div set up); prog_swap says which compile-time program it is.
int launch_nv12_x2(const ChainArgs& c, const PlaneParams* planes, int n_planes, bool prog_swap, void* stream, bool dry_run) {
    const ReadArgs& r = c.read;
    const WriteArgs& w = c.write;
    if (r.kind != CVGS_READ_NV12_RESIZE_LINEAR || (r.yuv_layout != CVGS_YUV_NV12 && r.yuv_layout != CVGS_YUV_NV21) || r.out_cn != 3) return 0;
    if (r.table || w.data2 || w.depth != CVGS_DEPTH_32F || (w.kind != CVGS_WRITE_TENSOR_SPLIT && w.kind != CVGS_WRITE_TENSOR_T_SPLIT)) return 0;
    if (r.batch < 1 || r.batch > kN2Planes || r.used != r.batch || n_planes != r.batch || r.dst_w < 2) return 0;
    N2Args a;
    for (int i = 0; i < kN2Planes; ++i) {
        const PlaneParams& p = planes[i < n_planes ? i : 0];
        if (i < n_planes && (p.w < 4 || p.x1 != 0 || p.y1 != 0 || p.x2 != r.dst_w - 1 || p.y2 != r.dst_h - 1)) return 0;
        a.plane[i] = N2Plane{p.data, p.w, p.h, p.step, p.uv_off, p.fx, p.fy};
    }
    if (dry_run) return 1;
    a.out = (float*)w.data;
    a.img_stride = w.img_stride;
    a.ch_stride = w.ch_stride;
    a.dst_w = r.dst_w;
    a.dst_h = r.dst_h;
    a.out_w = w.width;
    a.yuv_range = r.yuv_range;
    a.yuv_primaries = r.yuv_primaries;
    a.yuv_vu = r.yuv_layout == CVGS_YUV_NV21;
    a.prog = c.prog;
    // rows per wave: 1 (small / odd-width targets: maximum parallelism, the last lane may store ONE pixel) or kN2Rows, walked with the
    // next row's loads in flight (even widths, tensors below 4 GB: the stores go through one buffer descriptor)
    const int64_t total_bytes = ((int64_t)(r.batch - 1) * w.img_stride + 2 * w.ch_stride + (int64_t)r.dst_h * w.width) * 4;
    const bool pipe = (r.dst_w & 1) == 0 && total_bytes > 0 && total_bytes < ((int64_t)1 << 32) - 65536 &&
                      (int64_t)r.batch * r.dst_h * ((r.dst_w + 127) / 128) >= 4096;
    a.out_bytes = pipe ? (uint32_t)total_bytes : 0u;
    const int rpw = pipe ? kN2Rows : 1, waves = pipe ? kN2PipeWaves : kN2Waves;
    const unsigned col_tiles = (unsigned)((r.dst_w + 127) / 128), row_groups = (unsigned)((r.dst_h + waves * rpw - 1) / (waves * rpw));
    const dim3 grid(col_tiles, row_groups, (unsigned)r.batch), block(64 * waves);
    hipStream_t s = (hipStream_t)stream;
    const N2Plane& p0 = a.plane[0];
#define N2_PRELOADED_VALUES p0.data, p0.w, p0.h, p0.step, p0.uv_off, p0.fx, p0.fy, a.dst_w, a.dst_h, a.out_w, a.yuv_range, a.yuv_primaries, a.yuv_vu
    if (prog_swap) {
        if (pipe) hipLaunchKernelGGL((k4_nv12_x2<ProgSwapMulSubDiv, kN2Rows, kN2PipeWaves>), grid, block, 0, s, N2_PRELOADED_VALUES, a);
        else hipLaunchKernelGGL((k4_nv12_x2<ProgSwapMulSubDiv, 1, kN2Waves>), grid, block, 0, s, N2_PRELOADED_VALUES, a);
    } else {
        if (pipe) hipLaunchKernelGGL((k4_nv12_x2<ProgMulSubDiv, kN2Rows, kN2PipeWaves>), grid, block, 0, s, N2_PRELOADED_VALUES, a);
        else hipLaunchKernelGGL((k4_nv12_x2<ProgMulSubDiv, 1, kN2Waves>), grid, block, 0, s, N2_PRELOADED_VALUES, a);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

} // namespace cvgs
