// k_generic64.hip -- the interpreted kernel for chains that touch CV_64F (source, intermediate or output type):
// the reference's K5/K6/K7 sweeps include CV_32F -> CV_64F pairs (tests/read/test_read_x_write.cu:139-141,
// tests/batchread/test_batchread_x_write3D.cu:222-224).  Work registers are doubles; while the value's type is
// CV_32F every operation is carried out in float and widened back (exact), so the result is bit-identical to a
// float pipeline up to the cast and to a double pipeline after it.  Integer values (incl. 32S) are held as exact
// doubles.  Reads: per-pixel reads of any depth, resize/NV12 reads of non-64F sources (their output is float).
#include "k_common.hpp"

namespace cvgs {

struct Px64 {
    double v[4];
};

__device__ __forceinline__ double sel4(const Px64& p, int k) {
    return k == 0 ? p.v[0] : (k == 1 ? p.v[1] : (k == 2 ? p.v[2] : p.v[3]));
}

__device__ __forceinline__ void reorder64(Px64& p, int aux, int out_cn) {
    const Px64 s = p;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < out_cn) p.v[c] = sel4(s, (aux >> (2 * c)) & 3);
}

__device__ __forceinline__ void int_range64(int depth, double& lo, double& hi) {
    switch (depth) {
    case CVGS_DEPTH_8U: lo = 0.0; hi = 255.0; break;
    case CVGS_DEPTH_8S: lo = -128.0; hi = 127.0; break;
    case CVGS_DEPTH_16U: lo = 0.0; hi = 65535.0; break;
    case CVGS_DEPTH_16S: lo = -32768.0; hi = 32767.0; break;
    default: lo = -2147483648.0; hi = 2147483647.0; break;
    }
}

template <bool TRUNC = false>
__device__ __forceinline__ void cast64(Px64& p, int cn, int src, int dst) {
    if (src == CVGS_DEPTH_16F) src = CVGS_DEPTH_32F; // a half value is carried as the float (double) it equals
    if (src == dst || dst == CVGS_DEPTH_64F) return; // widening to double is exact
    if (dst == CVGS_DEPTH_16F) { // through float, then round to nearest even once more (the oracle's order: (float)d, then half)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < cn) p.v[c] = (double)round_half((float)p.v[c]);
        return;
    }
    if (dst == CVGS_DEPTH_32F) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < cn) p.v[c] = (double)(float)p.v[c];
        return;
    }
    double lo, hi;
    int_range64(dst, lo, hi);
    const bool from_float = src == CVGS_DEPTH_32F || src == CVGS_DEPTH_64F;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c < cn) {
            double v = p.v[c];
            // nearest even; NaN -> 0; '+ 0.0' turns the -0.0 that rint(-0.3) gives into the +0 an integer type holds
            if (from_float) v = (v != v) ? 0.0 : (TRUNC ? trunc(v) : rint(v)) + 0.0; // fk::Cast truncates
            p.v[c] = fmin(fmax(v, lo), hi);
        }
    }
}

__device__ __forceinline__ void apply_op64(int opc, int aux, const float* of, const double* od, Px64& p, int& depth, int& cn) {
    switch (opc) {
    case CVGS_OP_CAST:
        cast64(p, cn, depth, aux);
        depth = aux;
        break;
    case CVGS_OP_CAST_TRUNC:
        cast64<true>(p, cn, depth, aux);
        depth = aux;
        break;
    case CVGS_OP_MUL: case CVGS_OP_ADD: case CVGS_OP_SUB: case CVGS_OP_DIV:
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < cn) {
                if (depth == CVGS_DEPTH_64F) {
                    const double a = p.v[c], b = od[c];
                    p.v[c] = opc == CVGS_OP_MUL ? a * b : opc == CVGS_OP_ADD ? a + b : opc == CVGS_OP_SUB ? a - b : a / b;
                } else if (is_int_depth(depth)) { // integer-typed value: k_common.hpp int_arith (the operand arrives as an exact integer)
                    long long lo, hi;
                    int_range(depth, lo, hi);
                    p.v[c] = (double)int_arith(opc, (long long)p.v[c], (long long)od[c], lo, hi);
                } else {
                    const float a = (float)p.v[c], b = of[c];
                    const float r = opc == CVGS_OP_MUL ? a * b : opc == CVGS_OP_ADD ? a + b : opc == CVGS_OP_SUB ? a - b : a / b;
                    p.v[c] = (double)r;
                }
            }
        }
        break;
    case CVGS_OP_REORDER: reorder64(p, aux, cn); break;
    case CVGS_OP_ADD_ALPHA:
        reorder64(p, aux, 3);
        p.v[3] = (double)of[0];
        cn = 4;
        break;
    case CVGS_OP_DROP_ALPHA:
        reorder64(p, aux, 3);
        cn = 3;
        break;
    case CVGS_OP_GRAY: {
        const float r = (float)sel4(p, aux & 3), g = (float)sel4(p, (aux >> 2) & 3), b = (float)sel4(p, (aux >> 4) & 3);
        float lum = (r * 0.299f + g * 0.587f) + b * 0.114f;
        if (depth != CVGS_DEPTH_32F) lum = rintf(lum) + 0.0f; // integer result: no -0
        p.v[0] = (double)lum;
        cn = 1;
        break;
    }
    default: break;
    }
}

__device__ __forceinline__ void load64(const uint8_t* row, int depth, int cn, int x, Px64& p) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c < cn) {
            const size_t e = (size_t)x * cn + c;
            switch (depth) {
            case CVGS_DEPTH_8U: p.v[c] = (double)row[e]; break;
            case CVGS_DEPTH_8S: p.v[c] = (double)((const int8_t*)row)[e]; break;
            case CVGS_DEPTH_16U: p.v[c] = (double)((const uint16_t*)row)[e]; break;
            case CVGS_DEPTH_16S: p.v[c] = (double)((const int16_t*)row)[e]; break;
            case CVGS_DEPTH_32S: p.v[c] = (double)((const int32_t*)row)[e]; break;
            case CVGS_DEPTH_32F: p.v[c] = (double)((const float*)row)[e]; break;
            case CVGS_DEPTH_16F: p.v[c] = (double)(float)((const _Float16*)row)[e]; break;
            default: p.v[c] = ((const double*)row)[e]; break;
            }
        }
    }
}

__device__ __forceinline__ void store64(uint8_t* base, size_t idx, int depth, double v) {
    switch (depth) {
    case CVGS_DEPTH_8U: base[idx] = (uint8_t)v; break;
    case CVGS_DEPTH_8S: ((int8_t*)base)[idx] = (int8_t)v; break;
    case CVGS_DEPTH_16U: ((uint16_t*)base)[idx] = (uint16_t)v; break;
    case CVGS_DEPTH_16S: ((int16_t*)base)[idx] = (int16_t)v; break;
    case CVGS_DEPTH_32S: ((int32_t*)base)[idx] = (int32_t)v; break;
    case CVGS_DEPTH_32F: ((float*)base)[idx] = (float)v; break;
    case CVGS_DEPTH_16F: ((_Float16*)base)[idx] = (_Float16)(float)v; break;
    default: ((double*)base)[idx] = v; break;
    }
}

__device__ __forceinline__ void write64(const WriteArgs& w, const DstPlane* dst_planes, int x, int y, int z, const Px64& p,
                                        int depth, int cn) {
    const size_t W = (size_t)w.width;
    switch (w.kind) {
    case CVGS_WRITE_PIXEL_2D: {
        uint8_t* row = w.data + (size_t)y * (size_t)w.step;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) store64(row, (size_t)x * cn + c, depth, p.v[c]);
        break;
    }
    case CVGS_WRITE_PIXEL_2D_BATCH: {
        const DstPlane d = dst_planes[z];
        uint8_t* row = d.data + (size_t)y * (size_t)d.step;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) store64(row, (size_t)x * cn + c, depth, p.v[c]);
        break;
    }
    case CVGS_WRITE_PIXEL_3D: {
        const size_t pix = (size_t)z * w.img_stride + (size_t)y * W + x;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) store64(w.data, pix * cn + c, depth, p.v[c]);
        if (w.data2) {
            const size_t pix2 = (size_t)z * w.img_stride2 + (size_t)y * W + x;
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < cn) store64(w.data2, pix2 * cn + c, depth, p.v[c]);
        }
        break;
    }
    case CVGS_WRITE_TENSOR_SPLIT:
    case CVGS_WRITE_TENSOR_T_SPLIT:
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < cn) store64(w.data, (size_t)z * w.img_stride + (size_t)c * w.ch_stride + (size_t)y * W + x, depth, p.v[c]);
        if (w.data2) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < cn) store64(w.data2, (size_t)z * w.img_stride2 + (size_t)c * w.ch_stride2 + (size_t)y * W + x, depth, p.v[c]);
        }
        break;
    case CVGS_WRITE_SPLIT_2D:
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < cn) {
                const DstPlane d = dst_planes[(size_t)z * cn + c];
                store64(d.data + (size_t)y * (size_t)d.step, (size_t)x, depth, p.v[c]);
            }
        }
        break;
    default: break;
    }
}

template <int NPL>
__global__ __launch_bounds__(256) void k_generic64(const KernArgs64<NPL> a) {
    const ChainArgs& c = a.c;
    const ReadArgs& r = c.read;
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int z = blockIdx.z;
    if (x >= r.dst_w || y >= r.dst_h) return;

    Px64 p;
    p.v[0] = p.v[1] = p.v[2] = p.v[3] = 0.0;
    int depth = r.is_resize || r.kind == CVGS_READ_NV12 ? CVGS_DEPTH_32F : r.depth;
    int cn = r.out_cn;
    if (z >= r.used) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            p.v[k] = depth == CVGS_DEPTH_32S ? (double)(int)r.bg[k] : (depth == CVGS_DEPTH_16F ? (double)round_half(r.bg[k]) : (double)r.bg[k]);
    } else {
        PlaneParams P;
        if constexpr (NPL == 0) P = r.table[z];
        else P = a.planes[z];
        YuvK yk = yuv_matrix(r.yuv_range, r.yuv_primaries, r.yuv_layout);
        if (!r.is_resize) {
            if (r.kind == CVGS_READ_NV12) {
                Px t;
                nv12_px(P, x, y, yk, t);
#pragma unroll
                for (int k = 0; k < 4; ++k) p.v[k] = (double)t.v[k];
            } else {
                load64(P.data + (size_t)y * (size_t)P.step, r.depth, r.cn, x, p);
            }
        } else if (x >= P.x1 && x <= P.x2 && y >= P.y1 && y <= P.y2) {
            const float sx = (float)(x - P.x1) * P.fx, sy = (float)(y - P.y1) * P.fy;
            const int x1 = (int)floorf(sx), y1 = (int)floorf(sy);
            const int x2 = x1 + 1, y2 = y1 + 1;
            const int x2r = min(x2, P.w - 1), y2r = min(y2, P.h - 1);
            float t[4][4];
            const int xs[4] = {x1, x2r, x1, x2r}, ys[4] = {y1, y1, y2r, y2r};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (r.kind == CVGS_READ_NV12_RESIZE_LINEAR) {
                    Px tp;
                    nv12_px(P, xs[q], ys[q], yk, tp);
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[q][k] = tp.v[k];
                } else {
                    Px64 tp;
                    tp.v[0] = tp.v[1] = tp.v[2] = tp.v[3] = 0.0;
                    load64(P.data + (size_t)ys[q] * (size_t)P.step, r.depth, r.cn, xs[q], tp);
#pragma unroll
                    for (int k = 0; k < 4; ++k) t[q][k] = (float)tp.v[k]; // taps are cast to float first
                }
            }
            const float w00 = ((float)x2 - sx) * ((float)y2 - sy), w10 = (sx - (float)x1) * ((float)y2 - sy);
            const float w01 = ((float)x2 - sx) * (sy - (float)y1), w11 = (sx - (float)x1) * (sy - (float)y1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float acc = t[0][k] * w00;
                acc = acc + t[1][k] * w10;
                acc = acc + t[2][k] * w01;
                acc = acc + t[3][k] * w11;
                p.v[k] = (double)acc;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) p.v[k] = (double)r.bg[k];
        }
    }
    for (int k = 0; k < c.prog.n; ++k)
        apply_op64(c.prog.opcode[k], c.prog.aux[k], c.prog.operand[k], a.p64.operand[k], p, depth, cn);
    const DstPlane* dst = c.write.table ? c.write.table : c.dst_inline;
    write64(c.write, dst, x, y, z, p, depth, cn);
}

int launch_generic64(const ChainArgs& c, const Prog64Args& p64, const PlaneParams* inline_planes, int n_inline, void* stream,
                     bool dry_run, LaunchInfo* info) {
    if (info) info->kernel = c.read.table ? "generic64_table" : "generic64_inline8";
    if (dry_run) return 0;
    const dim3 block(64, 4, 1);
    const dim3 grid((c.read.dst_w + 63) / 64, (c.read.dst_h + 3) / 4, c.read.batch);
    hipStream_t s = (hipStream_t)stream;
    if (c.read.table) {
        KernArgs64<0> a;
        a.c = c;
        a.p64 = p64;
        a.planes[0] = PlaneParams{};
        hipLaunchKernelGGL(k_generic64<0>, grid, block, 0, s, a);
    } else {
        KernArgs64<8> a;
        a.c = c;
        a.p64 = p64;
        for (int i = 0; i < 8; ++i) a.planes[i] = i < n_inline ? inline_planes[i] : PlaneParams{};
        hipLaunchKernelGGL(k_generic64<8>, grid, block, 0, s, a);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- warp reads in front of a CV_64F program ---------------------------------------------------------------------------
// cvGS::warp produces CV_32F values (k_warp.hip holds the read's arithmetic, restated here unchanged); a chain that then
// detours through CV_64F (convertTo<CV_32FC3, CV_64FC3>, arithmetic on doubles, ...) needs the double work registers of this
// file.  Not a hot path: one thread per output pixel, interpreted program.
template <int NPL>
struct WarpKernArgs64 {
    ChainArgs c;
    Prog64Args p64;
    WarpPlane planes[NPL > 0 ? NPL : 1];
};

template <int NPL>
__global__ __launch_bounds__(256) void k_warp64(const WarpKernArgs64<NPL> a, const WarpPlane* __restrict__ table) {
    const ChainArgs& c = a.c;
    const ReadArgs& r = c.read;
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int z = blockIdx.z;
    if (x >= r.dst_w || y >= r.dst_h) return;
    WarpPlane P;
    if constexpr (NPL == 0) P = table[z];
    else P = a.planes[z];
    if (x >= P.dw || y >= P.dh) return; // this plane's own destination size
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (z >= r.used) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = r.bg[k];
    } else {
        const float fx = (float)x, fy = (float)y;
        float sx = (P.m[0] * fx + P.m[1] * fy) + P.m[2];
        float sy = (P.m[3] * fx + P.m[4] * fy) + P.m[5];
        if (r.kind == CVGS_READ_WARP_PERSPECTIVE) {
            const float w = (P.m[6] * fx + P.m[7] * fy) + P.m[8];
            sx = sx / w;
            sy = sy / w;
        }
        if (sx >= 0.f && sx < (float)P.w && sy >= 0.f && sy < (float)P.h) {
            const int x1 = (int)floorf(sx), y1 = (int)floorf(sy);
            const int x2 = x1 + 1, y2 = y1 + 1;
            const int x2r = min(x2, P.w - 1), y2r = min(y2, P.h - 1);
            Px p00, p10, p01, p11;
            const uint8_t* ra = P.data + (size_t)y1 * (size_t)P.step;
            const uint8_t* rb = P.data + (size_t)y2r * (size_t)P.step;
            if (r.depth == CVGS_DEPTH_64F) { // CV_64F sources (reference warp<WT, InputType> takes any type, include/cvGPUSpeedup.cuh:285-292):
                                              // taps are cast to float first, as in the resize (:227)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < r.cn) {
                        p00.v[k] = (float)((const double*)ra)[x1 * r.cn + k];
                        p10.v[k] = (float)((const double*)ra)[x2r * r.cn + k];
                        p01.v[k] = (float)((const double*)rb)[x1 * r.cn + k];
                        p11.v[k] = (float)((const double*)rb)[x2r * r.cn + k];
                    }
            } else {
                load_px(ra, r.depth, r.cn, x1, p00);
                load_px(ra, r.depth, r.cn, x2r, p10);
                load_px(rb, r.depth, r.cn, x1, p01);
                load_px(rb, r.depth, r.cn, x2r, p11);
            }
            const float w00 = ((float)x2 - sx) * ((float)y2 - sy);
            const float w10 = (sx - (float)x1) * ((float)y2 - sy);
            const float w01 = ((float)x2 - sx) * (sy - (float)y1);
            const float w11 = (sx - (float)x1) * (sy - (float)y1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < r.cn) {
                    float acc = tap_f(p00.v[k], r.depth) * w00;
                    acc = acc + tap_f(p10.v[k], r.depth) * w10;
                    acc = acc + tap_f(p01.v[k], r.depth) * w01;
                    acc = acc + tap_f(p11.v[k], r.depth) * w11;
                    v[k] = acc;
                }
            }
        }
    }
    Px64 p;
#pragma unroll
    for (int k = 0; k < 4; ++k) p.v[k] = (double)v[k];
    int depth = CVGS_DEPTH_32F, cn = r.out_cn;
    for (int k = 0; k < c.prog.n; ++k)
        apply_op64(c.prog.opcode[k], c.prog.aux[k], c.prog.operand[k], a.p64.operand[k], p, depth, cn);
    const DstPlane* dst = c.write.table ? c.write.table : c.dst_inline;
    write64(c.write, dst, x, y, z, p, depth, cn);
}

// `planes`: host array of n (inline when n <= kInlineWarp64), else `dev_table` (device copy)
int launch_warp64(const ChainArgs& c, const Prog64Args& p64, const WarpPlane* planes, int n, const WarpPlane* dev_table, void* stream,
                  bool dry_run, LaunchInfo* info) {
    if (info) info->kernel = dev_table ? "warp64_table" : "warp64_inline8";
    if (dry_run) return 0;
    const dim3 block(64, 4, 1);
    const dim3 grid((c.read.dst_w + 63) / 64, (c.read.dst_h + 3) / 4, c.read.batch);
    hipStream_t s = (hipStream_t)stream;
    if (dev_table) {
        WarpKernArgs64<0> a;
        a.c = c;
        a.p64 = p64;
        a.planes[0] = WarpPlane{};
        hipLaunchKernelGGL(k_warp64<0>, grid, block, 0, s, a, dev_table);
    } else {
        WarpKernArgs64<kInlineWarp64> a;
        a.c = c;
        a.p64 = p64;
        for (int i = 0; i < kInlineWarp64; ++i) a.planes[i] = i < n ? planes[i] : WarpPlane{};
        hipLaunchKernelGGL(k_warp64<kInlineWarp64>, grid, block, 0, s, a, (const WarpPlane*)nullptr);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // namespace cvgs
