/*
 * cvgs_oracle.c -- scalar, strict-IEEE CPU restatement of the cvGPUSpeedup hot path.
 * TEST INFRASTRUCTURE ONLY (see cvgs_oracle.h).  Build: oracle/Makefile (gcc -O2 -ffp-contract=off).
 *
 * Citations are to /root/reference (v0.21.0).  "[FKL]" marks semantics that live in the
 * un-vendored FusedKernelLibrary 0.1.8 and are restated from its published algorithm and from the
 * OpenCV-CUDA operation the reference's tests compare it with.
 */
#include "cvgs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* work pixel: the value flowing between IOps.  32S values live in i[], everything else in f[] */
/* (8/16-bit integers are exact in fp32).                                                      */
typedef struct opx {
    float f[4];
    int32_t i[4];
    double d[4]; /* CV_64F values */
    int depth;
    int cn;
} opx;

static int g_threads = 1;

void oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int oracle_get_threads(void) { return g_threads; }
int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int depth_bytes(int depth) {
    switch (depth) {
    case CVGS_DEPTH_8U: case CVGS_DEPTH_8S: return 1;
    case CVGS_DEPTH_16U: case CVGS_DEPTH_16S: case CVGS_DEPTH_16F: return 2;
    case CVGS_DEPTH_32S: case CVGS_DEPTH_32F: return 4;
    case CVGS_DEPTH_64F: return 8;
    }
    return 0;
}

/* CV_16F (IEEE binary16), the half-precision hand-off option of include/cvgs_hip.h -- an extension: the reference
 * has no half type, so this restates IEEE 754 conversion (round to nearest even, overflow to infinity, subnormals
 * kept), which is what cv::saturate_cast<cv::float16_t>(float) and the GPU's v_cvt_f16_f32 do.  gcc 11 has no
 * _Float16 on x86-64, hence the bit-level spelling; tests/test_oracle_independent.py checks it against numpy. */
static uint16_t half_bits_from_float(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t mag = u & 0x7fffffffu;
    if (mag >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (mag > 0x7f800000u ? 0x200u | ((mag >> 13) & 0x3ffu) : 0u));
    if (mag >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* >= 65520: rounds to infinity */
    if (mag < 0x33000001u) return (uint16_t)sign;              /* <= 2^-25: rounds to zero (tie goes to even = 0) */
    int exp = (int)(mag >> 23) - 127;
    uint32_t man = (mag & 0x7fffffu) | 0x800000u; /* 24-bit significand */
    int shift = exp < -14 ? 13 + (-14 - exp) : 13; /* bits dropped; subnormal halves drop more */
    uint32_t q = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) ++q;
    uint32_t h;
    if (exp < -14) h = q;                                   /* subnormal (a carry into 0x400 is the smallest normal) */
    else h = ((uint32_t)(exp + 15) << 10) + (q - 0x400u);   /* a carry out of the significand bumps the exponent */
    return (uint16_t)(sign | h);
}

static float float_from_half_bits(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    uint32_t u;
    if (exp == 0x1fu) u = sign | 0x7f800000u | (man << 13);
    else if (exp) u = sign | ((exp + 112u) << 23) | (man << 13);
    else if (!man) u = sign;
    else { /* subnormal: value = man * 2^-24, exact in fp32 */
        float f = (float)man * 5.9604644775390625e-08f;
        memcpy(&u, &f, 4);
        u |= sign;
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

uint16_t oracle_float_to_half(float v) { return half_bits_from_float(v); }
float oracle_half_to_float(uint16_t h) { return float_from_half_bits(h); }
static float round_half(float v) { return float_from_half_bits(half_bits_from_float(v)); }

/* fk::PerThreadRead<_2D,T>::exec: *(T*)((char*)data + y*pitch + x*sizeof(T))   [FKL]
 * (the reference builds it from a GpuMat at include/cvGPUSpeedup.cuh:40-44,613-615). */
static void load_pixel(const cvgs_image2d* im, int type, int x, int y, opx* p) {
    const int depth = CVGS_TYPE_DEPTH(type), cn = CVGS_TYPE_CN(type);
    const uint8_t* row = (const uint8_t*)im->data + (size_t)y * (size_t)im->step;
    p->depth = depth;
    p->cn = cn;
    for (int c = 0; c < cn; ++c) {
        const size_t e = (size_t)x * cn + c;
        switch (depth) {
        case CVGS_DEPTH_8U: p->f[c] = (float)row[e]; break;
        case CVGS_DEPTH_8S: p->f[c] = (float)((const int8_t*)row)[e]; break;
        case CVGS_DEPTH_16U: p->f[c] = (float)((const uint16_t*)row)[e]; break;
        case CVGS_DEPTH_16S: p->f[c] = (float)((const int16_t*)row)[e]; break;
        case CVGS_DEPTH_32S: p->i[c] = ((const int32_t*)row)[e]; break;
        case CVGS_DEPTH_32F: p->f[c] = ((const float*)row)[e]; break;
        case CVGS_DEPTH_64F: p->d[c] = ((const double*)row)[e]; break;
        case CVGS_DEPTH_16F: p->f[c] = float_from_half_bits(((const uint16_t*)row)[e]); break;
        default: p->f[c] = 0.f;
        }
    }
}

static float tap_as_float(const opx* p, int c) {
    if (p->depth == CVGS_DEPTH_64F) return (float)p->d[c];
    return p->depth == CVGS_DEPTH_32S ? (float)p->i[c] : p->f[c];
}

/* ------------------------------------------------------------------------------------------ */
/* Resize geometry.
 * IGNORE_AR: OpenCV-CUDA's host code passes static_cast<float>(1.0 / fx) with
 * fx = (double)dsize.width / src.cols to its kernel, and fk::Resize::build does the same [FKL]
 * (facade call site: reference include/cvGPUSpeedup.cuh:204-216,234-243).
 * PRESERVE_AR: scale by height; if the scaled width overflows scale by width; centre; the
 * OpenCV side of the reference test spells the same geometry at
 * tests/batchresize/test_batchresize_aspectratio_x_split3D.cu:86-95,129-133.  The target extent
 * is rounded to nearest (fk compute_target_size [FKL]); the reference test truncates instead and
 * the two agree on its only tested case (30x120 -> 32x128).  RN_EVEN rounds the free extent down
 * to an even number, LEFT pins the window to x = 0 [FKL, unpinned]. */
void oracle_resize_geometry(int32_t src_w, int32_t src_h, int32_t dst_w, int32_t dst_h,
                            int32_t aspect_ratio, oracle_resize_geom* g) {
    if (aspect_ratio == CVGS_IGNORE_AR) {
        const double cfx = (double)dst_w / (double)src_w;
        const double cfy = (double)dst_h / (double)src_h;
        g->fx = (float)(1.0 / cfx);
        g->fy = (float)(1.0 / cfy);
        g->x1 = 0;
        g->y1 = 0;
        g->x2 = dst_w - 1;
        g->y2 = dst_h - 1;
        return;
    }
    float scale = (float)dst_h / (float)src_h;
    int th = dst_h;
    int tw = (int)roundf(scale * (float)src_w);
    if (aspect_ratio == CVGS_PRESERVE_AR_RN_EVEN) tw -= tw % 2;
    if (tw > dst_w) {
        scale = (float)dst_w / (float)src_w;
        tw = dst_w;
        th = (int)roundf(scale * (float)src_h);
        if (aspect_ratio == CVGS_PRESERVE_AR_RN_EVEN) th -= th % 2;
    }
    if (tw < 1) tw = 1;
    if (th < 1) th = 1;
    const int x1 = aspect_ratio == CVGS_PRESERVE_AR_LEFT ? 0 : (dst_w - tw) / 2;
    const int y1 = (dst_h - th) / 2;
    const double cfx = (double)tw / (double)src_w;
    const double cfy = (double)th / (double)src_h;
    g->fx = (float)(1.0 / cfx);
    g->fy = (float)(1.0 / cfy);
    g->x1 = x1;
    g->y1 = y1;
    g->x2 = x1 + tw - 1;
    g->y2 = y1 + th - 1;
}

/* ------------------------------------------------------------------------------------------ */
/* NV12 -> RGB(A) float.  fk::ReadYUV<NV12>: Y at data[y][x], interleaved UV plane right below the
 * luma rows: (U,V) = data[H + y/2][2*(x/2) + {0,1}] (allocation H + H/2 rows, reference
 * tests/resize/test_fused_resize.cu:39-40,112-113).  fk::ConvertYUVToRGB<NV12,range,primaries,
 * alpha,floatN> (:50-51,141-142): 3x3 YCbCr->RGB matrix of the named standard [FKL, unpinned:
 * the reference test asserts nothing about the values]. */
typedef struct yuv_coeffs {
    float ysub, yscale, rv, gu, gv, bu;
    float csub, amax; /* chroma centre and alpha (= full scale): 128 / 255 for 8-bit samples, 512 / 1023 for P010 */
} yuv_coeffs;

/* Coefficients written as the 6-decimal literals of the published derivation from the standards' luma weights
 * (BT.601 Kr 0.299 Kb 0.114, BT.709 Kr 0.2126 Kb 0.0722, BT.2020 non-constant luminance Kr 0.2627 Kb 0.0593):
 * R = Y' + 2(1-Kr) Cr, B = Y' + 2(1-Kb) Cb, G = Y' - 2Kb(1-Kb)/Kg Cb - 2Kr(1-Kr)/Kg Cr; limited range scales luma by
 * 255/219 and chroma by 255/224 on 8-bit codes, by 1023/876 and 1023/896 on 10-bit codes.  tests/test_independent_pins.py
 * re-derives every set in float64. */
static yuv_coeffs yuv_matrix(int range, int primaries, int ten_bit) {
    static const float full[3][4] = {{1.402f, -0.344136f, -0.714136f, 1.772f},
                                     {1.5748f, -0.187324f, -0.468124f, 1.8556f},
                                     {1.4746f, -0.164553f, -0.571353f, 1.8814f}};
    static const float lim8[3][4] = {{1.596027f, -0.391762f, -0.812968f, 2.017232f},
                                     {1.792741f, -0.213249f, -0.532909f, 2.112402f},
                                     {1.678674f, -0.187326f, -0.650424f, 2.141772f}};
    static const float lim10[3][4] = {{1.600721f, -0.392915f, -0.815359f, 2.023165f},
                                      {1.798014f, -0.213876f, -0.534477f, 2.118615f},
                                      {1.683611f, -0.187877f, -0.652337f, 2.148072f}};
    yuv_coeffs k;
    const float* m;
    if (range == CVGS_YUV_FULL) {
        k.ysub = 0.f;
        k.yscale = 1.f;
        m = full[primaries];
    } else {
        k.ysub = ten_bit ? 64.f : 16.f;
        k.yscale = ten_bit ? 1.167808f : 1.164383f;
        m = ten_bit ? lim10[primaries] : lim8[primaries];
    }
    k.rv = m[0]; k.gu = m[1]; k.gv = m[2]; k.bu = m[3];
    k.csub = ten_bit ? 512.f : 128.f;
    k.amax = ten_bit ? 1023.f : 255.f;
    return k;
}

static void nv12_pixel(const cvgs_image2d* im, int x, int y, const cvgs_read_desc* rd, opx* p) {
    const uint8_t* base = (const uint8_t*)im->data;
    /* a crop of a surface carries its own luma -> chroma offset (cvgs_image2d.uv_offset); 0 = the whole surface */
    const size_t uv_off = im->uv_offset ? (size_t)im->uv_offset : (size_t)im->height * (size_t)im->step;
    /* 4:2:0 layouts (cvgs_yuv_layout): interleaved (U,V) [NV12] or (V,U) [NV21] pairs, one per 2x2 luma block, in rows of
     * `step` bytes; or planar chroma [I420: U plane then V plane, YV12: V then U], (W/2) x (H/2) samples in rows of
     * step/2 bytes; or P010 = NV12 with 16-bit samples carrying a 10-bit code in their high bits.  The reference
     * instantiates fk::ReadYUV<fk::NV12> only (tests/resize/test_fused_resize.cu:50). */
    float Y, U, V;
    if (rd->yuv_layout == CVGS_YUV_P010) {
        const uint16_t* yrow = (const uint16_t*)(base + (size_t)y * im->step);
        const uint16_t* uv = (const uint16_t*)(base + uv_off + (size_t)(y / 2) * im->step) + 2 * (x / 2);
        Y = (float)(yrow[x] >> 6);
        U = (float)(uv[0] >> 6);
        V = (float)(uv[1] >> 6);
    } else if (rd->yuv_layout <= CVGS_YUV_NV21) {
        const uint8_t* uv = base + uv_off + (size_t)(y / 2) * im->step + 2 * (x / 2);
        Y = (float)base[(size_t)y * im->step + x];
        U = (float)uv[rd->yuv_layout == CVGS_YUV_NV21 ? 1 : 0];
        V = (float)uv[rd->yuv_layout == CVGS_YUV_NV21 ? 0 : 1];
    } else {
        const size_t cstep = (size_t)(im->step / 2);
        const uint8_t* first = base + uv_off + (size_t)(y / 2) * cstep + (size_t)(x / 2);
        const uint8_t* second = first + (size_t)(im->height / 2) * cstep;
        Y = (float)base[(size_t)y * im->step + x];
        U = (float)*(rd->yuv_layout == CVGS_YUV_YV12 ? second : first);
        V = (float)*(rd->yuv_layout == CVGS_YUV_YV12 ? first : second);
    }
    const yuv_coeffs k = yuv_matrix(rd->yuv_range, rd->yuv_primaries, rd->yuv_layout == CVGS_YUV_P010);
    const float cb = U - k.csub;
    const float cr = V - k.csub;
    const float yv = (Y - k.ysub) * k.yscale;
    p->f[0] = yv + k.rv * cr;
    p->f[1] = (yv + k.gu * cb) + k.gv * cr;
    p->f[2] = yv + k.bu * cb;
    p->f[3] = k.amax;
    p->depth = CVGS_DEPTH_32F;
    p->cn = rd->yuv_alpha ? 4 : 3;
}

/* source pixel fetch used by the interpolator: the Resize BackIOp */
static void back_read(const cvgs_read_desc* rd, const cvgs_image2d* im, int x, int y, opx* p) {
    if (rd->kind == CVGS_READ_NV12 || rd->kind == CVGS_READ_NV12_RESIZE_LINEAR) nv12_pixel(im, x, y, rd, p);
    else load_pixel(im, rd->src_type, x, y, p);
}

/* fk::Interpolate<INTER_LINEAR> [FKL], identical to OpenCV-CUDA's resize_linear kernel which the
 * reference tests compare against (cv::cuda::resize(..., INTER_LINEAR),
 * tests/batchresize/test_batchresize_x_split3D.cu:297): no half-pixel offset, floor, +1 clamped
 * to the LAST source column/row, weights from the unclamped x2/y2, taps cast to float first, the
 * four products summed in the order 00,10,01,11 in fp32. */
static void interpolate_linear(const cvgs_read_desc* rd, const cvgs_image2d* im, float src_x,
                               float src_y, opx* out) {
    const int x1 = (int)floorf(src_x);
    const int y1 = (int)floorf(src_y);
    const int x2 = x1 + 1;
    const int y2 = y1 + 1;
    const int x2r = x2 < im->width - 1 ? x2 : im->width - 1;
    const int y2r = y2 < im->height - 1 ? y2 : im->height - 1;
    opx p00, p10, p01, p11;
    back_read(rd, im, x1, y1, &p00);
    back_read(rd, im, x2r, y1, &p10);
    back_read(rd, im, x1, y2r, &p01);
    back_read(rd, im, x2r, y2r, &p11);
    const float w00 = ((float)x2 - src_x) * ((float)y2 - src_y);
    const float w10 = (src_x - (float)x1) * ((float)y2 - src_y);
    const float w01 = ((float)x2 - src_x) * (src_y - (float)y1);
    const float w11 = (src_x - (float)x1) * (src_y - (float)y1);
    out->depth = CVGS_DEPTH_32F;
    out->cn = p00.cn;
    for (int c = 0; c < p00.cn; ++c) {
        float acc = tap_as_float(&p00, c) * w00;
        acc = acc + tap_as_float(&p10, c) * w10;
        acc = acc + tap_as_float(&p01, c) * w01;
        acc = acc + tap_as_float(&p11, c) * w11;
        out->f[c] = acc;
    }
}

static void background_pixel(const cvgs_read_desc* rd, int depth, int cn, opx* p) {
    p->depth = depth;
    p->cn = cn;
    for (int c = 0; c < cn; ++c) {
        p->f[c] = depth == CVGS_DEPTH_16F ? round_half(rd->background[c]) : rd->background[c];
        p->i[c] = (int32_t)rd->background[c];
        p->d[c] = (double)rd->background[c];
    }
}

/* The read stage for output element (x,y) of plane z.
 * fk::BatchRead<N,CONDITIONAL_WITH_DEFAULT>: z >= usedPlanes -> default value [FKL]
 * (reference include/cvGPUSpeedup.cuh:240-243,506-516).
 * fk::Resize<LINEAR,AR>: inside the AR window interpolate at ((x-x1)*fx,(y-y1)*fy), outside ->
 * background [FKL] (:238-244). */
static void read_stage(const cvgs_read_desc* rd, const oracle_resize_geom* geoms, int x, int y, int z,
                       opx* p) {
    const cvgs_image2d* im = (const cvgs_image2d*)rd->src + z;
    const int is_resize = rd->kind == CVGS_READ_RESIZE_LINEAR || rd->kind == CVGS_READ_NV12_RESIZE_LINEAR;
    const int is_warp = rd->kind == CVGS_READ_WARP_AFFINE || rd->kind == CVGS_READ_WARP_PERSPECTIVE;
    int out_depth, out_cn;
    if (rd->kind == CVGS_READ_PIXEL) {
        out_depth = CVGS_TYPE_DEPTH(rd->src_type);
        out_cn = CVGS_TYPE_CN(rd->src_type);
    } else if (rd->kind == CVGS_READ_RESIZE_LINEAR || is_warp) {
        out_depth = CVGS_DEPTH_32F;
        out_cn = CVGS_TYPE_CN(rd->src_type);
    } else {
        out_depth = CVGS_DEPTH_32F;
        out_cn = rd->yuv_alpha ? 4 : 3;
    }
    if (z >= rd->used_planes) {
        background_pixel(rd, out_depth, out_cn, p);
        return;
    }
    if (is_warp) {
        /* fk::Warping<WT, BackIOp> [FKL; unpinned beyond the translation case of the reference test, which must equal
         * cv::cuda::warpAffine(..., INTER_LINEAR, BORDER_CONSTANT 0), tests/warping/test_warping_opencv.cu:80-117]:
         * source position = M * (x, y, 1) with the inverse transform the facade narrowed to float
         * (include/cvGPUSpeedup.cuh:269-284), perspective divides by the third row; inside the source the
         * INTER_LINEAR interpolation of the resize, outside zero. */
        const float* m = rd->warp_matrices + (size_t)z * 9;
        const float fx = (float)x, fy = (float)y;
        float sx = (m[0] * fx + m[1] * fy) + m[2];
        float sy = (m[3] * fx + m[4] * fy) + m[5];
        if (rd->kind == CVGS_READ_WARP_PERSPECTIVE) {
            const float w = (m[6] * fx + m[7] * fy) + m[8];
            sx = sx / w;
            sy = sy / w;
        }
        if (sx >= 0.f && sx < (float)im->width && sy >= 0.f && sy < (float)im->height) {
            interpolate_linear(rd, im, sx, sy, p);
        } else {
            p->depth = out_depth;
            p->cn = out_cn;
            for (int c = 0; c < 4; ++c) { p->f[c] = 0.f; p->i[c] = 0; p->d[c] = 0.0; }
        }
        return;
    }
    if (!is_resize) {
        back_read(rd, im, x, y, p);
        return;
    }
    const oracle_resize_geom* g = geoms + z;
    if (x >= g->x1 && x <= g->x2 && y >= g->y1 && y <= g->y2) {
        const float src_x = (float)(x - g->x1) * g->fx;
        const float src_y = (float)(y - g->y1) * g->fy;
        interpolate_linear(rd, im, src_x, src_y, p);
    } else {
        background_pixel(rd, out_depth, out_cn, p);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* fk::SaturateCast<I,O> [FKL] == cv::saturate_cast on the CUDA side: float -> integer rounds to
 * nearest EVEN then clamps (KAT 20*0.5+0.5=10.5 -> 10, 15.5 -> 16: reference
 * tests/single_operation/test_convertTo.cu:69-73), NaN -> 0; integer -> narrower integer clamps;
 * anything -> float is a plain conversion. */
static void int_range(int depth, int64_t* lo, int64_t* hi) {
    switch (depth) {
    case CVGS_DEPTH_8U: *lo = 0; *hi = 255; break;
    case CVGS_DEPTH_8S: *lo = -128; *hi = 127; break;
    case CVGS_DEPTH_16U: *lo = 0; *hi = 65535; break;
    case CVGS_DEPTH_16S: *lo = -32768; *hi = 32767; break;
    default: *lo = INT32_MIN; *hi = INT32_MAX; break;
    }
}

/* trunc != 0: fk::Cast<I,O> = static_cast per channel (reference tests/warping/test_warping_opencv.cu:63): float ->
 * integer truncates toward zero; out-of-range values (undefined in C++) saturate, NaN -> 0. */
static void op_cast_mode(opx* p, int dst_depth, int trunc_mode) {
    int src_depth = p->depth;
    if (src_depth == dst_depth) return;
    if (src_depth == CVGS_DEPTH_16F) src_depth = CVGS_DEPTH_32F; /* a half value is carried as the float it equals */
    p->depth = dst_depth;
    if (src_depth == dst_depth) return;
    int64_t lo, hi;
    int_range(dst_depth, &lo, &hi);
    for (int c = 0; c < p->cn; ++c) {
        if (dst_depth == CVGS_DEPTH_16F) {
            p->f[c] = round_half(src_depth == CVGS_DEPTH_32S ? (float)p->i[c] : src_depth == CVGS_DEPTH_64F ? (float)p->d[c] : p->f[c]);
            continue;
        }
        if (dst_depth == CVGS_DEPTH_64F) { /* every narrower type converts exactly */
            p->d[c] = src_depth == CVGS_DEPTH_32S ? (double)p->i[c] : (double)p->f[c];
            continue;
        }
        if (dst_depth == CVGS_DEPTH_32F) {
            p->f[c] = src_depth == CVGS_DEPTH_32S ? (float)p->i[c] : src_depth == CVGS_DEPTH_64F ? (float)p->d[c] : p->f[c];
            continue;
        }
        int64_t iv;
        if (src_depth == CVGS_DEPTH_32F || src_depth == CVGS_DEPTH_64F) {
            const double v = src_depth == CVGS_DEPTH_64F ? p->d[c] : (double)p->f[c];
            if (v != v) iv = 0;
            else if (v >= 2147483648.0) iv = INT32_MAX;
            else if (v <= -2147483649.0) iv = INT32_MIN;
            else iv = (int64_t)(trunc_mode ? trunc(v) : nearbyint(v)); /* default rounding mode: nearest even (exact for float inputs too) */
        } else if (src_depth == CVGS_DEPTH_32S) {
            iv = p->i[c];
        } else {
            iv = (int64_t)p->f[c];
        }
        if (iv < lo) iv = lo;
        if (iv > hi) iv = hi;
        if (dst_depth == CVGS_DEPTH_32S) p->i[c] = (int32_t)iv;
        else p->f[c] = (float)iv;
    }
    p->depth = dst_depth;
}

static void op_cast(opx* p, int dst_depth) { op_cast_mode(p, dst_depth, 0); }

static void op_reorder(opx* p, int aux, int out_cn) {
    opx s = *p;
    for (int c = 0; c < out_cn; ++c) {
        const int k = (aux >> (2 * c)) & 3;
        p->f[c] = s.f[k];
        p->i[c] = s.i[k];
        p->d[c] = s.d[k];
    }
    p->cn = out_cn;
}

/* One Unary/Binary IOp.  fk::Mul/Add/Sub/Div<floatN>: per-channel IEEE fp32, true division (the
 * reference builds with fast-math OFF, cmake/libs/cuda/target_generation.cmake:11-12); operands
 * were narrowed double->float by cvScalar2CUDAV (include/cvGPUSpeedupHelpers.cuh:38-54). */
static int apply_op(const cvgs_op* op, opx* p) {
    switch (op->opcode) {
    case CVGS_OP_NOP: return 0;
    case CVGS_OP_CAST: op_cast(p, op->aux); return 0;
    case CVGS_OP_CAST_TRUNC:
        op_cast_mode(p, op->aux, 1);
        return 0;
    case CVGS_OP_MUL: case CVGS_OP_ADD: case CVGS_OP_SUB: case CVGS_OP_DIV:
        if (p->depth == CVGS_DEPTH_64F) { /* fk::Mul/Add/Sub/Div<doubleN>: IEEE fp64 with the double operand */
            for (int c = 0; c < p->cn; ++c) {
                const double a = p->d[c], b = op->operand_d[c];
                p->d[c] = op->opcode == CVGS_OP_MUL ? a * b : op->opcode == CVGS_OP_ADD ? a + b
                        : op->opcode == CVGS_OP_SUB ? a - b : a / b;
            }
            return 0;
        }
        if (p->depth == CVGS_DEPTH_16F) return CVGS_ERR_UNSUPPORTED;
        if (p->depth != CVGS_DEPTH_32F) {
            /* Integer-typed values (cvGS::multiply / add / subtract / divide<I>, integer I; reference include/cvGPUSpeedup.cuh:
             * 131-149 instantiates fk::Mul<uchar3> etc., whose definition lives in the absent FKL and which no reference test
             * uses).  Semantics fixed by this build: the scalar as the pixel's own type (cvScalar2CUDAV<I>:
             * truncation, saturated), 64-bit integer arithmetic, division truncating toward zero with x / 0 = 0, result
             * saturated to the type. */
            static const double lo[5] = {0, -128, 0, -32768, -2147483648.0}, hi[5] = {255, 127, 65535, 32767, 2147483647.0};
            const int d = p->depth;
            for (int c = 0; c < p->cn; ++c) {
                double sv = op->operand_d[c];
                if (op->operand_d[0] == 0 && op->operand_d[1] == 0 && op->operand_d[2] == 0 && op->operand_d[3] == 0) sv = op->operand[c];
                sv = sv != sv ? 0.0 : trunc(sv);
                sv = sv < lo[d] ? lo[d] : (sv > hi[d] ? hi[d] : sv);
                const long long b = (long long)sv;
                const long long a = d == CVGS_DEPTH_32S ? (long long)p->i[c] : (long long)p->f[c];
                long long r = op->opcode == CVGS_OP_MUL ? a * b : op->opcode == CVGS_OP_ADD ? a + b : op->opcode == CVGS_OP_SUB ? a - b
                              : (b == 0 ? 0 : a / b);
                r = r < (long long)lo[d] ? (long long)lo[d] : (r > (long long)hi[d] ? (long long)hi[d] : r);
                if (d == CVGS_DEPTH_32S) p->i[c] = (int32_t)r;
                else p->f[c] = (float)r;
            }
            return 0;
        }
        for (int c = 0; c < p->cn; ++c) {
            const float a = p->f[c], b = op->operand[c];
            p->f[c] = op->opcode == CVGS_OP_MUL ? a * b
                    : op->opcode == CVGS_OP_ADD ? a + b
                    : op->opcode == CVGS_OP_SUB ? a - b
                                                : a / b;
        }
        return 0;
    /* fk::ColorConversion channel permutations (codes listed at reference
     * include/cv2cuda_types.cuh:77-85); RGB2BGR = swap 0<->2, RGBA2BGRA keeps 3. */
    case CVGS_OP_REORDER: op_reorder(p, op->aux, p->cn); return 0;
    case CVGS_OP_ADD_ALPHA:
        op_reorder(p, op->aux, 3);
        p->f[3] = op->operand[0];
        p->i[3] = (int32_t)op->operand[0];
        p->d[3] = (double)op->operand[0];
        p->cn = 4;
        return 0;
    case CVGS_OP_DROP_ALPHA: op_reorder(p, op->aux, 3); return 0;
    /* *2GRAY: CCIR 601 luma, KATs RGB(10,100,200) -> 84, BGR -> 120
     * (reference tests/color/test_cvtColor.cu:36,115-123). */
    case CVGS_OP_GRAY: {
        if (p->depth == CVGS_DEPTH_32S || p->depth == CVGS_DEPTH_64F || p->depth == CVGS_DEPTH_16F) return CVGS_ERR_UNSUPPORTED;
        const float r = p->f[op->aux & 3], g = p->f[(op->aux >> 2) & 3], b = p->f[(op->aux >> 4) & 3];
        float lum = (r * 0.299f + g * 0.587f) + b * 0.114f;
        if (p->depth != CVGS_DEPTH_32F) lum = nearbyintf(lum) + 0.0f; /* integer result: no -0 */
        p->f[0] = lum;
        p->cn = 1;
        return 0;
    }
    }
    return CVGS_ERR_INVALID;
}

/* ------------------------------------------------------------------------------------------ */
static void store_elem(void* base, size_t elem_index, int depth, const opx* p, int c) {
    switch (depth) {
    case CVGS_DEPTH_8U: ((uint8_t*)base)[elem_index] = (uint8_t)p->f[c]; break;
    case CVGS_DEPTH_8S: ((int8_t*)base)[elem_index] = (int8_t)p->f[c]; break;
    case CVGS_DEPTH_16U: ((uint16_t*)base)[elem_index] = (uint16_t)p->f[c]; break;
    case CVGS_DEPTH_16S: ((int16_t*)base)[elem_index] = (int16_t)p->f[c]; break;
    case CVGS_DEPTH_32S: ((int32_t*)base)[elem_index] = p->i[c]; break;
    case CVGS_DEPTH_32F: ((float*)base)[elem_index] = p->f[c]; break;
    case CVGS_DEPTH_64F: ((double*)base)[elem_index] = p->d[c]; break;
    case CVGS_DEPTH_16F: ((uint16_t*)base)[elem_index] = half_bits_from_float(p->f[c]); break;
    }
}

/* The write stage.
 * TensorSplit:  out[z][c][y][x], dense NCHW (gpuMat2Tensor, reference include/cvGPUSpeedup.cuh:67-71,
 *               185-192; layout checked by tests/batchresize/test_batchresize_x_split3D.cu:337-345).
 * TensorTSplit: out[c][z][y][x] (:199-202; tests/batchread/test_circularbatchread_x_write3D.cu:324-337).
 * PerThreadWrite<_3D>: packed pixels [z][y][x] (:454-462).  PerThreadWrite<_2D>: one pitched image.
 * SplitWrite<_2D>: C pitched planes per batch element (:163-183). */
static void write_stage(const cvgs_write_desc* wr, int x, int y, int z, const opx* p) {
    const int depth = p->depth, cn = p->cn;
    const size_t W = (size_t)wr->width, H = (size_t)wr->height;
    switch (wr->kind) {
    case CVGS_WRITE_PIXEL_2D: {
        uint8_t* row = (uint8_t*)wr->data + (size_t)y * (size_t)wr->step;
        for (int c = 0; c < cn; ++c) store_elem(row, (size_t)x * cn + c, depth, p, c);
        break;
    }
    case CVGS_WRITE_PIXEL_2D_BATCH: {
        const cvgs_image2d* im = wr->planes2d + z;
        uint8_t* row = (uint8_t*)im->data + (size_t)y * (size_t)im->step;
        for (int c = 0; c < cn; ++c) store_elem(row, (size_t)x * cn + c, depth, p, c);
        break;
    }
    case CVGS_WRITE_PIXEL_3D: {
        const size_t pix = ((size_t)z * H + y) * W + x;
        for (int c = 0; c < cn; ++c) store_elem(wr->data, pix * cn + c, depth, p, c);
        break;
    }
    case CVGS_WRITE_TENSOR_SPLIT:
        for (int c = 0; c < cn; ++c)
            store_elem(wr->data, (((size_t)z * cn + c) * H + y) * W + x, depth, p, c);
        break;
    case CVGS_WRITE_TENSOR_T_SPLIT:
        for (int c = 0; c < cn; ++c)
            store_elem(wr->data, (((size_t)c * wr->planes + z) * H + y) * W + x, depth, p, c);
        break;
    case CVGS_WRITE_SPLIT_2D:
        for (int c = 0; c < cn; ++c) {
            const cvgs_image2d* im = wr->planes2d + (size_t)z * cn + c;
            uint8_t* row = (uint8_t*)im->data + (size_t)y * (size_t)im->step;
            store_elem(row, (size_t)x, depth, p, c);
        }
        break;
    }
}

/* ------------------------------------------------------------------------------------------ */
static int chain_extent(const cvgs_chain_desc* ch, int* w, int* h) {
    const cvgs_read_desc* rd = &ch->read;
    if (rd->batch < 1 || !rd->src) return CVGS_ERR_INVALID;
    if (rd->kind == CVGS_READ_RESIZE_LINEAR || rd->kind == CVGS_READ_NV12_RESIZE_LINEAR ||
        rd->kind == CVGS_READ_WARP_AFFINE || rd->kind == CVGS_READ_WARP_PERSPECTIVE) {
        const int warp = rd->kind == CVGS_READ_WARP_AFFINE || rd->kind == CVGS_READ_WARP_PERSPECTIVE;
        if (warp && !rd->warp_matrices) return CVGS_ERR_INVALID;
        *w = rd->dst_width;
        *h = rd->dst_height;
        if (warp && rd->warp_dst_sizes) { /* per-plane destination sizes (reference include/cvGPUSpeedup.cuh:381-401): the loop
                                            * below covers the largest plane, smaller ones skip the pixels outside theirs */
            *w = *h = 0;
            for (int z = 0; z < rd->batch; ++z) {
                if (rd->warp_dst_sizes[2 * z] > *w) *w = rd->warp_dst_sizes[2 * z];
                if (rd->warp_dst_sizes[2 * z + 1] > *h) *h = rd->warp_dst_sizes[2 * z + 1];
            }
        }
    } else {
        const cvgs_image2d* im = (const cvgs_image2d*)rd->src;
        *w = im->width;
        *h = im->height;
    }
    return (*w > 0 && *h > 0) ? 0 : CVGS_ERR_INVALID;
}

/* cvGS::executeOperations(stream, iops...) -> fk::executeOperations (reference
 * include/cvGPUSpeedup.cuh:464-468): thread (x,y,z) = one output element of plane z:
 * Read -> Unary/Binary... -> Write, the output of IOp k feeding IOp k+1. */
int oracle_execute(const cvgs_chain_desc* ch) {
    if (!ch || ch->struct_size != sizeof(cvgs_chain_desc)) return CVGS_ERR_INVALID;
    if (ch->n_ops < 0 || ch->n_ops > CVGS_MAX_OPS) return CVGS_ERR_INVALID;
    if (ch->read.flags & CVGS_READ_FLAG_TABLE_ON_DEVICE) return CVGS_ERR_UNSUPPORTED;
    if (ch->write.n_mirrors < 0 || ch->write.n_mirrors > CVGS_MAX_MIRRORS || (ch->write.n_mirrors && !ch->write.mirrors)) return CVGS_ERR_INVALID;
    if (ch->write.n_mirrors && ch->write.kind != CVGS_WRITE_PIXEL_3D && ch->write.kind != CVGS_WRITE_TENSOR_SPLIT &&
        ch->write.kind != CVGS_WRITE_TENSOR_T_SPLIT) return CVGS_ERR_INVALID;
    int W, H;
    int rc = chain_extent(ch, &W, &H);
    if (rc) return rc;
    const cvgs_read_desc* rd = &ch->read;
    const int N = rd->batch;
    oracle_resize_geom* geoms = (oracle_resize_geom*)calloc((size_t)N, sizeof(*geoms));
    if (!geoms) return CVGS_ERR_INVALID;
    if (rd->kind == CVGS_READ_RESIZE_LINEAR || rd->kind == CVGS_READ_NV12_RESIZE_LINEAR) {
        const int used = rd->used_planes < N ? rd->used_planes : N;
        for (int z = 0; z < used; ++z) {
            const cvgs_image2d* im = (const cvgs_image2d*)rd->src + z;
            oracle_resize_geometry(im->width, im->height, rd->dst_width, rd->dst_height,
                                   rd->aspect_ratio, geoms + z);
        }
    }
    int err = 0;
    const long total_rows = (long)N * H;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
#endif
    for (long zr = 0; zr < total_rows; ++zr) {
        const int z = (int)(zr / H), y = (int)(zr % H);
        const int warp_sized = (rd->kind == CVGS_READ_WARP_AFFINE || rd->kind == CVGS_READ_WARP_PERSPECTIVE) && rd->warp_dst_sizes;
        if (warp_sized && y >= rd->warp_dst_sizes[2 * z + 1]) continue;
        for (int x = 0; x < W; ++x) {
            if (warp_sized && x >= rd->warp_dst_sizes[2 * z]) break;
            opx p;
            memset(&p, 0, sizeof(p));
            read_stage(rd, geoms, x, y, z, &p);
            int e = 0;
            for (int k = 0; k < ch->n_ops && !e; ++k) e = apply_op(&ch->ops[k], &p);
            if (e) {
                err = e;
                continue;
            }
            if (p.depth != CVGS_TYPE_DEPTH(ch->write.dst_type) || p.cn != CVGS_TYPE_CN(ch->write.dst_type)) {
                err = CVGS_ERR_INVALID;
                continue;
            }
            write_stage(&ch->write, x, y, z, &p);
            /* cvgs_write_desc.mirrors: the same value at the same offsets of every further tensor (the peers' copies
             * of a sharded tensor; include/cvgs_hip.h) */
            for (int m = 0; m < ch->write.n_mirrors; ++m) {
                cvgs_write_desc w = ch->write;
                w.data = ch->write.mirrors[m];
                write_stage(&w, x, y, z, &p);
            }
        }
    }
    free(geoms);
    return err;
}

/* ------------------------------------------------------------------------------------------ */
/* The headline chain as a plain CPU loop nest -- what a hand-written CPU implementation of K1 looks like, for the
 * timed cpu_baseline of bench.py (the interpreter above pays a descriptor walk per pixel, which is not a fair CPU
 * baseline).  Same arithmetic, same order, same results as oracle_execute (tests/test_oracle_independent.py checks the
 * two agree bit for bit); only:  u8 C3 sources, stretch geometry, every plane used,
 * [REORDER(swap R,B)] MUL SUB DIV, fp32 TensorSplit.  Anything else: CVGS_ERR_UNSUPPORTED. */
int oracle_k1_fast(const cvgs_chain_desc* ch) { return oracle_k1_fast_repeat(ch, 1); }

/* `reps` passes over the same batch in ONE parallel region (threads split reps x crops x rows), so that the timed
 * baseline measures the host's throughput and not the start-up of a thread team per 50-crop batch; every pass writes
 * the same values to the same output. */
int oracle_k1_fast_repeat(const cvgs_chain_desc* ch, int reps) {
    if (!ch || ch->struct_size != sizeof(cvgs_chain_desc) || reps < 1) return CVGS_ERR_INVALID;
    const cvgs_read_desc* rd = &ch->read;
    const cvgs_write_desc* wr = &ch->write;
    if (rd->kind != CVGS_READ_RESIZE_LINEAR || rd->src_type != CVGS_MAKETYPE(CVGS_DEPTH_8U, 3) || rd->aspect_ratio != CVGS_IGNORE_AR ||
        rd->used_planes != rd->batch || (rd->flags & CVGS_READ_FLAG_TABLE_ON_DEVICE))
        return CVGS_ERR_UNSUPPORTED;
    if (wr->kind != CVGS_WRITE_TENSOR_SPLIT || wr->dst_type != CVGS_MAKETYPE(CVGS_DEPTH_32F, 3)) return CVGS_ERR_UNSUPPORTED;
    int k = 0, swap = 0;
    if (ch->n_ops == 4 && ch->ops[0].opcode == CVGS_OP_REORDER && ch->ops[0].aux == (2 | (1 << 2) | (0 << 4))) { swap = 1; k = 1; }
    if (ch->n_ops != k + 3 || ch->ops[k].opcode != CVGS_OP_MUL || ch->ops[k + 1].opcode != CVGS_OP_SUB || ch->ops[k + 2].opcode != CVGS_OP_DIV)
        return CVGS_ERR_UNSUPPORTED;
    const float* mul = ch->ops[k].operand; const float* sub = ch->ops[k + 1].operand; const float* dv = ch->ops[k + 2].operand;
    const int W = rd->dst_width, H = rd->dst_height, N = rd->batch;
    const size_t plane = (size_t)W * H;
    float* out = (float*)wr->data;
    /* per-crop column tables: x1 offset, clamped x2 offset, the two x weights */
    int* xa = (int*)malloc(sizeof(int) * 2 * (size_t)N * W);
    float* wx = (float*)malloc(sizeof(float) * 2 * (size_t)N * W);
    if (!xa || !wx) { free(xa); free(wx); return CVGS_ERR_INVALID; }
    for (int z = 0; z < N; ++z) {
        const cvgs_image2d* im = (const cvgs_image2d*)rd->src + z;
        oracle_resize_geom g;
        oracle_resize_geometry(im->width, im->height, W, H, CVGS_IGNORE_AR, &g);
        for (int x = 0; x < W; ++x) {
            const float sx = (float)x * g.fx;
            const int x1 = (int)floorf(sx), x2 = x1 + 1;
            xa[((size_t)z * W + x) * 2] = 3 * x1;
            xa[((size_t)z * W + x) * 2 + 1] = 3 * (x2 < im->width - 1 ? x2 : im->width - 1);
            wx[((size_t)z * W + x) * 2] = (float)x2 - sx;
            wx[((size_t)z * W + x) * 2 + 1] = sx - (float)x1;
        }
    }
    const long rows = (long)N * H * reps;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
#endif
    for (long rzr = 0; rzr < rows; ++rzr) {
        const long zr = rzr % ((long)N * H);
        const int z = (int)(zr / H), y = (int)(zr % H);
        const cvgs_image2d* im = (const cvgs_image2d*)rd->src + z;
        oracle_resize_geom g;
        oracle_resize_geometry(im->width, im->height, W, H, CVGS_IGNORE_AR, &g);
        const float sy = (float)y * g.fy;
        const int y1 = (int)floorf(sy), y2 = y1 + 1;
        const int y2r = y2 < im->height - 1 ? y2 : im->height - 1;
        const float wya = (float)y2 - sy, wyb = sy - (float)y1;
        const uint8_t* ra = (const uint8_t*)im->data + (size_t)y1 * im->step;
        const uint8_t* rb = (const uint8_t*)im->data + (size_t)y2r * im->step;
        const int* xz = xa + (size_t)z * W * 2;
        const float* wz = wx + (size_t)z * W * 2;
        float* o = out + (size_t)z * 3 * plane + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            const int o1 = xz[2 * x], o2 = xz[2 * x + 1];
            const float w00 = wz[2 * x] * wya, w10 = wz[2 * x + 1] * wya, w01 = wz[2 * x] * wyb, w11 = wz[2 * x + 1] * wyb;
            float v[3];
            for (int c = 0; c < 3; ++c) {
                float acc = (float)ra[o1 + c] * w00;
                acc = acc + (float)ra[o2 + c] * w10;
                acc = acc + (float)rb[o1 + c] * w01;
                acc = acc + (float)rb[o2 + c] * w11;
                v[c] = acc;
            }
            if (swap) { const float t = v[0]; v[0] = v[2]; v[2] = t; }
            for (int c = 0; c < 3; ++c) o[(size_t)c * plane + x] = ((v[c] * mul[c]) - sub[c]) / dv[c];
        }
    }
    free(xa);
    free(wx);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* distinct tapped source pixels of one K1 plane = ux * uy (taps are separable). */
static int64_t distinct_taps(int src, int n_out, float f) {
    int64_t count = 0;
    int last = -1;
    for (int d = 0; d < n_out; ++d) {
        const float s = (float)d * f;
        const int a = (int)floorf(s);
        const int b = a + 1 < src - 1 ? a + 1 : src - 1;
        if (a > last) { ++count; last = a; }
        if (b > last) { ++count; last = b; }
    }
    return count;
}

int64_t oracle_resize_tapped_bytes(int32_t src_w, int32_t src_h, int32_t dst_w, int32_t dst_h,
                                   int32_t aspect_ratio, int32_t bytes_per_pixel) {
    oracle_resize_geom g;
    oracle_resize_geometry(src_w, src_h, dst_w, dst_h, aspect_ratio, &g);
    const int64_t ux = distinct_taps(src_w, g.x2 - g.x1 + 1, g.fx);
    const int64_t uy = distinct_taps(src_h, g.y2 - g.y1 + 1, g.fy);
    return ux * uy * bytes_per_pixel;
}

/* ------------------------------------------------------------------------------------------ */
/* cvGS::CircularTensor (reference include/cvGPUSpeedup.cuh:600-627) -> fk::CircularTensor [FKL].
 * Observable contract, pinned by tests/batchread/test_circularbatchread_x_write3D.cu:263-279
 * (NewestFirst: after 100 updates with value i+1, slot z == 100 - z) and :382-395 (OldestFirst:
 * slot z == 100 - (BATCH-1-z)): every update the new frame = ops(input) becomes slot 0
 * (NewestFirst) or BATCH-1 (OldestFirst), every older frame moves one slot, the whole tensor at
 * data() is rewritten from an INTERNAL history (writes the caller makes into data() never
 * propagate).  History slots never written read as 0 (the reference leaves them uninitialised). */
struct oracle_circular_s {
    int32_t width, height, elem_type, color_planes, batch, order, cp_mode;
    size_t image_bytes; /* one image: color_planes * H * W elements */
    uint8_t* out;       /* the tensor at data()                     */
    uint8_t* hist;      /* batch images, update k at slot k % batch */
    uint8_t* tmp;       /* one image in standard [c][y][x] order    */
    int64_t count;
};

int oracle_circular_create(oracle_circular_t* out, int32_t width, int32_t height, int32_t elem_type,
                           int32_t color_planes, int32_t batch, int32_t order, int32_t cp_mode) {
    if (!out || width < 1 || height < 1 || color_planes < 1 || batch < 1) return CVGS_ERR_INVALID;
    const int esz = depth_bytes(CVGS_TYPE_DEPTH(elem_type)) * CVGS_TYPE_CN(elem_type);
    if (!esz) return CVGS_ERR_INVALID;
    struct oracle_circular_s* ct = (struct oracle_circular_s*)calloc(1, sizeof(*ct));
    ct->width = width; ct->height = height; ct->elem_type = elem_type;
    ct->color_planes = color_planes; ct->batch = batch; ct->order = order; ct->cp_mode = cp_mode;
    ct->image_bytes = (size_t)esz * width * height * color_planes;
    ct->out = (uint8_t*)calloc((size_t)batch, ct->image_bytes);
    ct->hist = (uint8_t*)calloc((size_t)batch, ct->image_bytes);
    ct->tmp = (uint8_t*)calloc(1, ct->image_bytes);
    *out = ct;
    return 0;
}

int oracle_circular_update(oracle_circular_t ct, const cvgs_chain_desc* chain) {
    if (!ct || !chain) return CVGS_ERR_INVALID;
    /* 1. new frame = chain applied to the input, as a single standard-order image */
    cvgs_chain_desc one = *chain;
    one.read.batch = 1;
    one.read.used_planes = 1;
    one.write.data = ct->tmp;
    one.write.width = ct->width;
    one.write.height = ct->height;
    one.write.planes = 1;
    if (one.write.kind == CVGS_WRITE_TENSOR_T_SPLIT) one.write.kind = CVGS_WRITE_TENSOR_SPLIT;
    int rc = oracle_execute(&one);
    if (rc) return rc;
    memcpy(ct->hist + (size_t)(ct->count % ct->batch) * ct->image_bytes, ct->tmp, ct->image_bytes);
    /* 2. rebuild the ordered tensor from the history */
    const size_t plane_bytes = ct->image_bytes / (size_t)ct->color_planes;
    for (int z = 0; z < ct->batch; ++z) {
        const int64_t age = ct->order == CVGS_NEWEST_FIRST ? z : ct->batch - 1 - z;
        const int64_t src_update = ct->count - age;
        for (int c = 0; c < ct->color_planes; ++c) {
            uint8_t* dst = ct->cp_mode == CVGS_PLANES_TRANSPOSED
                               ? ct->out + ((size_t)c * ct->batch + z) * plane_bytes
                               : ct->out + ((size_t)z * ct->color_planes + c) * plane_bytes;
            if (src_update < 0) memset(dst, 0, plane_bytes);
            else memcpy(dst, ct->hist + (size_t)(src_update % ct->batch) * ct->image_bytes + (size_t)c * plane_bytes,
                        plane_bytes);
        }
    }
    ct->count++;
    return 0;
}

void* oracle_circular_data(oracle_circular_t ct) { return ct ? ct->out : NULL; }
size_t oracle_circular_bytes(oracle_circular_t ct) { return ct ? ct->image_bytes * (size_t)ct->batch : 0; }

int oracle_circular_destroy(oracle_circular_t ct) {
    if (!ct) return CVGS_ERR_INVALID;
    free(ct->out); free(ct->hist); free(ct->tmp); free(ct);
    return 0;
}


/* ------------------------------------------------------------------------------------------ */
/* Checker for div_by_uniform (csrc/k_taps.hpp); see cvgs_oracle.h.  Test infrastructure only. */
typedef float (*fma_fn)(float, float, float);
static float fma_soft(float a, float b, float c) { return fmaf(a, b, c); }
#if defined(__x86_64__)
__attribute__((target("fma"))) static float fma_hw(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#endif
static inline uint64_t fd_rnd(uint64_t* s) {
    *s += 0x9E3779B97F4A7C15ull;
    uint64_t z = *s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline float fd_make(uint32_t sig, int e, uint32_t sign) {
    const uint32_t b = (sign << 31) | ((uint32_t)(e + 127) << 23) | (sig & 0x7fffffu);
    float f;
    memcpy(&f, &b, 4);
    return f;
}
int64_t oracle_fastdiv_mismatches(uint32_t sig_begin, uint32_t sig_end, int32_t per_divisor, uint64_t seed, int32_t steps) {
    fma_fn f = fma_soft;
#if defined(__x86_64__)
    if (__builtin_cpu_supports("fma")) f = fma_hw;
#endif
    int64_t bad = 0;
    uint64_t s = seed;
    for (uint32_t sig = sig_begin; sig < sig_end && sig < (1u << 23); ++sig) {
        if (sig == 0x7fffffu) continue;
        for (int k = 0; k < per_divisor; ++k) {
            const uint64_t a = fd_rnd(&s);
            const float d = fd_make(sig, (int)((a >> 12) % 41) - 20, (uint32_t)(a >> 11) & 1u);
            volatile float rv = 1.0f / d;
            const float r = rv;
            float x;
            switch (k & 3) {
            case 0: x = fd_make((uint32_t)(a >> 30) & 7u, (int)((a >> 40) % 129) - 90, (uint32_t)(a >> 63)); break;            /* just above 2^e */
            case 1: x = fd_make(0x7fffffu - ((uint32_t)(a >> 30) & 7u), (int)((a >> 40) % 129) - 90, (uint32_t)(a >> 63)); break; /* just below */
            case 2: x = fd_make((uint32_t)(a >> 30), (int)((a >> 40) % 129) - 90, (uint32_t)(a >> 63)); break;
            default: { /* near ties: x = RN(q d) moved by -8..7 ulp for a random q */
                const float q = fd_make((uint32_t)(a >> 30), (int)((a >> 54) % 41) - 20, 0);
                volatile float pv = q * d;
                float pp = pv;
                uint32_t b;
                memcpy(&b, &pp, 4);
                b += (uint32_t)((int)((a >> 5) & 15) - 8);
                memcpy(&pp, &b, 4);
                x = pp;
            }
            }
            if (x == 0.0f || !(fabsf(x) >= 0x1p-90f && fabsf(x) <= 0x1p38f)) continue;
            volatile float tv = x / d;
            const float t = tv;
            const float q0 = x * r;
            const float e0 = f(-d, q0, x);
            float q = f(e0, r, q0);
            if (steps >= 2) {
                const float e1 = f(-d, q, x);
                q = f(e1, r, q);
            }
            if (memcmp(&q, &t, 4) != 0) ++bad;
        }
    }
    return bad;
}
