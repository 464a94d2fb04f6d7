/*
 * cvgs_oracle.h -- CPU restatement of the reference's hot-path arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  The product (cvgpuspeedup_amd/) never links or calls it.
 *
 * The oracle interprets the same POD chain descriptor as the product's C-ABI
 * (include/cvgs_hip.h) but with every pointer being a HOST pointer, in plain scalar C compiled
 * with -O2 -ffp-contract=off (strict IEEE fp32, no FMA, SSE2 so no x87 excess precision).
 *
 * Provenance of the arithmetic: the reference's device code lives in the
 * un-vendored submodule FusedKernelLibrary 0.1.8 (reference .gitmodules:1-3, cmake/libs/fkl.cmake:1-8)
 * and its comparison side is OpenCV-CUDA 4.8-4.11; neither is present in /root/reference, so the
 * reference itself is UNBUILDABLE here (needs nvcc + OpenCV-CUDA + FKL) and there is no oracle/_ref.
 * Every function below restates the published algorithm and cites the reference call site it is
 * anchored on.  Pinning: tests/test_oracle_kat.py checks this oracle against every known-answer
 * vector the reference's own tests hold for the path.  The bilinear tap
 * geometry on NON-constant images, the NV12 coefficients and PRESERVE_AR_RN_EVEN/LEFT are not
 * constrained by any reference test: for those rows parity is "unpinned" (spec-based).
 */
#ifndef CVGS_ORACLE_H
#define CVGS_ORACLE_H

#include "../include/cvgs_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Resize geometry for one plane: the kernel-side scale factors and the aspect-ratio window. */
typedef struct oracle_resize_geom {
    float fx, fy;           /* source step per destination pixel                         */
    int32_t x1, y1, x2, y2; /* inclusive destination window that receives source pixels  */
} oracle_resize_geom;

void oracle_resize_geometry(int32_t src_w, int32_t src_h, int32_t dst_w, int32_t dst_h,
                            int32_t aspect_ratio, oracle_resize_geom* out);

/* Run a whole chain on host memory. All pointers in `chain` are host pointers. Returns 0 or a
 * negative cvgs_status. */
int oracle_execute(const cvgs_chain_desc* chain);

/* The headline K1 chain (u8 C3 crops -> resize -> [swap] mul sub div -> NCHW fp32) as a plain loop nest: the CPU
 * baseline bench.py times.  Bit-identical to oracle_execute on the chains it accepts; CVGS_ERR_UNSUPPORTED otherwise. */
int oracle_k1_fast(const cvgs_chain_desc* chain);
int oracle_k1_fast_repeat(const cvgs_chain_desc* chain, int reps); /* `reps` passes in one parallel region (throughput timing) */

/* Threads used by oracle_execute (OpenMP over planes x rows). 1 = scalar port. */
void oracle_set_threads(int n);
int oracle_get_threads(void);
int oracle_max_threads(void);

/* CircularTensor restated as "shift by one slot, then write the new frame" on a host tensor. */
typedef struct oracle_circular_s* oracle_circular_t;
int oracle_circular_create(oracle_circular_t* out, int32_t width, int32_t height, int32_t elem_type,
                           int32_t color_planes, int32_t batch, int32_t order, int32_t cp_mode);
int oracle_circular_update(oracle_circular_t ct, const cvgs_chain_desc* chain);
void* oracle_circular_data(oracle_circular_t ct);
size_t oracle_circular_bytes(oracle_circular_t ct);
int oracle_circular_destroy(oracle_circular_t ct);

/* Algorithmic read bytes of a K1 plane: 3 * ux * uy distinct tapped source
 * pixels times bytes per pixel. */
/* CV_16F conversions (round to nearest even; see cvgs_oracle.c) */
uint16_t oracle_float_to_half(float v);
float oracle_half_to_float(uint16_t h);

/* Checker for the product's division by a wave-uniform divisor (cvgpuspeedup_amd/csrc/k_taps.hpp, div_by_uniform:
 * q0 = x*r, two FMA correction steps, r = RN(1/d)): restates the formula with fmaf and counts the dividends for which
 * it differs from the IEEE quotient x / d.  For every divisor significand in [sig_begin, sig_end) (exponents swept over
 * the product's guarded ranges: 2^-20 <= |d| <= 2^20, 2^-90 <= |x| <= 2^38; beyond them the quotient or a residual
 * can be subnormal and the identity does fail) it tries `per_divisor` dividends: powers-of-two neighbours, random significands and
 * near-tie dividends (x = RN(q d) +- a few ulp).  all-ones significands are skipped like the product's host guard does.
 * `steps` = 2: the product's formula; 1: a single correction step (reported for the record, not used). */
int64_t oracle_fastdiv_mismatches(uint32_t sig_begin, uint32_t sig_end, int32_t per_divisor, uint64_t seed, int32_t steps);

int64_t oracle_resize_tapped_bytes(int32_t src_w, int32_t src_h, int32_t dst_w, int32_t dst_h,
                                   int32_t aspect_ratio, int32_t bytes_per_pixel);

#ifdef __cplusplus
}
#endif
#endif
