// k_common.hpp -- device-side building blocks shared by every kernel: the work pixel, the
// pointwise program (interpreted or compile-time), saturating casts, pixel loads and the write
// stages.  All arithmetic is strict IEEE fp32 in call order: the translation units are built with
// -ffp-contract=off so that no product/sum is ever fused into an FMA, and HIP's default
// correctly-rounded fp32 division is kept (the reference builds with fast-math off,
// cmake/libs/cuda/target_generation.cmake:11-12).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstring>

#include "cvgs_device.h"

namespace cvgs {

// The value flowing between stages: up to 4 channels of 32 bits.  8/16-bit integers and fp32 live
// as floats (exact); 32S values live as raw bits (use as_int / from_int).
struct Px {
    float v[4];
};

__device__ __forceinline__ int as_int(float f) { return __float_as_int(f); }
__device__ __forceinline__ float from_int(int i) { return __int_as_float(i); }

// ---- fk::SaturateCast -----------------------------------------------------------------------
// float -> integer depth: round to nearest even, clamp, NaN -> 0 (cv::saturate_cast on the GPU).
__device__ __forceinline__ float sat_round(float v, float lo, float hi) {
    // v_rndne_f32; integer-typed values live in float registers, and an integer has no -0: rint(-0.3) = -0.0 must become
    // +0 (it would otherwise survive a later cast back to float; found by tests/test_gpu_fuzz.py)
    float r = rintf(v) + 0.0f;
    r = (v != v) ? 0.f : r;
    return fminf(fmaxf(r, lo), hi);
}

// The SaturateCast in front of a STORE (the value leaves as bits, it does not continue as a float): single instructions whose
// hardware semantics ARE round to nearest even + clamp + NaN -> 0.  tools/sat_probe.cpp compares each against sat_round over
// all 2^32 float bit patterns on the GPU (no mismatch; profiles/r02_n_sat_probe.txt).
//   u8 : v_cvt_pk_u8_f32 converts AND inserts the byte into `old` (1 instruction instead of 7 + shift/or)
//   u16: v_rndne_f32, v_cvt_u32_f32 (saturating, NaN -> 0), v_min_u32      s16: v_rndne_f32, v_cvt_i32_f32, v_med3_i32
__device__ __forceinline__ uint32_t sat_u8_insert(float v, uint32_t byte, uint32_t old) { return __builtin_amdgcn_cvt_pk_u8_f32(v, byte, old); }
__device__ __forceinline__ uint32_t sat_u16_bits(float v) {
    uint32_t u;
    const float r = rintf(v);
    asm("v_cvt_u32_f32 %0, %1" : "=v"(u) : "v"(r));
    return u < 65535u ? u : 65535u;
}
__device__ __forceinline__ uint32_t sat_s16_bits(float v) {
    int32_t i;
    const float r = rintf(v);
    asm("v_cvt_i32_f32 %0, %1" : "=v"(i) : "v"(r));
    i = i < -32768 ? -32768 : (i > 32767 ? 32767 : i);
    return (uint32_t)i & 0xffffu;
}

__device__ __forceinline__ int sat_round_s32(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return -2147483647 - 1;
    return (int)rintf(v);
}

__device__ __forceinline__ void depth_range(int depth, float& lo, float& hi) {
    switch (depth) {
    case CVGS_DEPTH_8U: lo = 0.f; hi = 255.f; break;
    case CVGS_DEPTH_8S: lo = -128.f; hi = 127.f; break;
    case CVGS_DEPTH_16U: lo = 0.f; hi = 65535.f; break;
    default: lo = -32768.f; hi = 32767.f; break; // 16S
    }
}

// fp32 -> binary16 -> fp32: round to nearest even, overflow to +-inf (v_cvt_f16_f32); a CV_16F value lives in the
// work pixel as the float it converts to exactly
__device__ __forceinline__ float round_half(float v) { return (float)(_Float16)v; }

template <bool TRUNC = false>
__device__ __forceinline__ void cast_px(Px& p, int cn, int src_depth, int dst_depth) {
    if (src_depth == dst_depth) return;
    if (src_depth == CVGS_DEPTH_16F) src_depth = CVGS_DEPTH_32F; // already an exact float
    if (src_depth == dst_depth) return;
    if (dst_depth == CVGS_DEPTH_16F) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < cn) p.v[c] = round_half(src_depth == CVGS_DEPTH_32S ? (float)as_int(p.v[c]) : p.v[c]);
        return;
    }
    if (dst_depth == CVGS_DEPTH_32F) {
        if (src_depth == CVGS_DEPTH_32S) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < cn) p.v[c] = (float)as_int(p.v[c]);
        }
        return; // 8/16-bit integers are already exact floats
    }
    if (dst_depth == CVGS_DEPTH_32S) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < cn) p.v[c] = from_int(src_depth == CVGS_DEPTH_32F ? sat_round_s32(TRUNC ? truncf(p.v[c]) : p.v[c]) : (int)p.v[c]);
        return;
    }
    float lo, hi;
    depth_range(dst_depth, lo, hi);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c < cn) {
            float v = p.v[c];
            if (src_depth == CVGS_DEPTH_32S) {
                int iv = as_int(v);
                iv = iv < (int)lo ? (int)lo : (iv > (int)hi ? (int)hi : iv);
                p.v[c] = (float)iv;
            } else if (src_depth == CVGS_DEPTH_32F) {
                p.v[c] = sat_round(TRUNC ? truncf(v) : v, lo, hi); // fk::Cast: toward zero, then the same clamp
            } else {
                p.v[c] = fminf(fmaxf(v, lo), hi);
            }
        }
    }
}

__device__ __forceinline__ void reorder_px(Px& p, int aux, int out_cn) {
    const Px s = p;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c < out_cn) {
            const int k = (aux >> (2 * c)) & 3;
            p.v[c] = k == 0 ? s.v[0] : (k == 1 ? s.v[1] : (k == 2 ? s.v[2] : s.v[3]));
        }
    }
}

// Arithmetic on INTEGER-typed values (cvGS::multiply / add / subtract / divide<I> with an integer pixel type I, reference
// include/cvGPUSpeedup.cuh:131-149).  FKL, which defines fk::Mul<uchar3> etc., is not in the reference tree and no reference
// test uses these instantiations; the semantics chosen here: the scalar was converted to the pixel's own type by
// the facade (cvScalar2CUDAV<I>: truncation) -- the host hands it over as an exact integer --, the operation runs in 64-bit
// integer arithmetic (C++ promotion, no intermediate wrap), division truncates toward zero and x / 0 = 0 (cv::divide's
// convention), and the result is SATURATED back to the type (cv::saturate_cast), so the value stays a valid value of its type.
__device__ __forceinline__ void int_range(int depth, long long& lo, long long& hi) {
    switch (depth) {
    case CVGS_DEPTH_8U: lo = 0; hi = 255; break;
    case CVGS_DEPTH_8S: lo = -128; hi = 127; break;
    case CVGS_DEPTH_16U: lo = 0; hi = 65535; break;
    case CVGS_DEPTH_16S: lo = -32768; hi = 32767; break;
    default: lo = -2147483648ll; hi = 2147483647ll; break;
    }
}
__device__ __forceinline__ long long int_arith(int opc, long long a, long long b, long long lo, long long hi) {
    long long r;
    switch (opc) {
    case CVGS_OP_MUL: r = a * b; break;
    case CVGS_OP_ADD: r = a + b; break;
    case CVGS_OP_SUB: r = a - b; break;
    default: r = b == 0 ? 0 : a / b; break;
    }
    return r < lo ? lo : (r > hi ? hi : r);
}
__device__ __forceinline__ bool is_int_depth(int depth) { return depth <= CVGS_DEPTH_32S; }

// One pointwise stage.  `opc`, `aux` and the operands are wave-uniform (kernel arguments), so the
// switch is a scalar branch, never a divergent one.  INTA: the value may be integer-typed when an arithmetic stage meets it
// (the interpreted kernels only: every fast kernel's host-side plan keeps such chains out).
template <bool INTA = false>
__device__ __forceinline__ void apply_op(int opc, int aux, const float* operand, Px& p, int& depth, int& cn) {
    if constexpr (INTA) {
        if ((opc == CVGS_OP_MUL || opc == CVGS_OP_ADD || opc == CVGS_OP_SUB || opc == CVGS_OP_DIV) && is_int_depth(depth)) {
            long long lo, hi;
            int_range(depth, lo, hi);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < cn) {
                    const long long a = depth == CVGS_DEPTH_32S ? (long long)as_int(p.v[c]) : (long long)p.v[c];
                    const long long b = depth == CVGS_DEPTH_32S ? (long long)as_int(operand[c]) : (long long)operand[c];
                    const long long r = int_arith(opc, a, b, lo, hi);
                    p.v[c] = depth == CVGS_DEPTH_32S ? from_int((int)r) : (float)r;
                }
            return;
        }
    }
    switch (opc) {
    case CVGS_OP_CAST:
        cast_px(p, cn, depth, aux);
        depth = aux;
        break;
    case CVGS_OP_CAST_TRUNC:
        cast_px<true>(p, cn, depth, aux);
        depth = aux;
        break;
    case CVGS_OP_MUL:
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) p.v[c] = p.v[c] * operand[c];
        break;
    case CVGS_OP_ADD:
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) p.v[c] = p.v[c] + operand[c];
        break;
    case CVGS_OP_SUB:
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) p.v[c] = p.v[c] - operand[c];
        break;
    case CVGS_OP_DIV:
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) p.v[c] = p.v[c] / operand[c];
        break;
    case CVGS_OP_REORDER:
        reorder_px(p, aux, cn);
        break;
    case CVGS_OP_ADD_ALPHA:
        reorder_px(p, aux, 3);
        p.v[3] = depth == CVGS_DEPTH_32S ? from_int((int)operand[0]) : operand[0];
        cn = 4;
        break;
    case CVGS_OP_DROP_ALPHA:
        reorder_px(p, aux, 3);
        cn = 3;
        break;
    case CVGS_OP_GRAY: {
        const int kr = aux & 3, kg = (aux >> 2) & 3, kb = (aux >> 4) & 3;
        const float r = kr == 0 ? p.v[0] : (kr == 1 ? p.v[1] : (kr == 2 ? p.v[2] : p.v[3]));
        const float g = kg == 0 ? p.v[0] : (kg == 1 ? p.v[1] : (kg == 2 ? p.v[2] : p.v[3]));
        const float b = kb == 0 ? p.v[0] : (kb == 1 ? p.v[1] : (kb == 2 ? p.v[2] : p.v[3]));
        float lum = (r * 0.299f + g * 0.587f) + b * 0.114f;
        if (depth != CVGS_DEPTH_32F) lum = rintf(lum) + 0.0f; // integer result: no -0
        p.v[0] = lum;
        cn = 1;
        break;
    }
    default: break;
    }
}

// Interpreted program, integer-typed arithmetic included (k_generic / k_warp's interpreted kernels)
struct InterpProgInt {
    static __device__ __forceinline__ void run(const ProgArgs& prog, Px& p, int& depth, int& cn) {
        for (int k = 0; k < prog.n; ++k) apply_op<true>(prog.opcode[k], prog.aux[k], prog.operand[k], p, depth, cn);
    }
};

// Division by a WAVE-UNIFORM divisor d with r = RN(1/d) from the host: q0 = x*r and two FMA correction steps give RN(x/d) bit for bit
// (Markstein) when d's significand is not all ones and no intermediate leaves the normal range -- guaranteed by bounds on the
// operands (launch_k1: k1_fast_div_ok) and by construction for x: everything finite and far from the exponent range's
// ends; x == 0 is excluded by the caller (the sign of a zero quotient needs the real
// division).  tests/test_fast_division.py checks the identity against IEEE division for EVERY divisor significand.
__device__ __forceinline__ float div_by_uniform(float x, float d, float r) {
    const float q0 = x * r;
    const float e0 = __builtin_fmaf(-d, q0, x);
    const float q1 = __builtin_fmaf(e0, r, q0);
    const float e1 = __builtin_fmaf(-d, q1, x);
    return __builtin_fmaf(e1, r, q1);
}

// The DIV stage of a pointwise program on a thread's four pixels, RUN-TIME GUARDED (round 6): prog.fast_div == 2 + k says the host found stage
// k's divisors fit (k_taps.hpp: guarded_div_setup) and left their reciprocals in prog.rdiv.  The dividends are checked here: the largest and
// the smallest magnitude of the thread's values as integers (NaN and infinity compare above every finite float, zero below every normal one),
// one ballot over the wave -- inside [2^-90, 2^38] the five-instruction form is the IEEE quotient, otherwise the wave divides for real.
// An IEEE fp32 division costs ~10 instructions + a quarter-rate reciprocal per value; the reference's test_read_x_write chain on a 4K u8c3 frame
// spent 8 of its 23 us dividing.
__device__ __forceinline__ bool div4_guarded(bool stage_fits, const float (&d)[4], const float (&r)[4], Px (&px)[4], int cn) {
    if (!stage_fits) return false; // wave-uniform
    uint32_t mx = 0u, mn = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < cn) {
                const uint32_t u = __float_as_uint(px[i].v[c]) & 0x7fffffffu;
                mx = u > mx ? u : mx;
                mn = u < mn ? u : mn;
            }
    const bool outside = mn < 0x12800000u || mx > 0x52800000u; // 2^-90, 2^38
    if (__builtin_amdgcn_ballot_w64(outside) != 0) return false;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < cn) px[i].v[c] = div_by_uniform(px[i].v[c], d[c], r[c]);
    return true;
}
__device__ __forceinline__ bool div4_guarded(const ProgArgs& prog, int k, Px (&px)[4], int cn) {
    const float d[4] = {prog.operand[k][0], prog.operand[k][1], prog.operand[k][2], prog.operand[k][3]};
    const float r[4] = {prog.rdiv[0], prog.rdiv[1], prog.rdiv[2], prog.rdiv[3]};
    return div4_guarded(prog.fast_div == 2 + k, d, r, px, cn);
}

// The DIV stage on ONE pixel (the one-pixel-per-lane kernels: K1 / K4 / warp with an interpreted program), guarded like div4_guarded.
__device__ __forceinline__ bool div1_guarded(bool stage_fits, const float (&d)[4], const float (&r)[4], Px& p, int cn) {
    if (!stage_fits) return false; // wave-uniform
    uint32_t mx = 0u, mn = 0xffffffffu;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < cn) {
            const uint32_t u = __float_as_uint(p.v[c]) & 0x7fffffffu;
            mx = u > mx ? u : mx;
            mn = u < mn ? u : mn;
        }
    const bool outside = mn < 0x12800000u || mx > 0x52800000u; // 2^-90, 2^38
    if (__builtin_amdgcn_ballot_w64(outside) != 0) return false;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < cn) p.v[c] = div_by_uniform(p.v[c], d[c], r[c]);
    return true;
}

// Interpreted program: any valid op list whose arithmetic meets float values.
// Round 6: programs made of MUL / ADD / SUB / DIV / REORDER stages on CV_32F values only (the host says so: prog.fast_div & 0x100,
// interp_arith_setup) -- every resize chain that normalises in another order or with one more stage than the compile-time programs -- take
// run_arith: the opcodes, aux words and operands of the first kUnrolled stages are fetched BEFORE the first stage (scalar loads at fixed
// kernel-argument offsets, one wait; the rolled loop pays a scalar-memory round trip per stage and wave: + ~5 us per stage on a K1 tick of 16 x 50
// crops, 60 us against the compile-time program's 39 for one more `add`), the stages are a few instructions each, and the DIV stage the host vetted
// divides by reciprocal when every dividend of the wave fits (div1_guarded).  Everything else keeps the rolled loop over the full stage switch
// (inlining that switch six times into ~300 kernels would add 18 MB of code).  ARITH = false: the rolled loop only.
template <bool ARITH>
struct InterpProgT {
    static constexpr int kUnrolled = 6;
    // MUL, ADD and SUB are ONE fused multiply-add with scalar-selected operands -- p * o = fma(p, o, -0), p + o = fma(p, 1, o), p - o = fma(p, 1, -o),
    // each the same single rounding as the plain operation (adding -0 changes no value and no zero's sign) -- so the common stages cost one compare
    // and one branch instead of a compare-and-branch chain per opcode (a wave pays ~20 cycles per taken branch; five stages of four tests each
    // were most of the 18 us a K1 tick lost to its interpreted program).
    static __device__ __forceinline__ void arith_stage(int op, int aux, const float (&o)[4], bool div_fits, const float (&r)[4], Px& p, int cn) {
        if (op == CVGS_OP_MUL || op == CVGS_OP_ADD || op == CVGS_OP_SUB) { // wave-uniform
            const bool mul = op == CVGS_OP_MUL, sub = op == CVGS_OP_SUB;
#pragma unroll
            for (int c = 0; c < 4; ++c) { // (channels at and beyond cn hold nothing anybody stores)
                const float m = mul ? o[c] : 1.0f;
                const float a = mul ? -0.0f : (sub ? -o[c] : o[c]);
                p.v[c] = __builtin_fmaf(p.v[c], m, a);
            }
        } else if (op == CVGS_OP_DIV) {
            if (div1_guarded(div_fits, o, r, p, cn)) return;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < cn) p.v[c] = p.v[c] / o[c];
        } else { // CVGS_OP_REORDER
            reorder_px(p, aux, cn);
        }
    }
    // The arithmetic path's scalar state: requested by the kernel BEFORE its tap loads (prefetch: plain loads), pinned in registers AFTER they are
    // issued (settle) -- the program's words arrive in the shadow of the taps instead of after them.
    struct State {
        int op[ARITH ? kUnrolled : 1], aux[ARITH ? kUnrolled : 1];
        float o[ARITH ? kUnrolled : 1][4];
        float r[4];
        int n, div_at;
        bool arith;
    };
    static __device__ __forceinline__ State prefetch(const ProgArgs& prog) {
        State st;
        st.arith = false;
        if constexpr (ARITH) {
            st.arith = (prog.fast_div & 0x100) != 0;
            st.n = prog.n;
            st.div_at = (prog.fast_div & 0xff) - 2;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) st.r[ch] = prog.rdiv[ch];
#pragma unroll
            for (int k = 0; k < kUnrolled; ++k) {
                st.op[k] = prog.opcode[k];
                st.aux[k] = prog.aux[k];
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) st.o[k][ch] = prog.operand[k][ch];
            }
        }
        return st;
    }
    static __device__ __forceinline__ void settle(State& st) {
        if constexpr (ARITH) {
            // (pinned in scalar registers HERE: left alone, the compiler sinks each load into the stage that uses it)
#pragma unroll
            for (int k = 0; k < kUnrolled; ++k) {
                asm volatile("" : "+s"(st.op[k]), "+s"(st.aux[k]));
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) asm volatile("" : "+s"(st.o[k][ch]));
            }
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) asm volatile("" : "+s"(st.r[ch]));
        }
    }
    static __device__ __forceinline__ void run_arith(const ProgArgs& prog, const State& st, Px& p, int cn) {
        if constexpr (ARITH) {
#pragma unroll
            for (int k = 0; k < kUnrolled; ++k)
                if (k < st.n) arith_stage(st.op[k], st.aux[k], st.o[k], st.div_at == k, st.r, p, cn);
            for (int k = kUnrolled; k < st.n; ++k) {
                const float ok[4] = {prog.operand[k][0], prog.operand[k][1], prog.operand[k][2], prog.operand[k][3]};
                arith_stage(prog.opcode[k], prog.aux[k], ok, st.div_at == k, st.r, p, cn);
            }
        }
    }
    static __device__ __forceinline__ void run(const ProgArgs& prog, const State& st, Px& p, int& depth, int& cn) {
        if constexpr (ARITH) {
            if (st.arith && depth == CVGS_DEPTH_32F) { // wave-uniform
                run_arith(prog, st, p, cn);
                return;
            }
        }
        for (int k = 0; k < prog.n; ++k) apply_op(prog.opcode[k], prog.aux[k], prog.operand[k], p, depth, cn);
    }
    static __device__ __forceinline__ void run(const ProgArgs& prog, Px& p, int& depth, int& cn) {
        for (int k = 0; k < prog.n; ++k) apply_op(prog.opcode[k], prog.aux[k], prog.operand[k], p, depth, cn);
    }
    // four pixels at once: the opcode loop stays outermost so the pixel array is only indexed by constants
    static __device__ __forceinline__ void run4(const ProgArgs& prog, Px (&px)[4], int& depth, int& cn) {
        for (int k = 0; k < prog.n; ++k) {
            int d = depth, c = cn;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d = depth;
                c = cn;
                apply_op(prog.opcode[k], prog.aux[k], prog.operand[k], px[i], d, c);
            }
            depth = d;
            cn = c;
        }
    }
};
using InterpProg = InterpProgT<false>;      // the rolled loop only (K4, warps, the CircularTensor push, K1's packed / 4-rows-per-wave modes)
using InterpProgArith = InterpProgT<true>;  // + run_arith (K1's planar-tensor kernels: ~36 more scalar registers, which the others do not have)

// HOST side of the RUN-TIME GUARDED division of the pointwise programs (k_common.hpp: div4_guarded): the first DIV stage whose divisors qualify
// (finite, 2^-20 <= |d| <= 2^20, significand not all ones) gets its correctly rounded reciprocals and p.fast_div = 2 + its index.  Nothing is
// assumed about the dividends: the kernel checks them (every value of the wave inside [2^-90, 2^38], which also says "finite" and "not zero")
// and takes the real division otherwise -- the range tests/test_fast_division.py sweeps.
inline void guarded_div_setup(ProgArgs& p, int cn);
// HOST side of InterpProg::run_arith: an arithmetic-only program (MUL / ADD / SUB / DIV / REORDER: the value stays CV_32F with cn channels)
inline void interp_arith_setup(ProgArgs& p, int cn) {
    guarded_div_setup(p, cn);
    for (int k = 0; k < p.n; ++k)
        if (p.opcode[k] != CVGS_OP_MUL && p.opcode[k] != CVGS_OP_ADD && p.opcode[k] != CVGS_OP_SUB && p.opcode[k] != CVGS_OP_DIV && p.opcode[k] != CVGS_OP_REORDER) return;
    p.fast_div |= 0x100;
}
inline void guarded_div_setup(ProgArgs& p, int cn) {
    p.fast_div = 0;
    for (int k = 0; k < p.n; ++k) {
        if (p.opcode[k] == CVGS_OP_ADD_ALPHA) cn = 4; // (the interpreted one-pixel kernels: the channel count the DIV stage meets)
        else if (p.opcode[k] == CVGS_OP_DROP_ALPHA) cn = 3;
        else if (p.opcode[k] == CVGS_OP_GRAY) cn = 1;
        if (p.opcode[k] != CVGS_OP_DIV) continue;
        for (int c = 0; c < cn; ++c) {
            const float d = p.operand[k][c], a = std::fabs(d);
            uint32_t bits;
            std::memcpy(&bits, &d, 4);
            if (!std::isfinite(d) || !(a >= std::ldexp(1.0f, -20) && a <= std::ldexp(1.0f, 20)) || (bits & 0x7fffffu) == 0x7fffffu) return;
        }
        for (int c = 0; c < 4; ++c) {
            volatile float r = c < cn ? 1.0f / p.operand[k][c] : 0.0f; // IEEE single division on the host: the correctly rounded reciprocal
            p.rdiv[c] = r;
        }
        p.fast_div = 2 + k;
        return;
    }
}

// Compile-time program: the opcode list is a template pack (operands stay run-time), so the chain
// is straight-line code the compiler can schedule against the loads and stores.
template <int... OPS>
struct StaticProg {
    struct State {};
    static __device__ __forceinline__ State prefetch(const ProgArgs&) { return {}; }
    static __device__ __forceinline__ void settle(State&) {}
    static __device__ __forceinline__ void run(const ProgArgs& prog, const State&, Px& p, int& depth, int& cn) { run(prog, p, depth, cn); }
    static __device__ __forceinline__ void run(const ProgArgs& prog, Px& p, int& depth, int& cn) {
        int k = 0;
        ((apply_op(OPS, prog.aux[k], prog.operand[k], p, depth, cn), ++k), ...);
    }
    // four pixels, stage by stage (the stages are per-pixel: the same operations on the same values as pixel by pixel)
    static __device__ __forceinline__ void run4(const ProgArgs& prog, Px (&px)[4], int& depth, int& cn) {
        [[maybe_unused]] int k = 0;
        ((step4<OPS>(prog, k, px, depth, cn), ++k), ...);
    }
    template <int OP>
    static __device__ __forceinline__ void step4(const ProgArgs& prog, int k, Px (&px)[4], int& depth, int& cn) {
        if constexpr (OP == CVGS_OP_DIV) {
            if (depth == CVGS_DEPTH_32F && div4_guarded(prog, k, px, cn)) return;
        }
        int d = depth, c = cn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d = depth;
            c = cn;
            apply_op(OP, prog.aux[k], prog.operand[k], px[i], d, c);
        }
        depth = d;
        cn = c;
    }
};

// ---- fk::PerThreadRead<_2D,T> ------------------------------------------------------------------
__device__ __forceinline__ void load_px(const uint8_t* row, int depth, int cn, int x, Px& p) {
    switch (depth) {
    case CVGS_DEPTH_8U:
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) p.v[c] = (float)row[x * cn + c];
        break;
    case CVGS_DEPTH_8S:
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) p.v[c] = (float)((const int8_t*)row)[x * cn + c];
        break;
    case CVGS_DEPTH_16U:
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) p.v[c] = (float)((const uint16_t*)row)[x * cn + c];
        break;
    case CVGS_DEPTH_16S:
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) p.v[c] = (float)((const int16_t*)row)[x * cn + c];
        break;
    case CVGS_DEPTH_16F:
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) p.v[c] = (float)((const _Float16*)row)[x * cn + c];
        break;
    default: // 32S raw bits, 32F
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) p.v[c] = ((const float*)row)[x * cn + c];
        break;
    }
}

// a tap as the float the interpolator multiplies (taps are cast to float first)
__device__ __forceinline__ float tap_f(float v, int depth) {
    return depth == CVGS_DEPTH_32S ? (float)as_int(v) : v;
}

// ---- NV12 -> RGB(A) float ----------------------------------------------------------------------
struct YuvK {
    float ysub, yscale, rv, gu, gv, bu;
    float csub, amax; // chroma centre and alpha (= full scale): 128 / 255 for 8-bit samples, 512 / 1023 for P010
    int layout;       // cvgs_yuv_layout of the source
};

// 6-decimal literals of the derivation from the standards' luma weights (BT.601, BT.709, BT.2020 NCL); limited range scales
// luma by 255/219 and chroma by 255/224 on 8-bit codes, by 1023/876 and 1023/896 on P010's 10-bit codes
// (tests/test_independent_pins.py re-derives every set in float64; the oracle holds the same literals).
__device__ __forceinline__ YuvK yuv_matrix(int range, int primaries, int layout = CVGS_YUV_NV12) {
    YuvK k;
    k.layout = layout;
    const bool ten = layout == CVGS_YUV_P010;
    k.csub = ten ? 512.f : 128.f;
    k.amax = ten ? 1023.f : 255.f;
    if (range == CVGS_YUV_FULL) {
        k.ysub = 0.f; k.yscale = 1.f;
        if (primaries == CVGS_BT601) { k.rv = 1.402f; k.gu = -0.344136f; k.gv = -0.714136f; k.bu = 1.772f; }
        else if (primaries == CVGS_BT709) { k.rv = 1.5748f; k.gu = -0.187324f; k.gv = -0.468124f; k.bu = 1.8556f; }
        else { k.rv = 1.4746f; k.gu = -0.164553f; k.gv = -0.571353f; k.bu = 1.8814f; }
    } else if (!ten) {
        k.ysub = 16.f; k.yscale = 1.164383f;
        if (primaries == CVGS_BT601) { k.rv = 1.596027f; k.gu = -0.391762f; k.gv = -0.812968f; k.bu = 2.017232f; }
        else if (primaries == CVGS_BT709) { k.rv = 1.792741f; k.gu = -0.213249f; k.gv = -0.532909f; k.bu = 2.112402f; }
        else { k.rv = 1.678674f; k.gu = -0.187326f; k.gv = -0.650424f; k.bu = 2.141772f; }
    } else {
        k.ysub = 64.f; k.yscale = 1.167808f;
        if (primaries == CVGS_BT601) { k.rv = 1.600721f; k.gu = -0.392915f; k.gv = -0.815359f; k.bu = 2.023165f; }
        else if (primaries == CVGS_BT709) { k.rv = 1.798014f; k.gu = -0.213876f; k.gv = -0.534477f; k.bu = 2.118615f; }
        else { k.rv = 1.683611f; k.gu = -0.187877f; k.gv = -0.652337f; k.bu = 2.148072f; }
    }
    return k;
}

__device__ __forceinline__ void yuv_to_rgb(float Y, float U, float V, const YuvK& k, Px& p) {
    const float cb = U - k.csub, cr = V - k.csub;
    const float yv = (Y - k.ysub) * k.yscale;
    p.v[0] = yv + k.rv * cr;
    p.v[1] = (yv + k.gu * cb) + k.gv * cr;
    p.v[2] = yv + k.bu * cb;
    p.v[3] = k.amax;
}

// one tap of the fast NV12 resize kernels (K4, k_nv12.hip; the queue's NV12 worker, k_queue.hip): yuv_to_rgb with the channel
// count and the range known at compile time (full range: (Y - 0) * 1 is Y itself, bit for bit, so the two instructions are
// dropped; the alpha lane only exists for CN 4)
template <int CN, bool FULL>
__device__ __forceinline__ void k4_tap(float Y, float U, float V, const YuvK& k, float* t) {
    const float cb = U - k.csub, cr = V - k.csub;
    const float yv = FULL ? Y : (Y - k.ysub) * k.yscale;
    t[0] = yv + k.rv * cr;
    t[1] = (yv + k.gu * cb) + k.gv * cr;
    t[2] = yv + k.bu * cb;
    if constexpr (CN == 4) t[3] = k.amax;
}

__device__ __forceinline__ void nv12_px(const PlaneParams& P, int x, int y, const YuvK& k, Px& p) {
    float Y, U, V;
    if (k.layout == CVGS_YUV_P010) { // NV12's geometry, 16-bit samples, 10-bit code in the high bits
        const uint16_t* yrow = (const uint16_t*)(P.data + (size_t)y * P.step);
        const uint16_t* uv = (const uint16_t*)(P.data + (size_t)P.uv_off + (size_t)(y >> 1) * P.step) + 2 * (x >> 1);
        Y = (float)(yrow[x] >> 6);
        U = (float)(uv[0] >> 6);
        V = (float)(uv[1] >> 6);
    } else if (k.layout <= CVGS_YUV_NV21) { // interleaved chroma: one (U,V) or (V,U) pair per 2x2 luma block
        const uint8_t* uv = P.data + (size_t)P.uv_off + (size_t)(y >> 1) * P.step + 2 * (x >> 1);
        Y = (float)P.data[(size_t)y * P.step + x];
        U = (float)uv[k.layout == CVGS_YUV_NV21 ? 1 : 0];
        V = (float)uv[k.layout == CVGS_YUV_NV21 ? 0 : 1];
    } else {                         // planar chroma: (W/2) x (H/2) planes with rows of step/2 bytes, one after the other
        const size_t cstep = (size_t)(P.step >> 1);
        const uint8_t* first = P.data + (size_t)P.uv_off + (size_t)(y >> 1) * cstep + (x >> 1);
        const uint8_t* second = first + (size_t)(P.h >> 1) * cstep;
        Y = (float)P.data[(size_t)y * P.step + x];
        U = (float)*(k.layout == CVGS_YUV_YV12 ? second : first);
        V = (float)*(k.layout == CVGS_YUV_YV12 ? first : second);
    }
    yuv_to_rgb(Y, U, V, k, p);
}

// ---- write stages ------------------------------------------------------------------------------
__device__ __forceinline__ void store_elem(uint8_t* base, size_t idx, int depth, float v) {
    switch (depth) {
    case CVGS_DEPTH_8U: base[idx] = (uint8_t)v; break;
    case CVGS_DEPTH_8S: ((int8_t*)base)[idx] = (int8_t)v; break;
    case CVGS_DEPTH_16U: ((uint16_t*)base)[idx] = (uint16_t)v; break;
    case CVGS_DEPTH_16S: ((int16_t*)base)[idx] = (int16_t)v; break;
    case CVGS_DEPTH_16F: ((_Float16*)base)[idx] = (_Float16)v; break;
    default: ((float*)base)[idx] = v; break; // 32S raw bits, 32F
    }
}

__device__ __forceinline__ void write_px(const WriteArgs& w, const DstPlane* dst_planes, int x, int y, int z,
                                         const Px& p, int depth, int cn) {
    const size_t W = (size_t)w.width;
    switch (w.kind) {
    case CVGS_WRITE_PIXEL_2D: {
        uint8_t* row = w.data + (size_t)y * (size_t)w.step;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) store_elem(row, (size_t)x * cn + c, depth, p.v[c]);
        break;
    }
    case CVGS_WRITE_PIXEL_2D_BATCH: {
        const DstPlane d = dst_planes[z];
        uint8_t* row = d.data + (size_t)y * (size_t)d.step;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) store_elem(row, (size_t)x * cn + c, depth, p.v[c]);
        break;
    }
    case CVGS_WRITE_PIXEL_3D: {
        const size_t pix = (size_t)z * w.img_stride + (size_t)y * W + x;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < cn) store_elem(w.data, pix * cn + c, depth, p.v[c]);
        if (w.data2) {
            const size_t pix2 = (size_t)z * w.img_stride2 + (size_t)y * W + x;
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < cn) store_elem(w.data2, pix2 * cn + c, depth, p.v[c]);
        }
        break;
    }
    case CVGS_WRITE_TENSOR_SPLIT:
    case CVGS_WRITE_TENSOR_T_SPLIT:
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < cn) store_elem(w.data, (size_t)z * w.img_stride + (size_t)c * w.ch_stride + (size_t)y * W + x, depth, p.v[c]);
        if (w.data2) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < cn)
                    store_elem(w.data2, (size_t)z * w.img_stride2 + (size_t)c * w.ch_stride2 + (size_t)y * W + x, depth, p.v[c]);
        }
        break;
    case CVGS_WRITE_SPLIT_2D:
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < cn) {
                const DstPlane d = dst_planes[(size_t)z * cn + c];
                store_elem(d.data + (size_t)y * (size_t)d.step, (size_t)x, depth, p.v[c]);
            }
        }
        break;
    default: break;
    }
}

// XCD-aware remap of a linear workgroup id: the dispatcher places workgroup b on XCD b % 8
// (MI355X_MICROARCH, "Workgroup dispatch"), so consecutive LOGICAL tiles (same crop, neighbouring
// rows, shared source rows) are steered to the same XCD and share its L2.  Speed only.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t total) {
    constexpr uint32_t kXcd = 8;
    const uint32_t per = (total + kXcd - 1) / kXcd;
    const uint32_t logical = (bid % kXcd) * per + bid / kXcd;
    return logical;
}

} // namespace cvgs
