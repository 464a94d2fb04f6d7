// k_taps.hpp -- the tap machinery shared by the fast resize kernel (k_k1.hip) and the fast warp kernel (k_warp.hip):
// compile-time pointwise programs, the pixel-pair window (both horizontal taps of a source row in ONE unaligned load),
// its unpacking, non-temporal element stores.
#pragma once

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "k_common.hpp"

namespace cvgs {

// Compile-time programs for the chains the reference's tests spell.  kOpSwapRB is an internal opcode: REORDER whose
// permutation is the RGB<->BGR swap (aux 2,1,0[,3]) resolved at compile time instead of per-pixel selects.
constexpr int kOpSwapRB = 100;

// x / d for a wave-uniform divisor d whose correctly rounded reciprocal r = RN(1/d) came with the kernel arguments:
//   q0 = RN(x r);  e0 = RN(x - d q0);  q1 = RN(q0 + e0 r);  e1 = x - d q1 (exact);  q = RN(q1 + e1 r) = RN(x / d).
// q1 is within half an ulp (+ 2^-45 relative) of x/d, so e1 is exactly representable and the last step is Markstein's
// correction (P. Markstein, "Computation of elementary functions on the IBM RISC System/6000 processor", 1990, Thm 2):
// with r within half an ulp of 1/d and q1 faithful, RN(q1 + e1 r) IS the correctly rounded quotient -- the same bits the
// hardware's v_div_scale / v_rcp / v_div_fmas / v_div_fixup sequence (10 instructions, half of them on the uniform
// divisor again for every pixel) and the oracle's IEEE division produce.  Preconditions, checked on the HOST for the
// operands (launch_k1: k1_fast_div_ok) and by construction for x: everything finite and far from the exponent range's
// ends, d's significand not all ones; x == 0 is excluded by the caller (the sign of a zero quotient needs the real
// division).  tests/test_fast_division.py checks the identity against IEEE division for EVERY divisor significand.
// (div_by_uniform itself lives in k_common.hpp: the pointwise programs use it too)

// HOST side of div_by_uniform: can the DIV stage of a [swap] MUL SUB DIV program use it?  Integer-valued sources only
// (8U / 16U / 16S pixels, NV12 bytes through the YCbCr matrix): the interpolated value v is finite and bounded
// (|v| < 2^17), so bounds on the operands bound the dividend x = v * mul - sub: |x| < 2^38, and a non-zero x is never
// smaller than 2^-90 (v is a sum of products of two weights >= 2^-46 with values that are integers or >= 2^-24 after the
// matrix; times |mul| >= 2^-20; or a difference of two floats one of which is >= 2^-20), far inside the range where every
// intermediate of the FMA corrections is a normal number and the residuals are exact: with 2^-20 <= |d| <= 2^20 the
// quotient stays above 2^-110 and the lowest bit of d*q above 2^-136 (the oracle's checker shows the identity FAILING
// once quotients may be subnormal, e.g. |x| = 2^-100 with |d| = 2^40).  A divisor whose significand is
// all ones is left to the real division (the one case where RN(1/d) is not good enough for Markstein's theorem), and so
// is a background value outside [2^-20, 2^20] (it is pushed through the same program).
inline void fast_div_setup(ProgArgs& p, int div_at, int mul_at, int cn, const float* bg) {
    auto in_range = [](float v, int lo_exp, int hi_exp) {
        const float a = std::fabs(v);
        return std::isfinite(v) && a >= std::ldexp(1.0f, lo_exp) && a <= std::ldexp(1.0f, hi_exp);
    };
    for (int c = 0; c < cn; ++c) {
        const float mul = p.operand[mul_at][c], sub = p.operand[mul_at + 1][c], d = p.operand[div_at][c];
        if (!in_range(mul, -20, 20) || !(sub == 0.0f || in_range(sub, -20, 20)) || !in_range(d, -20, 20)) return;
        // the background value (aspect-ratio padding, unused planes) runs through the same program
        if (!(bg[c] == 0.0f || in_range(bg[c], -20, 20))) return;
        uint32_t bits;
        std::memcpy(&bits, &d, 4);
        if ((bits & 0x7fffffu) == 0x7fffffu) return;
    }
    for (int c = 0; c < cn; ++c) {
        volatile float r = 1.0f / p.operand[div_at][c]; // IEEE single division on the host: the correctly rounded reciprocal
        p.rdiv[c] = r;
    }
    p.fast_div = 1;
}

template <int... OPS>
struct K1Prog {
    struct State {};
    static __device__ __forceinline__ State prefetch(const ProgArgs&) { return {}; }
    static __device__ __forceinline__ void settle(State&) {}
    static __device__ __forceinline__ void run(const ProgArgs& prog, const State&, Px& p, int& depth, int& cn) { run(prog, p, depth, cn); }
    static __device__ __forceinline__ void run(const ProgArgs& prog, Px& p, int& depth, int& cn) {
        [[maybe_unused]] int k = 0;
        ((step<OPS>(prog, k, p, depth, cn), ++k), ...);
    }
    template <int OP>
    static __device__ __forceinline__ void step(const ProgArgs& prog, int k, Px& p, int& depth, int& cn) {
        if constexpr (OP == kOpSwapRB) {
            const float t = p.v[0];
            p.v[0] = p.v[2];
            p.v[2] = t;
        } else if constexpr (OP == CVGS_OP_DIV) {
            if (prog.fast_div) { // wave-uniform
                float mn = fabsf(p.v[0]);
#pragma unroll
                for (int c = 1; c < 4; ++c)
                    if (c < cn) mn = fminf(mn, fabsf(p.v[c]));
                // a zero dividend (its quotient's sign) goes through the real division: the whole wave takes that path
                if (__builtin_amdgcn_ballot_w64(mn == 0.0f) == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < cn) p.v[c] = div_by_uniform(p.v[c], prog.operand[k][c], prog.rdiv[c]);
                    return;
                }
            }
            apply_op(OP, prog.aux[k], prog.operand[k], p, depth, cn);
        } else {
            apply_op(OP, prog.aux[k], prog.operand[k], p, depth, cn);
        }
    }
};
using ProgSwapMulSubDiv = K1Prog<kOpSwapRB, CVGS_OP_MUL, CVGS_OP_SUB, CVGS_OP_DIV>;
using ProgMulSubDiv = K1Prog<CVGS_OP_MUL, CVGS_OP_SUB, CVGS_OP_DIV>;
// the empty program: resize -> [trailing cast folded into the store] -> write, the reference's single-image resize chains
// (tests/resize/test_resize_write.cu: resize -> convertTo<CV_32F, O> -> write).  Channel count and depth stay compile-time
// constants, so the whole-frame kernels carry no interpreter at all.
using ProgNone = K1Prog<>;

// ---- the CANONICAL arithmetic program (round 6): [swap R,B] -> two fma stages -> [DIV] -> two fma stages, straight-line --------------------------
// Every chain of the shape  [cvtColor(R<->B)]  {mul | add | sub} x 0..2  [div]  {mul | add | sub} x 0..2  -- "subtract the mean, divide by the
// deviation", "scale and shift", "divide by 255", one more stage behind the reference's normalisation ... -- is rewritten ON THE HOST
// (k1_canonicalise) into this fixed pipeline: a MUL / ADD / SUB stage is ONE fma with the operands (o, -0) / (1, o) / (1, -o) -- the same single
// rounding as the plain operation, zero signs included -- and a missing stage is the identity (1, -0).  Nothing is decoded at run time: the
// interpreted kernel's tick of 16 x 50 crops cost 60 us (51 with its arithmetic path) for such a chain against 39 for the reference's own.
//   prog.aux[0] = swap R,B first; prog.aux[1] = a DIV stage sits between the two fma pairs (divisors in operand[8], reciprocals in rdiv when
//   prog.fast_div says the divisors fit: then the dividends are checked per wave, k_common.hpp div1_guarded);
//   fma stage s (0..3): multipliers operand[s], addends operand[4 + s].
struct K1CanonProg {
    struct State {};
    static __device__ __forceinline__ State prefetch(const ProgArgs&) { return {}; }
    static __device__ __forceinline__ void settle(State&) {}
    static __device__ __forceinline__ void run(const ProgArgs& prog, const State&, Px& p, int& depth, int& cn) { run(prog, p, depth, cn); }
    static __device__ __forceinline__ void fma_stage(const ProgArgs& prog, int s, Px& p) {
#pragma unroll
        for (int c = 0; c < 4; ++c) p.v[c] = __builtin_fmaf(p.v[c], prog.operand[s][c], prog.operand[4 + s][c]); // (channels at and beyond cn hold nothing anybody stores)
    }
    static __device__ __forceinline__ void run(const ProgArgs& prog, Px& p, int& depth, int& cn) {
        if (prog.aux[0]) { // wave-uniform
            const float t = p.v[0];
            p.v[0] = p.v[2];
            p.v[2] = t;
        }
        fma_stage(prog, 0, p);
        fma_stage(prog, 1, p);
        if (prog.aux[1]) { // wave-uniform
            const float d[4] = {prog.operand[8][0], prog.operand[8][1], prog.operand[8][2], prog.operand[8][3]};
            const float r[4] = {prog.rdiv[0], prog.rdiv[1], prog.rdiv[2], prog.rdiv[3]};
            if (!div1_guarded(prog.fast_div != 0, d, r, p, cn)) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < cn) p.v[c] = p.v[c] / d[c];
            }
        }
        fma_stage(prog, 2, p);
        fma_stage(prog, 3, p);
    }
};
// HOST: rewrite `in` into the canonical pipeline if it has that shape (false: it does not)
inline bool k1_canonicalise(const ProgArgs& in, int cn, ProgArgs& out) {
    const int swap = cn == 3 ? (2 | (1 << 2) | (0 << 4)) : (2 | (1 << 2) | (0 << 4) | (3 << 6));
    int k = 0;
    bool has_swap = false;
    if (cn >= 3 && k < in.n && in.opcode[k] == CVGS_OP_REORDER && in.aux[k] == swap) {
        has_swap = true;
        ++k;
    }
    auto is_lin = [](int op) { return op == CVGS_OP_MUL || op == CVGS_OP_ADD || op == CVGS_OP_SUB; };
    int lin[4] = {-1, -1, -1, -1}, div_at = -1;
    for (int s = 0; s < 2 && k < in.n && is_lin(in.opcode[k]); ++s) lin[s] = k++;
    if (k < in.n && in.opcode[k] == CVGS_OP_DIV) div_at = k++;
    for (int s = 2; s < 4 && k < in.n && is_lin(in.opcode[k]); ++s) lin[s] = k++;
    if (k != in.n) return false;
    out = in; // (n stays: "is there a program at all" is still asked of it; nothing is interpreted)
    for (int i = 0; i < CVGS_MAX_OPS; ++i) {
        out.opcode[i] = 0;
        out.aux[i] = 0;
        for (int c = 0; c < 4; ++c) out.operand[i][c] = 0.f;
    }
    out.aux[0] = has_swap ? 1 : 0;
    out.aux[1] = div_at >= 0 ? 1 : 0;
    for (int s = 0; s < 4; ++s)
        for (int c = 0; c < 4; ++c) {
            float m = 1.0f, a = -0.0f;
            if (lin[s] >= 0) {
                const float o = in.operand[lin[s]][c];
                if (in.opcode[lin[s]] == CVGS_OP_MUL) m = o;
                else a = in.opcode[lin[s]] == CVGS_OP_ADD ? o : -o;
            }
            out.operand[s][c] = m;
            out.operand[4 + s][c] = a;
        }
    out.fast_div = 0;
    for (int c = 0; c < 4; ++c) out.rdiv[c] = 0.f;
    if (div_at >= 0) {
        bool fits = true;
        for (int c = 0; c < 4; ++c) {
            const float d = in.operand[div_at][c];
            out.operand[8][c] = d;
            if (c < cn) {
                uint32_t bits;
                std::memcpy(&bits, &d, 4);
                const float a = std::fabs(d);
                fits = fits && std::isfinite(d) && a >= std::ldexp(1.0f, -20) && a <= std::ldexp(1.0f, 20) && (bits & 0x7fffffu) != 0x7fffffu;
            }
        }
        if (fits) {
            for (int c = 0; c < cn; ++c) {
                volatile float r = 1.0f / in.operand[div_at][c]; // IEEE single division on the host: the correctly rounded reciprocal
                out.rdiv[c] = r;
            }
            out.fast_div = 1;
        }
    }
    return true;
}

// program shape: [REORDER(swap R,B)] MUL SUB DIV, with the swap's permutation checked on the host
inline int k1_classify_program(const ProgArgs& p, int cn) {
    if (cn < 3) return (p.n == 3 && p.opcode[0] == CVGS_OP_MUL && p.opcode[1] == CVGS_OP_SUB && p.opcode[2] == CVGS_OP_DIV) ? 1 : 2;
    const int swap = cn == 3 ? (2 | (1 << 2) | (0 << 4)) : (2 | (1 << 2) | (0 << 4) | (3 << 6));
    if (p.n == 4 && p.opcode[0] == CVGS_OP_REORDER && p.aux[0] == swap && p.opcode[1] == CVGS_OP_MUL &&
        p.opcode[2] == CVGS_OP_SUB && p.opcode[3] == CVGS_OP_DIV)
        return 0;
    if (p.n == 3 && p.opcode[0] == CVGS_OP_MUL && p.opcode[1] == CVGS_OP_SUB && p.opcode[2] == CVGS_OP_DIV) return 1;
    return 2;
}


typedef uint64_t u64_unaligned __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) u64_unaligned* gptr_u64;
typedef const __attribute__((address_space(1))) uint8_t* gptr_u8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_unaligned __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) u32x4_unaligned* gptr_u32x4;

// Source element kinds of the fast path (the reference sweeps K1 over 8U, 16U and 16S sources,
// tests/batchresize/test_batchresize_x_split3D.cu:427-432).
enum { SRC_U8 = 0, SRC_U16 = 1, SRC_S16 = 2, SRC_F32 = 3 };
template <int SRC> constexpr int elem_bytes = SRC == SRC_U8 ? 1 : (SRC == SRC_F32 ? 4 : 2);

// The tap window of one lane and one source row: both horizontal taps (a pixel pair: 2*CN elements) arrive in ONE
// unaligned load -- 8 bytes for u8 pixels (6 or 8 used), 16 bytes for 16-bit pixels (12 or 16 used).
template <int EB> struct Win;
template <> struct Win<1> { uint64_t lo; };
template <> struct Win<2> { uint64_t lo, hi; };
// CV_32F pixels: the window is exactly the pixel pair (2*CN floats, 8..32 bytes); no sub-dword shifting is ever needed
template <> struct Win<4> { float e[8]; };
typedef float f32_unaligned __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) f32_unaligned* gptr_f32u;
template <int CN>
__device__ __forceinline__ Win<4> load_win_f32(gptr_u8 p) {
    Win<4> w;
#pragma unroll
    for (int k = 0; k < 2 * CN; ++k) w.e[k] = *(gptr_f32u)(p + 4 * k);
    return w;
}
// one-pixel-wide rows: both taps are pixel 0
template <int CN>
__device__ __forceinline__ Win<4> gather_win_f32(gptr_u8 row) {
    Win<4> w;
#pragma unroll
    for (int k = 0; k < CN; ++k) w.e[k] = w.e[CN + k] = *(gptr_f32u)(row + 4 * k);
    return w;
}
// second_half: the window was clamped back by one pixel (x1 is the row's last pixel), so pixel x1 is the window's 2nd
template <int CN>
__device__ __forceinline__ void unpack_pair_f32(const Win<4>& w, bool second_half, bool edge, float* a, float* b) {
#pragma unroll
    for (int k = 0; k < CN; ++k) {
        a[k] = second_half ? w.e[CN + k] : w.e[k];
        b[k] = edge ? a[k] : w.e[CN + k];
    }
}

template <int EB>
__device__ __forceinline__ Win<EB> load_win(gptr_u8 p) {
    Win<EB> w;
    if constexpr (EB == 1) {
#if defined(CVGS_K1_LDSCOPE) && CVGS_K1_LDSCOPE // probe (tools/probes/build_ablate.sh): sc0 / sc1 / sc0 sc1 on the tap loads -- does the L2 then fetch less than 128-byte lines?
        w.lo = __hip_atomic_load((const uint64_t*)p, __ATOMIC_RELAXED, CVGS_K1_LDSCOPE == 1 ? __HIP_MEMORY_SCOPE_WORKGROUP : (CVGS_K1_LDSCOPE == 2 ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_SYSTEM));
#else
        w.lo = *(gptr_u64)p; // plain, cached: neighbouring lanes and rows share lines (non-temporal loads measured 8 % slower at 16 x 50 crops, 24 % at 3200)
#endif
    } else {
        const u32x4 v = *(gptr_u32x4)p;
        w.lo = ((uint64_t)v.y << 32) | v.x;
        w.hi = ((uint64_t)v.w << 32) | v.z;
    }
    return w;
}

// rows narrower than the window (1-2 pixels): gather byte by byte with clamped, always-valid addresses
template <int CN, int EB>
__device__ __forceinline__ Win<EB> gather_win(gptr_u8 row, int o, int row_bytes) {
    uint64_t part[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 2 * CN * EB; ++k) {
        const uint64_t b = row[min(o + k, row_bytes - 1)];
        part[k >> 3] |= b << (8 * (k & 7));
    }
    Win<EB> w;
    w.lo = part[0];
    if constexpr (EB == 2) w.hi = part[1];
    return w;
}

// window >> sh bits (sh is a multiple of the pixel size; non-zero only for the last columns of a row)
template <int EB>
__device__ __forceinline__ Win<EB> shift_win(Win<EB> w, int sh) {
    if constexpr (EB == 1) {
        w.lo >>= sh;
    } else {
        if (sh >= 64) {
            w.lo = w.hi >> (sh - 64);
            w.hi = 0;
        } else if (sh > 0) {
            w.lo = (w.lo >> sh) | (w.hi << (64 - sh));
            w.hi >>= sh;
        }
    }
    return w;
}

template <int SRC>
__device__ __forceinline__ float elem_to_float(uint32_t bits) {
    if constexpr (SRC == SRC_S16) return (float)(int16_t)(uint16_t)bits;
    else return (float)bits;
}

// pixel pair -> floats; at the right edge (x2 clamped onto x1) the second pixel IS the first one
template <int CN, int SRC>
__device__ __forceinline__ void unpack_pair(const Win<elem_bytes<SRC>>& w, bool edge, float* a, float* b) {
    if constexpr (SRC == SRC_U8) {
        // pixel 0 = bytes 0..CN-1, pixel 1 = bytes CN..2CN-1 of the 8-byte window
        const uint32_t lo = (uint32_t)w.lo;
        const uint32_t second = (uint32_t)(w.lo >> (8 * CN));
        const uint32_t s = edge ? lo : second;
#pragma unroll
        for (int k = 0; k < CN; ++k) {
            a[k] = (float)((lo >> (8 * k)) & 0xffu);
            b[k] = (float)((s >> (8 * k)) & 0xffu);
        }
    } else {
        // 16-bit elements e0..e(2CN-1): e0..e3 in lo, e4.. in hi
        const uint64_t first = w.lo;                                                             // pixel 0: e0..e(CN-1)
        const uint64_t second = CN == 4 ? w.hi : ((w.lo >> (16 * CN)) | (w.hi << (64 - 16 * CN))); // pixel 1: e(CN)..e(2CN-1)
        const uint64_t s = edge ? first : second;
#pragma unroll
        for (int k = 0; k < CN; ++k) {
            a[k] = elem_to_float<SRC>((uint32_t)(first >> (16 * k)) & 0xffffu);
            b[k] = elem_to_float<SRC>((uint32_t)(s >> (16 * k)) & 0xffffu);
        }
    }
}

// A wave-uniform pointer pinned in SGPRs: stores / loads through (pinned base + 32-bit lane offset) then use the
// scalar-base addressing form (global_store_dword v_off, v_data, s[base:base+1]); without the pin LLVM folds the lane
// offset into ONE 64-bit vector base and pays a 64-bit VALU add per access for the uniform row / channel strides.
template <typename T>
__device__ __forceinline__ T* pin_uniform(T* p) {
#ifdef CVGS_NO_PIN
    return p;
#else
    uint64_t v = (uint64_t)p;
    asm volatile("" : "+s"(v));
    return (T*)v;
#endif
}
__device__ __forceinline__ gptr_u8 pin_uniform(gptr_u8 p) {
#ifdef CVGS_NO_PIN
    return p;
#else
    uint64_t v = (uint64_t)p;
    asm volatile("" : "+s"(v));
    return (gptr_u8)v;
#endif
}
// element `off_bytes / sizeof(T)` of a pinned row: the lane offset stays a zero-extended 32-bit byte offset
template <typename T>
__device__ __forceinline__ T* lane_elem(T* row, uint32_t off_bytes) {
    return (T*)((__attribute__((address_space(1))) char*)(__attribute__((address_space(1))) T*)row + off_bytes);
}

// one planar element: the row pointer is wave-uniform and pinned in SGPRs, the lane adds its 32-bit byte offset
template <typename OT>
__device__ __forceinline__ void st_row(OT* row_uniform, uint32_t x_bytes, float v) {
    typedef __attribute__((address_space(1))) char* gchar;
    typedef __attribute__((address_space(1))) OT* got;
    const gchar r = (gchar)(got)pin_uniform(row_uniform);
#if defined(CVGS_K1_STORE) && CVGS_K1_STORE != 0 // tools/probes/tick_ablation.py only: 1 plain, 2 agent-scope (sc1), 3 system-scope write-through (sc0 sc1)
    if constexpr (std::is_same_v<OT, float>) {
        if constexpr (CVGS_K1_STORE == 1) *(got)(r + x_bytes) = v;
        else __hip_atomic_store((__attribute__((address_space(1))) uint32_t*)(r + x_bytes), __float_as_uint(v), __ATOMIC_RELAXED, CVGS_K1_STORE == 2 ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
#endif
    if constexpr (std::is_same_v<OT, float>) __builtin_nontemporal_store(v, (got)(r + x_bytes));
    else __builtin_nontemporal_store((OT)v, (got)(r + x_bytes));
}

// The same element into a PEER's tensor (cvgs_write_desc.mirrors: P2P-mapped memory of another GPU): a SYSTEM-scope write-through
// store (sc0 sc1), the flavour the descriptor queue's workers publish their rows with.  The arrival flag that follows
// (k_exchange.hip) is a relaxed system-scope store behind a kernel boundary; with write-through rows nothing depends on which
// scope the runtime gives that boundary's release between two back-to-back kernels.
template <typename OT>
__device__ __forceinline__ void st_row_sys(OT* row_uniform, uint32_t x_bytes, float v) {
    typedef __attribute__((address_space(1))) char* gchar;
    typedef __attribute__((address_space(1))) OT* got;
    const gchar r = (gchar)(got)pin_uniform(row_uniform);
    if constexpr (std::is_same_v<OT, float>) {
        __hip_atomic_store((__attribute__((address_space(1))) uint32_t*)(r + x_bytes), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        const OT h = (OT)v;
        uint16_t bits;
        __builtin_memcpy(&bits, &h, 2);
        __hip_atomic_store((__attribute__((address_space(1))) uint16_t*)(r + x_bytes), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
template <typename OT>
__device__ __forceinline__ void st_sys(OT* p, float v) {
    if constexpr (std::is_same_v<OT, float>) {
        __hip_atomic_store((uint32_t*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        const OT h = (OT)v;
        uint16_t bits;
        __builtin_memcpy(&bits, &h, 2);
        __hip_atomic_store((uint16_t*)p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__device__ __forceinline__ void st_nt(float* p, float v) { __builtin_nontemporal_store(v, p); }
// fp16 output: the chain's trailing CAST(CV_16F) is this one round-to-nearest-even conversion
__device__ __forceinline__ void st_nt(_Float16* p, float v) { __builtin_nontemporal_store((_Float16)v, p); }

__device__ __forceinline__ void st_plain(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st_plain(_Float16* p, float v) { __builtin_nontemporal_store((_Float16)v, p); }
// u8 targets: the chain's trailing SaturateCast (round to nearest even, clamp, NaN -> 0) is this conversion
__device__ __forceinline__ void st_plain(uint8_t* p, float v) { __builtin_nontemporal_store((uint8_t)sat_u8_insert(v, 0, 0), p); }
// 16-bit integer targets (the reference's resize -> convertTo<32F, 16U / 16S> -> write chains, tests/resize/test_resize_write.cu)
__device__ __forceinline__ void st_plain(uint16_t* p, float v) { __builtin_nontemporal_store((uint16_t)sat_u16_bits(v), p); }
__device__ __forceinline__ void st_plain(int16_t* p, float v) { __builtin_nontemporal_store((int16_t)(uint16_t)sat_s16_bits(v), p); }

// u8c3 packed pixels of a FULL 64-column tile: the wave's 192 output bytes leave as 48 dword stores instead of 192 byte
// stores.  Lane j < 48 assembles bytes 4j..4j+3 from the pixels of lanes p0 = 4j/3 and p0+1 (wave shuffles).
__device__ __forceinline__ void store_u8c3_tile(uint8_t* tile_row, int lane, const float* v) {
    const uint32_t mine = sat_u8_insert(v[2], 2, sat_u8_insert(v[1], 1, sat_u8_insert(v[0], 0, 0)));
    const int p0 = (4 * lane) / 3, o = 4 * lane - 3 * p0;
    const uint32_t a = (uint32_t)__shfl((int)mine, min(p0, 63)), b = (uint32_t)__shfl((int)mine, min(p0 + 1, 63));
    const uint64_t s = (uint64_t)a | ((uint64_t)b << 24);
    if (lane < 48) {
        typedef uint32_t u32a1 __attribute__((aligned(1)));
        __builtin_nontemporal_store((uint32_t)(s >> (8 * o)), (u32a1*)(tile_row + 4 * lane));
    }
}

// one packed pixel: a single vector store when the channel count is the compile-time one (the usual case), element
// stores when the chain changed it (e.g. *2GRAY after the resize)
template <int CN, typename OT>
__device__ __forceinline__ void store_packed_px(OT* px, const float* v, int cn) {
    if (CN >= 3 && cn == CN) {
        if constexpr (std::is_same_v<OT, float>) {
            typedef float vf __attribute__((ext_vector_type(CN)));
            typedef vf vfu __attribute__((aligned(4)));
            vf q;
#pragma unroll
            for (int k = 0; k < CN; ++k) {
                // The empty asm pins each channel in a VGPR before the vector is assembled.  Without it LLVM folds the four reads
                // into ONE <4 x float> load of the caller's Px; next to the interpreter's integer-typed accesses of the same
                // slots (CV_32S values travel as raw bits) that whole-array access keeps the pixel in SCRATCH (K4's interpreted
                // C4 kernels: 32 / 48 bytes per lane; tools/kernel_resources.py).  No instruction is emitted for it.
                float e = v[k];
                asm("" : "+v"(e));
                q[k] = e;
            }
            __builtin_nontemporal_store(q, (vfu*)px); // global_store_dwordx3 / x4
            return;
        } else if constexpr (std::is_same_v<OT, _Float16>) {
            // pairs of halves as one 32-bit store (a 3-element half vector would be stored as 8 bytes)
            typedef _Float16 vh2 __attribute__((ext_vector_type(2)));
            typedef vh2 vh2u __attribute__((aligned(2)));
            vh2 lo = {(_Float16)v[0], (_Float16)v[1]};
            __builtin_nontemporal_store(lo, (vh2u*)px);
            if constexpr (CN == 4) {
                vh2 hi = {(_Float16)v[2], (_Float16)v[3]};
                __builtin_nontemporal_store(hi, (vh2u*)(px + 2));
            } else {
                __builtin_nontemporal_store((_Float16)v[2], px + 2);
            }
            return;
        } else if constexpr (sizeof(OT) == 2) { // 16-bit integers: pairs of elements as one 32-bit store
            typedef uint32_t u32a2 __attribute__((aligned(2)));
            auto bits = [](float x) { return std::is_same_v<OT, uint16_t> ? sat_u16_bits(x) : sat_s16_bits(x); };
            const uint32_t e0 = bits(v[0]), e1 = bits(v[1]);
            __builtin_nontemporal_store(e0 | (e1 << 16), (u32a2*)px);
            if constexpr (CN == 4) {
                const uint32_t e2 = bits(v[2]), e3 = bits(v[3]);
                __builtin_nontemporal_store(e2 | (e3 << 16), (u32a2*)(px + 2));
            } else {
                st_plain(px + 2, v[2]);
            }
            return;
        } else if constexpr (CN == 4) { // u8c4: one dword
            typedef uint32_t u32a1 __attribute__((aligned(1)));
            uint32_t q = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) q = sat_u8_insert(v[k], (uint32_t)k, q);
            __builtin_nontemporal_store(q, (u32a1*)px);
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < cn) st_plain(px + k, v[k]);
}

} // namespace cvgs
