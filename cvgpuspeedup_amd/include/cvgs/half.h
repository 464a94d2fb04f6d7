// cvgs/half.h -- the host-side spelling of CV_16F elements (engine extension: the half-precision hand-off).  hipcc / clang and GCC >= 12
// have _Float16; older GCCs (this image's g++ 11 in C++ mode) do not: there the type is 16 bits of storage with correctly rounded
// (round-to-nearest-even, straight from double: no double rounding) conversions, which is all the host side does with a half value
// (cv::Mat::store / read-back in tests; the arithmetic is the device's).
#pragma once

#include <cstdint>
#include <cstring>

#if defined(__FLT16_MANT_DIG__) && !defined(CVGS_HALF_FORCE_SOFT) // (the macro: tests compare the stand-in with the compiler's type)
namespace cvgs {
using half_t = _Float16;
}
#else
namespace cvgs {
struct half_t {
    uint16_t bits = 0;
    half_t() = default;
    half_t(double v) : bits(from_double(v)) {}
    operator float() const { return to_float(bits); }

    static uint16_t from_double(double v) {
        uint64_t u;
        std::memcpy(&u, &v, 8);
        const uint16_t sign = (uint16_t)((u >> 48) & 0x8000u);
        const int64_t e = (int64_t)((u >> 52) & 0x7ff);
        const uint64_t m = u & 0xfffffffffffffull;
        if (e == 0x7ff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u | (uint16_t)(m >> 42) : 0u)); // inf / nan (quiet, payload's top bits)
        const int64_t he = e - 1023 + 15; // half's biased exponent
        if (he >= 31) return (uint16_t)(sign | 0x7c00u); // overflow -> inf
        uint64_t sig = e ? (m | (1ull << 52)) : m; // 53-bit significand (double subnormals are far below half's range anyway)
        int shift = 42;                            // 52 -> 10 fraction bits
        if (he <= 0) {                             // half subnormal (or zero): the implicit bit moves into the fraction
            if (he < -10) return sign;             // below half of the smallest subnormal: +-0 (ties handled below for he == -10)
            shift += (int)(1 - he);
        }
        const uint64_t kept = sig >> shift, rest = sig & ((1ull << shift) - 1), halfway = 1ull << (shift - 1);
        uint64_t r = kept + ((rest > halfway || (rest == halfway && (kept & 1))) ? 1 : 0);
        if (he <= 0) return (uint16_t)(sign | (uint16_t)r);               // (a carry into bit 10 IS the smallest normal)
        r += (uint64_t)(he - 1) << 10;                                      // kept holds the implicit bit at position 10: add exponent - 1
        return r >= 0x7c00u ? (uint16_t)(sign | 0x7c00u) : (uint16_t)(sign | (uint16_t)r);
    }
    static float to_float(uint16_t h) {
        const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
        uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, out;
        if (e == 0) {
            if (!m) out = sign;
            else {
                int s = 0;
                while (!(m & 0x400u)) { m <<= 1; ++s; }
                out = sign | ((uint32_t)(127 - 15 + 1 - s) << 23) | ((m & 0x3ffu) << 13);
            }
        } else if (e == 31) out = sign | 0x7f800000u | (m << 13);
        else out = sign | ((e + 127 - 15) << 23) | (m << 13);
        float f;
        std::memcpy(&f, &out, 4);
        return f;
    }
};
static_assert(sizeof(half_t) == 2, "half_t is 16 bits of storage");
} // namespace cvgs
#endif
