// common.h -- harness shared by the C++ facade tests (the role of the reference's tests/testsCommon.cuh):
// result checks with the reference's tolerances, deterministic inputs, and the CPU oracle as the checker.
#pragma once

#include <cvGPUSpeedup.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../oracle/cvgs_oracle.h" // checker only: tests/ may use the oracle, the product never does

inline int g_failures = 0;

#define CHECK(cond, what)                                                           \
    do {                                                                            \
        if (!(cond)) {                                                              \
            std::cout << "  FAILED: " << what << " (" << __FILE__ << ":" << __LINE__ << ")" << std::endl; \
            ++g_failures;                                                           \
        }                                                                           \
    } while (0)

#define HIP_OK(call)                                                                \
    do {                                                                            \
        const hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                                     \
            std::cout << "  FAILED: " #call " -> " << hipGetErrorString(e_) << " (" << __FILE__ << ":" << __LINE__ << ")" << std::endl; \
            ++g_failures;                                                           \
        }                                                                           \
    } while (0)

inline int report(const char* name) {
    std::cout << name << (g_failures ? " failed!!" : " passed!!") << std::endl;
    return g_failures ? 1 : 0;
}

// splitmix64 (seed 0xC0FFEE family): same generator as cvgpuspeedup_amd/workloads.py
inline void fill_random(cv::Mat& m, uint64_t seed) {
    uint64_t s = seed;
    const size_t row_bytes = (size_t)m.cols * m.elemSize();
    for (int y = 0; y < m.rows; ++y) {
        uint8_t* row = m.data + (size_t)y * m.step;
        for (size_t i = 0; i < row_bytes; i += 8) {
            s += 0x9E3779B97F4A7C15ull;
            uint64_t z = s;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            std::memcpy(row + i, &z, std::min<size_t>(8, row_bytes - i));
        }
    }
}

// reference tolerance (tests/testsCommon.cuh:36-61): float |diff| <= 1e-4, integers exact
template <typename T>
inline bool close_enough(T a, T b) {
    if constexpr (std::is_floating_point_v<T>) return std::fabs((double)a - (double)b) <= 1e-4;
    else return a == b;
}

// every element of a dense host buffer equals `expected`
template <typename T>
inline bool all_close(const T* p, size_t n, double expected) {
    for (size_t i = 0; i < n; ++i)
        if (!close_enough<T>(p[i], (T)expected)) {
            std::cout << "    element " << i << " = " << (double)p[i] << ", expected " << expected << std::endl;
            return false;
        }
    return true;
}

inline bool bit_equal(const void* a, const void* b, size_t bytes) { return std::memcmp(a, b, bytes) == 0; }

// download a device buffer of `bytes` bytes
inline std::vector<uint8_t> fetch(const void* dev, size_t bytes) {
    std::vector<uint8_t> h(bytes);
    cv::cvgs_hip_check(hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
    return h;
}

// A host-backed "GpuMat": lets the SAME facade code build a chain whose pointers are host pointers, which is
// what the CPU oracle consumes.
inline cv::cuda::GpuMat host_view(cv::Mat& m) { return cv::cuda::GpuMat(m.rows, m.cols, m.type(), m.data, m.step); }

template <typename... IOps>
inline void run_oracle(const IOps&... iops) {
    fk::ChainBuilder b;
    fk::lowerChain(b, iops...);
    const int rc = oracle_execute(&b.d);
    if (rc != 0) {
        std::cout << "  oracle_execute failed: " << rc << std::endl;
        ++g_failures;
    }
}
