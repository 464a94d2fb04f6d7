// any_order_probe.cpp -- does hipExtAnyOrderLaunch (AQL barrier bit cleared) let two kernels of ONE stream overlap on gfx950?
//   hipcc -O2 --offload-arch=gfx950 tools/probes/any_order_probe.cpp -o /tmp/any_order_probe && /tmp/any_order_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t = wall_clock64();
    const unsigned long long end = t + ticks;
    while (wall_clock64() < end) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && out) out[blockIdx.x] = wall_clock64();
}
__global__ void empty_k(int* p) { if (p && threadIdx.x == 1024) *p = 1; }

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    hipStream_t s;
    OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long* out;
    OK(hipMalloc(&out, 4096));
    const unsigned long long us100 = 100ull * 100; // wall_clock64: 100 MHz
    for (int mode = 0; mode < 2; ++mode) {
        const unsigned flags = mode ? hipExtAnyOrderLaunch : 0;
        for (int rep = 0; rep < 3; ++rep) {
            OK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 8; ++k) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, k ? flags : 0u, us100, out);
            OK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            std::printf("%s: 8 x 100 us spin kernels on one stream: %.1f us\n", mode ? "hipExtAnyOrderLaunch" : "in order            ", us);
        }
    }
    // dispatch rate of tiny kernels: in order vs any order
    for (int mode = 0; mode < 2; ++mode) {
        const unsigned flags = mode ? hipExtAnyOrderLaunch : 0;
        for (int rep = 0; rep < 3; ++rep) {
            OK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < 2000; ++k) hipExtLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, s, nullptr, nullptr, flags, (int*)nullptr);
            const double us_h = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            OK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            std::printf("%s: 2000 empty 256x256 kernels: host %.2f us per launch, total %.2f us per launch\n", mode ? "hipExtAnyOrderLaunch" : "in order            ", us_h / 2000, us / 2000);
        }
    }
    return 0;
}
