// What does the HOST pay per cvgs_queue_submit?  The pieces of the direct-slot publish (csrc/k_queue.hip: slot + index entry ->
// sfence -> tail -> sfence, all stores into uncached device memory through the PCIe BAR), timed one by one on the box:
//   hipcc -O2 tools/probes/submit_cost_probe.cpp -o /tmp/submit_cost_probe && /tmp/submit_cost_probe
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <chrono>

template <typename F> static double per_call_ns(int n, F f) {
    for (int i = 0; i < 200; ++i) f(i);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) f(i);
    return std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
    uint8_t* ring = nullptr;
    const int R = 128, SLOT = 4096;
    if (hipExtMallocWithFlags((void**)&ring, (size_t)R * SLOT + 4096, hipDeviceMallocUncached) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(ring, 0, (size_t)R * SLOT + 4096);
    hipDeviceSynchronize();
    volatile uint64_t* tail = (volatile uint64_t*)(ring + (size_t)R * SLOT);
    static uint8_t src[4096];
    for (int i = 0; i < 4096; ++i) src[i] = (uint8_t)i;
    const int N = 20000;
    for (int bytes : {256, 768, 1536, 2816, 4032}) {
        const double a = per_call_ns(N, [&](int i) { memcpy(ring + (size_t)(i % R) * SLOT, src, bytes); _mm_sfence(); *tail = i; _mm_sfence(); });
        const double b = per_call_ns(N, [&](int i) { memcpy(ring + (size_t)(i % R) * SLOT, src, bytes); _mm_sfence(); *tail = i; });
        const double c = per_call_ns(N, [&](int i) { memcpy(ring + (size_t)(i % R) * SLOT, src, bytes); });
        const double d = per_call_ns(N, [&](int i) {
            uint8_t* dst = ring + (size_t)(i % R) * SLOT;
            for (int o = 0; o < bytes; o += 32) _mm256_stream_si256((__m256i*)(dst + o), _mm256_loadu_si256((const __m256i*)(src + o)));
            _mm_sfence(); *tail = i; _mm_sfence(); });
        printf("slot %4d B: copy + sfence + tail + sfence %7.1f ns | without the last sfence %7.1f ns | copy only %7.1f ns | streaming stores, both fences %7.1f ns\n", bytes, a, b, c, d);
    }
    const double e = per_call_ns(N, [&](int i) { *tail = i; _mm_sfence(); });
    const double f = per_call_ns(N, [&](int i) { _mm_sfence(); });
    volatile uint64_t sink = 0;
    const double g = per_call_ns(N, [&](int i) { sink += (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count(); });
    printf("tail + sfence %7.1f ns | sfence alone %7.1f ns | steady_clock::now %7.1f ns\n", e, f, g);
    return 0;
}
