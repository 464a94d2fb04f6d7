// Can the HOST write device memory directly (large BAR)?  If it can, cvgs_queue_submit could place a batch's slot in the
// device ring itself instead of going through the pinned host ring + the feeder's copy.  Probes hipMalloc,
// hipExtMallocWithFlags(fine-grained / uncached) and hipMallocManaged pointers: host write (SIGSEGV-guarded), device read-back.
#include <hip/hip_runtime.h>
#include <setjmp.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <chrono>

static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

__global__ void sum_kernel(const uint32_t* p, int n, uint32_t* out) {
    uint32_t s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    atomicAdd(out, s);
}

static void probe(const char* name, void* p, size_t bytes) {
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    uint32_t* out;
    hipMalloc((void**)&out, 4);
    hipMemset(out, 0, 4);
    hipMemset(p, 0, bytes);
    hipDeviceSynchronize();
    if (sigsetjmp(jb, 1)) {
        printf("%-28s host write: FAULT (not host-accessible)\n", name);
        return;
    }
    const int n = (int)(bytes / 4);
    auto t0 = std::chrono::steady_clock::now();
    for (int rep = 0; rep < 64; ++rep)
        for (int i = 0; i < n; ++i) ((volatile uint32_t*)p)[i] = (uint32_t)(i + 1);
    _mm_sfence();
    auto t1 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, 0, (const uint32_t*)p, n, out);
    uint32_t got = 0;
    hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost);
    uint32_t want = 0;
    for (int i = 0; i < n; ++i) want += (uint32_t)(i + 1);
    const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / 64;
    printf("%-28s host write: ok, %.2f us per %zu-byte slot (%.2f GB/s), device sees %s\n", name, us, bytes, bytes / us / 1e3, got == want ? "the data" : "STALE/WRONG data");
}

#include <immintrin.h>
int main() {
    const size_t bytes = 4096;
    void* p;
    if (hipMalloc(&p, bytes) == hipSuccess) probe("hipMalloc", p, bytes);
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) == hipSuccess) probe("hipDeviceMallocFinegrained", p, bytes);
    else printf("hipDeviceMallocFinegrained: allocation failed\n");
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) == hipSuccess) probe("hipDeviceMallocUncached", p, bytes);
    else printf("hipDeviceMallocUncached: allocation failed\n");
    if (hipMallocManaged(&p, bytes) == hipSuccess) {
        hipMemAdvise(p, bytes, hipMemAdviseSetPreferredLocation, 0);
        probe("hipMallocManaged(pref dev)", p, bytes);
    }
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess) probe("hipHostMalloc (reference)", p, bytes);
    return 0;
}
