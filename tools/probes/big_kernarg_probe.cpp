// How large may a kernel-argument block be, and what does the host pay to launch with it?  (hipcc --offload-arch=gfx950 -O2 big_kernarg_probe.cpp)
// Each kernel sums one dword per 48-byte record of its argument block (wave-uniform scalar loads, as the engine's descriptor reads).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
template <int BYTES> struct Blk { uint32_t w[BYTES / 4]; };
template <int BYTES> __global__ void k(const Blk<BYTES> a, uint32_t* out) {
    uint32_t s = 0;
    for (int i = blockIdx.x; i < BYTES / 48; i += gridDim.x) s += a.w[i * 12];
    if (threadIdx.x == 0) atomicAdd(out, s);
}
template <int BYTES> static void run(hipStream_t st, uint32_t* d_out) {
    static Blk<BYTES> a;
    uint32_t want = 0;
    for (int i = 0; i < BYTES / 4; ++i) a.w[i] = (uint32_t)i * 2654435761u;
    for (int i = 0; i < BYTES / 48; ++i) want += a.w[i * 12];
    hipMemsetAsync(d_out, 0, 4, st);
    hipLaunchKernelGGL((k<BYTES>), dim3(64), dim3(64), 0, st, a, d_out);
    hipError_t e = hipGetLastError();
    uint32_t got = 0;
    hipStreamSynchronize(st);
    hipMemcpy(&got, d_out, 4, hipMemcpyDeviceToHost);
    const int N = 2000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL((k<BYTES>), dim3(64), dim3(64), 0, st, a, d_out);
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(st);
    auto t2 = std::chrono::steady_clock::now();
    std::printf("kernarg %6d B: launch %s, sum %s, host %.2f us per launch, %.2f us per launch incl. drain\n", BYTES, hipGetErrorString(e),
                got == want ? "ok" : "WRONG", std::chrono::duration<double, std::micro>(t1 - t0).count() / N,
                std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
}
int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    uint32_t* d_out;
    hipMalloc(&d_out, 4);
    run<4032>(st, d_out);
    run<16320>(st, d_out);
    run<32640>(st, d_out);
    run<52800>(st, d_out);
    run<65280>(st, d_out);
    run<130560>(st, d_out);
    return 0;
}
