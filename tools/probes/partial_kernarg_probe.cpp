// Can a kernel whose argument block is DECLARED large be launched with only the front of the block supplied (hipModuleLaunchKernel +
// HIP_LAUNCH_PARAM_BUFFER_SIZE < kernarg segment size), and does the host then pay for the bytes supplied only?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Blk { uint32_t n; uint32_t pad[3]; uint32_t w[16320]; uint32_t* out; }; // 65,304 B declared
__global__ void k(const Blk a) {
    uint32_t s = 0;
    for (uint32_t i = blockIdx.x; i < a.n; i += gridDim.x) s += a.w[i * 12];
    if (threadIdx.x == 0) atomicAdd(*(uint32_t* const*)&a.w[a.n * 12], s); // the out pointer travels right behind the used records
}
int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    uint32_t* d_out;
    hipMalloc(&d_out, 4);
    hipFunction_t f;
    hipError_t e = hipGetFuncBySymbol(&f, (const void*)k);
    std::printf("hipGetFuncBySymbol: %s\n", hipGetErrorString(e)); std::fflush(stdout);
    static Blk a;
    for (int full = 1; full >= 0; --full) for (uint32_t recs : {4u, 50u, 320u, 800u, 1300u}) {
        a.n = recs;
        uint32_t want = 0;
        for (uint32_t i = 0; i < recs * 12; ++i) a.w[i] = i * 2654435761u;
        for (uint32_t i = 0; i < recs; ++i) want += a.w[i * 12];
        *(uint32_t**)&a.w[recs * 12] = d_out;
        size_t size = full ? sizeof(Blk) : 16 + (size_t)recs * 48 + 8;
        std::printf("launching with %zu bytes\n", size); std::fflush(stdout);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
        hipMemsetAsync(d_out, 0, 4, st);
        e = hipModuleLaunchKernel(f, 64, 1, 1, 64, 1, 1, 0, st, nullptr, extra);
        hipStreamSynchronize(st);
        uint32_t got = 0;
        hipMemcpy(&got, d_out, 4, hipMemcpyDeviceToHost);
        const int N = 2000;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) hipModuleLaunchKernel(f, 64, 1, 1, 64, 1, 1, 0, st, nullptr, extra);
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(st);
        std::printf("%5u records, %6zu B supplied of %zu declared: launch %s, sum %s, host %.2f us per launch\n", recs, size, sizeof(Blk), hipGetErrorString(e),
                    got == want ? "ok" : "WRONG", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
    return 0;
}
