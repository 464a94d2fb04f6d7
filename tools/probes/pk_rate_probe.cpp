// Issue rate of v_pk_mul_f32 / v_pk_add_f32 against v_mul_f32 / v_cvt_f32_ubyte0 / v_perm_b32 on gfx950: a wave runs N
// iterations of 16 independent instructions of one kind; 8 waves per SIMD on every CU.  Prints cycles per instruction per
// SIMD (wall time x clock / instructions per SIMD): 4 = one pass of a 64-lane wave over 16 lanes.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pk_rate_probe.cpp -o /tmp/pk_rate && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int n, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[16];
    for (int i = 0; i < 16; ++i) a[i] = f2{seed + i, seed - i};
    f2 m = {1.0000001f, 0.9999999f};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (KIND == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if constexpr (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
            else if constexpr (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            else if constexpr (KIND == 3) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i].x));
            else if constexpr (KIND == 4) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(m.x));
            else if constexpr (KIND == 5) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a[i].x) : "v"(m.x));
            else if constexpr (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
            else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i].x) : "v"(m.x));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, float* out) {
    const int n = 4096, blocks = 256 * 8; // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 64, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, n, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); // kHz
    const double insts_per_simd = (double)blocks * 4 / (256.0 * 4) * n * 16;
    printf("%-18s %8.3f ms  %6.2f cycles per instruction per SIMD (at %d MHz)\n", name, ms, ms * 1e-3 * clk * 1e3 / insts_per_simd, clk / 1000);
}

int main() {
    float* out;
    hipMalloc((void**)&out, 4096);
    run<1>("v_mul_f32", out);
    run<0>("v_pk_mul_f32", out);
    run<2>("v_pk_add_f32", out);
    run<7>("v_fma_f32", out);
    run<6>("v_pk_fma_f32", out);
    run<3>("v_cvt_f32_ubyte0", out);
    run<4>("v_perm_b32", out);
    run<5>("v_cvt_pk_u8_f32", out);
    return 0;
}
