// sat_probe.cpp -- exhaustive check (all 2^32 float bit patterns) of cheaper instruction sequences for the final
// SaturateCast of a store (float -> u8 / u16 / s16: round to nearest even, clamp, NaN -> 0) against the engine's reference
// sequence sat_round() (k_common.hpp: v_rndne, +0, NaN select, max, min, v_cvt_i32 -- 7 VALU instructions per channel):
//   A  v_cvt_pk_u8_f32                                  (1 instruction, packs the byte as well)
//   B  v_rndne_f32, v_cvt_u32_f32, v_min_u32            (the hardware conversion saturates and maps NaN to 0)
//   C  v_rndne_f32, v_cvt_i32_f32, v_med3_i32           (signed 16-bit)
//   hipcc -O2 --offload-arch=gfx950 tools/sat_probe.cpp -o tools/bin/sat_probe && tools/bin/sat_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

__device__ __forceinline__ float sat_round_ref(float v, float lo, float hi) {
    float r = rintf(v) + 0.0f;
    r = (v != v) ? 0.f : r;
    return fminf(fmaxf(r, lo), hi);
}
__device__ __forceinline__ uint32_t cvt_u32_hw(float v) {
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ int32_t cvt_i32_hw(float v) {
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

struct Counts {
    unsigned long long a_u8, b_u8, b_u16, c_s16, first[4], control;
};

__global__ void probe(Counts* out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long ma = 0, mb = 0, mb16 = 0, mc = 0, md = 0;
    for (uint64_t bits = tid; bits < (1ull << 32); bits += stride) {
        const float v = __uint_as_float((uint32_t)bits);
        const uint32_t ref8 = (uint32_t)sat_round_ref(v, 0.f, 255.f);
        const uint32_t ref16 = (uint32_t)sat_round_ref(v, 0.f, 65535.f);
        const int32_t refs16 = (int32_t)sat_round_ref(v, -32768.f, 32767.f);
        const uint32_t a = __builtin_amdgcn_cvt_pk_u8_f32(v, 0, 0);
        const float r = rintf(v);
        const uint32_t u = cvt_u32_hw(r);
        const uint32_t b8 = u < 255u ? u : 255u, b16 = u < 65535u ? u : 65535u;
        int32_t i = cvt_i32_hw(r);
        i = i < -32768 ? -32768 : (i > 32767 ? 32767 : i);
        md += (uint32_t)fminf(fmaxf(truncf(v == v ? v : 0.f), 0.f), 255.f) != ref8; // control: truncation must differ
        if (a != ref8 && !ma++) atomicMin(&out->first[0], (unsigned long long)bits);
        if (b8 != ref8 && !mb++) atomicMin(&out->first[1], (unsigned long long)bits);
        if (b16 != ref16 && !mb16++) atomicMin(&out->first[2], (unsigned long long)bits);
        if (i != refs16 && !mc++) atomicMin(&out->first[3], (unsigned long long)bits);
    }
    atomicAdd(&out->a_u8, ma);
    atomicAdd(&out->b_u8, mb);
    atomicAdd(&out->b_u16, mb16);
    atomicAdd(&out->c_s16, mc);
    atomicAdd(&out->control, md);
}

int main() {
    Counts* d;
    CK(hipMalloc(&d, sizeof(Counts)));
    Counts h{0, 0, 0, 0, {~0ull, ~0ull, ~0ull, ~0ull}, 0};
    CK(hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, d);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
    std::printf("mismatches over all 2^32 bit patterns, first mismatching bit pattern:\n");
    std::printf("A v_cvt_pk_u8_f32            vs sat_round u8 : %llu  first 0x%08llx\n", h.a_u8, h.first[0]);
    std::printf("B rndne+cvt_u32+min 255      vs sat_round u8 : %llu  first 0x%08llx\n", h.b_u8, h.first[1]);
    std::printf("B rndne+cvt_u32+min 65535    vs sat_round u16: %llu  first 0x%08llx\n", h.b_u16, h.first[2]);
    std::printf("C rndne+cvt_i32+clamp        vs sat_round s16: %llu  first 0x%08llx\n", h.c_s16, h.first[3]);
    std::printf("control (truncation instead of rounding) vs sat_round u8: %llu mismatches (must be > 0)\n", h.control);
    return 0;
}
