// k_circular.hip -- K9, the CircularTensor shift: plane-to-plane copies from the history ring into
// the ordered output tensor ("pure bandwidth").  One launch moves every OLD plane of the
// tensor (the new frame is written to both places by the chain kernel): blockIdx.y selects the (src,dst)
// job, blockIdx.x strides over the plane with non-temporal 16-byte accesses, eight loads in flight per
// lane before the first store.
#include <cstdio>
#include <cstdlib>

#include "k_common.hpp"

namespace cvgs {

struct CopyArgs {
    CopyJob jobs[kMaxCopyJobs];
};

template <typename V, int UNROLL, bool NT = true>
__global__ __launch_bounds__(256) void k_plane_copy(const CopyArgs a, const size_t n_vec) {
    const CopyJob job = a.jobs[blockIdx.y];
    const V* __restrict__ src = (const V*)job.src;
    V* __restrict__ dst = (V*)job.dst;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n_vec; i += UNROLL * stride) {
        V v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if constexpr (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n_vec; i += stride) dst[i] = src[i];
}

typedef float vec4f __attribute__((ext_vector_type(4)));

int launch_plane_copies(const CopyJob* jobs, int n_jobs, size_t bytes, void* stream) {
    if (n_jobs < 1 || n_jobs > kMaxCopyJobs) return -1;
    CopyArgs a;
    for (int i = 0; i < n_jobs; ++i) a.jobs[i] = jobs[i];
    for (int i = n_jobs; i < kMaxCopyJobs; ++i) a.jobs[i] = CopyJob{nullptr, nullptr};
    bool al16 = bytes % 16 == 0, al4 = bytes % 4 == 0;
    for (int i = 0; i < n_jobs; ++i) {
        const uintptr_t m = (uintptr_t)jobs[i].src | (uintptr_t)jobs[i].dst;
        al16 = al16 && (m % 16 == 0);
        al4 = al4 && (m % 4 == 0);
    }
    hipStream_t s = (hipStream_t)stream;
    // ~8192 workgroups in total, 8 x 16-byte loads in flight per lane (measured best on cfg #4: 124 us vs 136 us
    // at 4096 workgroups x 4 loads; tools/bench_more.py with CVGS_COPY_TUNE)
    auto blocks_for = [&](size_t n_vec, int unroll) {
        size_t want = (n_vec + 256 * (size_t)unroll - 1) / (256 * (size_t)unroll);
        size_t cap = (size_t)(8192 / n_jobs > 1 ? 8192 / n_jobs : 1);
        size_t b = want < cap ? want : cap;
        return (unsigned)(b < 1 ? 1 : b);
    };
    if (al16) {
        const size_t n = bytes / 16;
        // tuning hook (benchmarks only): CVGS_COPY_TUNE=<variant>,<total workgroups>
        static const char* tune = getenv("CVGS_COPY_TUNE");
        if (tune) {
            int variant = 0, total = 4096;
            sscanf(tune, "%d,%d", &variant, &total);
            auto blocks = [&](int unroll) {
                size_t want = (n + 256 * (size_t)unroll - 1) / (256 * (size_t)unroll);
                size_t cap = (size_t)(total / n_jobs > 1 ? total / n_jobs : 1);
                return (unsigned)(want < cap ? want : cap);
            };
            switch (variant) {
            case 1: hipLaunchKernelGGL((k_plane_copy<vec4f, 8>), dim3(blocks(8), n_jobs), dim3(256), 0, s, a, n); break;
            case 2: hipLaunchKernelGGL((k_plane_copy<vec4f, 4, false>), dim3(blocks(4), n_jobs), dim3(256), 0, s, a, n); break;
            case 3: hipLaunchKernelGGL((k_plane_copy<vec4f, 2>), dim3(blocks(2), n_jobs), dim3(256), 0, s, a, n); break;
            case 4: hipLaunchKernelGGL((k_plane_copy<vec4f, 1>), dim3(blocks(1), n_jobs), dim3(256), 0, s, a, n); break;
            default: hipLaunchKernelGGL((k_plane_copy<vec4f, 4>), dim3(blocks(4), n_jobs), dim3(256), 0, s, a, n); break;
            }
            return hipGetLastError() == hipSuccess ? 0 : -1;
        }
        hipLaunchKernelGGL((k_plane_copy<vec4f, 8>), dim3(blocks_for(n, 8), n_jobs), dim3(256), 0, s, a, n);
    } else if (al4) {
        const size_t n = bytes / 4;
        hipLaunchKernelGGL((k_plane_copy<uint32_t, 4>), dim3(blocks_for(n, 4), n_jobs), dim3(256), 0, s, a, n);
    } else {
        hipLaunchKernelGGL((k_plane_copy<uint8_t, 4>), dim3(blocks_for(bytes, 4), n_jobs), dim3(256), 0, s, a, bytes);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // namespace cvgs
