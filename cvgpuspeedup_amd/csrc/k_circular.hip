// k_circular.hip -- K9, the CircularTensor shift: plane-to-plane copies from the history ring into
// the ordered output tensor ("pure bandwidth").  One launch moves every OLD plane of the
// tensor (the new frame is written to both places by the chain kernel): blockIdx.y selects the (src,dst)
// job, blockIdx.x strides over the plane with non-temporal 16-byte accesses, eight loads in flight per
// lane before the first store.
#include <cstdio>
#include <cstdlib>

#include "k_pointwise_body.hpp"

namespace cvgs {

struct CopyArgs {
    CopyJob jobs[kMaxCopyJobs];
};

// Single-launch CircularTensor update for per-pixel u8 pushes (the form the reference tests,
// include/cvGPUSpeedup.cuh:612-617: one kernel computes the new plane and shifts the others).  Workgroups
// [0, pw_blocks) run the thread-fused pointwise chain on the new frame (written to its tensor slot AND its ring slot),
// the rest stream the BATCH-1 older frames ring -> tensor exactly like k_plane_copy.
struct PushGeom {
    uint32_t pw_blocks, col_groups; // compute part: pw_blocks = col_groups * row_groups
    uint32_t blocks_per_job, n_jobs;
    size_t n_vec;                   // 16-byte vectors per plane
};

// DEV (capturable handles, CVGS_CIRCULAR_CAPTURABLE): nothing that depends on the update count comes with the kernel arguments --
// the kernel reads the count and derives the new frame's ring slot (both slots of a mirrored ring) and every copy job itself,
// so a captured update replays as the NEXT update with the traffic of an eager one (no staging image).
template <int CN, class Prog, typename OT, bool DEV = false>
__global__ __launch_bounds__(256) void k_circular_push(const KernArgs<1> a, const PwGeom g, const CopyArgs jobs, const PushGeom pg, const CircDev d) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    int64_t count = 0;
    if constexpr (DEV) count = (int64_t)*d.count; // updates completed before this one (wave-uniform)
    if (blockIdx.x < pg.pw_blocks) {
        const uint32_t by = blockIdx.x / pg.col_groups, bx = blockIdx.x - by * pg.col_groups;
        if constexpr (DEV) {
            PwGeom g2 = g;
            const size_t image_bytes = d.plane_bytes * (size_t)d.color_planes;
            const int64_t km = count % d.batch;
            if (d.mirrored) {
                const int64_t p = d.order == CVGS_NEWEST_FIRST ? d.batch - 1 - km : km;
                g2.out = d.ring + (size_t)p * image_bytes;
                g2.out2 = d.ring + (size_t)(p + d.batch) * image_bytes;
            } else {
                g2.out2 = d.ring + (size_t)km * image_bytes; // (g.out: the new frame's tensor slot, the same for every update)
            }
            pw4_body<CN, Prog, OT>(a.c, a.planes[0], g2, (int)bx, (int)by, 0);
        } else {
            pw4_body<CN, Prog, OT>(a.c, a.planes[0], g, (int)bx, (int)by, 0);
        }
        return;
    }
    const uint32_t b = blockIdx.x - pg.pw_blocks;
    const uint32_t j = b / pg.blocks_per_job, bx = b - j * pg.blocks_per_job;
    CopyJob job;
    if constexpr (DEV) { // job j: plane c of the frame of age >= 1 shown at tensor slot z (k_circular_dev's derivation)
        const int B = d.batch, CP = d.color_planes;
        const int c = (int)j % CP, zi = (int)j / CP;
        const int z = d.order == CVGS_NEWEST_FIRST ? zi + 1 : zi; // every slot but the new frame's (0 / B - 1)
        const int64_t age = d.order == CVGS_NEWEST_FIRST ? z : B - 1 - z;
        int64_t slot = (count - age) % B;
        if (slot < 0) slot += B; // never-written history slots hold zeros
        job.src = d.ring + ((size_t)slot * CP + c) * d.plane_bytes;
        job.dst = d.transposed ? d.out + ((size_t)c * B + z) * d.plane_bytes : d.out + ((size_t)z * CP + c) * d.plane_bytes;
    } else {
        job = jobs.jobs[j];
    }
    const v4* __restrict__ src = (const v4*)job.src;
    v4* __restrict__ dst = (v4*)job.dst;
    const size_t stride = (size_t)pg.blocks_per_job * 256;
    size_t i = (size_t)bx * 256 + threadIdx.x;
    constexpr int UNROLL = 8;
    for (; i + (UNROLL - 1) * stride < pg.n_vec; i += UNROLL * stride) {
        v4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
    }
    for (; i < pg.n_vec; i += stride) dst[i] = src[i];
}

template <int CN, typename OT>
static hipError_t launch_push_prog(int prog_id, const KernArgs<1>& a, const PwGeom& g, const CopyArgs& jobs, const PushGeom& pg,
                                   const CircDev* dev, hipStream_t s) {
    const dim3 grid(pg.pw_blocks + pg.blocks_per_job * pg.n_jobs);
    if (dev) {
        if (prog_id == 0) hipLaunchKernelGGL((k_circular_push<CN, ProgCastMulSubDiv, OT, true>), grid, dim3(256), 0, s, a, g, jobs, pg, *dev);
        else if (prog_id == 1) hipLaunchKernelGGL((k_circular_push<CN, ProgCast, OT, true>), grid, dim3(256), 0, s, a, g, jobs, pg, *dev);
        else hipLaunchKernelGGL((k_circular_push<CN, InterpProg, OT, true>), grid, dim3(256), 0, s, a, g, jobs, pg, *dev);
        return hipGetLastError();
    }
    const CircDev none{};
    if (prog_id == 0) hipLaunchKernelGGL((k_circular_push<CN, ProgCastMulSubDiv, OT>), grid, dim3(256), 0, s, a, g, jobs, pg, none);
    else if (prog_id == 1) hipLaunchKernelGGL((k_circular_push<CN, ProgCast, OT>), grid, dim3(256), 0, s, a, g, jobs, pg, none);
    else hipLaunchKernelGGL((k_circular_push<CN, InterpProg, OT>), grid, dim3(256), 0, s, a, g, jobs, pg, none);
    return hipGetLastError();
}

// Returns 1 if it took the update (new frame + all copies in ONE launch), 0 if not eligible, <0 on error.
// `dev` (capturable handles): the kernel derives the count-dependent destinations and the n_jobs copy jobs itself (copy_jobs unused;
// n_jobs = (BATCH - 1) * planes, or 0 for a mirrored ring); the caller advances the device-side count behind it.
int launch_circular_push(const ChainArgs& c_in, const PlaneParams& plane, const CopyJob* copy_jobs, int n_jobs, size_t plane_bytes,
                         uint32_t chain_flags, void* stream, const CircDev* dev) {
    ChainArgs c;
    PwGeom g;
    int prog_id = 0;
    bool f16 = false;
    if (!pointwise4_plan(c_in, 1, chain_flags, c, g, prog_id, f16)) return 0;
    if (prog_id == 3) return 0; // non-u8 sources: chain kernel + copy kernel
    if (n_jobs < (dev ? 0 : 1) || n_jobs > kMaxCopyJobs || plane_bytes % 16) return 0;
    if (dev) {
        if ((((uintptr_t)dev->out | (uintptr_t)dev->ring) & 15) != 0) return 0;
    } else {
        for (int i = 0; i < n_jobs; ++i)
            if ((((uintptr_t)copy_jobs[i].src | (uintptr_t)copy_jobs[i].dst) & 15) != 0) return 0;
    }
    KernArgs<1> a;
    a.c = c;
    a.planes[0] = plane;
    CopyArgs jobs;
    for (int i = 0; i < kMaxCopyJobs; ++i) jobs.jobs[i] = (!dev && i < n_jobs) ? copy_jobs[i] : CopyJob{nullptr, nullptr};
    PushGeom pg;
    pg.col_groups = (uint32_t)((g.w + 255) / 256);
    pg.pw_blocks = pg.col_groups * (uint32_t)((g.h + 3) / 4);
    pg.n_jobs = (uint32_t)n_jobs;
    pg.n_vec = plane_bytes / 16;
    const size_t want = (pg.n_vec + 256 * 8 - 1) / (256 * 8);
    const size_t cap = (size_t)(n_jobs > 0 && 8192 / n_jobs > 1 ? 8192 / n_jobs : 1);
    pg.blocks_per_job = (uint32_t)(want < cap ? want : cap);
    if (pg.blocks_per_job < 1) pg.blocks_per_job = 1;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (f16) {
        switch (c.read.cn) {
        case 1: e = launch_push_prog<1, _Float16>(prog_id, a, g, jobs, pg, dev, s); break;
        case 2: e = launch_push_prog<2, _Float16>(prog_id, a, g, jobs, pg, dev, s); break;
        case 3: e = launch_push_prog<3, _Float16>(prog_id, a, g, jobs, pg, dev, s); break;
        default: e = launch_push_prog<4, _Float16>(prog_id, a, g, jobs, pg, dev, s); break;
        }
    } else {
        switch (c.read.cn) {
        case 1: e = launch_push_prog<1, float>(prog_id, a, g, jobs, pg, dev, s); break;
        case 2: e = launch_push_prog<2, float>(prog_id, a, g, jobs, pg, dev, s); break;
        case 3: e = launch_push_prog<3, float>(prog_id, a, g, jobs, pg, dev, s); break;
        default: e = launch_push_prog<4, float>(prog_id, a, g, jobs, pg, dev, s); break;
        }
    }
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

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

// ---- device-indexed update (CVGS_CIRCULAR_CAPTURABLE) ----------------------------------------------------------------------
// The reference's update is an ordinary stream launch (include/cvGPUSpeedup.cuh:612-622), so a serving loop may capture it into a
// graph.  The default path resolves the ring slot on the HOST (kernel arguments hold the resolved pointers): replaying such a
// capture would write the same slots for ever.  Here the update count lives in DEVICE memory: the push chain writes the new
// frame into a fixed staging image, and this kernel -- every plane job derived from *count -- moves it to its tensor slot and
// its history slot and shifts the older frames; k_circular_bump then advances the count.  Nothing depends on host state, so
// N captured updates replay as the next N updates.  Costs one extra pass over ONE image (+6 % of cfg #4's traffic).
template <typename V, int UNROLL>
__global__ __launch_bounds__(256) void k_circular_dev(const CircDev a, const size_t n_vec) {
    const int64_t count = (int64_t)*a.count; // updates completed before this one (wave-uniform)
    const int B = a.batch, CP = a.color_planes;
    const int job = (int)blockIdx.y;
    const uint8_t* src;
    uint8_t* dst;
    if (a.mirrored) { // 2 * CP jobs: staging plane c -> ring slots p and p + B
        const int c = job % CP, half = job / CP;
        const int64_t km = count % B;
        const int64_t p = a.order == CVGS_NEWEST_FIRST ? B - 1 - km : km;
        src = a.stage + (size_t)c * a.plane_bytes;
        dst = a.ring + ((size_t)(p + (half ? B : 0)) * CP + c) * a.plane_bytes;
    } else if (job >= B * CP) { // CP jobs: staging plane c -> history slot count % B
        const int c = job - B * CP;
        src = a.stage + (size_t)c * a.plane_bytes;
        dst = a.ring + ((size_t)(count % B) * CP + c) * a.plane_bytes;
    } else { // B * CP jobs: tensor slot z, plane c
        const int z = job / CP, c = job % CP;
        const int z_new = a.order == CVGS_NEWEST_FIRST ? 0 : B - 1;
        if (z == z_new) {
            src = a.stage + (size_t)c * a.plane_bytes;
        } else {
            const int64_t age = a.order == CVGS_NEWEST_FIRST ? z : B - 1 - z;
            int64_t slot = (count - age) % B;
            if (slot < 0) slot += B; // never-written history slots hold zeros
            src = a.ring + ((size_t)slot * CP + c) * a.plane_bytes;
        }
        dst = a.transposed ? a.out + ((size_t)c * B + z) * a.plane_bytes : a.out + ((size_t)z * CP + c) * a.plane_bytes;
    }
    const V* __restrict__ sp = (const V*)src;
    V* __restrict__ dp = (V*)dst;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n_vec; i += UNROLL * stride) {
        V v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(sp + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) __builtin_nontemporal_store(v[u], dp + i + u * stride);
    }
    for (; i < n_vec; i += stride) dp[i] = sp[i];
}
__global__ void k_circular_bump(uint64_t* count) {
    if (threadIdx.x == 0) *count += 1;
}

int launch_circular_bump(const uint64_t* count, void* stream) {
    hipLaunchKernelGGL(k_circular_bump, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint64_t*)count);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_circular_dev(const CircDev& a, void* stream) {
    const int n_jobs = a.mirrored ? 2 * a.color_planes : (a.batch + 1) * a.color_planes;
    if (n_jobs < 1 || n_jobs > 65535) return -1;
    hipStream_t s = (hipStream_t)stream;
    auto blocks_for = [&](size_t n_vec, int unroll) {
        size_t want = (n_vec + 256 * (size_t)unroll - 1) / (256 * (size_t)unroll);
        size_t cap = (size_t)(8192 / n_jobs > 1 ? 8192 / n_jobs : 1);
        size_t b = want < cap ? want : cap;
        return (unsigned)(b < 1 ? 1 : b);
    };
    typedef float v4 __attribute__((ext_vector_type(4)));
    if (a.plane_bytes % 16 == 0) hipLaunchKernelGGL((k_circular_dev<v4, 8>), dim3(blocks_for(a.plane_bytes / 16, 8), n_jobs), dim3(256), 0, s, a, a.plane_bytes / 16);
    else if (a.plane_bytes % 4 == 0) hipLaunchKernelGGL((k_circular_dev<uint32_t, 4>), dim3(blocks_for(a.plane_bytes / 4, 4), n_jobs), dim3(256), 0, s, a, a.plane_bytes / 4);
    else hipLaunchKernelGGL((k_circular_dev<uint8_t, 4>), dim3(blocks_for(a.plane_bytes, 4), n_jobs), dim3(256), 0, s, a, a.plane_bytes);
    if (hipGetLastError() != hipSuccess) return -1;
    hipLaunchKernelGGL(k_circular_bump, dim3(1), dim3(64), 0, s, (uint64_t*)a.count);
    return hipGetLastError() == hipSuccess ? 0 : -1;
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
    // at 4096 workgroups x 4 loads; round 1 sweep)
    auto blocks_for = [&](size_t n_vec, int unroll) {
        size_t want = (n_vec + 256 * (size_t)unroll - 1) / (256 * (size_t)unroll);
        size_t cap = (size_t)(8192 / n_jobs > 1 ? 8192 / n_jobs : 1);
        size_t b = want < cap ? want : cap;
        return (unsigned)(b < 1 ? 1 : b);
    };
    if (al16) {
        const size_t n = bytes / 16;
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
