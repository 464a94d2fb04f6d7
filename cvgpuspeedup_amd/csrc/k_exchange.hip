// k_exchange.hip -- device-side arrival flags for the sharded batched-crop path (BASELINE cfg #5, SURVEY.md 8e option 2).
//
// The P2P fused write stores every rank's rows of the [N,C,H,W] tensor straight into every peer's copy (k1_resize_split's MIR
// instantiations, cvgs_write_desc.mirrors).  What was left on the host was the "all rows have landed" barrier: one RCCL
// all-reduce per step issued from Python, 32 us per step against 5 us of compute.  Here the barrier is two tiny
// kernels on the producing stream and flags in the tensors' own allocations (which every peer has IPC-mapped already):
//   cvgs_exchange_signal  after the step's K1 launch: writes the step number into THIS rank's word of every peer's flag block
//                         (system-scope stores over xGMI).  The kernel boundary in front of it is the release: K1's stores to
//                         the peers are complete and visible before any flag is.
//   cvgs_exchange_wait    makes the stream wait until every listed flag word has reached a step number (one wave polls with
//                         system-scope loads; a watchdog gives up after `timeout_ms` and reports instead of hanging the box --
//                         and once err[0] is set every later wait behind the same error words returns at once: a lost peer
//                         costs ONE timeout, not one per step of a replayed graph).
// No collective, no host round trip; the waits may lag the signals by a few steps so that they never block a well-fed stream.
// The reference has no multi-GPU code (include/cvGPUSpeedup.cuh:605-610 is its only device selector).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cvgs_device.h"

namespace cvgs {

struct XPtrs {
    uint64_t* p[CVGS_MAX_EXCHANGE_PEERS];
    int32_t n, pad;
};

typedef __attribute__((address_space(1))) uint64_t* xg_u64;

// counter != nullptr: the step number lives on the device (graph replays: a captured constant would repeat) -- the kernel
// advances *counter and publishes the new count
__global__ void k_exchange_signal(const XPtrs peers, uint64_t value, uint64_t* counter) {
    const int i = (int)threadIdx.x;
    if (counter) {
        value = *counter + 1; // every lane reads the old count before lane 0 stores the new one (one wave, in order)
        __builtin_amdgcn_wave_barrier();
        if (i == 0) *counter = value;
    }
    if (i < peers.n) __hip_atomic_store((xg_u64)peers.p[i], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// err[0] = 1 when the watchdog expired (a flag that was still behind in err[1]).  counter != nullptr: wait for *counter - lag.
__global__ void k_exchange_wait(const XPtrs flags, uint64_t value, const uint64_t* counter, const uint64_t lag, const uint64_t timeout_ticks,
                                uint64_t* err) {
    const int i = (int)threadIdx.x;
    if (counter) {
        const uint64_t c = *counter;
        if (c <= lag) return; // nothing that old has been signalled yet
        value = c - lag;
    }
    if (err && __hip_atomic_load((xg_u64)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return; // a peer was lost earlier: fail fast
    const uint64_t t0 = wall_clock64();
    for (;;) {
        const bool behind = i < flags.n && __hip_atomic_load((xg_u64)flags.p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < value;
        const uint64_t mask = __builtin_amdgcn_ballot_w64(behind);
        if (mask == 0) break;
        if (wall_clock64() - t0 > timeout_ticks) {
            if (i == 0 && err) {
                __hip_atomic_store((xg_u64)err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store((xg_u64)(err + 1), (uint64_t)__builtin_ctzll(mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            break;
        }
        __builtin_amdgcn_s_sleep(16);
    }
    // the arrival is an acquire for whatever runs next on this stream: the kernel boundary behind this kernel invalidates the caches
}

// signal + lagged wait in ONE launch per step (the step counter on the device): what a steady exchange loop enqueues behind each K1
__global__ void k_exchange_step(const XPtrs peers, const XPtrs flags, uint64_t* counter, const uint64_t lag, const uint64_t timeout_ticks, uint64_t* err) {
    const int i = (int)threadIdx.x;
    const uint64_t value = *counter + 1;
    __builtin_amdgcn_wave_barrier();
    if (i == 0) *counter = value;
    if (i < peers.n) __hip_atomic_store((xg_u64)peers.p[i], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (value <= lag) return;
    if (err && __hip_atomic_load((xg_u64)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return; // a peer was lost earlier: fail fast
    const uint64_t want = value - lag;
    const uint64_t t0 = wall_clock64();
    for (;;) {
        const bool behind = i < flags.n && __hip_atomic_load((xg_u64)flags.p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want;
        const uint64_t mask = __builtin_amdgcn_ballot_w64(behind);
        if (mask == 0) break;
        if (wall_clock64() - t0 > timeout_ticks) {
            if (i == 0 && err) {
                __hip_atomic_store((xg_u64)err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store((xg_u64)(err + 1), (uint64_t)__builtin_ctzll(mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            break;
        }
        __builtin_amdgcn_s_sleep(16);
    }
}

int launch_exchange_step(void* const* peer_flags, const void* const* own_flags, int n, uint64_t* counter, uint64_t lag, double timeout_ms, void* err_words,
                         void* stream) {
    if (n < 0 || n > CVGS_MAX_EXCHANGE_PEERS || !counter) return -1;
    if (n == 0) return 0; // no peers: nothing to tell, nothing to wait for
    XPtrs a{}, b{};
    a.n = b.n = n;
    for (int i = 0; i < n; ++i) {
        a.p[i] = (uint64_t*)peer_flags[i];
        b.p[i] = (uint64_t*)own_flags[i];
    }
    const uint64_t ticks = (uint64_t)((timeout_ms <= 0 ? 2000.0 : timeout_ms) * 1e-3 * 100e6);
    hipLaunchKernelGGL(k_exchange_step, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, counter, lag, ticks, (uint64_t*)err_words);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_exchange_signal(void* const* peer_flags, int n, uint64_t value, uint64_t* counter, void* stream) {
    if (n < 0 || n > CVGS_MAX_EXCHANGE_PEERS) return -1;
    if (n == 0 && !counter) return 0;
    XPtrs a{};
    a.n = n;
    for (int i = 0; i < n; ++i) a.p[i] = (uint64_t*)peer_flags[i];
    hipLaunchKernelGGL(k_exchange_signal, dim3(1), dim3(64), 0, (hipStream_t)stream, a, value, counter);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_exchange_wait(const void* const* flags, int n, uint64_t value, const uint64_t* counter, uint64_t lag, double timeout_ms, void* err_words,
                         void* stream) {
    if (n < 0 || n > CVGS_MAX_EXCHANGE_PEERS) return -1;
    if (n == 0) return 0;
    XPtrs a{};
    a.n = n;
    for (int i = 0; i < n; ++i) a.p[i] = (uint64_t*)flags[i];
    const uint64_t ticks = (uint64_t)((timeout_ms <= 0 ? 2000.0 : timeout_ms) * 1e-3 * 100e6);
    hipLaunchKernelGGL(k_exchange_wait, dim3(1), dim3(64), 0, (hipStream_t)stream, a, value, counter, lag, ticks, (uint64_t*)err_words);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // namespace cvgs
