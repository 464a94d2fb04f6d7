// cvgs_rccl.cpp -- libcvgs_rccl.so: thin C-ABI over RCCL for the in-place all-gather that assembles the sharded
// crop tensor (include/cvgs_rccl.h).  Links librccl.so.1 (the same soname torch bundles, so one RCCL per process).
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/cvgs_rccl.h"

namespace {
thread_local std::string g_err;
int fail(const std::string& m) {
    g_err = m;
    return -5; // CVGS_ERR_RCCL
}
int check(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return 0;
    return fail(std::string(what) + ": " + ncclGetErrorString(r));
}
} // namespace

struct cvgs_comm_s {
    ncclComm_t comm = nullptr;
    int rank = 0, size = 1;
};

extern "C" {

const char* cvgs_rccl_last_error(void) { return g_err.c_str(); }

int cvgs_comm_unique_id(void* id_out) {
    static_assert(sizeof(ncclUniqueId) == CVGS_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!id_out) return fail("null id buffer");
    return check(ncclGetUniqueId((ncclUniqueId*)id_out), "ncclGetUniqueId");
}

int cvgs_comm_init_rank(cvgs_comm_t* out, int32_t n_ranks, int32_t rank, const void* id) {
    if (!out || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail("bad arguments");
    cvgs_comm_s* c = new cvgs_comm_s;
    ncclUniqueId uid = *(const ncclUniqueId*)id;
    int rc = check(ncclCommInitRank(&c->comm, n_ranks, uid, rank), "ncclCommInitRank");
    if (rc) {
        delete c;
        return rc;
    }
    c->rank = rank;
    c->size = n_ranks;
    *out = c;
    return 0;
}

int cvgs_comm_init_all(cvgs_comm_t* out, int32_t n_devices, const int32_t* devices) {
    if (!out || n_devices < 1) return fail("bad arguments");
    std::vector<ncclComm_t> comms((size_t)n_devices);
    int rc = check(ncclCommInitAll(comms.data(), n_devices, (const int*)devices), "ncclCommInitAll");
    if (rc) return rc;
    for (int i = 0; i < n_devices; ++i) {
        cvgs_comm_s* c = new cvgs_comm_s;
        c->comm = comms[(size_t)i];
        c->rank = i;
        c->size = n_devices;
        out[i] = c;
    }
    return 0;
}

int32_t cvgs_comm_rank(cvgs_comm_t c) { return c ? c->rank : -1; }
int32_t cvgs_comm_size(cvgs_comm_t c) { return c ? c->size : -1; }

int cvgs_allgather_inplace(cvgs_comm_t c, void* full, size_t bytes_per_rank, void* stream) {
    if (!c || !full) return fail("bad arguments");
    const char* send = (const char*)full + (size_t)c->rank * bytes_per_rank;
    return check(ncclAllGather(send, full, bytes_per_rank, ncclChar, c->comm, (hipStream_t)stream), "ncclAllGather");
}

int cvgs_group_start(void) { return check(ncclGroupStart(), "ncclGroupStart"); }
int cvgs_group_end(void) { return check(ncclGroupEnd(), "ncclGroupEnd"); }

// ---- P2P fused write: making the peers' tensors addressable ------------------------------------------------------------
static int hip_check(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return -3; // CVGS_ERR_HIP
}

int cvgs_ipc_alloc(void** dev_ptr, size_t bytes) {
    if (!dev_ptr || !bytes) return fail("bad arguments");
    int rc = hip_check(hipMalloc(dev_ptr, bytes), "hipMalloc");
    if (rc) return rc;
    return hip_check(hipMemset(*dev_ptr, 0, bytes), "hipMemset");
}
int cvgs_ipc_free(void* dev_ptr) { return hip_check(hipFree(dev_ptr), "hipFree"); }

int cvgs_ipc_export(const void* dev_ptr, void* handle_out) {
    static_assert(sizeof(hipIpcMemHandle_t) == CVGS_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
    if (!dev_ptr || !handle_out) return fail("bad arguments");
    return hip_check(hipIpcGetMemHandle((hipIpcMemHandle_t*)handle_out, const_cast<void*>(dev_ptr)), "hipIpcGetMemHandle");
}
int cvgs_ipc_open(const void* handle, void** dev_ptr) {
    if (!handle || !dev_ptr) return fail("bad arguments");
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof(h));
    return hip_check(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle");
}
int cvgs_ipc_close(void* dev_ptr) { return hip_check(hipIpcCloseMemHandle(dev_ptr), "hipIpcCloseMemHandle"); }

int cvgs_peer_can_access(int32_t device, int32_t peer) {
    int can = 0;
    int rc = hip_check(hipDeviceCanAccessPeer(&can, device, peer), "hipDeviceCanAccessPeer");
    return rc ? rc : can;
}
int cvgs_peer_enable(int32_t device, int32_t peer) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = hip_check(hipSetDevice(device), "hipSetDevice");
    if (rc) return rc;
    hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
    if (e == hipErrorPeerAccessAlreadyEnabled) {
        (void)hipGetLastError();
        e = hipSuccess;
    }
    (void)hipSetDevice(prev);
    return hip_check(e, "hipDeviceEnablePeerAccess");
}

int cvgs_comm_destroy(cvgs_comm_t c) {
    if (!c) return fail("null communicator");
    int rc = check(ncclCommDestroy(c->comm), "ncclCommDestroy");
    delete c;
    return rc;
}

} // extern "C"
