/*
 * cvgs_rccl.h -- C-ABI of the multi-GPU assembly step (libcvgs_rccl.so): the RCCL all-gather over xGMI that
 * assembles the sharded crop tensor on every GPU (BASELINE.json north_star).  The reference has no
 * multi-GPU code at all ("Parallelism strategies / distributed backend in the reference:
 * none"), so these entry points have no reference counterpart; they complete the C-ABI of include/cvgs_hip.h for
 * hosts that do not want torch.distributed.
 *
 * Layout contract (in-place all-gather): the full tensor is [n_ranks * items_per_rank] items of item_bytes each;
 * rank r's kernel writes rows [r*items_per_rank, (r+1)*items_per_rank) of ITS copy of the full tensor, then
 * cvgs_allgather_inplace() makes every copy complete.  Asynchronous on the given HIP stream.
 */
#ifndef CVGS_RCCL_H
#define CVGS_RCCL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVGS_UNIQUE_ID_BYTES 128

typedef struct cvgs_comm_s* cvgs_comm_t;

/* rank 0 creates the id (ncclGetUniqueId) and ships the 128 bytes to the other ranks by any means */
int cvgs_comm_unique_id(void* id_out);
/* one process per GPU: every rank calls this with the same id (ncclCommInitRank on the current HIP device) */
int cvgs_comm_init_rank(cvgs_comm_t* out, int32_t n_ranks, int32_t rank, const void* id);
/* one process driving n_devices GPUs (ncclCommInitAll); out[] receives n_devices communicators */
int cvgs_comm_init_all(cvgs_comm_t* out, int32_t n_devices, const int32_t* devices);
int32_t cvgs_comm_rank(cvgs_comm_t comm);
int32_t cvgs_comm_size(cvgs_comm_t comm);
/* ncclAllGather(send = full + rank*bytes_per_rank, recv = full, bytes_per_rank) on `stream` */
int cvgs_allgather_inplace(cvgs_comm_t comm, void* full, size_t bytes_per_rank, void* stream);
/* for single-process multi-GPU use: brackets a set of per-device cvgs_allgather_inplace calls */
int cvgs_group_start(void);
int cvgs_group_end(void);
int cvgs_comm_destroy(cvgs_comm_t comm);
const char* cvgs_rccl_last_error(void);

/* ---- P2P fused write (SURVEY.md 8e option 2) ---------------------------------------------------------------------
 * Instead of a collective that moves every shard a second time, rank r's K1 launch stores its rows of the
 * [N,C,H,W] tensor into its own copy AND into every peer's copy (cvgs_write_desc.mirrors, include/cvgs_hip.h); one
 * small barrier per step replaces the all-gather.  These calls make the peers' tensors addressable:
 *  - one process per GPU (torch.distributed / MPI hosts): allocate the tensor with cvgs_ipc_alloc (a dedicated
 *    hipMalloc allocation, so that the handle covers exactly it), cvgs_ipc_export its 64-byte handle, ship the handles
 *    by any means, cvgs_ipc_open every peer's handle (hipIpcOpenMemHandle, lazy peer access) -> device pointers valid
 *    in THIS process;
 *  - one process driving several GPUs (cvgs_comm_init_all hosts): cvgs_peer_enable(device, peer) once per ordered
 *    pair (hipDeviceEnablePeerAccess); peers' hipMalloc pointers are then directly usable as mirrors.               */
#define CVGS_IPC_HANDLE_BYTES 64
int cvgs_ipc_alloc(void** dev_ptr, size_t bytes);            /* hipMalloc on the current device, zero-filled */
int cvgs_ipc_free(void* dev_ptr);
int cvgs_ipc_export(const void* dev_ptr, void* handle_out);  /* handle_out: CVGS_IPC_HANDLE_BYTES bytes        */
int cvgs_ipc_open(const void* handle, void** dev_ptr);       /* maps a PEER process's allocation               */
int cvgs_ipc_close(void* dev_ptr);
int cvgs_peer_enable(int32_t device, int32_t peer);          /* 0 also when access was already enabled         */
int cvgs_peer_can_access(int32_t device, int32_t peer);      /* 1 / 0, or a negative status                    */

#ifdef __cplusplus
}
#endif
#endif
