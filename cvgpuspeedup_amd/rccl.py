"""ctypes mirror of include/cvgs_rccl.h (libcvgs_rccl.so): the native RCCL all-gather of the sharded crop tensor."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcvgs_rccl.so")
UNIQUE_ID_BYTES = 128

SYMBOLS = [
    ("cvgs_comm_unique_id", C.c_int, [C.c_void_p]),
    ("cvgs_comm_init_rank", C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_void_p]),
    ("cvgs_comm_init_all", C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_int32)]),
    ("cvgs_comm_rank", C.c_int32, [C.c_void_p]),
    ("cvgs_comm_size", C.c_int32, [C.c_void_p]),
    ("cvgs_allgather_inplace", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("cvgs_group_start", C.c_int, []),
    ("cvgs_group_end", C.c_int, []),
    ("cvgs_comm_destroy", C.c_int, [C.c_void_p]),
    ("cvgs_rccl_last_error", C.c_char_p, []),
    ("cvgs_ipc_alloc", C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    ("cvgs_ipc_free", C.c_int, [C.c_void_p]),
    ("cvgs_ipc_export", C.c_int, [C.c_void_p, C.c_void_p]),
    ("cvgs_ipc_open", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    ("cvgs_ipc_close", C.c_int, [C.c_void_p]),
    ("cvgs_peer_enable", C.c_int, [C.c_int32, C.c_int32]),
    ("cvgs_peer_can_access", C.c_int, [C.c_int32, C.c_int32]),
]
IPC_HANDLE_BYTES = 64

_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libcvgs_rccl.so is missing: run `make -C cvgpuspeedup_amd/csrc`")
    try:
        import torch  # noqa: F401  (same reason as capi.load_library: one HIP runtime / one RCCL per process)
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RuntimeError("cvgs_rccl error %d: %s" % (rc, load_library().cvgs_rccl_last_error().decode()))


class Communicator:
    """One rank of the all-gather group (one process per GPU)."""

    def __init__(self, n_ranks, rank, unique_id):
        self.lib = load_library()
        self.handle = C.c_void_p(0)
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        check(self.lib.cvgs_comm_init_rank(C.byref(self.handle), n_ranks, rank, buf))

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
        check(lib.cvgs_comm_unique_id(buf))
        return bytes(buf)

    def allgather_inplace(self, full_ptr, bytes_per_rank, stream):
        check(self.lib.cvgs_allgather_inplace(self.handle, full_ptr, bytes_per_rank, stream))

    def destroy(self):
        if self.handle:
            self.lib.cvgs_comm_destroy(self.handle)
            self.handle = C.c_void_p(0)


class DeviceBuffer:
    """A dedicated device allocation (cvgs_ipc_alloc) that torch can view and peers can map: the sharded tensor of the
    P2P fused-write exchange.  `tensor(dtype, shape)` wraps it without copying (__cuda_array_interface__)."""

    def __init__(self, nbytes):
        self.lib = load_library()
        self.nbytes = int(nbytes)
        p = C.c_void_p(0)
        check(self.lib.cvgs_ipc_alloc(C.byref(p), self.nbytes))
        self.ptr = int(p.value)

    def handle(self):
        buf = (C.c_uint8 * IPC_HANDLE_BYTES)()
        check(self.lib.cvgs_ipc_export(C.c_void_p(self.ptr), buf))
        return bytes(buf)

    def tensor(self, offset_bytes, shape, typestr="<f4", itemsize=4):
        import torch

        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (self.ptr + offset_bytes, False),
                                      "version": 2}
        v._owner = self
        return torch.as_tensor(v, device="cuda")

    def free(self):
        if self.ptr:
            self.lib.cvgs_ipc_free(C.c_void_p(self.ptr))
            self.ptr = 0


def view(ptr, shape, typestr="<f4"):
    """A torch tensor over raw device memory (own or a mapped peer's), no copy."""
    import torch

    class _View:
        pass
    v = _View()
    v.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(v, device="cuda")


def open_peer(handle_bytes):
    """Map a peer process's DeviceBuffer: returns its device address in this process."""
    lib = load_library()
    buf = (C.c_uint8 * IPC_HANDLE_BYTES).from_buffer_copy(handle_bytes)
    p = C.c_void_p(0)
    check(lib.cvgs_ipc_open(buf, C.byref(p)))
    return int(p.value)


def close_peer(ptr):
    load_library().cvgs_ipc_close(C.c_void_p(ptr))
