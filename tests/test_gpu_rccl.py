"""libcvgs_rccl.so on one GPU: a 1-rank communicator (the multi-rank path cannot run on the 1-GPU test box; the
N > 1 layout logic is covered by tests/test_sharding_gloo.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_allgather_inplace():
    import torch
    from cvgpuspeedup_amd import rccl
    torch.cuda.set_device(0)
    uid = rccl.Communicator.unique_id()
    assert len(uid) == rccl.UNIQUE_ID_BYTES
    comm = rccl.Communicator(1, 0, uid)
    assert comm.lib.cvgs_comm_size(comm.handle) == 1 and comm.lib.cvgs_comm_rank(comm.handle) == 0
    t = torch.arange(4096, dtype=torch.float32, device="cuda:0")
    comm.allgather_inplace(t.data_ptr(), t.numel() * 4, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), np.arange(4096, dtype=np.float32))
    comm.destroy()


def test_device_buffer_is_viewable_and_exportable():
    """The sharded tensor of the P2P fused-write exchange: a dedicated allocation torch can view (no copy) and whose
    64-byte IPC handle a peer process can open (the open itself needs a second process: tests/test_sharding_gloo.py
    covers the addressing, bench_dist.py the real exchange)."""
    import torch
    from cvgpuspeedup_amd import rccl
    torch.cuda.set_device(0)
    buf = rccl.DeviceBuffer(2 * 8 * 16 * 4)
    a = buf.tensor(0, (8, 16))
    b = buf.tensor(8 * 16 * 4, (8, 16))
    assert a.data_ptr() == buf.ptr and b.data_ptr() == buf.ptr + 512 and not a.any() and not b.any()
    b.fill_(3.0)
    a[2:4].fill_(1.0)
    torch.cuda.synchronize()
    assert float(a.sum()) == 32.0 and float(b.sum()) == 3.0 * 128
    h = buf.handle()
    assert len(h) == rccl.IPC_HANDLE_BYTES and any(h)
    assert rccl.load_library().cvgs_peer_can_access(0, 0) in (0, 1)
    del a, b
    buf.free()
