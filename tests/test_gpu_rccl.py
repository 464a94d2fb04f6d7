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
