"""One rank of the 2-process exchange test (tests/test_gpu_exchange.py starts two of these ON THE SAME GPU): the sharded
batched-crop path of BASELINE cfg #5 in small -- every rank owns a frame and a crop list, its K1 launch stores its rows of the
[2*n,3,128,64] tensor into its own copy AND into the peer's copy through an IPC-mapped pointer (cvgs_write_desc.mirrors), and
"all rows have landed" is the device-side flag exchange (cvgs_exchange_signal / cvgs_exchange_wait).  gloo carries the 64-byte
IPC handles only.  Each rank checks ITS copy of every step's full tensor bit for bit against the oracle."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cvgpuspeedup_amd import capi, cvgs, rccl  # noqa: E402
from oracle import oracle_binding  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    half = os.environ.get("EXCHANGE_HALF") == "1"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    lib = capi.load_library()
    oracle_binding.load_oracle()
    n, T, dst = 7, 6, (64, 128)
    plane = 3 * dst[0] * dst[1]
    esz = 2 if half else 4
    tensor_bytes = world * n * plane * esz
    flag_off = T * tensor_bytes
    buf = rccl.DeviceBuffer(flag_off + world * 128 + 256)
    outs = [buf.tensor(t * tensor_bytes, (world * n, plane), "<f2" if half else "<f4", esz) for t in range(T)]
    handles = [None] * world
    dist.all_gather_object(handles, buf.handle())
    bases = [buf.ptr if r == rank else rccl.open_peer(handles[r]) for r in range(world)]
    others = [r for r in range(world) if r != rank]
    frames = {r: H.random_u8((720, 1280, 3), seed=777 + r) for r in range(world)}
    crops = {(r, t): H.random_crops(n, 1280, 720, wmax=300, hmax=400, seed=10 * t + r) for r in range(world) for t in range(T)}
    frame_t = torch.from_numpy(frames[rank]).to(dev)
    out_type = cvgs.CV_16FC1 if half else cvgs.CV_32FC1
    sig = (C.c_void_p * len(others))(*[bases[r] + flag_off + rank * 128 for r in others])
    own = (C.c_void_p * len(others))(*[buf.ptr + flag_off + r * 128 for r in others])
    err = torch.zeros(2, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream()

    def chain(t, mat, out_mat, mirrors=None):
        ops = H.k1_chain(mat, crops[(rank, t)], out_mat, dst, 3, half=half)
        if mirrors:
            ops[-1].mirrored_to(mirrors)
        return ops

    def my_rows(t):
        return cvgs.GpuMat.from_tensor(outs[t][rank * n:(rank + 1) * n], out_type)

    def peer_rows(t):
        return [bases[r] + t * tensor_bytes + rank * n * plane * esz for r in others]

    dist.barrier()
    # ---- phase 1: eager steps, explicit step numbers ----
    for t in range(T):
        ops = chain(t, cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), my_rows(t), peer_rows(t))
        assert "k1_" in cvgs.kernel_name(*ops), cvgs.kernel_name(*ops)
        cvgs.executeOperations(s, *ops)
        capi.check(lib.cvgs_exchange_signal(sig, len(others), t + 1, None, s.cuda_stream))
    capi.check(lib.cvgs_exchange_wait(own, len(others), T, None, 0, 5000.0, err.data_ptr(), s.cuda_stream))
    s.synchronize()
    assert int(err[0].item()) == 0, "flag wait timed out"
    want = []
    for t in range(T):
        ref = np.zeros((world * n, plane), np.float16 if half else np.float32)
        for r in range(world):
            sl = ref[r * n:(r + 1) * n]
            oracle_binding.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frames[r], cvgs.CV_8UC3), crops[(r, t)],
                                                          cvgs.GpuMat.from_array(sl, out_type), dst, 3, half=half)))
        want.append(ref)
        H.assert_bit_exact(outs[t].cpu().numpy(), ref, "rank %d, eager step %d" % (rank, t))
    dist.barrier()
    # ---- phase 2: the T steps captured into ONE graph (step number on the device), replayed twice ----
    for o in outs:
        o.zero_()
    torch.cuda.synchronize()
    dist.barrier()
    # (the flags still hold T from phase 1: the device counter starts there)
    counter = torch.full((1,), T, dtype=torch.int64, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cs = torch.cuda.current_stream()
        for t in range(T):
            cvgs.executeOperations(cs, *chain(t, cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), my_rows(t), peer_rows(t)))
            if t % 2:  # both spellings: the fused step, and signal + wait as two launches
                capi.check(lib.cvgs_exchange_step(sig, own, len(others), counter.data_ptr(), 2, 5000.0, err.data_ptr(), cs.cuda_stream))
            else:
                capi.check(lib.cvgs_exchange_signal(sig, len(others), 0, counter.data_ptr(), cs.cuda_stream))
                capi.check(lib.cvgs_exchange_wait(own, len(others), 0, counter.data_ptr(), 2, 5000.0, err.data_ptr(), cs.cuda_stream))
        capi.check(lib.cvgs_exchange_wait(own, len(others), 0, counter.data_ptr(), 0, 5000.0, err.data_ptr(), cs.cuda_stream))
    for rep in range(2):
        g.replay()
        torch.cuda.synchronize()
        assert int(err[0].item()) == 0, "flag wait timed out in replay %d" % rep
        assert int(counter.item()) == T * (rep + 2)
        for t in range(T):
            H.assert_bit_exact(outs[t].cpu().numpy(), want[t], "rank %d, replay %d, step %d" % (rank, rep, t))
        dist.barrier()
        if rep == 0:
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()
            dist.barrier()
    for r in others:
        rccl.close_peer(bases[r])
    dist.barrier()
    print("exchange worker rank %d OK" % rank)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
