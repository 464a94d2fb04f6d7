#!/usr/bin/env python3
"""Known-byte-count launches for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on OUR access pattern
(MI355X_MICROARCH.md, HBM section: FETCH_SIZE halves wide 16 B/lane streams on gfx950; other widths must be
calibrated before an absolute is trusted).

Launch A (K1 pattern: 8 B/lane unaligned taps, 4 B/lane planar nt stores): identity-scale K1 over FOUR whole 4K
frames -> every source byte is tapped exactly once per output row pair, so the HBM read volume is the frames'
bytes (4 x 24,883,200 B, + <= 1/16 for the row each 16-row tile shares with its neighbour), the write volume
4 x 99,532,800 B.
Launch B (K9 pattern: 16 B/lane streaming copy): cvgs_stream_copy (the CircularTensor's plane-copy kernel) over
373,248,000 B = the 45 planes a depth-16 1080p fp32x3 update shifts.

Run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python tools/calibrate_pmc.py
            rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -- python tools/calibrate_pmc.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cvgpuspeedup_amd import cvgs  # noqa: E402
from cvgpuspeedup_amd import workloads as W  # noqa: E402

# `python tools/calibrate_pmc.py A|B|C` runs one launch kind only (A and C are the same kernel instantiation: separate
# profiler runs keep their counters apart); no argument = all three.
WHICH = sys.argv[1] if len(sys.argv) > 1 else "ABC"
dev = torch.device("cuda:0")
fw, fh = W.FRAME_4K
N = 4
frames = [W.random_u8_torch((fh, fw, 3), 99 + i, dev) for i in range(N)]
out = torch.zeros((N, 3 * fw * fh), dtype=torch.float32, device=dev)
mats = [cvgs.GpuMat.from_tensor(f, cvgs.CV_8UC3) for f in frames]
f3 = cvgs.CV_32FC3
ops = [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, mats, (fw, fh), N), cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f3),
       cvgs.multiply(f3, [W.K1_ALPHA] * 3), cvgs.subtract(f3, W.K1_SUB[3]), cvgs.divide(f3, W.K1_DIV[3]),
       cvgs.split(f3, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (fw, fh))]
s = torch.cuda.current_stream()
for _ in range(6 if "A" in WHICH else 0):
    cvgs.executeOperations(s, *ops)
torch.cuda.synchronize()
if "A" in WHICH:
    print("A: K1 identity over %d frames: read >= %d B (frames), write %d B per launch; kernel %s" % (
    N, N * fw * fh * 3, N * fw * fh * 12, cvgs.kernel_name(*ops)))

# Launch C (K1's SPARSE taps): the same four frames shrunk 4:1 horizontally and 1:1 vertically.  Every source row is
# tapped and a tap pair falls into every 12-byte stride, so every 64-byte sector of the frames holds tapped bytes: the
# sector-granular read volume is again the frames' bytes (4 x 24,883,200 B), although only half of the bytes are tapped.
out_c = torch.zeros((N, 3 * (fw // 4) * fh), dtype=torch.float32, device=dev)
ops_c = [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, mats, (fw // 4, fh), N), cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f3),
         cvgs.multiply(f3, [W.K1_ALPHA] * 3), cvgs.subtract(f3, W.K1_SUB[3]), cvgs.divide(f3, W.K1_DIV[3]),
         cvgs.split(f3, cvgs.GpuMat.from_tensor(out_c, cvgs.CV_32FC1), (fw // 4, fh))]
for _ in range(6 if "C" in WHICH else 0):
    cvgs.executeOperations(s, *ops_c)
torch.cuda.synchronize()
if "C" in WHICH:
    print("C: K1 4:1 x 1:1 over %d frames (sparse taps, every sector touched): read >= %d B, write %d B per launch; dst %dx%d" % (
    N, N * fw * fh * 3, N * (fw // 4) * fh * 12, fw // 4, fh))

from cvgpuspeedup_amd import capi  # noqa: E402

lib = capi.load_library()
nbytes = 45 * 1920 * 1080 * 4
src = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
for _ in range(6 if "B" in WHICH else 0):
    capi.check(lib.cvgs_stream_copy(dst.data_ptr(), src.data_ptr(), nbytes, s.cuda_stream))
torch.cuda.synchronize()
if "B" in WHICH:
    print("B: streaming copy (k_plane_copy): read %d B, write %d B per launch" % (nbytes, nbytes))
