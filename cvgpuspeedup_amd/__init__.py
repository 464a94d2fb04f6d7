"""cvgpuspeedup_amd -- MI355X-native fused image-preprocessing engine (cvGS-compatible hot path).

Layout:
  csrc/      hand-written HIP kernels for gfx950 + the C-ABI (include/cvgs_hip.h) -> lib/libcvgs_hip.so
  include/   C++17 facade with the reference's cvGS:: / cv2cuda names (drop-in for C++ callers)
  capi.py    ctypes mirror of the C-ABI
  cvgs.py    Python spelling of the cvGS:: builders for the test/bench harnesses
The compute path is the HIP library only; importing this package never falls back to CPU code.
"""
from . import capi, cvgs  # noqa: F401
from .capi import CvgsError, build_library, load_library  # noqa: F401

__version__ = "0.1.0"
