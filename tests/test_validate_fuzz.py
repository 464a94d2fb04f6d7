"""Robustness of the C-ABI's front door against a binding's mistakes (CPU only: cvgs_validate and the dry-run dispatcher dereference no
device memory): a VALID descriptor with its scalar fields mutated at random -- kinds, types, batch and plane counts, sizes, pitches,
flags, stage opcodes and selectors, extreme integers -- must come back with a status code (0 or a CVGS_ERR_*), never crash, hang or read
outside the arrays the descriptor names.  Descriptors the validator ACCEPTS must also get through the dispatcher's dry run (a kernel name or
a status).  CVGS_FUZZ_VALIDATE_N=300000 for a long hunt."""
import ctypes as C
import os

import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

N = int(os.environ.get("CVGS_FUZZ_VALIDATE_N", "6000"))
EXTREME = [0, 1, -1, 2, 3, 7, 8, 63, 64, 65, 74, 75, 255, 256, 320, 321, 4095, 4096, 65535, 65536, 1 << 20, (1 << 31) - 1, -(1 << 31)]


def _bases():
    """a few valid chains over host arrays large enough for every batch value the mutations keep (<= 128 planes)"""
    frame = np.zeros((480, 640, 3), np.uint8)
    src = cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3)
    out = np.zeros((128, 3 * 64 * 128), np.float32)
    crops = H.random_crops(128, 640, 480, wmax=100, hmax=200)
    k1 = cvgs.lower(H.k1_chain(src, crops, cvgs.GpuMat.from_array(out, cvgs.CV_32FC1)))
    img_out = np.zeros((480, 640, 3), np.float32)
    pw = cvgs.lower([cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [src], 1), cvgs.convertTo(cvgs.CV_8UC3, cvgs.CV_32FC3),
                     cvgs.multiply(cvgs.CV_32FC3, [0.5, 0.25, 2.0]), cvgs.write(cvgs.CV_32FC3, cvgs.GpuMat.from_array(img_out, cvgs.CV_32FC3))])
    return [(k1, (frame, out, crops)), (pw, (frame, img_out))]


def _mutate(rng, d):
    r, w = d.read, d.write
    fields = [(r, "kind"), (r, "src_type"), (r, "batch"), (r, "used_planes"), (r, "dst_width"), (r, "dst_height"), (r, "aspect_ratio"),
              (r, "flags"), (r, "yuv_range"), (r, "yuv_primaries"), (r, "yuv_alpha"), (r, "yuv_layout"), (w, "kind"), (w, "dst_type"),
              (w, "width"), (w, "height"), (w, "step"), (w, "planes"), (w, "n_mirrors"), (d, "n_ops"), (d, "flags")]
    for _ in range(int(rng.integers(1, 4))):
        k = int(rng.integers(len(fields) + 2))
        if k < len(fields):
            obj, name = fields[k]
            v = int(rng.choice(EXTREME)) if rng.uniform() < 0.6 else int(rng.integers(-4, 40))
            if name in ("batch", "used_planes", "planes", "n_mirrors") and v > 128:
                v = 128  # (the arrays a descriptor names are the CALLER's promise: read.src has `batch` entries -- the base chains hold 128)
            if name in ("flags", "struct_size"):
                v &= 0xffffffff
            try:
                setattr(obj, name, v)
            except (TypeError, OverflowError):
                setattr(obj, name, v & 0x7fffffff)
        elif k == len(fields):
            op = d.ops[int(rng.integers(0, capi.MAX_OPS))]
            op.opcode = int(rng.integers(-2, 14))
            op.aux = int(rng.choice(EXTREME)) & 0x7fffffff if rng.uniform() < 0.5 else int(rng.integers(0, 256))
        else:
            op = d.ops[int(rng.integers(0, capi.MAX_OPS))]
            for c in range(4):
                op.operand[c] = float(rng.choice([0.0, -0.0, 1.0, float("inf"), float("nan"), 1e38, -1e-38]))
                op.operand_d[c] = float(op.operand[c])


def test_mutated_descriptors_get_a_status_never_a_crash(lib):
    rng = np.random.default_rng(20260930)
    bases = _bases()
    name = C.create_string_buffer(128)
    accepted = refused = 0
    for i in range(N):
        base, keep = bases[i % len(bases)]
        d = capi.ChainDesc()
        C.memmove(C.byref(d), C.byref(base.desc), C.sizeof(capi.ChainDesc))
        _mutate(rng, d)
        rc = lib.cvgs_validate(C.byref(d))
        assert rc <= 0, rc
        if rc == 0:
            accepted += 1
            rc2 = lib.cvgs_kernel_name(C.byref(d), name, 128)
            assert rc2 <= 0
            if rc2 == 0:
                assert len(name.value) > 0
        else:
            refused += 1
            assert len(lib.cvgs_last_error()) > 0
    assert accepted > N // 50 and refused > N // 10, (accepted, refused)
