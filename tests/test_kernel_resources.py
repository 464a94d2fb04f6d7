"""Static perf gate (CPU, no GPU): the resource footprint of every gfx950 kernel in the built libraries.

Round 2 shipped a 1.4-7x slowdown of the thread-fused pointwise kernels with 2435 bit-exactness tests green:
a refactor made LLVM park part of a per-thread array in LDS (3-channel kernels) / scratch (4-channel kernels).  Those
symptoms are in the code object's metadata, which tools/kernel_resources.py reads straight out of the .so.  Hard rules for
every kernel of the engine: no scratch, no register spills, no dynamic stack, no dispatch/queue-packet reads (the mark of an
alloca promoted to LDS); and against the committed table (tools/kernel_resources.json): the LDS footprint may not change and
the VGPR count may not grow by more than 4 without the table being regenerated on purpose
(`python tools/kernel_resources.py --json tools/kernel_resources.json`).
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as KR  # noqa: E402

LIBS = [os.path.join(ROOT, "cvgpuspeedup_amd", "lib", n) for n in ("libcvgs_hip.so",)]


@pytest.fixture(scope="module")
def kernels():
    for lib in LIBS:
        if not os.path.exists(lib):
            pytest.fail("%s is missing: run build() first (there is no fallback path)" % lib)
    return KR.kernels_of(LIBS[0])


def test_library_holds_gfx950_kernels_only(kernels):
    assert len(kernels) > 500  # K1 / K4 / pointwise / circular / generic / warp instantiations


def test_no_kernel_uses_scratch_spills_or_promoted_allocas(kernels):
    bad = KR.violations(kernels)
    names = KR.demangle([b[0] for b in bad])
    assert not bad, "\n".join("%s: %s" % (n[:200], b[1]) for n, b in zip(names, bad))


def test_lds_and_vgpr_footprints_match_the_committed_table(kernels):
    table = json.load(open(KR.TABLE))
    changed, new = KR.drift(kernels, table)
    msg = ["%s: [lds, vgpr, sgpr] %s -> %s" % (KR.demangle([n])[0][:200], ref, now) for n, ref, now in changed]
    msg += ["not in the table: %s" % KR.demangle([n])[0][:200] for n in new]
    assert not changed and not new, "\n".join(msg[:40])


HOT = (  # mangled-name fragments of the kernels behind BASELINE's configs
    "k1_resize_splitILi3ELi64ELi1ENS_6K1ProgIJLi100ELi2ELi4ELi5EEEELi0EfLi0E",   # cfg #2: the 50-crop headline (1 row per wave)
    "k1_resize_splitILi3ELi64ELi2ENS_6K1ProgIJLi100ELi2ELi4ELi5EEEELi0EfLi0E",   # the same, 2 rows per wave (large batches)
    "k4_nv12_resizeILi64ENS_6K1ProgIJLi100ELi2ELi4ELi5EEEEfLi1ELi3E",            # cfg #3: NV12 -> BGR -> resize -> normalize -> split
    "k4_nv12_x2INS_6K1ProgIJLi100ELi2ELi4ELi5EEEELi1E",                          # the same at frame size, two pixels per lane
    "k_pointwise4ILi3ELi64ENS_10StaticProgIJLi1ELi2ELi4ELi5EEEEfLi0E",           # K5/K6: the regression's kernel
    "k_circular_pushILi3ENS_10StaticProgIJLi1ELi2ELi4ELi5EEEEfLb0EE",                # cfg #4: CircularTensor push
    "k_circular_pushILi3ENS_10StaticProgIJLi1ELi2ELi4ELi5EEEEfLb1EE",  # the same, device-indexed (capturable handles)
    "k_plane_copyIDv4_fLi8E",                                                    # cfg #4: the shift
)


def test_hot_kernels_keep_full_occupancy(kernels):
    """The kernels behind BASELINE's configs are latency / bandwidth bound and rely on 8 waves per SIMD: <= 64 VGPRs
    (512 / 8), and a 256-thread workgroup's LDS must leave room for 8 workgroups per CU (160 KB / 8 = 20 KB)."""
    for frag in HOT:
        found = [k for k in kernels if frag in k["name"]]
        assert found, "hot kernel not in the library any more: " + frag
        for k in found:
            assert k["vgpr"] <= 64 and k["lds"] <= 20480, (KR.demangle([k["name"]])[0][:160], k["vgpr"], k["lds"])


def test_k1_and_frame_size_k4_kernels_get_their_leading_scalars_preloaded(kernels):
    """k1_resize_split / k4_nv12_x2 take 14 leading scalar parameters that the hardware delivers in user SGPRs with the dispatch
    (csrc/Makefile: PRELOAD; DESIGN 9: tick of 16 39.58 -> 38.97 us).  A build that loses the flag is still correct and 1.5 % slower:
    the kernel descriptor says which one this is."""
    hot = [k for k in kernels if "k1_resize_split" in k["name"] or "k4_nv12_x2" in k["name"]]
    assert len(hot) > 100
    wrong = [k["name"] for k in hot if k["preload"] != 14]
    assert not wrong, KR.demangle(wrong[:5])
    # ... and nothing else asks for it (the flag is scoped to those translation units)
    others = [k["name"] for k in kernels if k["preload"] and k not in hot]
    assert not others, KR.demangle(others[:5])
