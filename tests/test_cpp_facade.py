"""The C++ cvGS facade (cvgpuspeedup_amd/include, the drop-in mirror of the reference's include/*.cuh):
tests/cpp/*.cpp restate the reference's own tests on it.  Without a GPU we check they COMPILE against the facade
(every template instantiation the reference's tests need); on the GPU box they run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
PROGRAMS = ["test_batchresize", "test_resize", "test_pointwise", "test_circulartensor", "test_warping"]


def _build():
    subprocess.run(["make", "-C", os.path.join(ROOT, "cvgpuspeedup_amd", "csrc"), "-j8"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CPP, "-j8"], check=True, stdout=subprocess.DEVNULL)


def test_facade_tests_compile():
    _build()
    for p in PROGRAMS:
        assert os.path.exists(os.path.join(CPP, "bin", p)), p


def test_facade_rejects_type_mismatch_at_compile_time(tmp_path):
    """The reference's compile-time errors: a chain whose types do not follow, convertTo changing channels."""
    inc = os.path.join(ROOT, "cvgpuspeedup_amd", "include")
    bad = {
        "chain": "cvGS::executeOperations(s, cvGS::resize<CV_8UC3, cv::INTER_LINEAR>(m, cv::Size(8, 8), 0., 0.), "
                 "cvGS::multiply<CV_32FC4>(cv::Scalar(1)), cvGS::write<CV_32FC4>(m));",
        "convert": "auto x = cvGS::convertTo<CV_8UC1, CV_32FC2>(); (void)x;",
        "interp": "auto x = cvGS::resize<CV_8UC3, cv::INTER_CUBIC>(m, cv::Size(8, 8), 0., 0.); (void)x;",
    }
    for name, body in bad.items():
        src = tmp_path / (name + ".cpp")
        src.write_text('#include <cvGPUSpeedup.h>\nint main() { cv::cuda::Stream s; cv::cuda::GpuMat m; %s return 0; }\n' % body)
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-x", "c++", "-std=c++17", "-fsyntax-only", "-I" + inc, "-I/opt/rocm/include",
                            "-D__HIP_PLATFORM_AMD__", str(src)], capture_output=True, text=True)
        assert r.returncode != 0, "%s compiled but must not" % name
        assert "static assertion failed" in r.stderr or "static_assert" in r.stderr, r.stderr[-500:]


@pytest.mark.gpu
def test_readme_example_runs():
    """examples/readme_example.cpp: the reference README's 50-detection example on the facade."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "examples", "bin", "readme_example")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("prog", PROGRAMS)
def test_facade_program_passes(prog):
    exe = os.path.join(CPP, "bin", prog)
    if not os.path.exists(exe):
        _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "passed!!" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
