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
def test_serving_ticks_example_runs():
    """examples/serving_ticks.cpp: the reference's loop (one executeOperations per frame behind a producer) against ChainBatch ticks on
    streams ATTACHED to a descriptor queue (deferred waits / strictly ordered / the loop unchanged with recorded ticks): the ticks' tensors
    must equal the loop's bit for bit."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "examples", "bin", "serving_ticks")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count("bit for bit") == 5, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("prog", PROGRAMS)
def test_facade_program_passes(prog):
    exe = os.path.join(CPP, "bin", prog)
    if not os.path.exists(exe):
        _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "passed!!" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_small_vec_semantics(tmp_path):
    """fk::detail::SmallVec (the builders' inline plane tables): inline -> heap spill, copy, move, aliasing push_back,
    assign / resize -- host-only, runs without a GPU."""
    _build()
    inc = os.path.join(ROOT, "cvgpuspeedup_amd", "include")
    src = tmp_path / "smallvec.cpp"
    src.write_text(r'''
#include <cvgs/fk_compat.h>
#include <cstdio>
#include <vector>
using fk::detail::SmallVec;
struct P { int a; float b; };
#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)
template <size_t N> static bool same(const SmallVec<P, N>& v, const std::vector<P>& r) {
    if (v.size() != r.size()) return false;
    for (size_t i = 0; i < r.size(); ++i) if (v[i].a != r[i].a || v[i].b != r[i].b) return false;
    return true;
}
int main() {
    SmallVec<P, 4> v; std::vector<P> r;
    CHECK(v.empty() && v.size() == 0);
    for (int i = 0; i < 3; ++i) { v.push_back({i, i * .5f}); r.push_back({i, i * .5f}); }
    const P* inl = v.data();
    CHECK(same(v, r));
    SmallVec<P, 4> c = v;                    // copy while inline
    CHECK(same(c, r) && c.data() != v.data());
    for (int i = 3; i < 40; ++i) { v.push_back(v[0]); r.push_back(r[0]); v.back().a = i; r.back().a = i; } // spills; aliasing source
    CHECK(same(v, r) && v.data() != inl);
    SmallVec<P, 4> c2 = v;                   // copy of a spilled vector
    CHECK(same(c2, r) && c2.data() != v.data());
    const P* heap = v.data();
    SmallVec<P, 4> m = std::move(v);         // move steals the heap block
    CHECK(same(m, r) && m.data() == heap && v.empty());
    v.push_back({7, 7.f});                   // the moved-from vector is usable (inline again)
    CHECK(v.size() == 1 && v[0].a == 7);
    SmallVec<P, 4> mi = std::move(c);        // move of an inline vector copies the elements
    CHECK(mi.size() == 3 && mi[2].a == 2 && c.empty());
    c2 = mi;  r.resize(3);                   // copy-assign a shorter vector over a spilled one
    CHECK(same(c2, r));
    c2 = std::move(m); r.clear(); for (int i = 0; i < 40; ++i) r.push_back({i, i < 3 ? i * .5f : 0.f});
    CHECK(c2.size() == 40 && c2[39].a == 39 && m.empty());
    c2.assign(2, P{9, 1.f});
    CHECK(c2.size() == 2 && c2[1].a == 9);
    c2.resize(70, P{5, 2.f});
    CHECK(c2.size() == 70 && c2[0].a == 9 && c2[2].a == 5 && c2[69].b == 2.f);
    c2.resize(1);
    CHECK(c2.size() == 1 && c2[0].a == 9);
    std::vector<P> src = {{1, 1.f}, {2, 2.f}, {3, 3.f}, {4, 4.f}, {5, 5.f}};
    c2.assign(src.begin(), src.end());
    CHECK(same(c2, src));
    c2.clear();
    CHECK(c2.empty());
    int n = 0; for (const P& p : mi) n += p.a;
    CHECK(n == 3);
    std::printf("ok\n");
    return 0;
}
''')
    exe = tmp_path / "smallvec"
    subprocess.run(["/opt/rocm/bin/hipcc", "-x", "c++", "-std=c++17", "-O1", "-Wall", "-Werror", "-Wno-unused-command-line-argument",
                    "-I" + inc, "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", str(src), "-o", str(exe),
                    "-L" + os.path.join(ROOT, "cvgpuspeedup_amd", "lib"), "-lcvgs_hip", "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + os.path.join(ROOT, "cvgpuspeedup_amd", "lib")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
