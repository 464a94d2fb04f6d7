"""The C++ cvGS facade (cvgpuspeedup_amd/include, the drop-in mirror of the reference's include/*.cuh):
tests/cpp/*.cpp restate the reference's own tests on it.  Without a GPU we check they COMPILE against the facade
(every template instantiation the reference's tests need); on the GPU box they run."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
PROGRAMS = ["test_batchresize", "test_resize", "test_pointwise", "test_circulartensor", "test_warping", "test_divergent", "test_sharded"]


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
def test_sharded_crops_example_runs():
    """examples/sharded_crops.cpp: BASELINE cfg #5 from ONE C++ process through the C-ABI -- cvgs_comm_init_all, one fused
    K1 launch per GPU into its rows, cvgs_allgather_inplace inside cvgs_group_start / _end, then the P2P mirror variant; every GPU's copy must be
    the tensor one GPU computes alone, bit for bit.  N = cvgs_device_count() (1 on the test boxes; tests/cpp/test_sharded.cpp adds the oracle)."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "examples", "bin", "sharded_crops"), "--iters", "10"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count("bit for bit") == 2 and "rccl_ranks_seen" in r.stdout, r.stdout + r.stderr


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


def test_soft_half_matches_the_compilers_float16(tmp_path):
    """cvgs/half.h: on compilers without _Float16 (g++ 11 in C++ mode) CV_16F elements are 16 bits of storage with software conversions.
    Built here with a compiler that HAS _Float16, the stand-in forced: every one of the 65,536 bit patterns widens to the same float, and
    doubles -- every exponent of interest, ties, subnormals, overflow, 2,000,000 random ones -- narrow to the same bits (round to nearest
    even straight from double).  Host-only."""
    inc = os.path.join(ROOT, "cvgpuspeedup_amd", "include")
    src = tmp_path / "half.cpp"
    src.write_text(r'''
#define CVGS_HALF_FORCE_SOFT
#include <cvgs/half.h>
#include <cmath>
#include <cstdio>
static uint16_t native(double v) { _Float16 h = (_Float16)v; uint16_t b; std::memcpy(&b, &h, 2); return b; }
static uint64_t s = 0x1234567ull;
static uint64_t rnd() { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
int main() {
    long bad = 0;
    for (uint32_t b = 0; b < 65536; ++b) {
        _Float16 h; uint16_t bb = (uint16_t)b; std::memcpy(&h, &bb, 2);
        const float a = (float)h, c = cvgs::half_t::to_float(bb);
        if (std::isnan(a) ? !std::isnan(c) : std::memcmp(&a, &c, 4) != 0) ++bad;
        // and back: a half value narrows to itself
        if (!std::isnan(a) && cvgs::half_t::from_double((double)a) != bb) ++bad;
    }
    auto check = [&](double v) {
        const uint16_t a = native(v), c = cvgs::half_t::from_double(v);
        if (std::isnan(v) ? ((c & 0x7c00u) != 0x7c00u || !(c & 0x3ffu)) : a != c) { if (bad < 5) std::printf("%.17g: native %04x soft %04x\n", v, a, c); ++bad; }
    };
    for (uint32_t b = 0; b < 65536; ++b) { // every half value, its neighbours' midpoints and a hair to either side of them
        uint16_t bb = (uint16_t)b;
        const float f = cvgs::half_t::to_float(bb);
        if (std::isnan(f) || std::isinf(f)) continue;
        const float g = cvgs::half_t::to_float((uint16_t)(bb + 1));
        if (std::isnan(g) || std::isinf(g) || ((bb + 1) & 0x8000u) != (bb & 0x8000u)) continue;
        const double mid = ((double)f + (double)g) / 2;
        check(mid); check(std::nextafter(mid, 1e300)); check(std::nextafter(mid, -1e300)); check((double)f); check(std::nextafter((double)f, 1e300));
    }
    const double special[] = {0.0, -0.0, 65504.0, 65519.999, 65520.0, 65520.001, 1e300, -1e300, 5.9604644775390625e-8, 2.98023223876953125e-8,
                              2.9802322387695316e-8, 2.9802322387695309e-8, 1e-30, -1e-30, 6.103515625e-5, 6.0975551605224609375e-5, INFINITY, -INFINITY, NAN};
    for (double v : special) check(v);
    for (int i = 0; i < 2000000; ++i) {
        const uint64_t r = rnd();
        double v;
        const uint64_t bits = (r & 0x800fffffffffffffull) | ((uint64_t)(1023 - 30 + (r >> 52) % 50) << 52); // exponents -30 .. +19
        std::memcpy(&v, &bits, 8);
        check(v);
    }
    std::printf("%ld mismatches\n", bad);
    return bad ? 1 : 0;
}
''')
    exe = tmp_path / "half"
    subprocess.run(["/opt/rocm/bin/hipcc", "-x", "c++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-command-line-argument", "-I" + inc, str(src), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-1000:]


def test_facade_headers_compile_with_plain_gcc():
    """The facade is host C++17: the five reference-test programs must get through g++ (this image's 11.4, no _Float16 in C++ mode) as well as
    through hipcc -x c++ -- a maintainer's host compiler need not be clang.  Syntax + semantics only (no link: the box decides)."""
    inc = os.path.join(ROOT, "cvgpuspeedup_amd", "include")
    for prog in PROGRAMS:
        subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-I" + inc,
                        os.path.join(CPP, prog + ".cpp")], check=True, cwd=CPP)


def test_c_abi_headers_are_plain_c99(tmp_path):
    """include/cvgs_hip.h and include/cvgs_rccl.h are what a cgo / JNI / ctypes binding includes: they must be C, not C++ (gcc -std=c99
    -pedantic, no warnings)."""
    src = tmp_path / "abi.c"
    src.write_text('#include "include/cvgs_hip.h"\n#include "include/cvgs_rccl.h"\nint main(void) { return (int)sizeof(cvgs_chain_desc) == 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I" + ROOT, str(src)], check=True)


@pytest.mark.gpu
def test_the_boundary_from_plain_c():
    """tests/cpp/c_abi_k1.c: the headline chain described as ONE cvgs_chain_desc in C99 (gcc, no C++), run with cvgs_execute on a HIP
    stream, against the oracle running the same descriptor on host memory -- what a cgo / JNI / N-API binding of the C-ABI does."""
    subprocess.run(["make", "-C", CPP, "bin/c_abi_k1"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(CPP, "bin", "c_abi_k1")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "bit-identical" in r.stdout and "k1_u8c3_swap_mul_sub_div" in r.stdout, r.stdout + r.stderr


def test_readme_spelling_and_the_new_fk_slice_lower_as_specified(tmp_path):
    """Host-only lowering checks (no GPU): (1) the reference README's call verbatim -- convertTo<CV_8UC3, CV_32FC3>() behind the batched
    resize, `substract` -- lowers to the SAME descriptor as the chain without the redundant cast (README.md:123-130);
    (2) fk::CircularBatchRead<Ascendent / Descendent> rotates its planes where the descriptor is built
    (tests/batchread/test_circularbatchread_x_write3D.cu:59-66); (3) fk::executeDivergentBatch's plane selection: a tensor write of plane z
    starts z planes in, a batched read keeps its plane z."""
    _build()
    inc = os.path.join(ROOT, "cvgpuspeedup_amd", "include")
    src = tmp_path / "lowering.cpp"
    src.write_text(r'''
#include <cvGPUSpeedup.cuh>
#include <cstdio>
#include <cstring>
#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)
int main() {
    constexpr int N = 50;
    static unsigned char frame[480 * 640 * 3];
    static float out[N * 64 * 128 * 3];
    cv::cuda::GpuMat f(480, 640, CV_8UC3, frame, 640 * 3), o(N, 64 * 128 * 3, CV_32FC1, out, 64 * 128 * 3 * 4);
    std::array<cv::cuda::GpuMat, N> crops;
    for (int i = 0; i < N; ++i) crops[i] = f(cv::Rect(i, i, 60, 120));
    const cv::Size resDims(64, 128);
    const cv::Scalar substract_val(1, 4, 6), divide_val(255, 255, 255);
    const double alpha = 0.5;
    fk::ChainBuilder a, b;
    fk::lowerChain(a, cvGS::resize<CV_8UC3, cv::INTER_LINEAR, N>(crops, resDims, 50), cvGS::convertTo<CV_8UC3, CV_32FC3>(),
                   cvGS::multiply<CV_32FC3>(cv::Scalar(alpha, alpha, alpha)), cvGS::substract<CV_32FC3>(substract_val), cvGS::divide<CV_32FC3>(divide_val),
                   cvGS::split<CV_32FC3>(o, resDims));
    fk::lowerChain(b, cvGS::resize<CV_8UC3, cv::INTER_LINEAR, N>(crops, resDims, 50),
                   cvGS::multiply<CV_32FC3>(cv::Scalar(alpha, alpha, alpha)), cvGS::subtract<CV_32FC3>(substract_val), cvGS::divide<CV_32FC3>(divide_val),
                   cvGS::split<CV_32FC3>(o, resDims));
    CHECK(a.d.n_ops == 3 && b.d.n_ops == 3);
    CHECK(std::memcmp(a.d.ops, b.d.ops, sizeof(a.d.ops)) == 0);
    CHECK(a.d.read.batch == N && a.d.read.kind == CVGS_READ_RESIZE_LINEAR && a.d.write.kind == CVGS_WRITE_TENSOR_SPLIT);
    CHECK(cvgs_validate(&a.d) == CVGS_OK);
    // ---- CircularBatchRead ----
    constexpr int B = 5;
    static uchar3 planes[B][4 * 4];
    fk::Read<fk::CircularBatchRead<fk::Ascendent, fk::PerThreadRead<fk::_2D, uchar3>, B>> asc;
    fk::Read<fk::CircularBatchRead<fk::Descendent, fk::PerThreadRead<fk::_2D, uchar3>, B>> desc;
    asc.params.first = desc.params.first = 2;
    for (int i = 0; i < B; ++i) asc.params.opData[i].params = desc.params.opData[i].params = fk::RawPtr<fk::_2D, uchar3>{planes[i], {4, 4, 12}};
    static uchar3 tensor[B * 16];
    fk::Write<fk::PerThreadWrite<fk::_3D, uchar3>> w{fk::RawPtr<fk::_3D, uchar3>{tensor, {4, 4, B, 1, 12, 48}}};
    fk::ChainBuilder ca, cd;
    fk::lowerChain(ca, asc, w);
    fk::lowerChain(cd, desc, w);
    const cvgs_image2d* sa = (const cvgs_image2d*)ca.d.read.src;
    const cvgs_image2d* sd = (const cvgs_image2d*)cd.d.read.src;
    for (int z = 0; z < B; ++z) {
        CHECK(sa[z].data == planes[(z + 2) % B]);
        CHECK(sd[z].data == planes[(2 + B - z) % B]);
    }
    CHECK(ca.d.read.batch == B && cvgs_validate(&ca.d) == CVGS_OK && cvgs_validate(&cd.d) == CVGS_OK);
    // ---- plane selection of the divergent batch ----
    fk::ChainBuilder p3;
    fk::lowerChain(p3, asc, w);
    fk::detail::select_plane(p3, 3);
    p3.finish();
    CHECK(p3.d.read.batch == 1 && ((const cvgs_image2d*)p3.d.read.src)[0].data == planes[(3 + 2) % B]);
    CHECK((unsigned char*)p3.d.write.data == (unsigned char*)tensor + 3 * 16 * 3);
    CHECK(cvgs_validate(&p3.d) == CVGS_OK);
    std::printf("ok\n");
    return 0;
}
''')
    exe = tmp_path / "lowering"
    subprocess.run(["/opt/rocm/bin/hipcc", "-x", "c++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-command-line-argument", "-Wno-unused-variable",
                    "-I" + inc, "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", str(src), "-o", str(exe),
                    "-L" + os.path.join(ROOT, "cvgpuspeedup_amd", "lib"), "-lcvgs_hip", "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + os.path.join(ROOT, "cvgpuspeedup_amd", "lib")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
