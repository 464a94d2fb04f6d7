// test_pointwise.cpp -- mirrors reference tests/read/test_read_x_write.cu (K6), tests/read/test_read_x_split.cu (K7),
// tests/batchread/test_batchread_x_write3D.cu (K5), tests/color/test_cvtColor.cu (K8),
// tests/single_operation/test_convertTo.cu and tests/unit_tests/test_split.cu on the cvGS facade.
#include "common.h"

struct P6 { cv::Scalar init, sub, mul, div; };
static const P6 kP6[4] = {{{2}, {0.3f}, {1.f}, {3.2f}},
                          {{2, 37}, {0.3f, 0.3f}, {1.f, 4.f}, {3.2f, 0.6f}},
                          {{2, 37, 128}, {0.3f, 0.3f, 0.3f}, {1.f, 4.f, 3.2f}, {3.2f, 0.6f, 11.8f}},
                          {{2, 37, 128, 20}, {0.3f, 0.3f, 0.3f, 0.3f}, {1.f, 4.f, 3.2f, 0.5f}, {3.2f, 0.6f, 11.8f, 33.f}}};

template <int I, int OC>
static void test_read_x_write(cv::cuda::Stream& stream) {
    constexpr int CN = CV_MAT_CN(OC);
    const P6& p = kP6[CN - 1];
    const int W = 3840, H = 2160;
    cv::cuda::GpuMat d_input(H, W, I, p.init), d_out(H, W, OC);
    cvGS::executeOperations(d_input, d_out, stream, cvGS::convertTo<I, OC>(), cvGS::subtract<OC>(p.sub), cvGS::multiply<OC>(p.mul),
                            cvGS::divide<OC>(p.div), cvGS::add<OC>(p.div));
    stream.waitForCompletion();
    cv::Mat h, src(1, 1, I, p.init); // src holds the SATURATED init (e.g. 128 -> 127 for CV_8S)
    d_out.download(h);
    bool ok = true;
    for (int c = 0; c < CN && ok; ++c) {
        double v0;
        switch (CV_MAT_DEPTH(I)) {
        case CV_8U: v0 = src.ptr<uchar>()[c]; break;
        case CV_8S: v0 = src.ptr<schar>()[c]; break;
        case CV_16U: v0 = src.ptr<ushort>()[c]; break;
        case CV_16S: v0 = src.ptr<short>()[c]; break;
        case CV_32S: v0 = src.ptr<int>()[c]; break;
        default: v0 = src.ptr<float>()[c]; break;
        }
        using OB = BASE_CUDA_T(OC); // float or double output
        const double e = CV_MAT_DEPTH(OC) == CV_64F
                             ? ((v0 - p.sub[c]) * p.mul[c]) / p.div[c] + p.div[c]
                             : ((v0 - (float)p.sub[c]) * (float)p.mul[c]) / (float)p.div[c] + (float)p.div[c];
        for (int y = 0; y < H && ok; y += 97)
            for (int x = 0; x < W && ok; x += 13) ok = close_enough<OB>(h.ptr<OB>(y)[x * CN + c], (OB)e);
    }
    CHECK(ok, "K6 read x write, types " << I << " -> " << OC);
}

template <int I, int O>
static void test_read_convert_split(cv::cuda::Stream& stream) {
    constexpr int CN = CV_MAT_CN(O);
    const int W = 1920, H = 1080;
    cv::Mat h_in(H, W, I);
    fill_random(h_in, 5 + I);
    cv::cuda::GpuMat d_in(h_in), hv_in = host_view(h_in);
    std::vector<cv::cuda::GpuMat> d_out(CN), hv_out(CN);
    std::vector<cv::Mat> h_ref(CN);
    for (int c = 0; c < CN; ++c) { d_out[c].create(H, W, CV_MAT_DEPTH(O)); h_ref[c].create(H, W, CV_MAT_DEPTH(O)); hv_out[c] = host_view(h_ref[c]); }
    cvGS::executeOperations(d_in, stream, cvGS::convertTo<I, O>(), cvGS::split<O>(d_out));
    const fk::Read<fk::PerThreadRead<fk::_2D, CUDA_T(I)>> h_read{cvGS::gpuMat2RawPtr2D<CUDA_T(I)>(hv_in)};
    run_oracle(h_read, cvGS::convertTo<I, O>(), cvGS::split<O>(hv_out));
    stream.waitForCompletion();
    for (int c = 0; c < CN; ++c) {
        cv::Mat h;
        d_out[c].download(h);
        bool same = true;
        for (int y = 0; y < H; ++y) same = same && bit_equal(h.ptr<uchar>(y), h_ref[c].ptr<uchar>(y), (size_t)W * h.elemSize());
        CHECK(same, "K7 read x convert x split vs oracle, type " << I << " plane " << c);
    }
}

template <int I, int O, int BATCH>
static void test_batchread_x_write3D(cv::cuda::Stream& stream) {
    constexpr int CN = CV_MAT_CN(O);
    static const cv::Scalar init[4] = {{10}, {10, 20}, {10, 20, 30}, {10, 20, 30, 40}};
    static const cv::Scalar sub[4] = {{1.5f}, {1.5f, 4.f}, {1.5f, 4.f, 3.2f}, {1.5f, 4.f, 3.2f, 0.5f}};
    const cv::Scalar& div = kP6[CN - 1].div;
    const cv::Size cropSize(60, 120);
    std::array<cv::cuda::GpuMat, BATCH> crops;
    for (auto& c : crops) c = cv::cuda::GpuMat(cropSize, I, init[CN - 1]);
    cv::cuda::GpuMat d_tensor(BATCH, cropSize.width * cropSize.height, O);
    d_tensor.step = (size_t)cropSize.width * cropSize.height * d_tensor.elemSize();
    cvGS::executeOperations<false>(crops, stream, cvGS::convertTo<I, O>(1.0f), cvGS::subtract<O>(sub[CN - 1]), cvGS::divide<O>(div),
                                   cvGS::write<O>(d_tensor, cropSize));
    stream.waitForCompletion();
    const auto h = fetch(d_tensor.data, (size_t)BATCH * cropSize.width * cropSize.height * CN * sizeof(float));
    const float* t = (const float*)h.data();
    bool ok = true;
    for (size_t i = 0; i < (size_t)BATCH * cropSize.width * cropSize.height && ok; ++i)
        for (int c = 0; c < CN && ok; ++c) {
            const double e = ((double)init[CN - 1][c] - (float)sub[CN - 1][c]) / (float)div[c];
            ok = close_enough<float>(t[i * CN + c], (float)e);
        }
    CHECK(ok, "K5 batch read x write3D, types " << I << " -> " << O);
}

template <int I, int O, cv::ColorConversionCodes CC>
static void test_cvtColor(cv::cuda::Stream& stream, std::initializer_list<double> expected) {
    static const cv::Scalar init[4] = {{1}, {1, 2}, {10, 100, 200}, {1, 2, 3, 4}};
    cv::cuda::GpuMat d_in(2160, 3840, I, init[CV_MAT_CN(I) - 1]), d_out(2160, 3840, O);
    cvGS::executeOperations<false>(d_in, d_out, stream, cvGS::cvtColor<CC, I, O>());
    stream.waitForCompletion();
    cv::Mat h;
    d_out.download(h);
    cv::Scalar e;
    int k = 0;
    for (double v : expected) e[k++] = v;
    cv::Mat ref(1, 3840, O, e);
    bool same = true;
    for (int y = 0; y < h.rows; y += 41) same = same && bit_equal(h.ptr<uchar>(y), ref.ptr<uchar>(0), (size_t)h.cols * h.elemSize());
    CHECK(same, "cvtColor code " << (int)CC << " type " << I);
}

static void test_convertTo(cv::cuda::Stream& stream) {
    cv::cuda::GpuMat v1(16, 16, CV_8UC1, cv::Scalar(20)), v3(16, 16, CV_8UC3, cv::Scalar(20, 30, 40)), v4(16, 16, CV_8UC4, cv::Scalar(20, 30, 40, 50));
    cv::cuda::GpuMat f1(16, 16, CV_32FC1), f3(16, 16, CV_32FC3), s4(16, 16, CV_32SC4), s4b(16, 16, CV_32SC4);
    cvGS::executeOperations(v1, f1, stream, cvGS::convertTo<CV_8UC1, CV_32FC1>());
    cvGS::executeOperations(v3, f3, stream, cvGS::convertTo<CV_8UC3, CV_32FC3>(0.5f, 0.5f));
    cvGS::executeOperations(v4, s4, stream, cvGS::convertTo<CV_8UC4, CV_32SC4>(0.5f, 0.5f));
    cvGS::executeOperations(v4, s4b, stream, cvGS::convertTo<CV_8UC4, CV_32SC4>(0.5f));
    stream.waitForCompletion();
    cv::Mat h1, h3, h4, h4b;
    f1.download(h1); f3.download(h3); s4.download(h4); s4b.download(h4b);
    CHECK(h1.at<float>(5, 5) == 20.f, "convertTo 8UC1 -> 32FC1");
    CHECK(h3.at<float>(3, 0) == 10.5f && h3.at<float>(3, 1) == 15.5f && h3.at<float>(3, 2) == 20.5f, "convertTo(0.5, 0.5) -> 32FC3");
    // cv::GpuMat::convertTo rounds half to even: 10.5 -> 10, 15.5 -> 16, 20.5 -> 20, 25.5 -> 26
    CHECK(h4.at<int>(2, 0) == 10 && h4.at<int>(2, 1) == 16 && h4.at<int>(2, 2) == 20 && h4.at<int>(2, 3) == 26, "convertTo(0.5, 0.5) -> 32SC4");
    CHECK(h4b.at<int>(2, 0) == 10 && h4b.at<int>(2, 1) == 15 && h4b.at<int>(2, 2) == 20 && h4b.at<int>(2, 3) == 25, "convertTo(0.5) -> 32SC4");
}

static void test_split(cv::cuda::Stream& stream) {
    constexpr size_t BATCH = 10;
    cv::Mat h_in(cv::Size(16, 16), CV_8UC3, cv::Scalar(1, 2, 3));
    std::array<cv::cuda::GpuMat, BATCH> in;
    std::array<std::vector<cv::cuda::GpuMat>, BATCH> out;
    for (size_t i = 0; i < BATCH; ++i) {
        in[i].upload(h_in);
        for (int c = 0; c < 3; ++c) out[i].emplace_back(cv::Size(16, 16), CV_8UC1);
    }
    cvGS::executeOperations(in, stream, cvGS::split<CV_8UC3>(out));
    cvGS::executeOperations(in[0], stream, cvGS::split<CV_8UC3>(out[0]));
    stream.waitForCompletion();
    bool ok = true;
    for (size_t i = 0; i < BATCH; ++i)
        for (int c = 0; c < 3; ++c) {
            cv::Mat h;
            out[i][c].download(h);
            for (int y = 0; y < 16; ++y)
                for (int x = 0; x < 16; ++x) ok = ok && h.at<uchar>(y, x) == 1 + c;
        }
    CHECK(ok, "split / batch split: every element of plane c equals 1 + c");
}

// arithmetic IOps on INTEGER pixel types (reference include/cvGPUSpeedup.cuh:131-149 defines them for every type; no reference
// test uses them): known answers of the semantics DESIGN.md 7 fixes -- integer arithmetic, truncating division, saturation
template <int T>
static void test_integer_arithmetic(cv::cuda::Stream& stream) {
    using B = BASE_CUDA_T(T);
    constexpr int CN = CV_MAT_CN(T);
    cv::Mat h_in(24, 40, T, cv::Scalar(200, 100, 50, 7));
    cv::cuda::GpuMat d_in(h_in), d_out(24, 40, T);
    cvGS::executeOperations(d_in, d_out, stream, cvGS::multiply<T>(cv::Scalar(2, 3, 1, 1)), cvGS::subtract<T>(cv::Scalar(100, 0.9, 60, 0)),
                            cvGS::divide<T>(cv::Scalar(3, 0, 2, 2)));
    stream.waitForCompletion();
    const auto h = fetch(d_out.data, d_out.step * d_out.rows);
    // per channel, in the type's own range: ch0 (200*2 sat) - 100, / 3; ch1 anything / 0 = 0; ch2 50 - 60 (sat at 0 for unsigned) / 2; ch3 7 / 2 = 3
    const long long hi = std::is_same_v<B, uchar> ? 255 : (std::is_same_v<B, ushort> ? 65535 : (std::is_same_v<B, short> ? 32767 : 2147483647ll));
    const long long lo = std::is_unsigned_v<B> ? 0 : -hi - 1;
    auto sat = [&](long long v) { return v < lo ? lo : (v > hi ? hi : v); };
    const long long want[4] = {sat(sat(200 * 2) - 100) / 3, 0, sat(50 - 60) / 2, 3};
    bool ok = true;
    for (int y = 0; y < 24 && ok; ++y)
        for (int x = 0; x < 40 && ok; ++x)
            for (int c = 0; c < CN; ++c) {
                const B v = ((const B*)(h.data() + (size_t)y * d_out.step))[x * CN + c];
                if ((long long)v != want[c]) {
                    std::cout << "    (" << x << "," << y << "," << c << "): " << (long long)v << " vs " << want[c] << std::endl;
                    ok = false;
                    break;
                }
            }
    CHECK(ok, "integer-typed multiply / subtract / divide, type " << T);
}

int main() {
    cv::cuda::Stream stream;
#define RW(I, O) test_read_x_write<I, O>(stream);
    RW(CV_8UC1, CV_32FC1) RW(CV_8SC1, CV_32FC1) RW(CV_16UC1, CV_32FC1) RW(CV_16SC1, CV_32FC1) RW(CV_32SC1, CV_32FC1) RW(CV_32FC1, CV_32FC1)
    RW(CV_8UC2, CV_32FC2) RW(CV_8UC3, CV_32FC3) RW(CV_8UC4, CV_32FC4) RW(CV_8SC2, CV_32FC2) RW(CV_8SC3, CV_32FC3) RW(CV_8SC4, CV_32FC4)
    RW(CV_16UC2, CV_32FC2) RW(CV_16UC3, CV_32FC3) RW(CV_16UC4, CV_32FC4) RW(CV_16SC2, CV_32FC2) RW(CV_16SC3, CV_32FC3) RW(CV_16SC4, CV_32FC4)
    RW(CV_32SC2, CV_32FC2) RW(CV_32SC3, CV_32FC3) RW(CV_32SC4, CV_32FC4)
    RW(CV_32FC2, CV_64FC2) RW(CV_32FC3, CV_64FC3) RW(CV_32FC4, CV_64FC4) // double-precision outputs (reference :139-141)
#undef RW
    test_read_convert_split<CV_8UC2, CV_32FC2>(stream);
    test_read_convert_split<CV_8UC3, CV_32FC3>(stream);
    test_read_convert_split<CV_8UC4, CV_32FC4>(stream);
    test_read_convert_split<CV_8SC3, CV_32FC3>(stream);
    test_read_convert_split<CV_16UC3, CV_32FC3>(stream);
    test_read_convert_split<CV_16SC4, CV_32FC4>(stream);
    test_read_convert_split<CV_32SC3, CV_32FC3>(stream);
    test_batchread_x_write3D<CV_8UC1, CV_32FC1, 50>(stream);
    test_batchread_x_write3D<CV_8UC3, CV_32FC3, 50>(stream);
    test_batchread_x_write3D<CV_8UC4, CV_32FC4, 50>(stream);
    test_batchread_x_write3D<CV_16SC2, CV_32FC2, 50>(stream);
    test_batchread_x_write3D<CV_32SC3, CV_32FC3, 50>(stream);
    // reference tests/color/test_cvtColor.cu:105-123
    test_cvtColor<CV_8UC3, CV_8UC3, cv::COLOR_RGB2BGR>(stream, {200, 100, 10});
    test_cvtColor<CV_8UC4, CV_8UC4, cv::COLOR_RGBA2BGRA>(stream, {3, 2, 1, 4});
    test_cvtColor<CV_16UC3, CV_16UC3, cv::COLOR_RGB2BGR>(stream, {200, 100, 10});
    test_cvtColor<CV_16UC4, CV_16UC4, cv::COLOR_RGBA2BGRA>(stream, {3, 2, 1, 4});
    test_cvtColor<CV_8UC3, CV_8UC3, cv::COLOR_BGR2RGB>(stream, {200, 100, 10});
    test_cvtColor<CV_8UC4, CV_8UC4, cv::COLOR_BGRA2RGBA>(stream, {3, 2, 1, 4});
    test_cvtColor<CV_8UC3, CV_8UC1, cv::COLOR_RGB2GRAY>(stream, {84});
    test_cvtColor<CV_8UC4, CV_8UC1, cv::COLOR_RGBA2GRAY>(stream, {2});
    test_cvtColor<CV_16UC3, CV_16UC1, cv::COLOR_RGB2GRAY>(stream, {84});
    test_cvtColor<CV_8UC3, CV_8UC1, cv::COLOR_BGR2GRAY>(stream, {120});
    test_cvtColor<CV_8UC4, CV_8UC1, cv::COLOR_BGRA2GRAY>(stream, {2});
    test_cvtColor<CV_16UC4, CV_16UC1, cv::COLOR_BGRA2GRAY>(stream, {2});
    test_convertTo(stream);
    test_split(stream);
    test_integer_arithmetic<CV_8UC3>(stream);
    test_integer_arithmetic<CV_8UC4>(stream);
    test_integer_arithmetic<CV_16UC3>(stream);
    test_integer_arithmetic<CV_16SC4>(stream);
    test_integer_arithmetic<CV_32SC3>(stream);
    return report("test_read_x_write + read_x_split + batchread_x_write3D + cvtColor + convertTo + split");
}
