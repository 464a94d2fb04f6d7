// test_resize.cpp -- mirrors reference tests/resize/test_resize_x_split.cu (K2), test_resize_write.cu (K3) and
// test_fused_resize.cu (K4, NV12 read-back fused into the resize) on the cvGS facade.
#include "common.h"

struct Params { cv::Scalar init, alpha, sub, div; };
static const double kAlpha = 0.3;
static const Params kParams[4] = {
    {{2}, {kAlpha}, {1.f}, {3.2f}},
    {{2, 37}, {kAlpha, kAlpha}, {1.f, 4.f}, {3.2f, 0.6f}},
    {{2, 37, 128}, {kAlpha, kAlpha, kAlpha}, {1.f, 4.f, 3.2f}, {3.2f, 0.6f, 11.8f}},
    {{2, 37, 128, 20}, {kAlpha, kAlpha, kAlpha, kAlpha}, {1.f, 4.f, 3.2f, 0.5f}, {3.2f, 0.6f, 11.8f, 33.f}}};

// K2: resize(ROI) -> multiply -> subtract -> divide -> split into C separate GpuMats
template <int TI, int TO>
static void test_resize_split_one(cv::cuda::Stream& stream) {
    constexpr int CN = CV_MAT_CN(TO);
    const Params& p = kParams[CN - 1];
    const cv::Size up(64, 128);
    const cv::Rect2d crop(cv::Point2d(200, 200), cv::Point2d(260, 320));
    cv::cuda::GpuMat d_input(2160, 3840, TI, p.init);
    std::vector<cv::cuda::GpuMat> d_out(CN);
    for (auto& m : d_out) m.create(up, CV_MAT_DEPTH(TO));
    cvGS::executeOperations(stream, cvGS::resize<TI, cv::INTER_LINEAR>(d_input(crop), up, 0., 0.), cvGS::multiply<TO>(p.alpha),
                            cvGS::subtract<TO>(p.sub), cvGS::divide<TO>(p.div), cvGS::split<TO>(d_out));
    stream.waitForCompletion();
    for (int c = 0; c < CN; ++c) {
        cv::Mat h;
        d_out[c].download(h);
        const double e = ((double)p.init[c] * (float)kAlpha - (float)p.sub[c]) / (float)p.div[c];
        bool ok = true;
        for (int y = 0; y < h.rows && ok; ++y) ok = all_close<float>(h.ptr<float>(y), (size_t)h.cols, e);
        CHECK(ok, "K2 known answer, type " << TI << " channel " << c);
    }
    // non-constant frame vs oracle
    cv::Mat h_frame(700, 900, TI);
    fill_random(h_frame, 77 + TI);
    cv::cuda::GpuMat d_frame(h_frame), hv = host_view(h_frame);
    std::vector<cv::Mat> h_ref(CN);
    std::vector<cv::cuda::GpuMat> hv_ref(CN);
    for (int c = 0; c < CN; ++c) { h_ref[c].create(up.height, up.width, CV_MAT_DEPTH(TO)); hv_ref[c] = host_view(h_ref[c]); }
    const cv::Rect roi(13, 7, 333, 555);
    cvGS::executeOperations(stream, cvGS::resize<TI, cv::INTER_LINEAR>(d_frame(roi), up, 0., 0.), cvGS::multiply<TO>(p.alpha),
                            cvGS::subtract<TO>(p.sub), cvGS::divide<TO>(p.div), cvGS::split<TO>(d_out));
    run_oracle(cvGS::resize<TI, cv::INTER_LINEAR>(hv(roi), up, 0., 0.), cvGS::multiply<TO>(p.alpha), cvGS::subtract<TO>(p.sub),
               cvGS::divide<TO>(p.div), cvGS::split<TO>(hv_ref));
    stream.waitForCompletion();
    for (int c = 0; c < CN; ++c) {
        cv::Mat h;
        d_out[c].download(h);
        bool same = true;
        for (int y = 0; y < h.rows; ++y) same = same && bit_equal(h.ptr<float>(y), h_ref[c].ptr<float>(y), (size_t)h.cols * 4);
        CHECK(same, "K2 non-constant vs oracle, type " << TI << " channel " << c);
    }
}

// K3: resize up / down -> convertTo<32F -> I> (saturate) -> write; a constant image stays that constant
template <int I>
static void test_resize_write(cv::cuda::Stream& stream) {
    constexpr int CN = CV_MAT_CN(I);
    constexpr int F = CV_MAKETYPE(CV_32F, CN);
    const cv::Scalar init = kParams[CN - 1].init;
    cv::cuda::GpuMat d_input(2160, 3840, I, init);
    for (const cv::Size sz : {cv::Size(3870, 2260), cv::Size(300, 500)}) {
        cv::cuda::GpuMat d_out(sz, I);
        cvGS::executeOperations(stream, cvGS::resize<I, cv::INTER_LINEAR>(d_input, sz, 0., 0.), cvGS::convertTo<F, I>(), cvGS::write<I>(d_out));
        stream.waitForCompletion();
        cv::Mat h;
        d_out.download(h);
        cv::Mat e(sz, I, init);
        bool same = true;
        for (int y = 0; y < h.rows; ++y) {
            if constexpr (CV_MAT_DEPTH(I) == CV_32F) // float: the reference's 1e-4 tolerance; integers: exact
                same = same && all_close<float>(h.ptr<float>(y), (size_t)h.cols * CN, init[0]);
            else
                same = same && bit_equal(h.ptr<uchar>(y), e.ptr<uchar>(y), (size_t)h.cols * h.elemSize());
        }
        CHECK(same, "K3 resize+saturate of a constant image, type " << I << " to " << sz.width << "x" << sz.height);
    }
}

// K4: Resize<LINEAR>(fuse(ReadYUV<NV12>, ConvertYUVToRGB<NV12,Full,bt709,true,float4>)) -> SaturateCast -> VectorReorder -> write
static void test_fused_nv12_resize(cv::cuda::Stream& stream) {
    const uint W = 1920, H = 1080;
    const fk::Size down(640, 360);
    cv::Mat h_nv12(H + H / 2, W, CV_8UC1);
    fill_random(h_nv12, 4242);
    cv::cuda::GpuMat d_nv12(h_nv12);
    auto run = [&](uchar* base, uint pitch, uchar4* out, uint out_pitch, bool gpu) {
        const fk::RawPtr<fk::_2D, uchar> src{base, {W, H, pitch}};
        const auto readBack = fk::fuse(fk::Read<fk::ReadYUV<fk::NV12>>{src},
                                       fk::Unary<fk::ConvertYUVToRGB<fk::NV12, fk::Full, fk::bt709, true, float4>>{});
        const auto readOp = fk::Resize<fk::INTER_LINEAR>::build(readBack, down);
        const fk::RawPtr<fk::_2D, uchar4> dst{out, {(uint)down.width, (uint)down.height, out_pitch}};
        const auto convertOp = fk::Unary<fk::SaturateCast<float4, uchar4>>{};
        const auto colorConvert = fk::Unary<fk::VectorReorder<uchar4, 2, 1, 0, 3>>{};
        const auto writeOp = fk::Write<fk::PerThreadWrite<fk::_2D, uchar4>>{dst};
        if (gpu) fk::executeOperations(stream.raw(), readOp, convertOp, colorConvert, writeOp);
        else run_oracle(readOp, convertOp, colorConvert, writeOp);
    };
    cv::cuda::GpuMat d_out(down.height, down.width, CV_8UC4);
    cv::Mat h_ref(down.height, down.width, CV_8UC4);
    run(d_nv12.data, (uint)d_nv12.step, (uchar4*)d_out.data, (uint)d_out.step, true);
    run(h_nv12.data, (uint)h_nv12.step, (uchar4*)h_ref.data, (uint)h_ref.step, false);
    stream.waitForCompletion();
    cv::Mat h;
    d_out.download(h);
    bool same = true;
    for (int y = 0; y < h.rows; ++y) same = same && bit_equal(h.ptr<uchar>(y), h_ref.ptr<uchar>(y), (size_t)h.cols * 4);
    CHECK(same, "K4 NV12 -> BGRA resize, bit-exact vs oracle");
}

// a software decoder's planar-chroma surface (yuv420p) -> BGR u8 thumbnail: Resize<LINEAR>(fuse(ReadYUV<PF>, ConvertYUVToRGB<PF,...,float3>))
// -> VectorReorder -> SaturateCast<float3, uchar3> -> write: the pixel-format parameter of the reference's reader
// (tests/resize/test_fused_resize.cu:50-51) beyond NV12, on K4's u8 image mode
template <fk::PixelFormat PF>
static void test_fused_planar_resize_u8(cv::cuda::Stream& stream, const char* what) {
    const uint W = 1280, H = 720;
    const fk::Size down(426, 240); // 6 full 64-column tiles + a ragged one per row
    cv::Mat h_yuv(H + H / 2, W, CV_8UC1);
    fill_random(h_yuv, 5150);
    cv::cuda::GpuMat d_yuv(h_yuv);
    auto run = [&](uchar* base, uint pitch, uchar3* out, uint out_pitch, bool gpu) {
        const fk::RawPtr<fk::_2D, uchar> src{base, {W, H, pitch}};
        const auto readBack = fk::fuse(fk::Read<fk::ReadYUV<PF>>{src}, fk::Unary<fk::ConvertYUVToRGB<PF, fk::Limited, fk::bt601, false, float3>>{});
        const auto readOp = fk::Resize<fk::INTER_LINEAR>::build(readBack, down);
        const fk::RawPtr<fk::_2D, uchar3> dst{out, {(uint)down.width, (uint)down.height, out_pitch}};
        const auto colorConvert = fk::Unary<fk::VectorReorder<float3, 2, 1, 0>>{};
        const auto convertOp = fk::Unary<fk::SaturateCast<float3, uchar3>>{};
        const auto writeOp = fk::Write<fk::PerThreadWrite<fk::_2D, uchar3>>{dst};
        if (gpu) fk::executeOperations(stream.raw(), readOp, colorConvert, convertOp, writeOp);
        else run_oracle(readOp, colorConvert, convertOp, writeOp);
    };
    cv::cuda::GpuMat d_out(down.height, down.width, CV_8UC3);
    cv::Mat h_ref(down.height, down.width, CV_8UC3);
    run(d_yuv.data, (uint)d_yuv.step, (uchar3*)d_out.data, (uint)d_out.step, true);
    run(h_yuv.data, (uint)h_yuv.step, (uchar3*)h_ref.data, (uint)h_ref.step, false);
    stream.waitForCompletion();
    cv::Mat h;
    d_out.download(h);
    bool same = true, varied = false;
    for (int y = 0; y < h.rows; ++y) {
        same = same && bit_equal(h.ptr<uchar>(y), h_ref.ptr<uchar>(y), (size_t)h.cols * 3);
        varied = varied || h_ref.ptr<uchar>(y)[0] != h_ref.ptr<uchar>(0)[0];
    }
    CHECK(same && varied, what);
}

// cfg #3 through the facade: cvtColorNV12 -> resize -> normalize -> split, ONE kernel
static void test_nv12_facade_cfg3(cv::cuda::Stream& stream) {
    const int W = 1280, H = 720;
    const cv::Size down(426, 240);
    cv::Mat h_nv12(H + H / 2, W, CV_8UC1);
    fill_random(h_nv12, 777);
    cv::cuda::GpuMat d_nv12(h_nv12), hv_nv12 = host_view(h_nv12);
    cv::cuda::GpuMat d_out(1, down.width * down.height * 3, CV_32F);
    cv::Mat h_ref(1, down.width * down.height * 3, CV_32F);
    cv::cuda::GpuMat hv_ref = host_view(h_ref);
    const cv::Scalar a(0.3, 0.3, 0.3), s(1.f, 4.f, 3.2f), d(3.2f, 0.6f, 11.8f);
    cvGS::executeOperations(stream, cvGS::resize<cv::INTER_LINEAR>(cvGS::cvtColorNV12<cv::COLOR_YUV2BGR_NV12>(d_nv12), down),
                            cvGS::multiply<CV_32FC3>(a), cvGS::subtract<CV_32FC3>(s), cvGS::divide<CV_32FC3>(d),
                            cvGS::split<CV_32FC3>(d_out, down));
    run_oracle(cvGS::resize<cv::INTER_LINEAR>(cvGS::cvtColorNV12<cv::COLOR_YUV2BGR_NV12>(hv_nv12), down), cvGS::multiply<CV_32FC3>(a),
               cvGS::subtract<CV_32FC3>(s), cvGS::divide<CV_32FC3>(d), cvGS::split<CV_32FC3>(hv_ref, down));
    stream.waitForCompletion();
    const auto h = fetch(d_out.data, (size_t)down.width * down.height * 3 * 4);
    CHECK(bit_equal(h.data(), h_ref.data, h.size()), "cfg3 through cvGS::cvtColorNV12 + resize, bit-exact vs oracle");
    // the same frame through the device-side descriptor queue (engine extension): decoder surfaces are its second kind
    cvGS::Queue queue;
    cv::cuda::GpuMat d_out_q(1, down.width * down.height * 3, CV_32F);
    const uint64_t ticket = cvGS::executeOperations(queue, cvGS::resize<cv::INTER_LINEAR>(cvGS::cvtColorNV12<cv::COLOR_YUV2BGR_NV12>(d_nv12), down),
                                                    cvGS::multiply<CV_32FC3>(a), cvGS::subtract<CV_32FC3>(s), cvGS::divide<CV_32FC3>(d),
                                                    cvGS::split<CV_32FC3>(d_out_q, down));
    queue.wait(ticket);
    const auto hq = fetch(d_out_q.data, (size_t)down.width * down.height * 3 * 4);
    CHECK(bit_equal(hq.data(), h_ref.data, hq.size()), "cfg3 through cvGS::executeOperations(queue, ...), bit-exact vs oracle");
}

// a decoder surface letterboxed into a detector input: cvtColorNV12<RGB> -> resize<LINEAR, PRESERVE_AR>(640x640 style) -> normalize -> split
static void test_nv12_facade_letterbox(cv::cuda::Stream& stream) {
    const int W = 1280, H = 720;
    const cv::Size box(320, 320);
    cv::Mat h_nv12(H + H / 2, W, CV_8UC1);
    fill_random(h_nv12, 778);
    cv::cuda::GpuMat d_nv12(h_nv12), hv_nv12 = host_view(h_nv12);
    cv::cuda::GpuMat d_out(1, box.width * box.height * 3, CV_32F);
    cv::Mat h_ref(1, box.width * box.height * 3, CV_32F);
    cv::cuda::GpuMat hv_ref = host_view(h_ref);
    const cv::Scalar a(1 / 255.0, 1 / 255.0, 1 / 255.0), s(0.485, 0.456, 0.406), d(0.229, 0.224, 0.225), pad(114, 100, 7);
    for (int bgr = 0; bgr < 2; ++bgr) {
        auto go = [&](auto read_dev, auto read_host) {
            cvGS::executeOperations(stream, cvGS::resize<cv::INTER_LINEAR, cvGS::PRESERVE_AR>(read_dev, box, pad), cvGS::multiply<CV_32FC3>(a),
                                    cvGS::subtract<CV_32FC3>(s), cvGS::divide<CV_32FC3>(d), cvGS::split<CV_32FC3>(d_out, box));
            run_oracle(cvGS::resize<cv::INTER_LINEAR, cvGS::PRESERVE_AR>(read_host, box, pad), cvGS::multiply<CV_32FC3>(a),
                       cvGS::subtract<CV_32FC3>(s), cvGS::divide<CV_32FC3>(d), cvGS::split<CV_32FC3>(hv_ref, box));
        };
        if (bgr) go(cvGS::cvtColorNV12<cv::COLOR_YUV2BGR_NV12, fk::Limited>(d_nv12), cvGS::cvtColorNV12<cv::COLOR_YUV2BGR_NV12, fk::Limited>(hv_nv12));
        else go(cvGS::cvtColorNV12<cv::COLOR_YUV2RGB_NV12, fk::Limited>(d_nv12), cvGS::cvtColorNV12<cv::COLOR_YUV2RGB_NV12, fk::Limited>(hv_nv12));
        stream.waitForCompletion();
        const auto h = fetch(d_out.data, (size_t)box.width * box.height * 3 * 4);
        CHECK(bit_equal(h.data(), h_ref.data, h.size()), (bgr ? "NV12 letterbox (BGR order), bit-exact vs oracle" : "NV12 letterbox (RGB order), bit-exact vs oracle"));
        // the padding rows carry the background pushed through the chain; backgroundValue is in the IOp's OUTPUT channel order, so plane 0,
        // row 0 is pad[0] for the RGB and for the BGR code alike
        const float want = (float)(((float)pad[0] * (float)a[0] - (float)s[0]) / (float)d[0]);
        CHECK(std::fabs(((const float*)h_ref.data)[0] - want) < 1e-5f, "letterbox padding = background through the chain");
    }
}

// the decode-side headline path: N crops of an NV12 surface -> BGR float -> 64x128 -> normalize -> NCHW, ONE kernel;
// each crop must equal the single-surface chain run on a copy of that crop (same taps, same arithmetic)
static void test_nv12_crops_batch(cv::cuda::Stream& stream) {
    const int W = 1280, H = 720;
    constexpr size_t N = 12;
    const cv::Size up(64, 128);
    cv::Mat h_nv12(H + H / 2, W, CV_8UC1);
    fill_random(h_nv12, 999);
    cv::cuda::GpuMat d_nv12(h_nv12), hv_nv12 = host_view(h_nv12);
    std::array<cv::Rect, N> crops;
    for (size_t i = 0; i < N; ++i) crops[i] = cv::Rect(2 * (int)(i * 37 % 400), 2 * (int)(i * 23 % 200), 16 + 2 * (int)(i * 41 % 300), 32 + 2 * (int)(i * 29 % 150));
    const size_t n = N * 3 * up.width * up.height;
    cv::cuda::GpuMat d_out((int)N, up.width * up.height * 3, CV_32F);
    cv::Mat h_ref((int)N, up.width * up.height * 3, CV_32F);
    cv::cuda::GpuMat hv_ref = host_view(h_ref);
    const cv::Scalar a(0.3, 0.3, 0.3), s(1.f, 4.f, 3.2f), d(3.2f, 0.6f, 11.8f);
    cvGS::executeOperations(stream, cvGS::resize<cv::INTER_LINEAR>(cvGS::cvtColorNV12<cv::COLOR_YUV2BGR_NV12>(d_nv12, crops), up),
                            cvGS::multiply<CV_32FC3>(a), cvGS::subtract<CV_32FC3>(s), cvGS::divide<CV_32FC3>(d), cvGS::split<CV_32FC3>(d_out, up));
    run_oracle(cvGS::resize<cv::INTER_LINEAR>(cvGS::cvtColorNV12<cv::COLOR_YUV2BGR_NV12>(hv_nv12, crops), up), cvGS::multiply<CV_32FC3>(a),
               cvGS::subtract<CV_32FC3>(s), cvGS::divide<CV_32FC3>(d), cvGS::split<CV_32FC3>(hv_ref, up));
    stream.waitForCompletion();
    const auto h = fetch(d_out.data, n * 4);
    CHECK(bit_equal(h.data(), h_ref.data, n * 4), "NV12 crop batch -> NCHW, bit-exact vs oracle");
    // crop 5, copied into its own small NV12 surface, through the single-surface chain (host oracle): must be identical
    const cv::Rect r = crops[5];
    cv::Mat h_small(r.height + r.height / 2, r.width, CV_8UC1);
    for (int y = 0; y < r.height; ++y) std::memcpy(h_small.ptr<uchar>(y), h_nv12.ptr<uchar>(r.y + y) + r.x, (size_t)r.width);
    for (int y = 0; y < r.height / 2; ++y) std::memcpy(h_small.ptr<uchar>(r.height + y), h_nv12.ptr<uchar>(H + r.y / 2 + y) + r.x, (size_t)r.width);
    cv::Mat h_one(1, up.width * up.height * 3, CV_32F);
    cv::cuda::GpuMat hv_small = host_view(h_small), hv_one = host_view(h_one);
    run_oracle(cvGS::resize<cv::INTER_LINEAR>(cvGS::cvtColorNV12<cv::COLOR_YUV2BGR_NV12>(hv_small), up), cvGS::multiply<CV_32FC3>(a),
               cvGS::subtract<CV_32FC3>(s), cvGS::divide<CV_32FC3>(d), cvGS::split<CV_32FC3>(hv_one, up));
    CHECK(bit_equal(h_one.data, h_ref.ptr<uchar>(5), (size_t)up.width * up.height * 3 * 4), "a crop view == the same pixels as their own surface");
}

// the same decode-side path from a 10-bit (P010) decoder surface, BT.2020 limited range: crops -> RGB on the 0..1023 scale ->
// 64x128 -> x 1/1023 -> normalize -> NCHW, ONE kernel; plus the fk:: spelling (ReadYUV<P010> fused with ConvertYUVToRGB)
static void test_p010_crops_batch(cv::cuda::Stream& stream) {
    const int W = 1280, H = 720;
    constexpr size_t N = 9;
    const cv::Size up(64, 128);
    cv::Mat h_p010(H + H / 2, W, CV_16UC1);
    fill_random(h_p010, 1234); // random 16-bit samples: 10-bit codes plus garbage in the 6 low bits
    cv::cuda::GpuMat d_p010(h_p010), hv_p010 = host_view(h_p010);
    std::array<cv::Rect, N> crops;
    for (size_t i = 0; i < N; ++i) crops[i] = cv::Rect(2 * (int)(i * 53 % 400), 2 * (int)(i * 31 % 200), 16 + 2 * (int)(i * 43 % 300), 32 + 2 * (int)(i * 19 % 150));
    const size_t n = N * 3 * up.width * up.height;
    cv::cuda::GpuMat d_out((int)N, up.width * up.height * 3, CV_32F);
    cv::Mat h_ref((int)N, up.width * up.height * 3, CV_32F);
    cv::cuda::GpuMat hv_ref = host_view(h_ref);
    const double k = 1.0 / 1023.0;
    const cv::Scalar a(k, k, k), s(0.485, 0.456, 0.406), d(0.229, 0.224, 0.225);
    cvGS::executeOperations(stream, cvGS::resize<cv::INTER_LINEAR>(cvGS::cvtColorP010<cv::COLOR_YUV2RGB_NV12>(d_p010, crops), up),
                            cvGS::multiply<CV_32FC3>(a), cvGS::subtract<CV_32FC3>(s), cvGS::divide<CV_32FC3>(d), cvGS::split<CV_32FC3>(d_out, up));
    run_oracle(cvGS::resize<cv::INTER_LINEAR>(cvGS::cvtColorP010<cv::COLOR_YUV2RGB_NV12>(hv_p010, crops), up), cvGS::multiply<CV_32FC3>(a),
               cvGS::subtract<CV_32FC3>(s), cvGS::divide<CV_32FC3>(d), cvGS::split<CV_32FC3>(hv_ref, up));
    stream.waitForCompletion();
    const auto h = fetch(d_out.data, n * 4);
    CHECK(bit_equal(h.data(), h_ref.data, n * 4), "P010 crop batch -> NCHW, bit-exact vs oracle");
    {   // the same batch through the device-side descriptor queue (engine extension): P010 surfaces are its fourth kind
        cvGS::Queue queue;
        cv::cuda::GpuMat d_out_q((int)N, up.width * up.height * 3, CV_32F);
        const uint64_t ticket = cvGS::executeOperations(queue, cvGS::resize<cv::INTER_LINEAR>(cvGS::cvtColorP010<cv::COLOR_YUV2RGB_NV12>(d_p010, crops), up),
                                                        cvGS::multiply<CV_32FC3>(a), cvGS::subtract<CV_32FC3>(s), cvGS::divide<CV_32FC3>(d),
                                                        cvGS::split<CV_32FC3>(d_out_q, up));
        queue.wait(ticket);
        const auto hq = fetch(d_out_q.data, n * 4);
        CHECK(bit_equal(hq.data(), h_ref.data, n * 4), "P010 crop batch through cvGS::executeOperations(queue, ...), bit-exact vs oracle");
    }

    // fk spelling, whole surface -> 10-bit RGBA image (ushort4): Resize(fuse(ReadYUV<P010>, ConvertYUVToRGB<P010, ..., float4>)) -> SaturateCast
    const cv::Size down(320, 180);
    fk::RawPtr<fk::_2D, ushort> dsrc, hsrc;
    dsrc.data = (ushort*)d_p010.data; dsrc.dims = {(uint)W, (uint)H, (uint)d_p010.step};
    hsrc.data = (ushort*)h_p010.data; hsrc.dims = {(uint)W, (uint)H, (uint)h_p010.step};
    cv::cuda::GpuMat d_img(down.height, down.width, CV_16UC4);
    cv::Mat h_img_ref(down.height, down.width, CV_16UC4);
    cv::cuda::GpuMat hv_img = host_view(h_img_ref);
    using Conv = fk::ConvertYUVToRGB<fk::P010, fk::Limited, fk::bt2020, true, float4>;
    const auto rd_d = fk::Resize<fk::INTER_LINEAR>::build(fk::fuse(fk::Read<fk::ReadYUV<fk::P010>>{dsrc}, fk::Unary<Conv>{}), fk::Size(down.width, down.height));
    const auto rd_h = fk::Resize<fk::INTER_LINEAR>::build(fk::fuse(fk::Read<fk::ReadYUV<fk::P010>>{hsrc}, fk::Unary<Conv>{}), fk::Size(down.width, down.height));
    cvGS::executeOperations(stream, rd_d, cvGS::convertTo<CV_32FC4, CV_16UC4>(), cvGS::write<CV_16UC4>(d_img));
    run_oracle(rd_h, cvGS::convertTo<CV_32FC4, CV_16UC4>(), cvGS::write<CV_16UC4>(hv_img));
    stream.waitForCompletion();
    cv::Mat h_img;
    d_img.download(h_img);
    bool same = true;
    for (int y = 0; y < h_img.rows; ++y) same = same && bit_equal(h_img.ptr<uchar>(y), h_img_ref.ptr<uchar>(y), (size_t)h_img.cols * 8);
    CHECK(same, "fk::ReadYUV<P010> -> ushort4 image, bit-exact vs oracle");
}

int main() {
    cv::cuda::Stream stream;
    test_fused_planar_resize_u8<fk::I420>(stream, "K4 I420 -> BGR u8 thumbnail, bit-exact vs oracle");
    test_fused_planar_resize_u8<fk::YV12>(stream, "K4 YV12 -> BGR u8 thumbnail, bit-exact vs oracle");
    test_fused_planar_resize_u8<fk::NV21>(stream, "K4 NV21 -> BGR u8 thumbnail, bit-exact vs oracle");
    test_nv12_facade_cfg3(stream);
    test_nv12_facade_letterbox(stream);
    test_nv12_crops_batch(stream);
    test_p010_crops_batch(stream);
    test_resize_split_one<CV_8UC3, CV_32FC3>(stream);
    test_resize_split_one<CV_8UC4, CV_32FC4>(stream);
    test_resize_split_one<CV_16UC3, CV_32FC3>(stream);
    test_resize_split_one<CV_16UC4, CV_32FC4>(stream);
    test_resize_split_one<CV_16SC3, CV_32FC3>(stream);
    test_resize_split_one<CV_16SC4, CV_32FC4>(stream);
    test_resize_write<CV_8UC1>(stream);
    test_resize_write<CV_8UC3>(stream);
    test_resize_write<CV_8UC4>(stream);
    test_resize_write<CV_16UC1>(stream);
    test_resize_write<CV_16UC3>(stream);
    test_resize_write<CV_16UC4>(stream);
    test_resize_write<CV_16SC1>(stream);
    test_resize_write<CV_16SC3>(stream);
    test_resize_write<CV_16SC4>(stream);
    test_resize_write<CV_32FC1>(stream);
    test_fused_nv12_resize(stream);
    return report("test_resize_x_split + test_resize_write + test_fused_resize");
}
