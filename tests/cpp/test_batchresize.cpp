// test_batchresize.cpp -- mirrors reference tests/batchresize/test_batchresize_x_split3D.cu and
// test_batchresize_aspectratio_x_split3D.cu on the cvGS facade: N crops of a 4K frame -> resize -> cvtColor ->
// multiply -> subtract -> divide -> split into an NCHW tensor, ONE fused kernel.
//  (1) the reference's constant-colour known answers (tolerance 1e-4, as the reference),
//  (2) the same chain on a NON-constant frame, bit-exact against the CPU oracle,
//  (3) facade result == hand-built C-ABI descriptor result (the reference's cvGS == fk check,
//      benchmarks/benchmark_CPUandGPU_cvGS_vs_fk.cu:191-192).
#include "common.h"

#include <memory>

struct Params { cv::Scalar init, alpha, sub, div; };
static const double kAlpha = 0.3;
static const Params kParams[4] = {
    {{2}, {kAlpha}, {1.f}, {3.2f}},
    {{2, 37}, {kAlpha, kAlpha}, {1.f, 4.f}, {3.2f, 0.6f}},
    {{5, 5, 5}, {kAlpha, kAlpha, kAlpha}, {1.f, 4.f, 3.2f}, {3.2f, 0.6f, 11.8f}},
    {{2, 37, 128, 20}, {kAlpha, kAlpha, kAlpha, kAlpha}, {1.f, 4.f, 3.2f, 0.5f}, {3.2f, 0.6f, 11.8f, 33.f}}};

template <int TI, int TO, int BATCH, cvGS::AspectRatio AR>
static auto build_chain(const std::array<cv::cuda::GpuMat, BATCH>& crops, const cv::cuda::GpuMat& tensor, const cv::Size& up,
                        const Params& p) {
    const auto rd = cvGS::resize<TI, cv::INTER_LINEAR, BATCH, AR>(crops, up, BATCH, cvGS::cvScalar_set<TO>(128.f));
    if constexpr (CV_MAT_CN(TI) == 3)
        return std::make_tuple(rd, cvGS::cvtColor<cv::COLOR_RGB2BGR, TO>(), cvGS::multiply<TO>(p.alpha), cvGS::subtract<TO>(p.sub),
                               cvGS::divide<TO>(p.div), cvGS::split<TO>(tensor, up));
    else
        return std::make_tuple(rd, cvGS::cvtColor<cv::COLOR_RGBA2BGRA, TO>(), cvGS::multiply<TO>(p.alpha), cvGS::subtract<TO>(p.sub),
                               cvGS::divide<TO>(p.div), cvGS::split<TO>(tensor, up));
}

template <int TI, int TO, int BATCH, cvGS::AspectRatio AR>
static void test_constant(cv::cuda::Stream& stream, int cropW) {
    constexpr int CN = CV_MAT_CN(TO);
    const Params& p = kParams[CN - 1];
    const cv::Size up(64, 128);
    cv::cuda::GpuMat d_input(2160, 3840, TI, p.init);
    std::array<cv::cuda::GpuMat, BATCH> crops;
    for (int i = 0; i < BATCH; ++i) crops[i] = d_input(cv::Rect2d(cv::Point2d(i, i), cv::Point2d(i + cropW, i + 120)));
    cv::cuda::GpuMat d_tensor(BATCH, up.width * up.height * CN, CV_32F);
    d_tensor.step = (size_t)up.width * up.height * CN * sizeof(float);

    std::apply([&](const auto&... iops) { cvGS::executeOperations(stream, iops...); }, build_chain<TI, TO, BATCH, AR>(crops, d_tensor, up, p));
    stream.waitForCompletion();

    const auto h = fetch(d_tensor.data, (size_t)BATCH * CN * up.width * up.height * sizeof(float));
    const float* t = (const float*)h.data();
    // expected constants: channel swap (0 <-> 2), x alpha, - sub, / div; outside the AR window the background 128
    int x1 = 0, x2 = 63;
    if (AR != cvGS::IGNORE_AR) { x1 = 16; x2 = 47; } // 30x120 -> 32x128 centred
    bool ok = true;
    for (int z = 0; z < BATCH && ok; ++z)
        for (int c = 0; c < CN && ok; ++c) {
            const int sc = (c == 0) ? 2 : (c == 2 ? 0 : c);
            const double in = ((double)p.init[sc] * (float)kAlpha - (float)p.sub[c]) / (float)p.div[c];
            const double out = (128.0 * (float)kAlpha - (float)p.sub[c]) / (float)p.div[c];
            for (int y = 0; y < 128 && ok; ++y)
                for (int x = 0; x < 64 && ok; ++x) {
                    const float v = t[(((size_t)z * CN + c) * 128 + y) * 64 + x];
                    const double e = (x >= x1 && x <= x2) ? in : out;
                    if (std::fabs(v - e) > 1e-4) {
                        std::cout << "    z=" << z << " c=" << c << " y=" << y << " x=" << x << ": " << v << " vs " << e << std::endl;
                        ok = false;
                    }
                }
        }
    CHECK(ok, "constant-colour known answer, type " << TI << " batch " << BATCH);
}

template <int TI, int TO, int BATCH, cvGS::AspectRatio AR>
static void test_random_vs_oracle(cv::cuda::Stream& stream) {
    constexpr int CN = CV_MAT_CN(TO);
    const Params& p = kParams[CN - 1];
    const cv::Size up(64, 128);
    cv::Mat h_frame(1080, 1920, TI);
    fill_random(h_frame, 0xC0FFEEull + TI);
    cv::cuda::GpuMat d_frame(h_frame);
    cv::cuda::GpuMat hv_frame = host_view(h_frame);
    std::array<cv::cuda::GpuMat, BATCH> crops, h_crops;
    for (int i = 0; i < BATCH; ++i) {
        const int w = 8 + (i * 37) % 500, hgt = 16 + (i * 53) % 900, x = (i * 91) % (1920 - w), y = (i * 67) % (1080 - hgt);
        crops[i] = d_frame(cv::Rect(x, y, w, hgt));
        h_crops[i] = hv_frame(cv::Rect(x, y, w, hgt));
    }
    const size_t n = (size_t)BATCH * CN * up.width * up.height;
    cv::cuda::GpuMat d_tensor(BATCH, up.width * up.height * CN, CV_32F);
    d_tensor.step = (size_t)up.width * up.height * CN * sizeof(float);
    cv::Mat h_ref(BATCH, up.width * up.height * CN, CV_32F);
    cv::cuda::GpuMat hv_ref = host_view(h_ref);

    std::apply([&](const auto&... iops) { cvGS::executeOperations(stream, iops...); }, build_chain<TI, TO, BATCH, AR>(crops, d_tensor, up, p));
    std::apply([&](const auto&... iops) { run_oracle(iops...); }, build_chain<TI, TO, BATCH, AR>(h_crops, hv_ref, up, p));
    stream.waitForCompletion();
    const auto h = fetch(d_tensor.data, n * sizeof(float));
    CHECK(bit_equal(h.data(), h_ref.data, n * sizeof(float)), "non-constant frame, bit-exact vs oracle, type " << TI << " AR " << AR);

    // facade == hand-built C-ABI descriptor
    cvgs_chain_desc d;
    std::memset(&d, 0, sizeof(d));
    d.struct_size = sizeof(d);
    std::vector<cvgs_image2d> planes(BATCH);
    for (int i = 0; i < BATCH; ++i) planes[i] = cvgs_image2d{crops[i].data, crops[i].cols, crops[i].rows, (int32_t)crops[i].step, 0};
    d.read.kind = CVGS_READ_RESIZE_LINEAR; d.read.src_type = TI; d.read.batch = BATCH; d.read.used_planes = BATCH;
    d.read.src = planes.data(); d.read.dst_width = 64; d.read.dst_height = 128; d.read.aspect_ratio = (int)AR;
    for (int c = 0; c < CN; ++c) d.read.background[c] = 128.f;
    d.n_ops = 4;
    d.ops[0].opcode = CVGS_OP_REORDER; d.ops[0].aux = CN == 3 ? (2 | (1 << 2)) : (2 | (1 << 2) | (3 << 6));
    d.ops[1].opcode = CVGS_OP_MUL; d.ops[2].opcode = CVGS_OP_SUB; d.ops[3].opcode = CVGS_OP_DIV;
    for (int c = 0; c < CN; ++c) { d.ops[1].operand[c] = (float)p.alpha[c]; d.ops[2].operand[c] = (float)p.sub[c]; d.ops[3].operand[c] = (float)p.div[c]; }
    cv::cuda::GpuMat d_raw(BATCH, up.width * up.height * CN, CV_32F);
    d.write.kind = CVGS_WRITE_TENSOR_SPLIT; d.write.dst_type = TO; d.write.data = d_raw.data; d.write.width = 64; d.write.height = 128; d.write.planes = BATCH;
    CHECK(cvgs_execute(&d, stream.raw()) == CVGS_OK, "raw C-ABI launch: " << cvgs_last_error());
    stream.waitForCompletion();
    const auto hr = fetch(d_raw.data, n * sizeof(float));
    CHECK(bit_equal(h.data(), hr.data(), n * sizeof(float)), "facade == raw C-ABI descriptor");
}

// half-precision hand-off (engine extension): the same chain + convertTo<CV_32FCn, CV_16FCn> -> fp16 NCHW
template <int TI, int BATCH>
static void test_half_handoff(cv::cuda::Stream& stream) {
    constexpr int CN = CV_MAT_CN(TI);
    constexpr int TF = CV_MAKETYPE(CV_32F, CN), TH = CV_MAKETYPE(CV_16F, CN);
    const Params& p = kParams[CN - 1];
    const cv::Size up(64, 128);
    cv::Mat h_frame(720, 1280, TI);
    fill_random(h_frame, 0xC0FFEEull + 1600 + TI);
    cv::cuda::GpuMat d_frame(h_frame);
    cv::cuda::GpuMat hv_frame = host_view(h_frame);
    std::array<cv::cuda::GpuMat, BATCH> crops, h_crops;
    for (int i = 0; i < BATCH; ++i) {
        const int w = 4 + (i * 41) % 400, hgt = 9 + (i * 59) % 600, x = (i * 97) % (1280 - w), y = (i * 71) % (720 - hgt);
        crops[i] = d_frame(cv::Rect(x, y, w, hgt));
        h_crops[i] = hv_frame(cv::Rect(x, y, w, hgt));
    }
    const size_t n = (size_t)BATCH * CN * up.width * up.height;
    cv::cuda::GpuMat d_tensor(BATCH, up.width * up.height * CN, CV_16F);
    cv::Mat h_ref(BATCH, up.width * up.height * CN, CV_16F);
    cv::cuda::GpuMat hv_ref = host_view(h_ref);
    auto chain = [&](const std::array<cv::cuda::GpuMat, BATCH>& in, const cv::cuda::GpuMat& out) {
        return std::make_tuple(cvGS::resize<TI, cv::INTER_LINEAR, BATCH, cvGS::IGNORE_AR>(in, up, BATCH, cvGS::cvScalar_set<TF>(0.f)),
                               cvGS::multiply<TF>(p.alpha), cvGS::subtract<TF>(p.sub), cvGS::divide<TF>(p.div),
                               cvGS::convertTo<TF, TH>(), cvGS::split<TH>(out, up));
    };
    std::apply([&](const auto&... iops) { cvGS::executeOperations(stream, iops...); }, chain(crops, d_tensor));
    std::apply([&](const auto&... iops) { run_oracle(iops...); }, chain(h_crops, hv_ref));
    stream.waitForCompletion();
    const auto h = fetch(d_tensor.data, n * 2);
    CHECK(bit_equal(h.data(), h_ref.data, n * 2), "fp16 NCHW hand-off, bit-exact vs oracle, type " << TI);
}

// the read-composition spelling (reference include/cvGPUSpeedup.cuh:204-207,247-265,444-447, never exercised by its
// tests): read(frame).then(crop(rects)).then(resize<INTER_LINEAR>(size)) must be the same K1 launch as resize(array of ROIs)
static void test_then_spelling(cv::cuda::Stream& stream) {
    constexpr int BATCH = 12;
    const cv::Size up(64, 128);
    cv::Mat h_frame(720, 1280, CV_8UC3);
    fill_random(h_frame, 0xC0FFEEull + 4242);
    cv::cuda::GpuMat d_frame(h_frame);
    std::array<cv::Rect2d, BATCH> rects;
    std::array<cv::cuda::GpuMat, BATCH> crops;
    for (int i = 0; i < BATCH; ++i) {
        rects[i] = cv::Rect2d(3.7 + i * 31, 2.2 + i * 17, 40.9 + i * 23, 55.5 + i * 29); // doubles truncate
        crops[i] = d_frame(rects[i]);
    }
    const size_t n = (size_t)BATCH * 3 * up.width * up.height;
    cv::cuda::GpuMat d_a(BATCH, up.width * up.height * 3, CV_32F), d_b(BATCH, up.width * up.height * 3, CV_32F);
    const cv::Scalar sub(1.f, 4.f, 3.2f);
    cvGS::executeOperations(stream, cvGS::resize<CV_8UC3, cv::INTER_LINEAR, BATCH>(crops, up, BATCH), cvGS::subtract<CV_32FC3>(sub),
                            cvGS::split<CV_32FC3>(d_a, up));
    const fk::Read<fk::PerThreadRead<fk::_2D, uchar3>> read{cvGS::gpuMat2RawPtr2D<uchar3>(d_frame)};
    cvGS::executeOperations(stream, cvGS::crop(read, rects).then(cvGS::resize<cv::INTER_LINEAR>(up)), cvGS::subtract<CV_32FC3>(sub),
                            cvGS::split<CV_32FC3>(d_b, up));
    stream.waitForCompletion();
    const auto a = fetch(d_a.data, n * sizeof(float)), b = fetch(d_b.data, n * sizeof(float));
    CHECK(bit_equal(a.data(), b.data(), n * sizeof(float)), "read.then(crop).then(resize) == resize(array of ROIs)");
    // single crop + single resize
    cv::cuda::GpuMat d_c(1, up.width * up.height * 3, CV_32F), d_d(1, up.width * up.height * 3, CV_32F);
    cvGS::executeOperations(stream, cvGS::resize<CV_8UC3, cv::INTER_LINEAR>(crops[5], up, 0., 0.), cvGS::split<CV_32FC3>(d_c, up));
    cvGS::executeOperations(stream, cvGS::crop(read, rects[5]).then(cvGS::resize<cv::INTER_LINEAR>(up)), cvGS::split<CV_32FC3>(d_d, up));
    stream.waitForCompletion();
    const auto c = fetch(d_c.data, n / BATCH * sizeof(float)), d = fetch(d_d.data, n / BATCH * sizeof(float));
    CHECK(bit_equal(c.data(), d.data(), c.size()), "read.then(crop(rect)).then(resize) == resize(ROI)");
}

// the pointer adapters of reference include/cvGPUSpeedup.cuh:34-65 on an array of crops
static void test_pointer_adapters() {
    cv::cuda::GpuMat d_frame(240, 320, CV_8UC3, cv::Scalar(1, 2, 3));
    std::array<cv::cuda::GpuMat, 3> crops = {d_frame(cv::Rect(0, 0, 10, 20)), d_frame(cv::Rect(5, 7, 100, 50)), d_frame(cv::Rect(300, 200, 20, 40))};
    const auto owning = cvGS::gpuMat2Ptr2D_arr<uchar3, 3>(crops);
    const auto raw = cvGS::gpuMat2RawPtr2D_arr<uchar3>(crops);
    bool ok = true;
    for (size_t i = 0; i < 3; ++i) {
        const auto p = owning[i].ptr();
        ok = ok && p.data == (uchar3*)crops[i].data && p.dims.width == (uint)crops[i].cols && p.dims.height == (uint)crops[i].rows &&
             p.dims.pitch == (uint)crops[i].step && raw[i].data == p.data && raw[i].dims.pitch == p.dims.pitch;
    }
    CHECK(ok, "gpuMat2Ptr2D_arr / gpuMat2RawPtr2D_arr describe the crops (no copy)");
}

// cvGS::ChainBatch (cvgs_execute_many): four cameras' 50-crop chains in one launch == four executeOperations calls
static void test_chain_batch(cv::cuda::Stream& stream) {
    constexpr int CAMS = 4, N = 50;
    const cv::Size up(64, 128);
    const cv::Scalar mul(0.3, 0.3, 0.3), sub(1, 4, 3.2), div(3.2, 0.6, 11.8);
    std::vector<cv::cuda::GpuMat> frames, outs_a, outs_b;
    std::vector<std::array<cv::cuda::GpuMat, N>> crops(CAMS);
    for (int c = 0; c < CAMS; ++c) {
        cv::Mat h(720, 1280, CV_8UC3);
        fill_random(h, 900 + c);
        cv::cuda::GpuMat d(720, 1280, CV_8UC3);
        d.upload(h);
        frames.push_back(d);
        for (int i = 0; i < N; ++i) crops[c][i] = frames[c](cv::Rect(3 * i + c, 2 * i, 40 + 7 * i, 60 + 9 * i));
        outs_a.emplace_back(N, up.width * up.height * 3, CV_32F);
        outs_b.emplace_back(N, up.width * up.height * 3, CV_32F);
    }
    cvGS::ChainBatch batch;
    for (int c = 0; c < CAMS; ++c) {
        batch.add(cvGS::resize<CV_8UC3, cv::INTER_LINEAR, N>(crops[c], up, N), cvGS::cvtColor<cv::COLOR_RGB2BGR, CV_32FC3>(),
                  cvGS::multiply<CV_32FC3>(mul), cvGS::subtract<CV_32FC3>(sub), cvGS::divide<CV_32FC3>(div), cvGS::split<CV_32FC3>(outs_a[c], up));
        cvGS::executeOperations(stream, cvGS::resize<CV_8UC3, cv::INTER_LINEAR, N>(crops[c], up, N), cvGS::cvtColor<cv::COLOR_RGB2BGR, CV_32FC3>(),
                                cvGS::multiply<CV_32FC3>(mul), cvGS::subtract<CV_32FC3>(sub), cvGS::divide<CV_32FC3>(div), cvGS::split<CV_32FC3>(outs_b[c], up));
    }
    batch.execute(stream);
    stream.waitForCompletion();
    bool ok = batch.size() == CAMS;
    const size_t n = (size_t)N * up.width * up.height * 3 * sizeof(float);
    for (int c = 0; c < CAMS && ok; ++c) {
        const auto a = fetch(outs_a[c].data, n), b = fetch(outs_b[c].data, n);
        ok = bit_equal(a.data(), b.data(), n);
    }
    CHECK(ok, "ChainBatch of 4 x 50 crops == 4 executeOperations");
    // the same batch on a stream attached to a queue: the four chains go behind ONE gate on the stream (cvgs_queue_submit_many_on),
    // ordered behind the memsets in front of them -- same bits
    cvGS::Queue queue;
    cvGS::attachQueue(stream, queue, /*deferWait=*/false, /*minGroup=*/4); // (the default policy takes ticks of 8 chains and more)
    for (int c = 0; c < CAMS; ++c)
        HIP_OK(hipMemsetAsync(outs_a[c].data, 0xff, n, cv::cuda::StreamAccessor::getStream(stream)));
    batch.execute(stream);
    std::vector<std::vector<uint8_t>> got((size_t)CAMS, std::vector<uint8_t>(n));
    for (int c = 0; c < CAMS; ++c) // copies ON THE STREAM: ordered behind the batch by the gate kernel's wait
        HIP_OK(hipMemcpyAsync(got[(size_t)c].data(), outs_a[c].data, n, hipMemcpyDeviceToHost, cv::cuda::StreamAccessor::getStream(stream)));
    stream.waitForCompletion();
    cvGS::detachQueue(stream);
    ok = true;
    for (int c = 0; c < CAMS && ok; ++c) {
        const auto b = fetch(outs_b[c].data, n);
        ok = bit_equal(got[(size_t)c].data(), b.data(), n);
    }
    CHECK(ok, "ChainBatch on a stream attached to a queue (one gate for the four chains) == 4 executeOperations");
}

template <int TI, int TO>
static void sweep(cv::cuda::Stream& stream) {
    test_constant<TI, TO, 10, cvGS::IGNORE_AR>(stream, 60);
    test_constant<TI, TO, 30, cvGS::IGNORE_AR>(stream, 60);
    test_constant<TI, TO, 50, cvGS::IGNORE_AR>(stream, 60);
    test_constant<TI, TO, 20, cvGS::PRESERVE_AR>(stream, 30);
    test_constant<TI, TO, 50, cvGS::PRESERVE_AR>(stream, 30);
    test_random_vs_oracle<TI, TO, 50, cvGS::IGNORE_AR>(stream);
    test_random_vs_oracle<TI, TO, 24, cvGS::PRESERVE_AR>(stream);
}

// engine extension: the device-side descriptor queue -- executeOperations(queue, iops...) must give the bits of the stream form
template <int TI, int TO, int BATCH>
static void test_queue_vs_oracle(cv::cuda::Stream& stream) {
    constexpr int CN = CV_MAT_CN(TO);
    const Params& p = kParams[CN - 1];
    const cv::Size up(64, 128);
    cvGS::Queue queue;
    const int FRAMES = 6;
    std::vector<cv::Mat> h_frames;
    std::vector<cv::cuda::GpuMat> d_frames, d_tensors;
    std::vector<uint64_t> tickets;
    std::vector<std::array<cv::Rect, BATCH>> rects(FRAMES);
    for (int f = 0; f < FRAMES; ++f) {
        h_frames.emplace_back(720, 1280, TI);
        fill_random(h_frames.back(), 0xC0FFEEull + 7000 + f);
        d_frames.emplace_back(h_frames.back());
        cv::cuda::GpuMat t(BATCH, up.width * up.height * CN, CV_32F);
        d_tensors.push_back(t);
        for (int i = 0; i < BATCH; ++i) {
            const int w = 8 + ((i + f) * 37) % 400, hgt = 16 + ((i + 2 * f) * 53) % 600;
            rects[f][i] = cv::Rect(((i + f) * 91) % (1280 - w), (i * 67) % (720 - hgt), w, hgt);
        }
    }
    stream.waitForCompletion(); // uploads done: the queue is not ordered behind any stream
    for (int f = 0; f < FRAMES; ++f) { // six frames in flight, no wait in between
        std::array<cv::cuda::GpuMat, BATCH> crops;
        for (int i = 0; i < BATCH; ++i) crops[i] = d_frames[f](rects[f][i]);
        tickets.push_back(std::apply([&](const auto&... iops) { return cvGS::executeOperations(queue, iops...); },
                                     build_chain<TI, TO, BATCH, cvGS::IGNORE_AR>(crops, d_tensors[f], up, p)));
    }
    const size_t n = (size_t)BATCH * CN * up.width * up.height;
    for (int f = 0; f < FRAMES; ++f) {
        queue.wait(tickets[f]);
        cv::cuda::GpuMat hv_frame = host_view(h_frames[f]);
        std::array<cv::cuda::GpuMat, BATCH> h_crops;
        for (int i = 0; i < BATCH; ++i) h_crops[i] = hv_frame(rects[f][i]);
        cv::Mat h_ref(BATCH, up.width * up.height * CN, CV_32F);
        cv::cuda::GpuMat hv_ref = host_view(h_ref);
        std::apply([&](const auto&... iops) { run_oracle(iops...); }, build_chain<TI, TO, BATCH, cvGS::IGNORE_AR>(h_crops, hv_ref, up, p));
        const auto h = fetch(d_tensors[f].data, n * sizeof(float));
        CHECK(bit_equal(h.data(), h_ref.data, n * sizeof(float)), "queue: frame " << f << " bit-exact vs oracle, type " << TI);
    }
    // a chain the server does not take throws like any unsupported chain
    bool threw = false;
    try {
        cv::cuda::GpuMat out(up.height, up.width, TO);
        cvGS::executeOperations(queue, cvGS::resize<TI, cv::INTER_LINEAR>(d_frames[0], up, 0., 0.), cvGS::write<TO>(out));
    } catch (const std::exception&) {
        threw = true;
    }
    CHECK(threw, "queue refuses a chain that is not the batched resize -> normalize -> split shape");
}

// engine extension (ABI 5): the reference's OWN call shape -- cvGS::executeOperations(stream, iops...), include/cvGPUSpeedup.cuh:464-473 --
// on streams attached to a queue.  Per iteration and stream: a copy on the stream REWRITES the frame buffer, executeOperations follows
// with no synchronisation, a copy on the stream reads the tensor back; two streams share the queue so that batches overlap on the server
// (a lone batch takes the direct launch: the hybrid policy -- both paths must give the oracle's bits).
template <int TI, int TO, int BATCH>
static void test_attached_streams_vs_oracle() {
    constexpr int CN = CV_MAT_CN(TO);
    const Params& p = kParams[CN - 1];
    const cv::Size up(64, 128);
    const int STREAMS = 2, POOL = 4, ITERS = 120, RING = 4;
    const size_t n = (size_t)BATCH * CN * up.width * up.height;
    cvGS::Queue queue(0, 0, 5000.0);
    struct Cam {
        cv::cuda::Stream stream;
        std::vector<cv::Mat> h_pool;
        std::vector<cv::cuda::GpuMat> d_pool;
        std::vector<cv::Mat> refs;
        cv::cuda::GpuMat frame, tensor;
        std::array<cv::Rect, BATCH> rects;
        float* ring[4] = {nullptr, nullptr, nullptr, nullptr};
        hipEvent_t ev[4];
    };
    std::vector<std::unique_ptr<Cam>> cams;
    for (int c = 0; c < STREAMS; ++c) {
        cams.emplace_back(new Cam);
        Cam& cam = *cams.back();
        cam.frame = cv::cuda::GpuMat(720, 1280, TI);
        cam.tensor = cv::cuda::GpuMat(BATCH, up.width * up.height * CN, CV_32F);
        for (int i = 0; i < BATCH; ++i) {
            const int w = 8 + ((i + c) * 37) % 400, hgt = 16 + ((i + 2 * c) * 53) % 600;
            cam.rects[i] = cv::Rect(((i + c) * 91) % (1280 - w), (i * 67) % (720 - hgt), w, hgt);
        }
        for (int k = 0; k < POOL; ++k) {
            cam.h_pool.emplace_back(720, 1280, TI);
            fill_random(cam.h_pool.back(), 0xC0FFEEull + 9000 + 10 * c + k);
            cam.d_pool.emplace_back(cam.h_pool.back());
            cv::cuda::GpuMat hv_frame = host_view(cam.h_pool.back());
            std::array<cv::cuda::GpuMat, BATCH> h_crops;
            for (int i = 0; i < BATCH; ++i) h_crops[i] = hv_frame(cam.rects[i]);
            cam.refs.emplace_back(BATCH, up.width * up.height * CN, CV_32F);
            cv::cuda::GpuMat hv_ref = host_view(cam.refs.back());
            std::apply([&](const auto&... iops) { run_oracle(iops...); }, build_chain<TI, TO, BATCH, cvGS::IGNORE_AR>(h_crops, hv_ref, up, p));
        }
        for (int r = 0; r < RING; ++r) {
            HIP_OK(hipHostMalloc((void**)&cam.ring[r], n * sizeof(float), hipHostMallocDefault));
            HIP_OK(hipEventCreateWithFlags(&cam.ev[r], hipEventDisableTiming));
        }
        cvGS::attachQueue(cam.stream, queue, /*deferWait=*/false, /*minGroup=*/c == 0 ? 1 : 0); // stream 0: every call may go to the server; stream 1: the default policy (launches)
    }
    HIP_OK(hipDeviceSynchronize());
    int bad = 0;
    auto check_slot = [&](Cam& cam, int it) {
        HIP_OK(hipEventSynchronize(cam.ev[it % RING]));
        if (!bit_equal(cam.ring[it % RING], cam.refs[(size_t)(it % POOL)].data, n * sizeof(float))) ++bad;
    };
    for (int it = 0; it < ITERS; ++it)
        for (auto& cp : cams) {
            Cam& cam = *cp;
            hipStream_t s = cv::cuda::StreamAccessor::getStream(cam.stream);
            if (it >= RING) check_slot(cam, it - RING); // (the host lags four iterations behind: nothing orders producer -> chain -> consumer but the stream)
            HIP_OK(hipMemcpy2DAsync(cam.frame.data, cam.frame.step, cam.d_pool[(size_t)(it % POOL)].data, cam.d_pool[(size_t)(it % POOL)].step,
                                    (size_t)cam.frame.cols * cam.frame.elemSize(), (size_t)cam.frame.rows, hipMemcpyDeviceToDevice, s));
            std::array<cv::cuda::GpuMat, BATCH> crops;
            for (int i = 0; i < BATCH; ++i) crops[i] = cam.frame(cam.rects[i]);
            std::apply([&](const auto&... iops) { cvGS::executeOperations(cam.stream, iops...); },
                       build_chain<TI, TO, BATCH, cvGS::IGNORE_AR>(crops, cam.tensor, up, p));
            HIP_OK(hipMemcpyAsync(cam.ring[it % RING], cam.tensor.data, n * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_OK(hipEventRecord(cam.ev[it % RING], s));
        }
    for (auto& cp : cams)
        for (int it = ITERS - RING; it < ITERS; ++it) check_slot(*cp, it);
    CHECK(bad == 0, "executeOperations(stream, ...) on streams attached to a queue: " << bad << " of " << STREAMS * ITERS << " tensors differ from the oracle, type " << TI);
    for (auto& cp : cams) {
        cvGS::detachQueue(cp->stream);
        for (int r = 0; r < RING; ++r) {
            (void)hipHostFree(cp->ring[r]);
            (void)hipEventDestroy(cp->ev[r]);
        }
    }
}

// engine extension (ABI 5, recorded ticks): the reference's multi-camera loop UNCHANGED -- one executeOperations(stream, ...) per camera,
// one waitForCompletion() per tick -- on a stream attached with attachQueueTicks.  CAMS cameras and a tick of 16: the calls go behind
// gates 16 at a time and the rest at the fence (fewer than 8 left: launches, the hybrid policy) -- every tensor must have the oracle's bits,
// with the frames REWRITTEN on the stream between ticks.
template <int TI, int TO, int BATCH, int CAMS>
static void test_recorded_ticks_vs_oracle(bool fence_then_async, bool with_queue = true) {
    constexpr int CN = CV_MAT_CN(TO);
    const Params& p = kParams[CN - 1];
    const cv::Size up(64, 128);
    const int POOL = 3, TICKS = 12;
    const size_t n = (size_t)BATCH * CN * up.width * up.height;
    std::unique_ptr<cvGS::Queue> queue(with_queue ? new cvGS::Queue(0, 0, 5000.0) : nullptr); // (without: cvGS::recordTicks, one cvgs_execute_many launch per tick)
    cv::cuda::Stream stream;
    hipStream_t s = cv::cuda::StreamAccessor::getStream(stream);
    struct Cam {
        std::vector<cv::cuda::GpuMat> d_pool;
        std::vector<cv::Mat> refs;
        cv::cuda::GpuMat frame, tensor;
        std::array<cv::Rect, BATCH> rects;
        float* host = nullptr;
    };
    std::vector<std::unique_ptr<Cam>> cams;
    for (int c = 0; c < CAMS; ++c) {
        cams.emplace_back(new Cam);
        Cam& cam = *cams.back();
        cam.frame = cv::cuda::GpuMat(360, 640, TI);
        cam.tensor = cv::cuda::GpuMat(BATCH, up.width * up.height * CN, CV_32F);
        for (int i = 0; i < BATCH; ++i) {
            const int w = 8 + ((i + c) * 37) % 300, hgt = 16 + ((i + 2 * c) * 53) % 300;
            cam.rects[i] = cv::Rect(((i + c) * 91) % (640 - w), (i * 67 + c) % (360 - hgt), w, hgt);
        }
        for (int k = 0; k < POOL; ++k) {
            cv::Mat h(360, 640, TI);
            fill_random(h, 0xABCDull + 100 * c + k);
            cam.d_pool.emplace_back(h);
            cv::cuda::GpuMat hv_frame = host_view(h);
            std::array<cv::cuda::GpuMat, BATCH> h_crops;
            for (int i = 0; i < BATCH; ++i) h_crops[i] = hv_frame(cam.rects[i]);
            cam.refs.emplace_back(BATCH, up.width * up.height * CN, CV_32F);
            cv::cuda::GpuMat hv_ref = host_view(cam.refs.back());
            std::apply([&](const auto&... iops) { run_oracle(iops...); }, build_chain<TI, TO, BATCH, cvGS::IGNORE_AR>(h_crops, hv_ref, up, p));
        }
        HIP_OK(hipHostMalloc((void**)&cam.host, n * sizeof(float), hipHostMallocDefault));
    }
    if (with_queue) cvGS::attachQueueTicks(stream, *queue, 16);
    else cvGS::recordTicks(stream, 16);
    int bad = 0;
    for (int t = 0; t < TICKS; ++t) {
        for (auto& cp : cams) { // the producers: the frames of this tick arrive on the stream (the previous tick was fenced)
            Cam& cam = *cp;
            const cv::cuda::GpuMat& src = cam.d_pool[(size_t)(t % POOL)];
            HIP_OK(hipMemcpy2DAsync(cam.frame.data, cam.frame.step, src.data, src.step, (size_t)cam.frame.cols * cam.frame.elemSize(),
                                    (size_t)cam.frame.rows, hipMemcpyDeviceToDevice, s));
        }
        for (auto& cp : cams) {
            Cam& cam = *cp;
            std::array<cv::cuda::GpuMat, BATCH> crops;
            for (int i = 0; i < BATCH; ++i) crops[i] = cam.frame(cam.rects[i]);
            std::apply([&](const auto&... iops) { cvGS::executeOperations(stream, iops...); },
                       build_chain<TI, TO, BATCH, cvGS::IGNORE_AR>(crops, cam.tensor, up, p));
        } // (crops -- the GpuMat headers -- die here: the recorded chains own their descriptors)
        if (fence_then_async) {
            cvGS::fence(stream); // consumers enqueued from here on follow the tick
            for (auto& cp : cams) HIP_OK(hipMemcpyAsync(cp->host, cp->tensor.data, n * sizeof(float), hipMemcpyDeviceToHost, s));
            HIP_OK(hipStreamSynchronize(s));
        } else {
            stream.waitForCompletion(); // the reference's synchronisation: submits what is pending, fences, waits
            for (auto& cp : cams) HIP_OK(hipMemcpy(cp->host, cp->tensor.data, n * sizeof(float), hipMemcpyDeviceToHost));
        }
        for (auto& cp : cams)
            if (!bit_equal(cp->host, cp->refs[(size_t)(t % POOL)].data, n * sizeof(float))) ++bad;
    }
    CHECK(bad == 0, "recorded ticks (" << (with_queue ? "attachQueueTicks, " : "recordTicks with no queue, ") << CAMS << " cameras, " << (fence_then_async ? "fence + async consumers" : "waitForCompletion")
                        << "): " << bad << " of " << CAMS * TICKS << " tensors differ from the oracle, type " << TI);
    // recorded calls are not lost at detach: record, detach (submits), synchronise, compare
    for (auto& cp : cams) HIP_OK(hipMemsetAsync(cp->tensor.data, 0, n * sizeof(float), s));
    for (auto& cp : cams) {
        Cam& cam = *cp;
        std::array<cv::cuda::GpuMat, BATCH> crops;
        for (int i = 0; i < BATCH; ++i) crops[i] = cam.frame(cam.rects[i]);
        std::apply([&](const auto&... iops) { cvGS::executeOperations(stream, iops...); },
                   build_chain<TI, TO, BATCH, cvGS::IGNORE_AR>(crops, cam.tensor, up, p));
    }
    cvGS::fence(stream);
    cvGS::detachQueue(stream);
    HIP_OK(hipStreamSynchronize(s));
    int bad2 = 0;
    for (auto& cp : cams) {
        HIP_OK(hipMemcpy(cp->host, cp->tensor.data, n * sizeof(float), hipMemcpyDeviceToHost));
        if (!bit_equal(cp->host, cp->refs[(size_t)((TICKS - 1) % POOL)].data, n * sizeof(float))) ++bad2;
        (void)hipHostFree(cp->host);
    }
    CHECK(bad2 == 0, "recorded ticks: fence + detach leave " << bad2 << " tensors wrong");
    // a stream that dies takes its attachment (and submits what it recorded): record on a scoped stream, let it go, then the tensors must be there
    {
        for (auto& cp : cams) HIP_OK(hipMemset(cp->tensor.data, 0, n * sizeof(float)));
        cv::cuda::Stream scoped;
        if (with_queue) cvGS::attachQueueTicks(scoped, *queue, 64);
        else cvGS::recordTicks(scoped, 64);
        for (auto& cp : cams) {
            Cam& cam = *cp;
            std::array<cv::cuda::GpuMat, BATCH> crops;
            for (int i = 0; i < BATCH; ++i) crops[i] = cam.frame(cam.rects[i]);
            std::apply([&](const auto&... iops) { cvGS::executeOperations(scoped, iops...); },
                       build_chain<TI, TO, BATCH, cvGS::IGNORE_AR>(crops, cam.tensor, up, p));
        }
    } // ~Stream: flush + detach + hipStreamDestroy (which completes the stream's work)
    HIP_OK(hipDeviceSynchronize());
    int bad3 = 0;
    std::vector<float> back(n);
    for (auto& cp : cams) {
        HIP_OK(hipMemcpy(back.data(), cp->tensor.data, n * sizeof(float), hipMemcpyDeviceToHost));
        if (!bit_equal(back.data(), cp->refs[(size_t)((TICKS - 1) % POOL)].data, n * sizeof(float))) ++bad3;
    }
    CHECK(bad3 == 0, "recorded calls of a stream that was destroyed: " << bad3 << " tensors wrong");
    uint64_t tk = 0;
    cv::cuda::Stream fresh; // (may get the dead stream's handle: it must not inherit an attachment)
    CHECK(!cvGS::lastTicket(fresh, &tk), "a new stream starts unattached");
}

int main() {
    cv::cuda::Stream stream;
    // the type list of the reference's LAUNCH_TESTS (test_batchresize_x_split3D.cu:427-432)
    sweep<CV_8UC3, CV_32FC3>(stream);
    sweep<CV_8UC4, CV_32FC4>(stream);
    sweep<CV_16UC3, CV_32FC3>(stream);
    sweep<CV_16UC4, CV_32FC4>(stream);
    sweep<CV_16SC3, CV_32FC3>(stream);
    sweep<CV_16SC4, CV_32FC4>(stream);
    test_then_spelling(stream);
    test_pointer_adapters();
    test_chain_batch(stream);
    test_half_handoff<CV_8UC3, 50>(stream);
    test_half_handoff<CV_8UC4, 17>(stream);
    test_queue_vs_oracle<CV_8UC3, CV_32FC3, 50>(stream);
    test_queue_vs_oracle<CV_8UC4, CV_32FC4, 9>(stream);
    test_queue_vs_oracle<CV_16UC3, CV_32FC3, 50>(stream); // the 16-bit kind: a queue of its own (one kind per queue)
    test_queue_vs_oracle<CV_16SC4, CV_32FC4, 11>(stream);
    test_queue_vs_oracle<CV_8UC3, CV_32FC3, 100>(stream); // more crops than a ring slot holds (74): two slots behind one ticket
    test_attached_streams_vs_oracle<CV_8UC3, CV_32FC3, 40>();
    test_attached_streams_vs_oracle<CV_8UC4, CV_32FC4, 9>();
    test_recorded_ticks_vs_oracle<CV_8UC3, CV_32FC3, 20, 27>(/*fence_then_async=*/false); // 16 behind a gate + 11 at the fence
    test_recorded_ticks_vs_oracle<CV_8UC4, CV_32FC4, 7, 21>(/*fence_then_async=*/true);   // 16 behind a gate + 5 launches at the fence
    test_recorded_ticks_vs_oracle<CV_8UC3, CV_32FC3, 20, 27>(/*fence_then_async=*/false, /*with_queue=*/false); // one launch per 16 + one for 11
    test_recorded_ticks_vs_oracle<CV_8UC4, CV_32FC4, 7, 21>(/*fence_then_async=*/true, /*with_queue=*/false);
    return report("test_batchresize_x_split3D + aspectratio");
}
