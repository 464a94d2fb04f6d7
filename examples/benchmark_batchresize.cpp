// benchmark_batchresize.cpp -- the reference's own benchmark of the headline path, restated on this engine's facade:
// tests/batchresize/test_batchresize_x_split3D.cu built with ENABLE_BENCHMARK sweeps BATCH = 10, 20, ..., 300 crops
// (60x120 at (i,i) of a 4K frame -> 64x128, x0.3, RGB2BGR for 3 channels, subtract, divide, split into the [BATCH,C,128,64]
// tensor), six type pairs, 1 warm-up + ITERS = 100 iterations, an event pair around EVERY call (tests/testsCommon.cuh:122-195),
// and writes one CSV row per batch: mean / variance / max / min of the per-call time in milliseconds.  This program prints the
// same table for the cvGS columns (there is no OpenCV-CUDA on this platform to fill the other half).  The per-call time of an
// event pair around one eager call includes the launch latency -- that is what the reference's CSV holds too; the
// graph-replayed, back-to-back figure of the same kernel is bench.py's.  The last column is the reference's OTHER benchmark of
// this chain, benchmarks/benchmark_CPU_OpenCV_vs_cvGS.cu: the HOST time of the cvGS::executeOperations call itself (a
// steady_clock pair around the call, no synchronisation inside) -- building the IOps, lowering, cvgs_execute, hipLaunchKernel.
//   make -C examples && ./examples/bin/benchmark_batchresize > benchmark_batchresize_x_split3D.csv
#include <cvGPUSpeedup.cuh>

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <utility>

namespace {
constexpr int ITERS = 100;     // reference: tests/testsCommon.cuh:128
constexpr int FIRST = 10, STEP = 10, EXPERIMENTS = 30; // reference: test_batchresize_x_split3D.cu:384-392 (CUDA 12, benchmark build)
constexpr int FRAME_W = 3840, FRAME_H = 2160, CROP_W = 60, CROP_H = 120;

struct Stats { float mean, variance, max, min; };

Stats summarize(const std::array<float, ITERS>& ms) {
    Stats s{0.f, 0.f, ms[0], ms[0]};
    for (float v : ms) { s.mean += v; s.max = std::max(s.max, v); s.min = std::min(s.min, v); }
    s.mean /= ITERS;
    for (float v : ms) s.variance += (v - s.mean) * (v - s.mean);
    s.variance /= (ITERS - 1); // the reference's computeVariance: sample variance
    return s;
}

template <int TI, int TO> const char* pair_name() {
    if (TI == CV_8UC3) return "CV_8UC3XCV_32FC3";
    if (TI == CV_8UC4) return "CV_8UC4XCV_32FC4";
    if (TI == CV_16UC3) return "CV_16UC3XCV_32FC3";
    if (TI == CV_16UC4) return "CV_16UC4XCV_32FC4";
    if (TI == CV_16SC3) return "CV_16SC3XCV_32FC3";
    return "CV_16SC4XCV_32FC4";
}

template <int TI, int TO, int BATCH>
void one_batch(cv::cuda::Stream& stream, hipEvent_t e0, hipEvent_t e1) {
    constexpr int CN = CV_MAT_CN(TO);
    const cv::Scalar init[4] = {{2}, {2, 37}, {5, 5, 5}, {2, 37, 128, 20}};
    const cv::Scalar sub[4] = {{1.f}, {1.f, 4.f}, {1.f, 4.f, 3.2f}, {1.f, 4.f, 3.2f, 0.5f}};
    const cv::Scalar div[4] = {{3.2f}, {3.2f, 0.6f}, {3.2f, 0.6f, 11.8f}, {3.2f, 0.6f, 11.8f, 33.f}};
    const double alpha = 0.3;
    cv::cuda::GpuMat frame(FRAME_H, FRAME_W, TI, init[CN - 1]);
    std::array<cv::cuda::GpuMat, BATCH> crops;
    for (int i = 0; i < BATCH; ++i) crops[i] = frame(cv::Rect2d(cv::Point2d(i, i), cv::Point2d(i + CROP_W, i + CROP_H)));
    const cv::Size up(64, 128);
    cv::cuda::GpuMat tensor(BATCH, up.width * up.height * CN, CV_MAT_DEPTH(TO));
    const cv::Scalar a(alpha, alpha, alpha, alpha);

    auto call = [&] {
        if constexpr (CN == 3)
            cvGS::executeOperations(stream, cvGS::resize<TI, cv::INTER_LINEAR, BATCH>(crops, up, BATCH), cvGS::cvtColor<cv::COLOR_RGB2BGR, TO, TO>(),
                                    cvGS::multiply<TO>(a), cvGS::subtract<TO>(sub[CN - 1]), cvGS::divide<TO>(div[CN - 1]), cvGS::split<TO>(tensor, up));
        else
            cvGS::executeOperations(stream, cvGS::resize<TI, cv::INTER_LINEAR, BATCH>(crops, up, BATCH), cvGS::cvtColor<cv::COLOR_RGBA2BGRA, TO, TO>(),
                                    cvGS::multiply<TO>(a), cvGS::subtract<TO>(sub[CN - 1]), cvGS::divide<TO>(div[CN - 1]), cvGS::split<TO>(tensor, up));
    };
    std::array<float, ITERS> ms{};
    double host_ms = 0.0;
    for (int it = -1; it < ITERS; ++it) { // it == -1: the warm-up iteration (ITERS_W = 1)
        (void)hipEventRecord(e0, stream.raw());
        const auto h0 = std::chrono::steady_clock::now();
        call();
        const auto h1 = std::chrono::steady_clock::now();
        if (it >= 0) host_ms += std::chrono::duration<double, std::milli>(h1 - h0).count();
        (void)hipEventRecord(e1, stream.raw());
        stream.waitForCompletion();
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0, e1);
        if (it >= 0) ms[(size_t)it] = t;
    }
    const Stats s = summarize(ms);
    std::printf("%d, %.6f, %.3e, %.6f, %.6f, %.1f, %.6f", BATCH, s.mean, s.variance, s.max, s.min,
                (double)BATCH * up.width * up.height / (s.mean * 1e-3) / 1e6, host_ms / ITERS);
    // engine extension: the same call on the device-side descriptor queue (executeOperations(queue, iops...): no kernel launch per
    // call).  QITERS calls back to back, one wait at the end: the sustained time per batch and the host time of the call itself.
    { // (a ring slot holds 74 planes: larger batches go out as consecutive slots behind one ticket)
        constexpr int QITERS = 2000;
        cvGS::Queue queue;
        auto qcall = [&] {
            if constexpr (CN == 3)
                return cvGS::executeOperations(queue, cvGS::resize<TI, cv::INTER_LINEAR, BATCH>(crops, up, BATCH), cvGS::cvtColor<cv::COLOR_RGB2BGR, TO, TO>(),
                                               cvGS::multiply<TO>(a), cvGS::subtract<TO>(sub[CN - 1]), cvGS::divide<TO>(div[CN - 1]), cvGS::split<TO>(tensor, up));
            else
                return cvGS::executeOperations(queue, cvGS::resize<TI, cv::INTER_LINEAR, BATCH>(crops, up, BATCH), cvGS::cvtColor<cv::COLOR_RGBA2BGRA, TO, TO>(),
                                               cvGS::multiply<TO>(a), cvGS::subtract<TO>(sub[CN - 1]), cvGS::divide<TO>(div[CN - 1]), cvGS::split<TO>(tensor, up));
        };
        stream.waitForCompletion();
        for (int i = 0; i < 200; ++i) queue.wait(qcall()); // server up, code and descriptors warm
        double qhost_ms = 0.0;
        uint64_t last = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < QITERS; ++i) {
            const auto h0 = std::chrono::steady_clock::now();
            last = qcall();
            qhost_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
        }
        queue.wait(last);
        const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf(", %.6f, %.6f\n", total_ms / QITERS, qhost_ms / QITERS);
    }
}

template <int TI, int TO, size_t... Is>
void sweep(cv::cuda::Stream& stream, hipEvent_t e0, hipEvent_t e1, std::index_sequence<Is...>) {
    std::printf("BATCH (%s), cvGS MeanTime [ms], cvGS TimeVariance, cvGS MaxTime [ms], cvGS MinTime [ms], output Mpix/s at the mean, cvGS CPU MeanTime [ms], queue sustained [ms per batch], queue CPU MeanTime [ms]\n", pair_name<TI, TO>());
    (one_batch<TI, TO, FIRST + STEP * (int)Is>(stream, e0, e1), ...);
}
} // namespace

int main() {
    cv::cuda::Stream stream;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    constexpr auto seq = std::make_index_sequence<EXPERIMENTS>{};
    sweep<CV_8UC3, CV_32FC3>(stream, e0, e1, seq);
    sweep<CV_8UC4, CV_32FC4>(stream, e0, e1, seq);
    sweep<CV_16UC3, CV_32FC3>(stream, e0, e1, seq);
    sweep<CV_16UC4, CV_32FC4>(stream, e0, e1, seq);
    sweep<CV_16SC3, CV_32FC3>(stream, e0, e1, seq);
    sweep<CV_16SC4, CV_32FC4>(stream, e0, e1, seq);
    return 0;
}
