// readme_example.cpp -- the reference README's 50-detection example (reference README.md:104-134), compiled
// unchanged in shape against this engine's facade, plus the reference's benchmark protocol in small
// (tests/testsCommon.cuh:122-195: 1 warm-up, ITERS timed iterations, event timing on the stream, mean/min/max).
//   make -C examples && ./examples/bin/readme_example
#include <cvGPUSpeedup.cuh>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

int main() {
    constexpr int MAX_DETECTIONS = 50;
    constexpr int ITERS = 100;
    cv::cuda::Stream stream;

    // a 4K frame and 50 detections of assorted sizes (ROI views: no copies)
    cv::Mat h_frame(2160, 3840, CV_8UC3, cv::Scalar(50, 100, 150));
    cv::cuda::GpuMat frame(h_frame);
    std::array<cv::cuda::GpuMat, MAX_DETECTIONS> crops;
    for (int i = 0; i < MAX_DETECTIONS; ++i)
        crops[i] = frame(cv::Rect(40 * i, 20 * i, 60 + 8 * i, 120 + 16 * i));

    const cv::Scalar subtract_val(1, 4, 6), divide_val(255, 255, 255);
    const cv::Size resDims(64, 128);
    cv::cuda::GpuMat output(MAX_DETECTIONS, resDims.width * resDims.height * CV_MAT_CN(CV_32FC3), CV_32FC1);
    const double alpha = 0.5;
    const int activeDetections = 50;

    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<float> us;
    for (int it = -1; it < ITERS; ++it) { // it == -1: warm-up
        (void)hipEventRecord(e0, stream.raw());
        // single kernel -- the README's call verbatim (README.md:123-130), its convertTo<CV_8UC3, CV_32FC3>() behind the resize and its
        // `substract` included: the resize already yields CV_32FC3, the facade drops that one redundant cast at compile time
        cvGS::executeOperations(stream,
                                cvGS::resize<CV_8UC3, cv::INTER_LINEAR, MAX_DETECTIONS>(crops, resDims, activeDetections),
                                cvGS::convertTo<CV_8UC3, CV_32FC3>(),
                                cvGS::multiply<CV_32FC3>(cv::Scalar(alpha, alpha, alpha)),
                                cvGS::substract<CV_32FC3>(subtract_val),
                                cvGS::divide<CV_32FC3>(divide_val),
                                cvGS::split<CV_32FC3>(output, resDims));
        (void)hipEventRecord(e1, stream.raw());
        stream.waitForCompletion();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it >= 0) us.push_back(ms * 1000.f);
    }
    float mean = 0.f;
    for (float v : us) mean += v;
    mean /= (float)us.size();
    // host enqueue time per call: what the reference's "CPU" benchmark measures (benchmarks/benchmark_CPU_OpenCV_vs_cvGS.cu:
    // the time the host spends building the IOps and launching, not the kernel).  256 back-to-back calls, no sync.
    constexpr int CALLS = 256; // fewer than the queue holds: the host is never throttled by the device
    stream.waitForCompletion();
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < CALLS; ++it)
        cvGS::executeOperations(stream,
                                cvGS::resize<CV_8UC3, cv::INTER_LINEAR, MAX_DETECTIONS>(crops, resDims, activeDetections),
                                cvGS::multiply<CV_32FC3>(cv::Scalar(alpha, alpha, alpha)),
                                cvGS::subtract<CV_32FC3>(subtract_val),
                                cvGS::divide<CV_32FC3>(divide_val),
                                cvGS::split<CV_32FC3>(output, resDims));
    const auto t1 = std::chrono::steady_clock::now();
    stream.waitForCompletion();
    const auto t2 = std::chrono::steady_clock::now();
    std::printf("host enqueue: %.2f us per cvGS::executeOperations call (C++: build IOps + lower + cvgs_execute + hipLaunchKernel); "
                "%.2f us per call including the device drain\n",
                std::chrono::duration<double, std::micro>(t1 - t0).count() / CALLS,
                std::chrono::duration<double, std::micro>(t2 - t0).count() / CALLS);
    cv::Mat h_out;
    output.download(h_out);
    // constant frame: every pixel of plane c equals (init[c] * 0.5 - sub[c]) / 255
    const float expect0 = (50.f * 0.5f - 1.f) / 255.f;
    std::printf("cvGS::executeOperations (50 crops -> [50,3,128,64] fp32): mean %.2f us, min %.2f us, max %.2f us over %d iterations "
                "(event pair around each call: includes launch latency)\n",
                mean, *std::min_element(us.begin(), us.end()), *std::max_element(us.begin(), us.end()), ITERS);
    std::printf("output[0][0][0][0] = %.7f (expected %.7f)\n", h_out.at<float>(0, 0), expect0);
    return std::fabs(h_out.at<float>(0, 0) - expect0) < 1e-4f ? 0 : 1;
}
