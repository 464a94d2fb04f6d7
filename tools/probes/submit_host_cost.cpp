// Host cost of one executeOperations(queue, ...) -- the facade call of the reference's benchmark chain (60x120 crops of a 4K frame
// -> 64x128, benchmarks/benchmark_CPU_OpenCV_vs_cvGS.cu:68-124) -- per batch size, submits back to back:
//   hipcc -O2 -std=c++17 -x c++ -Icvgpuspeedup_amd/include tools/probes/submit_host_cost.cpp -o /tmp/shc -Lcvgpuspeedup_amd/lib -lcvgs_hip -Wl,-rpath,$PWD/cvgpuspeedup_amd/lib
#include <cvGPUSpeedup.cuh>
#include <array>
#include <chrono>
#include <cstdio>

template <int BATCH> void run() {
    constexpr int TI = CV_8UC3, TO = CV_32FC3;
    cv::cuda::GpuMat frame(2160, 3840, TI, cv::Scalar(5, 5, 5));
    std::array<cv::cuda::GpuMat, BATCH> crops;
    for (int i = 0; i < BATCH; ++i) crops[i] = frame(cv::Rect2d(cv::Point2d(i, i), cv::Point2d(i + 60, i + 120)));
    const cv::Size up(64, 128);
    cv::cuda::GpuMat tensor(BATCH, up.width * up.height * 3, CV_32F);
    const cv::Scalar a(0.3, 0.3, 0.3, 0.3), sub(1.f, 4.f, 3.2f), div(3.2f, 0.6f, 11.8f);
    (void)hipDeviceSynchronize();
    cvGS::Queue queue;
    auto call = [&] {
        return cvGS::executeOperations(queue, cvGS::resize<TI, cv::INTER_LINEAR, BATCH>(crops, up, BATCH), cvGS::cvtColor<cv::COLOR_RGB2BGR, TO, TO>(),
                                       cvGS::multiply<TO>(a), cvGS::subtract<TO>(sub), cvGS::divide<TO>(div), cvGS::split<TO>(tensor, up));
    };
    for (int i = 0; i < 500; ++i) queue.wait(call());
    const int N = 20000;
    uint64_t last = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) last = call();
    const auto t1 = std::chrono::steady_clock::now();
    queue.wait(last);
    const auto t2 = std::chrono::steady_clock::now();
    std::printf("batch %3d: host %.3f us per call, sustained %.3f us per batch\n", BATCH, std::chrono::duration<double, std::micro>(t1 - t0).count() / N,
                std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
}

int main() {
    run<10>();
    run<30>();
    run<50>();
    run<70>();
    return 0;
}
