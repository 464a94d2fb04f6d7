// reference_goldens.cu -- runs on an NVIDIA box WITH the reference built (see README.md in this directory).  Written against
// the reference's public cvGS:: API only; produces the raw tensors of three seeded, NON-constant cases.
#include <cvGPUSpeedup.cuh>

#include <array>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include <opencv2/core.hpp>
#include <opencv2/core/cuda.hpp>

static uint64_t mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t splitmix(uint64_t seed, uint64_t i) { return mix(seed + i * 0x9E3779B97F4A7C15ull); } // i = 1, 2, ...

constexpr int BATCH = 50, FW = 3840, FH = 2160;

template <cvGS::AspectRatio AR>
static void run_case(const char* name, const cv::cuda::GpuMat& d_frame, const std::array<cv::Rect, BATCH>& rects, const std::string& dir) {
    const cv::Size up(64, 128);
    std::array<cv::cuda::GpuMat, BATCH> crops;
    for (int i = 0; i < BATCH; ++i) crops[i] = d_frame(rects[i]);
    cv::cuda::GpuMat d_tensor(BATCH, up.width * up.height * 3, CV_32F);
    d_tensor.step = (size_t)up.width * up.height * 3 * sizeof(float);
    cv::cuda::Stream stream;
    const cv::Scalar alpha(0.3, 0.3, 0.3), sub(1.0, 4.0, 3.2), div(3.2, 0.6, 11.8), bg(128.0, 128.0, 128.0);
    cvGS::executeOperations(stream, cvGS::resize<CV_8UC3, cv::INTER_LINEAR, BATCH, AR>(crops, up, BATCH, bg),
                            cvGS::cvtColor<cv::COLOR_RGB2BGR, CV_32FC3>(), cvGS::multiply<CV_32FC3>(alpha),
                            cvGS::subtract<CV_32FC3>(sub), cvGS::divide<CV_32FC3>(div), cvGS::split<CV_32FC3>(d_tensor, up));
    cv::Mat h(BATCH, up.width * up.height * 3, CV_32F);
    d_tensor.download(h, stream);
    stream.waitForCompletion();
    const std::string path = dir + "/" + name + ".f32";
    FILE* f = std::fopen(path.c_str(), "wb");
    for (int i = 0; i < BATCH; ++i) std::fwrite(h.ptr<float>(i), sizeof(float), (size_t)up.width * up.height * 3, f);
    std::fclose(f);
    std::printf("%s: wrote %s, first values %.9g %.9g %.9g\n", name, path.c_str(), h.at<float>(0, 0), h.at<float>(0, 1), h.at<float>(0, 2));
}

// the frame of a case: consecutive little-endian splitmix64 outputs of its seed (tests/helpers.py: random_u8)
static cv::cuda::GpuMat seeded_frame(uint64_t seed) {
    cv::Mat h_frame(FH, FW, CV_8UC3);
    const size_t n = (size_t)FW * FH * 3;
    uint8_t* p = h_frame.data; // continuous: FW * 3 bytes per row
    for (size_t w = 0; w * 8 < n; ++w) {
        const uint64_t v = splitmix(seed, w + 1);
        for (size_t b = 0; b < 8 && w * 8 + b < n; ++b) p[w * 8 + b] = (uint8_t)(v >> (8 * b));
    }
    return cv::cuda::GpuMat(h_frame);
}
// variable crops of a case: splitmix64 outputs of seed + 1 (cvgpuspeedup_amd/workloads.py: random_crops)
static std::array<cv::Rect, BATCH> seeded_crops(uint64_t seed) {
    std::array<cv::Rect, BATCH> r;
    for (int i = 0; i < BATCH; ++i) {
        const uint64_t r0 = splitmix(seed + 1, 4 * i + 1), r1 = splitmix(seed + 1, 4 * i + 2), r2 = splitmix(seed + 1, 4 * i + 3),
                       r3 = splitmix(seed + 1, 4 * i + 4);
        const int w = 32 + (int)(r0 % 481), hgt = 64 + (int)(r1 % 961);
        r[i] = cv::Rect((int)(r2 % (uint64_t)(FW - w + 1)), (int)(r3 % (uint64_t)(FH - hgt + 1)), w, hgt);
    }
    return r;
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    std::array<cv::Rect, BATCH> fixed;
    for (int i = 0; i < BATCH; ++i) fixed[i] = cv::Rect(i, i, 60, 120);
    // (case, seed) as in tests/golden/seeded_fixtures.json
    run_case<cvGS::IGNORE_AR>("k1_cfg2a_fixed", seeded_frame(0xC0FFEEull), fixed, dir);
    run_case<cvGS::IGNORE_AR>("k1_cfg2b_variable", seeded_frame(0xC0FFF5ull), seeded_crops(0xC0FFF5ull), dir);
    run_case<cvGS::PRESERVE_AR>("k1_cfg2b_preserve_ar", seeded_frame(0xC0FFF7ull), seeded_crops(0xC0FFF7ull), dir);
    return 0;
}
