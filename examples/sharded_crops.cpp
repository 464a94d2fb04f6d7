// sharded_crops.cpp -- BASELINE cfg #5 from a C++ host, ONE process driving every GPU of the node through the C-ABI ("single
// process, ncclCommInitAll"; the reference itself is single-GPU, include/cvGPUSpeedup.cuh:605-610 is its only device selector):
//   every GPU g owns one resident 6K frame and CROPS = 64 crops of it; the step's tensor [G*64, 3, 128, 64] fp32 lives on EVERY GPU;
//   GPU g's ONE fused kernel (cvGS::executeOperations: crop -> resize -> RGB2BGR -> x0.3 -> -sub -> /div -> split, the K1 chain of
//   tests/batchresize/test_batchresize_x_split3D.cu:311-314) writes rows [64g, 64g+64) of its copy, then
//     leg A  an in-place RCCL all-gather per GPU inside cvgs_group_start / cvgs_group_end (include/cvgs_rccl.h) completes every copy;
//     leg B  no collective: the kernel itself stores its rows into every peer's copy through P2P-mapped pointers
//            (cvgs_write_desc.mirrors after cvgs_peer_enable), one device synchronisation per step.
// Both legs must leave, on every GPU, the tensor one GPU computes alone from all frames -- checked bit for bit.  Runs with G = 1 on a
// one-GPU box (a one-rank communicator, no mirrors).
//   make -C examples && ./examples/bin/sharded_crops [--gpus G] [--iters K]
#include <cvGPUSpeedup.cuh>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/cvgs_rccl.h"

namespace sharded {

constexpr int CROPS = 64;          // per GPU (BASELINE cfg #5)
constexpr int FRAME_W = 6144, FRAME_H = 3456;
constexpr int DST_W = 64, DST_H = 128;
constexpr size_t ROW_FLOATS = (size_t)3 * DST_W * DST_H; // one crop's planes: 24,576 floats
constexpr size_t SLICE_BYTES = (size_t)CROPS * ROW_FLOATS * sizeof(float); // 6,291,456 B per GPU and step

inline uint64_t splitmix(uint64_t& s) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline void fill(cv::Mat& m, uint64_t seed) {
    uint64_t s = seed;
    const size_t row_bytes = (size_t)m.cols * m.elemSize();
    for (int y = 0; y < m.rows; ++y) {
        uint8_t* row = m.data + (size_t)y * m.step;
        for (size_t i = 0; i < row_bytes; i += 8) {
            const uint64_t z = splitmix(s);
            std::memcpy(row + i, &z, row_bytes - i < 8 ? row_bytes - i : 8);
        }
    }
}
// cfg #2 / #5's crop distribution: w ~ U[32, 512], h ~ U[64, 1024], anywhere inside the frame
inline std::array<cv::Rect, CROPS> crop_list(uint64_t seed) {
    std::array<cv::Rect, CROPS> r;
    uint64_t s = seed;
    for (int i = 0; i < CROPS; ++i) {
        const int w = 32 + (int)(splitmix(s) % 481), h = 64 + (int)(splitmix(s) % 961);
        const int x = (int)(splitmix(s) % (uint64_t)(FRAME_W - w + 1)), y = (int)(splitmix(s) % (uint64_t)(FRAME_H - h + 1));
        r[(size_t)i] = cv::Rect(x, y, w, h);
    }
    return r;
}

#define SH_HIP(call)                                                                                     \
    do {                                                                                                 \
        const hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_)); return false; } \
    } while (0)
#define SH_RCCL(call)                                                                                    \
    do {                                                                                                 \
        if ((call) != 0) { std::fprintf(stderr, "%s -> %s\n", #call, cvgs_rccl_last_error()); return false; } \
    } while (0)

// the K1 chain into rows [first_row, first_row + CROPS) of `tensor` (a device pointer valid on the current device)
template <typename Exec>
inline void k1_chain(const std::array<cv::cuda::GpuMat, CROPS>& crops, float* tensor, int first_row, Exec&& exec) {
    const cv::Size dst(DST_W, DST_H);
    cv::cuda::GpuMat rows(CROPS, (int)ROW_FLOATS, CV_32FC1, tensor + (size_t)first_row * ROW_FLOATS);
    exec(cvGS::resize<CV_8UC3, cv::INTER_LINEAR, CROPS>(crops, dst, CROPS), cvGS::cvtColor<cv::COLOR_RGB2BGR, CV_32FC3>(),
         cvGS::multiply<CV_32FC3>(cv::Scalar(0.3, 0.3, 0.3)), cvGS::subtract<CV_32FC3>(cv::Scalar(1, 4, 3.2)),
         cvGS::divide<CV_32FC3>(cv::Scalar(3.2, 0.6, 11.8)), cvGS::split<CV_32FC3>(rows, dst));
}

struct Result {
    int n_dev = 0;
    std::vector<cv::Mat> frames;                         // host copies: what a checker needs
    std::vector<std::array<cv::Rect, CROPS>> rects;
    std::vector<float> single;                           // the tensor GPU 0 computes alone from every frame
    bool allgather_equal = false, mirrors_equal = false; // every GPU's copy == single, bit for bit
    int rccl_ranks = 0;
    bool p2p_ran = false;
    double us_compute = 0, us_allgather = 0, us_mirrors = 0; // per step, wall clock, all GPUs
};

inline bool run(int want_gpus, int iters, Result& R) {
    int have = cvgs_device_count();
    if (have < 1) { std::fprintf(stderr, "no HIP device\n"); return false; }
    const int G = want_gpus > 0 && want_gpus < have ? want_gpus : have;
    R.n_dev = G;
    const size_t full_floats = (size_t)G * CROPS * ROW_FLOATS, full_bytes = full_floats * sizeof(float);
    std::vector<hipStream_t> streams((size_t)G);
    std::vector<cv::cuda::GpuMat> frames((size_t)G);
    std::vector<std::array<cv::cuda::GpuMat, CROPS>> crops((size_t)G);
    std::vector<float*> full((size_t)G, nullptr);
    std::vector<int32_t> devices((size_t)G);
    for (int g = 0; g < G; ++g) {
        devices[(size_t)g] = g;
        SH_HIP(hipSetDevice(g));
        SH_HIP(hipStreamCreateWithFlags(&streams[(size_t)g], hipStreamNonBlocking));
        R.frames.emplace_back(FRAME_H, FRAME_W, CV_8UC3);
        fill(R.frames.back(), 0xC0FFEEull + 1000ull * (uint64_t)g);
        frames[(size_t)g].upload(R.frames.back()); // allocated on GPU g
        R.rects.push_back(crop_list(0xC0FFEEull + 500000ull + (uint64_t)g));
        for (int i = 0; i < CROPS; ++i) crops[(size_t)g][(size_t)i] = frames[(size_t)g](R.rects.back()[(size_t)i]);
        SH_HIP(hipMalloc((void**)&full[(size_t)g], full_bytes));
        SH_HIP(hipMemset(full[(size_t)g], 0, full_bytes));
    }
    auto sync_all = [&]() -> bool {
        for (int g = 0; g < G; ++g) { SH_HIP(hipSetDevice(g)); SH_HIP(hipStreamSynchronize(streams[(size_t)g])); }
        return true;
    };
    auto launch = [&](int g) { // GPU g's fused kernel into its rows of its own copy
        (void)hipSetDevice(g);
        cv::cuda::Stream s = cv::cuda::StreamAccessor::wrapStream(streams[(size_t)g]);
        k1_chain(crops[(size_t)g], full[(size_t)g], g * CROPS, [&](const auto&... iops) { cvGS::executeOperations(s, iops...); });
    };

    // ---- the tensor ONE GPU computes alone: every frame uploaded to GPU 0, G chains into one tensor ----
    {
        SH_HIP(hipSetDevice(0));
        float* ref = nullptr;
        SH_HIP(hipMalloc((void**)&ref, full_bytes));
        cv::cuda::Stream s0 = cv::cuda::StreamAccessor::wrapStream(streams[0]);
        for (int g = 0; g < G; ++g) {
            cv::cuda::GpuMat f0;
            f0.upload(R.frames[(size_t)g]);
            std::array<cv::cuda::GpuMat, CROPS> c0;
            for (int i = 0; i < CROPS; ++i) c0[(size_t)i] = f0(R.rects[(size_t)g][(size_t)i]);
            k1_chain(c0, ref, g * CROPS, [&](const auto&... iops) { cvGS::executeOperations(s0, iops...); });
            SH_HIP(hipStreamSynchronize(streams[0])); // f0 goes out of scope
        }
        R.single.resize(full_floats);
        SH_HIP(hipMemcpy(R.single.data(), ref, full_bytes, hipMemcpyDeviceToHost));
        SH_HIP(hipFree(ref));
    }
    auto every_copy_equals_single = [&](bool& equal) -> bool {
        std::vector<float> h(full_floats);
        equal = true;
        for (int g = 0; g < G; ++g) {
            SH_HIP(hipSetDevice(g));
            SH_HIP(hipMemcpy(h.data(), full[(size_t)g], full_bytes, hipMemcpyDeviceToHost));
            equal = equal && std::memcmp(h.data(), R.single.data(), full_bytes) == 0;
        }
        return true;
    };
    auto clear = [&]() -> bool {
        for (int g = 0; g < G; ++g) { SH_HIP(hipSetDevice(g)); SH_HIP(hipMemsetAsync(full[(size_t)g], 0, full_bytes, streams[(size_t)g])); }
        return sync_all();
    };
    using clock = std::chrono::steady_clock;
    auto us_per_step = [&](clock::time_point t0) { return std::chrono::duration<double, std::micro>(clock::now() - t0).count() / iters; };

    // ---- compute only ----
    for (int g = 0; g < G; ++g) launch(g);
    if (!sync_all()) return false;
    auto t0 = clock::now();
    for (int k = 0; k < iters; ++k)
        for (int g = 0; g < G; ++g) launch(g);
    if (!sync_all()) return false;
    R.us_compute = us_per_step(t0);

    // ---- leg A: in-place RCCL all-gather (ncclCommInitAll: one communicator per GPU in this process) ----
    std::vector<cvgs_comm_t> comms((size_t)G, nullptr);
    SH_RCCL(cvgs_comm_init_all(comms.data(), G, devices.data()));
    R.rccl_ranks = cvgs_comm_size(comms[0]);
    auto step_allgather = [&]() -> bool {
        for (int g = 0; g < G; ++g) launch(g);
        SH_RCCL(cvgs_group_start());
        for (int g = 0; g < G; ++g) {
            (void)hipSetDevice(g);
            SH_RCCL(cvgs_allgather_inplace(comms[(size_t)g], full[(size_t)g], SLICE_BYTES, streams[(size_t)g]));
        }
        SH_RCCL(cvgs_group_end());
        return true;
    };
    if (!clear() || !step_allgather() || !sync_all() || !every_copy_equals_single(R.allgather_equal)) return false;
    t0 = clock::now();
    for (int k = 0; k < iters; ++k)
        if (!step_allgather()) return false;
    if (!sync_all()) return false;
    R.us_allgather = us_per_step(t0);
    for (int g = 0; g < G; ++g) (void)cvgs_comm_destroy(comms[(size_t)g]);

    // ---- leg B: P2P fused write -- the kernel stores its rows into every peer's copy; no collective ----
    bool p2p = true;
    for (int g = 0; g < G && p2p; ++g)
        for (int p = 0; p < G && p2p; ++p)
            if (p != g) p2p = cvgs_peer_can_access(g, p) == 1 && cvgs_peer_enable(g, p) == 0;
    R.p2p_ran = p2p;
    if (p2p) {
        auto step_mirrors = [&]() -> bool {
            for (int g = 0; g < G; ++g) {
                (void)hipSetDevice(g);
                fk::ChainBuilder b; // the facade's lowering, then the one field it has no spelling for: the peers' copies of the same rows
                k1_chain(crops[(size_t)g], full[(size_t)g], g * CROPS, [&](const auto&... iops) { fk::lowerChain(b, iops...); });
                void* mirrors[CVGS_MAX_MIRRORS];
                int n = 0;
                for (int p = 0; p < G; ++p)
                    if (p != g) mirrors[n++] = full[(size_t)p] + (size_t)g * CROPS * ROW_FLOATS;
                b.d.write.mirrors = n ? mirrors : nullptr;
                b.d.write.n_mirrors = n;
                if (cvgs_execute(&b.d, streams[(size_t)g]) != CVGS_OK) { std::fprintf(stderr, "cvgs_execute: %s\n", cvgs_last_error()); return false; }
            }
            return sync_all(); // the step's barrier: every GPU's kernel has finished, so every copy is complete
        };
        if (!clear() || !step_mirrors() || !every_copy_equals_single(R.mirrors_equal)) return false;
        t0 = clock::now();
        for (int k = 0; k < iters; ++k)
            if (!step_mirrors()) return false;
        R.us_mirrors = us_per_step(t0);
    }
    for (int g = 0; g < G; ++g) {
        (void)hipSetDevice(g);
        (void)hipFree(full[(size_t)g]);
        (void)hipStreamDestroy(streams[(size_t)g]);
    }
    return true;
}

} // namespace sharded

#ifndef SHARDED_CROPS_NO_MAIN
int main(int argc, char** argv) {
    int gpus = 0, iters = 50;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!std::strcmp(argv[i], "--gpus")) gpus = std::atoi(argv[i + 1]);
        if (!std::strcmp(argv[i], "--iters")) iters = std::atoi(argv[i + 1]);
    }
    sharded::Result R;
    if (!sharded::run(gpus, iters, R)) return 2;
    const double px = (double)R.n_dev * sharded::CROPS * sharded::DST_W * sharded::DST_H;
    std::printf("sharded_crops: gpus_seen %d, rccl_ranks_seen %d, crops per GPU %d, tensor [%d,3,128,64]\n", R.n_dev, R.rccl_ranks, sharded::CROPS, R.n_dev * sharded::CROPS);
    std::printf("  compute only            %8.2f us per step  (%.0f Mpix/s)\n", R.us_compute, px / R.us_compute);
    std::printf("  + RCCL in-place gather  %8.2f us per step  (%.0f Mpix/s)  every copy == one-GPU tensor: %s\n", R.us_allgather, px / R.us_allgather,
                R.allgather_equal ? "bit for bit" : "MISMATCH");
    if (R.p2p_ran)
        std::printf("  P2P fused write         %8.2f us per step  (%.0f Mpix/s)  every copy == one-GPU tensor: %s\n", R.us_mirrors, px / R.us_mirrors,
                    R.mirrors_equal ? "bit for bit" : "MISMATCH");
    else
        std::printf("  P2P fused write         skipped: no peer access between the devices\n");
    return R.allgather_equal && (!R.p2p_ran || R.mirrors_equal) ? 0 : 1;
}
#endif
