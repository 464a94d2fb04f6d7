// test_sharded.cpp -- BASELINE cfg #5 from ONE C++ process through the C-ABI (examples/sharded_crops.cpp: ncclCommInitAll + in-place RCCL
// all-gather, and the P2P fused write), with the CPU oracle as the checker: the tensor one GPU computes alone from every frame must be the
// oracle's bit for bit, and every GPU's gathered / mirrored copy must be that tensor.  Runs on however many GPUs the box shows (1 on the
// test boxes: a one-rank communicator, no mirrors -- the same code path a node takes with G = 8).
#define SHARDED_CROPS_NO_MAIN
#include "../../examples/sharded_crops.cpp"

#include "common.h"

int main() {
    sharded::Result R;
    const bool ran = sharded::run(0, 3, R);
    CHECK(ran, "sharded::run");
    if (ran) {
        CHECK(R.n_dev == cvgs_device_count(), "every visible GPU takes part");
        CHECK(R.rccl_ranks == R.n_dev, "one RCCL rank per GPU (cvgs_comm_init_all)");
        CHECK(R.allgather_equal, "RCCL in-place all-gather: every GPU's copy equals the one-GPU tensor bit for bit");
        CHECK(!R.p2p_ran || R.mirrors_equal, "P2P fused write: every GPU's copy equals the one-GPU tensor bit for bit");
        CHECK(R.n_dev == 1 ? R.p2p_ran : true, "one GPU: the mirror leg runs (with no mirrors)");
        // the one-GPU tensor against the oracle, frame by frame (host views of the same frames and rectangles)
        std::vector<float> ref((size_t)R.n_dev * sharded::CROPS * sharded::ROW_FLOATS, 0.f);
        for (int g = 0; g < R.n_dev; ++g) {
            cv::cuda::GpuMat hv = host_view(R.frames[(size_t)g]);
            std::array<cv::cuda::GpuMat, sharded::CROPS> c;
            for (int i = 0; i < sharded::CROPS; ++i) c[(size_t)i] = hv(R.rects[(size_t)g][(size_t)i]);
            sharded::k1_chain(c, ref.data(), g * sharded::CROPS, [&](const auto&... iops) { run_oracle(iops...); });
        }
        CHECK(bit_equal(ref.data(), R.single.data(), ref.size() * sizeof(float)), "the sharded tensor is the oracle's, bit for bit");
        std::cout << "  gpus " << R.n_dev << ", compute " << R.us_compute << " us, + all-gather " << R.us_allgather << " us, P2P " << R.us_mirrors << " us per step" << std::endl;
    }
    return report("test_sharded");
}
