// test_circulartensor.cpp -- mirrors reference tests/batchread/test_circularbatchread_x_write3D.cu:176-460 on the facade:
// cvGS::CircularTensor in its four tested configurations; after ITERS updates with value i+1, slot z must hold
// ITERS - age(z).  Plus the "push with resize + normalize" form of BASELINE cfg #4 against the CPU oracle.
#include "common.h"

constexpr uint BATCH = 15, WIDTH = 128, HEIGHT = 128;
constexpr int ITERS = 100;

template <int IT, int OT>
static void testCircularTensorcvGS() {
    constexpr uint CP = CV_MAT_CN(IT);
    cvGS::CircularTensor<IT, CV_MAT_DEPTH(OT), CP, BATCH, fk::CircularTensorOrder::NewestFirst> myTensor(WIDTH, HEIGHT);
    cv::cuda::Stream cv_stream;
    cv::cuda::GpuMat input(HEIGHT, WIDTH, IT);
    for (int i = 0; i < ITERS; ++i) {
        input.setTo(cv::Scalar(i + 1, i + 1, i + 1));
        myTensor.update(cv_stream, input, fk::Unary<fk::SaturateCast<CUDA_T(IT), CUDA_T(OT)>>{}, fk::Write<fk::TensorSplit<CUDA_T(OT)>>{myTensor.ptr()});
        cv_stream.waitForCompletion();
    }
    const auto h = fetch(myTensor.data(), myTensor.sizeInBytes());
    const float* t = (const float*)h.data();
    bool ok = true;
    const size_t plane = (size_t)WIDTH * HEIGHT;
    for (uint z = 0; z < BATCH && ok; ++z)
        for (uint c = 0; c < CP && ok; ++c) ok = all_close<float>(t + ((size_t)z * CP + c) * plane, plane, ITERS - (int)z);
    CHECK(ok, "CircularTensor NewestFirst Standard: slot z == ITERS - z");
}

template <int IT, int OT, fk::CircularTensorOrder ORDER>
static void testTransposedCircularTensorcvGS() {
    constexpr uint CP = CV_MAT_CN(IT);
    cvGS::CircularTensor<IT, CV_MAT_DEPTH(OT), CP, BATCH, ORDER, fk::ColorPlanes::Transposed> myTensor(WIDTH, HEIGHT);
    cv::cuda::Stream cv_stream;
    cv::cuda::GpuMat input(HEIGHT, WIDTH, IT);
    for (int i = 0; i < ITERS; ++i) {
        input.setTo(cv::Scalar(i + 1, i + 1, i + 1));
        myTensor.update(cv_stream, input, fk::Unary<fk::SaturateCast<CUDA_T(IT), CUDA_T(OT)>>{}, fk::Write<fk::TensorTSplit<CUDA_T(OT)>>{myTensor.ptr()});
        cv_stream.waitForCompletion();
    }
    const auto h = fetch(myTensor.data(), myTensor.sizeInBytes());
    const float* t = (const float*)h.data();
    bool ok = true;
    const size_t plane = (size_t)WIDTH * HEIGHT;
    for (uint c = 0; c < CP && ok; ++c)
        for (uint z = 0; z < BATCH && ok; ++z) {
            const int e = ORDER == fk::CircularTensorOrder::NewestFirst ? ITERS - (int)z : ITERS - (int)(BATCH - z - 1);
            ok = all_close<float>(t + ((size_t)c * BATCH + z) * plane, plane, e);
        }
    CHECK(ok, "CircularTensor Transposed: plane layout [c][z][y][x], order " << (int)ORDER);
}

static void testOldestFirstCircularTensorcvGS_noSplit() {
    cvGS::CircularTensor<CV_8UC4, CV_32FC4, 1, BATCH, fk::CircularTensorOrder::OldestFirst> myTensor(WIDTH, HEIGHT);
    cv::cuda::Stream cv_stream;
    cv::cuda::GpuMat input(HEIGHT, WIDTH, CV_8UC4);
    for (int i = 0; i < ITERS; ++i) {
        input.setTo(cv::Scalar::all(i + 1));
        myTensor.update(cv_stream, input, fk::Unary<fk::SaturateCast<CUDA_T(CV_8UC4), CUDA_T(CV_32FC4)>>{},
                        fk::Write<fk::TensorWrite<CUDA_T(CV_32FC4)>>{myTensor.ptr()});
        cv_stream.waitForCompletion();
    }
    const auto h = fetch(myTensor.data(), myTensor.sizeInBytes());
    const float* t = (const float*)h.data();
    bool ok = true;
    const size_t plane = (size_t)WIDTH * HEIGHT * 4;
    for (uint z = 0; z < BATCH && ok; ++z) ok = all_close<float>(t + (size_t)z * plane, plane, ITERS - (int)(BATCH - z - 1));
    CHECK(ok, "CircularTensor OldestFirst packed (COLOR_PLANES = 1)");
}

// cfg #4 in small: every update resizes + normalizes a NEW frame into the tensor; checked against the oracle's
// CircularTensor restatement after every update.
// MIRRORED = true: the opt-in mirrored-ring layout must show the SAME tensor at data() (which then moves every update).
template <fk::CircularTensorOrder ORDER, bool MIRRORED>
static void testResizeNormalizePush() {
    constexpr uint B = 6, W = 96, H = 54;
    cvGS::CircularTensor<CV_8UC3, CV_32F, 3, B, ORDER, fk::ColorPlanes::Standard, MIRRORED> myTensor(W, H);
    oracle_circular_t oc = nullptr;
    CHECK(oracle_circular_create(&oc, W, H, CV_32FC1, 3, B, (int)ORDER, CVGS_PLANES_STANDARD) == 0, "oracle circular create");
    cv::cuda::Stream cv_stream;
    const cv::Scalar sub(1.f, 4.f, 3.2f), div(3.2f, 0.6f, 11.8f), alpha(0.3, 0.3, 0.3);
    bool ok = true;
    for (int i = 0; i < 2 * (int)B + 1; ++i) {
        cv::Mat h_frame(540, 960, CV_8UC3);
        fill_random(h_frame, 900 + i);
        cv::cuda::GpuMat d_frame(h_frame), hv = host_view(h_frame);
        myTensor.update(cv_stream, cvGS::resize<CV_8UC3, cv::INTER_LINEAR>(d_frame, cv::Size(W, H), 0., 0.), cvGS::multiply<CV_32FC3>(alpha),
                        cvGS::subtract<CV_32FC3>(sub), cvGS::divide<CV_32FC3>(div), fk::Write<fk::TensorSplit<float3>>{myTensor.ptr()});
        fk::ChainBuilder b;
        fk::lowerChain(b, cvGS::resize<CV_8UC3, cv::INTER_LINEAR>(hv, cv::Size(W, H), 0., 0.), cvGS::multiply<CV_32FC3>(alpha),
                       cvGS::subtract<CV_32FC3>(sub), cvGS::divide<CV_32FC3>(div), fk::Write<fk::TensorSplit<float3>>{myTensor.ptr()});
        ok = ok && oracle_circular_update(oc, &b.d) == 0;
        cv_stream.waitForCompletion();
        const auto h = fetch(myTensor.data(), myTensor.sizeInBytes());
        ok = ok && bit_equal(h.data(), oracle_circular_data(oc), h.size());
    }
    CHECK(ok, "CircularTensor push with resize + normalize, bit-exact vs oracle after every update (order " << (int)ORDER
                  << ", mirrored " << MIRRORED << ")");
    oracle_circular_destroy(oc);
}

int main() {
    testCircularTensorcvGS<CV_8UC3, CV_32FC3>();
    testTransposedCircularTensorcvGS<CV_8UC3, CV_32FC3, fk::CircularTensorOrder::NewestFirst>();
    testTransposedCircularTensorcvGS<CV_8UC3, CV_32FC3, fk::CircularTensorOrder::OldestFirst>();
    testOldestFirstCircularTensorcvGS_noSplit();
    testResizeNormalizePush<fk::CircularTensorOrder::NewestFirst, false>();
    testResizeNormalizePush<fk::CircularTensorOrder::OldestFirst, false>();
    testResizeNormalizePush<fk::CircularTensorOrder::NewestFirst, true>();
    testResizeNormalizePush<fk::CircularTensorOrder::OldestFirst, true>();
    return report("test_circulartensor");
}
