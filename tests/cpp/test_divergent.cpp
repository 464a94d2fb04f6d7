// test_divergent.cpp -- mirrors, on the facade, the three places the reference spells fk::CircularBatchRead, fk::buildOperationSequence and
// the divergent-batch launch (VERDICT r4 "What's missing" #3):
//   * tests/batchread/test_circularbatchread_x_write3D.cu:24-87   testCircularBatchRead: 15 planes of value i, FIRST = 4 -> out[z] == (z + 4) mod 15
//   * tests/batchread/test_circularbatchread_x_write3D.cu:95-174  testDivergentBatch: two sequences over a 2-plane uint tensor, selector z -> z + 1:
//                                                                 plane 0 = input[0] + 3, plane 1 = input[1]
//   * tests/resize/test_fused_resize.cu:73-92                      the tensor variant of the fused NV12 resize: one sequence for every plane
//                                                                 (PerPlaneSequenceSelector::at == 1), written into a Tensor<uchar4>
// plus what the reference does not test: Descendent order, K1-shaped planes through one / two sequences (one sequence: ONE fused launch), a selector that skips planes, and the oracle as checker.
// The raw kernel launch `launchDivergentBatchTransformDPP_Kernel<PA, Selector><<<grid(.., .., BATCH), block, 0, stream>>>(seqs...)` is spelled
// fk::executeDivergentBatch<Selector>(stream, BATCH, seqs...) here.
#include "common.h"

struct OneToOne { // the reference's selector, qualifiers included (:89-93)
    constexpr static __device__ __forceinline__ uint at(const uint& zIdx) { return zIdx + 1; }
};
struct PerPlaneSequenceSelector { // tests/resize/test_fused_resize.cu:22-26
    FK_HOST_DEVICE_FUSE uint at(const uint& index) { return 1; }
};
struct EverySecondPlane { // planes 1, 3, ... run nothing
    FK_HOST_DEVICE_FUSE uint at(const uint& z) { return z % 2 == 0 ? 1 : 0; }
};

template <fk::CircularDirection DIR>
static void testCircularBatchRead(hipStream_t stream) {
    constexpr uint WIDTH = 32, HEIGHT = 32, BATCH = 15, FIRST = 4;
    std::vector<fk::Ptr2D<uchar3>> inputs;
    fk::Read<fk::CircularBatchRead<DIR, fk::PerThreadRead<fk::_2D, uchar3>, BATCH>> circularBatchRead;
    circularBatchRead.params.first = FIRST;
    for (uint i = 0; i < BATCH; ++i) {
        fk::Ptr2D<uchar3> temp(WIDTH, HEIGHT);
        std::vector<uchar3> h(((size_t)temp.dims().pitch * HEIGHT + sizeof(uchar3) - 1) / sizeof(uchar3), fk::make_<uchar3>(i, i, i)); // (a 256-byte pitch is not a whole number of 3-byte pixels)
        HIP_OK(hipMemcpy(temp.ptr().data, h.data(), (size_t)temp.dims().pitch * HEIGHT, hipMemcpyHostToDevice));
        inputs.push_back(temp);
        circularBatchRead.params.opData[i].params = temp;
    }
    fk::Tensor<uchar3> output(WIDTH, HEIGHT, BATCH);
    fk::WriteInstantiableOperation<fk::PerThreadWrite<fk::_3D, uchar3>> write3D{{output}};
    fk::executeOperations(stream, circularBatchRead, write3D);
    HIP_OK(hipStreamSynchronize(stream));
    const auto h = fetch(output.ptr().data, output.sizeInBytes());
    bool correct = true;
    for (uint z = 0; z < BATCH; ++z) {
        const uint want = DIR == fk::Ascendent ? (z + FIRST) % BATCH : (FIRST + BATCH - z) % BATCH;
        for (size_t i = 0; i < (size_t)WIDTH * HEIGHT * 3; ++i) correct = correct && h[(size_t)z * WIDTH * HEIGHT * 3 + i] == (uint8_t)want;
    }
    CHECK(correct, "CircularBatchRead " << (DIR == fk::Ascendent ? "Ascendent: out[z] == in[(z + first) mod BATCH]" : "Descendent: out[z] == in[(first - z) mod BATCH]"));
}

static void testDivergentBatch(hipStream_t stream) {
    constexpr uint WIDTH = 32, HEIGHT = 32, BATCH = 2, VAL_SUM = 3;
    std::vector<fk::Ptr2D<uint>> inputAllocations;
    std::array<fk::RawPtr<fk::_2D, uint>, BATCH> input;
    for (uint i = 0; i < BATCH; ++i) {
        fk::Ptr2D<uint> temp(WIDTH, HEIGHT);
        std::vector<uint> h((size_t)temp.dims().pitch / sizeof(uint) * HEIGHT, i);
        HIP_OK(hipMemcpy(temp.ptr().data, h.data(), (size_t)temp.dims().pitch * HEIGHT, hipMemcpyHostToDevice));
        inputAllocations.push_back(temp);
        input[i] = temp;
    }
    fk::Tensor<uint> output(WIDTH, HEIGHT, BATCH);
    HIP_OK(hipMemset(output.ptr().data, 0xff, output.sizeInBytes()));
    auto opSeq1 = fk::buildOperationSequence(fk::Read<fk::PerThreadRead<fk::_2D, uint>>{input[0]}, fk::Binary<fk::Add<uint>>{VAL_SUM},
                                             fk::Write<fk::PerThreadWrite<fk::_3D, uint>>{output.ptr()});
    auto opSeq2 = fk::buildOperationSequence(fk::Read<fk::PerThreadRead<fk::_2D, uint>>{input[1]}, fk::Write<fk::PerThreadWrite<fk::_3D, uint>>{output.ptr()});
    fk::executeDivergentBatch<OneToOne>(stream, BATCH, opSeq1, opSeq2);
    HIP_OK(hipStreamSynchronize(stream));
    const auto h = fetch(output.ptr().data, output.sizeInBytes());
    const uint* t = (const uint*)h.data();
    bool correct = true;
    for (uint z = 0; z < BATCH; ++z)
        for (size_t i = 0; i < (size_t)WIDTH * HEIGHT; ++i) correct = correct && t[(size_t)z * WIDTH * HEIGHT + i] == (z == 0 ? VAL_SUM : z);
    CHECK(correct, "divergent batch: plane 0 = input[0] + 3 (sequence 1), plane 1 = input[1] (sequence 2)");
    // a selector that leaves planes out: those planes are not written
    HIP_OK(hipMemset(output.ptr().data, 0xff, output.sizeInBytes()));
    fk::executeDivergentBatch<EverySecondPlane>(stream, BATCH, opSeq1);
    HIP_OK(hipStreamSynchronize(stream));
    const auto h2 = fetch(output.ptr().data, output.sizeInBytes());
    const uint* t2 = (const uint*)h2.data();
    bool skipped = true;
    for (size_t i = 0; i < (size_t)WIDTH * HEIGHT; ++i) skipped = skipped && t2[i] == VAL_SUM && t2[(size_t)WIDTH * HEIGHT + i] == 0xffffffffu;
    CHECK(skipped, "divergent batch: a plane whose selector names no sequence is left alone");
}

// the tensor variant of the fused NV12 resize: ONE sequence, every plane of the tensor runs it with z as its plane
static void testFusedResizeIntoTensor(hipStream_t stream) {
    const uint W = 1920, H = 1080;
    constexpr uint OUTPUTS = 3;
    const fk::Size down(640, 360);
    cv::Mat h_nv12(H + H / 2, W, CV_8UC1);
    fill_random(h_nv12, 9191);
    cv::cuda::GpuMat d_nv12(h_nv12);
    fk::Tensor<uchar4> myTensor(down.width, down.height, OUTPUTS);
    auto sequence = [&](uchar* base, uint pitch, fk::RawPtr<fk::_3D, uchar4> out) {
        const fk::RawPtr<fk::_2D, uchar> src{base, {W, H, pitch}};
        const auto readBackOp = fk::fuse(fk::Read<fk::ReadYUV<fk::NV12>>{src}, fk::Unary<fk::ConvertYUVToRGB<fk::NV12, fk::Full, fk::bt709, true, float4>>{});
        const auto readOp = fk::Resize<fk::InterpolationType::INTER_LINEAR>::build(readBackOp, down);
        auto convertOp = fk::Unary<fk::SaturateCast<float4, uchar4>>{};
        auto colorConvert = fk::Unary<fk::VectorReorder<uchar4, 2, 1, 0, 3>>{};
        fk::Write<fk::TensorWrite<uchar4>> writesTensor;
        writesTensor.params = out;
        return fk::buildOperationSequence(readOp, convertOp, colorConvert, writesTensor);
    };
    auto OpSeqTensor = sequence(d_nv12.data, (uint)d_nv12.step, myTensor.ptr());
    fk::executeDivergentBatch<PerPlaneSequenceSelector>(stream, OUTPUTS, OpSeqTensor);
    HIP_OK(hipStreamSynchronize(stream));
    // the oracle: the same chain into a host image
    cv::Mat h_ref(down.height, down.width, CV_8UC4);
    {
        const fk::RawPtr<fk::_2D, uchar> src{h_nv12.data, {W, H, (uint)h_nv12.step}};
        const auto readBackOp = fk::fuse(fk::Read<fk::ReadYUV<fk::NV12>>{src}, fk::Unary<fk::ConvertYUVToRGB<fk::NV12, fk::Full, fk::bt709, true, float4>>{});
        const fk::RawPtr<fk::_2D, uchar4> dst{(uchar4*)h_ref.data, {(uint)down.width, (uint)down.height, (uint)h_ref.step}};
        run_oracle(fk::Resize<fk::InterpolationType::INTER_LINEAR>::build(readBackOp, down), fk::Unary<fk::SaturateCast<float4, uchar4>>{},
                   fk::Unary<fk::VectorReorder<uchar4, 2, 1, 0, 3>>{}, fk::Write<fk::PerThreadWrite<fk::_2D, uchar4>>{dst});
    }
    const auto h = fetch(myTensor.ptr().data, myTensor.sizeInBytes());
    bool same = true;
    const size_t plane = (size_t)down.width * down.height * 4;
    for (uint z = 0; z < OUTPUTS; ++z)
        for (int y = 0; y < down.height; ++y) same = same && bit_equal(h.data() + z * plane + (size_t)y * down.width * 4, h_ref.ptr<uchar>(y), (size_t)down.width * 4);
    CHECK(same, "fused NV12 resize as ONE operation sequence over a " << OUTPUTS << "-plane Tensor<uchar4>: every plane bit-exact vs the oracle");
}

struct EvenOdd { // even planes run sequence 1, odd planes sequence 2
    FK_HOST_DEVICE_FUSE uint at(const uint& z) { return 1 + z % 2; }
};

// K1-shaped planes: the crops of a frame -> resize -> normalize -> the planes of ONE tensor.  With one sequence for every plane the
// divergent batch is the batched chain itself (every plane = one crop: ONE fused launch, grid z = plane); with two sequences (different
// multipliers) even planes must carry sequence 1's values and odd planes sequence 2's.  Checked against cvGS::executeOperations.
static void testK1PlanesThroughSequences(cv::cuda::Stream& cv_stream) {
    constexpr int CROPS = 6;
    const cv::Size up(64, 128);
    cv::Mat h(480, 640, CV_8UC3);
    fill_random(h, 4711);
    cv::cuda::GpuMat frame(h);
    std::array<cv::cuda::GpuMat, CROPS> crops;
    for (int i = 0; i < CROPS; ++i) crops[i] = frame(cv::Rect(10 * i + 3, 7 * i, 60 + 20 * i, 120 + 10 * i));
    const size_t plane_bytes = (size_t)up.width * up.height * 3 * 4, bytes = CROPS * plane_bytes;
    cv::cuda::GpuMat tensor(CROPS, up.width * up.height * 3, CV_32FC1), refA(CROPS, up.width * up.height * 3, CV_32FC1), refB(CROPS, up.width * up.height * 3, CV_32FC1);
    const cv::Scalar a(0.5, 0.25, 2.0), b(3.0, 1.5, 0.125), sub(1, 4, 3.2), div(3.2, 0.6, 11.8);
    auto chain = [&](const cv::Scalar& mul, cv::cuda::GpuMat& out) {
        return fk::buildOperationSequence(cvGS::resize<CV_8UC3, cv::INTER_LINEAR, CROPS>(crops, up, CROPS), cvGS::multiply<CV_32FC3>(mul),
                                          cvGS::subtract<CV_32FC3>(sub), cvGS::divide<CV_32FC3>(div), cvGS::split<CV_32FC3>(out, up));
    };
    cvGS::executeOperations(cv_stream, cvGS::resize<CV_8UC3, cv::INTER_LINEAR, CROPS>(crops, up, CROPS), cvGS::multiply<CV_32FC3>(a),
                            cvGS::subtract<CV_32FC3>(sub), cvGS::divide<CV_32FC3>(div), cvGS::split<CV_32FC3>(refA, up));
    cvGS::executeOperations(cv_stream, cvGS::resize<CV_8UC3, cv::INTER_LINEAR, CROPS>(crops, up, CROPS), cvGS::multiply<CV_32FC3>(b),
                            cvGS::subtract<CV_32FC3>(sub), cvGS::divide<CV_32FC3>(div), cvGS::split<CV_32FC3>(refB, up));
    const auto seqA = chain(a, tensor), seqB = chain(b, tensor);
    fk::executeDivergentBatch<PerPlaneSequenceSelector>(cv_stream.raw(), CROPS, seqA);
    cv_stream.waitForCompletion();
    CHECK(fetch(tensor.data, bytes) == fetch(refA.data, bytes), "one K1 sequence for every plane == the batched chain, bit for bit");
    tensor.setTo(cv::Scalar(-1));
    fk::executeDivergentBatch<EvenOdd>(cv_stream.raw(), CROPS, seqA, seqB);
    cv_stream.waitForCompletion();
    const auto t = fetch(tensor.data, bytes), ra = fetch(refA.data, bytes), rb = fetch(refB.data, bytes);
    bool same = true;
    for (int z = 0; z < CROPS; ++z) same = same && bit_equal(t.data() + z * plane_bytes, (z % 2 ? rb : ra).data() + z * plane_bytes, plane_bytes);
    CHECK(same, "two K1 sequences: even planes carry sequence 1's values, odd planes sequence 2's");
}

int main() {
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    testCircularBatchRead<fk::Ascendent>(stream);
    testCircularBatchRead<fk::Descendent>(stream);
    testDivergentBatch(stream);
    testFusedResizeIntoTensor(stream);
    HIP_OK(hipStreamDestroy(stream));
    {
        cv::cuda::Stream cv_stream;
        testK1PlanesThroughSequences(cv_stream);
    }
    return report("test_divergent");
}
