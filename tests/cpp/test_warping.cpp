// test_warping.cpp -- mirrors reference tests/warping/test_warping_opencv.cu on the cvGS facade: cvGS::warp (affine,
// perspective, batched, batched with unused planes) -> fk::Cast<float3, uchar3> -> write, ONE fused kernel each.
// The reference compares against cv::cuda::warpAffine / warpPerspective on a PNG that is not part of its tree; here:
//  (1) affine translation (tx = 50, ty = 100), the reference's only must-PASS case: the result must be the shifted
//      image with a zero border (what cv::cuda::warpAffine(INTER_LINEAR, BORDER_CONSTANT 0) gives for an integer shift),
//  (2) every chain bit-exact against the CPU oracle on a non-constant image.
#include "common.h"

static cv::Mat random_image(int rows, int cols, int type, uint64_t seed) {
    cv::Mat m(rows, cols, type);
    fill_random(m, seed);
    return m;
}

static bool testAffine() {
    cv::Mat img = random_image(430, 470, CV_8UC3, 7001);
    cv::cuda::Stream stream;
    const cv::cuda::GpuMat d_img(img);
    const double tx = 50, ty = 100;
    cv::Mat affine_matrix = (cv::Mat_<double>(2, 3) << 1, 0, tx, 0, 1, ty);
    cv::cuda::GpuMat d_resultcvGS(img.size(), CV_8UC3);

    const auto warpFunc = cvGS::warp<fk::WarpType::Affine, CV_8UC3>(d_img, affine_matrix, img.size());
    auto writeFunc = cvGS::write<CV_8UC3>(d_resultcvGS);
    cvGS::executeOperations(stream, warpFunc, fk::Cast<float3, uchar3>::build(), writeFunc);
    stream.waitForCompletion();

    cv::Mat result;
    d_resultcvGS.download(result);
    bool ok = true;
    for (int y = 0; y < img.rows && ok; ++y)
        for (int x = 0; x < img.cols && ok; ++x)
            for (int c = 0; c < 3 && ok; ++c) {
                const int sx = x - 50, sy = y - 100;
                const uchar e = (sx >= 0 && sy >= 0) ? img.ptr<uchar>(sy)[sx * 3 + c] : 0;
                ok = result.ptr<uchar>(y)[x * 3 + c] == e;
            }
    CHECK(ok, "affine translation == shifted image with zero border");

    // the same chain on the oracle
    cv::Mat h_ref(img.rows, img.cols, CV_8UC3);
    cv::cuda::GpuMat hv_img = host_view(img), hv_ref = host_view(h_ref);
    run_oracle(cvGS::warp<fk::WarpType::Affine, CV_8UC3>(hv_img, affine_matrix, img.size()), fk::Cast<float3, uchar3>::build(),
               cvGS::write<CV_8UC3>(hv_ref));
    bool same = true;
    for (int y = 0; y < img.rows && same; ++y) same = bit_equal(result.ptr<uchar>(y), h_ref.ptr<uchar>(y), (size_t)img.cols * 3);
    CHECK(same, "affine chain bit-exact vs oracle");
    return ok && same;
}

static const cv::Point2f kSrc[5][4] = {
    {{56, 65}, {368, 52}, {28, 387}, {389, 390}}, {{50, 50}, {400, 50}, {50, 400}, {400, 400}},
    {{30, 30}, {350, 30}, {30, 350}, {350, 350}}, {{70, 70}, {370, 70}, {70, 370}, {370, 370}},
    {{20, 20}, {320, 20}, {20, 320}, {320, 320}}};
static const cv::Point2f kDst[5][4] = {
    {{0, 0}, {300, 0}, {0, 300}, {300, 300}}, {{0, 0}, {300, 0}, {0, 300}, {300, 300}}, {{0, 0}, {250, 0}, {0, 250}, {250, 250}},
    {{0, 0}, {280, 0}, {0, 280}, {280, 280}}, {{0, 0}, {200, 0}, {0, 200}, {200, 200}}};

static bool testPerspective() {
    cv::Mat img = random_image(430, 470, CV_8UC3, 7002);
    cv::cuda::Stream stream;
    const cv::cuda::GpuMat d_img(img);
    cv::Mat perspective_matrix = cv::getPerspectiveTransform(kSrc[0], kDst[0]);
    cv::cuda::GpuMat d_resultcvGS(img.size(), CV_8UC3);
    cvGS::executeOperations(stream, cvGS::warp<fk::WarpType::Perspective, CV_8UC3>(d_img, perspective_matrix, img.size()),
                            fk::Cast<float3, uchar3>::build(), cvGS::write<CV_8UC3>(d_resultcvGS));
    stream.waitForCompletion();
    cv::Mat result;
    d_resultcvGS.download(result);
    cv::Mat h_ref(img.rows, img.cols, CV_8UC3);
    cv::cuda::GpuMat hv_img = host_view(img), hv_ref = host_view(h_ref);
    run_oracle(cvGS::warp<fk::WarpType::Perspective, CV_8UC3>(hv_img, perspective_matrix, img.size()), fk::Cast<float3, uchar3>::build(),
               cvGS::write<CV_8UC3>(hv_ref));
    bool same = true;
    size_t nonzero = 0;
    for (int y = 0; y < img.rows && same; ++y) {
        same = bit_equal(result.ptr<uchar>(y), h_ref.ptr<uchar>(y), (size_t)img.cols * 3);
        for (int x = 0; x < img.cols * 3; ++x) nonzero += result.ptr<uchar>(y)[x] != 0;
    }
    CHECK(same, "perspective chain bit-exact vs oracle");
    CHECK(nonzero > (size_t)img.rows * img.cols, "perspective result is not empty");
    return same;
}

template <size_t NUM_IMGS>
static bool testPerspectiveBatch(int usedPlanes) {
    cv::Mat img = random_image(430, 470, CV_8UC3, 7003 + NUM_IMGS);
    cv::cuda::Stream stream;
    const cv::cuda::GpuMat d_img(img);
    cv::cuda::GpuMat hv_img = host_view(img);
    std::array<cv::cuda::GpuMat, NUM_IMGS> d_imgs, hv_imgs, d_results, hv_results;
    std::array<cv::Mat, NUM_IMGS> matrices, h_refs;
    for (size_t i = 0; i < NUM_IMGS; ++i) {
        d_imgs[i] = d_img;
        hv_imgs[i] = hv_img;
        if ((int)i < usedPlanes) matrices[i] = cv::getPerspectiveTransform(kSrc[i % 5], kDst[i % 5]);
        d_results[i] = cv::cuda::GpuMat(img.size(), CV_8UC3);
        h_refs[i] = cv::Mat(img.rows, img.cols, CV_8UC3);
        hv_results[i] = host_view(h_refs[i]);
    }
    auto chain = [&](const std::array<cv::cuda::GpuMat, NUM_IMGS>& in, const std::array<cv::cuda::GpuMat, NUM_IMGS>& out) {
        const auto warpFunc = usedPlanes == (int)NUM_IMGS
                                  ? cvGS::warp<fk::WarpType::Perspective, CV_8UC3, NUM_IMGS>(in, matrices, img.size())
                                  : cvGS::warp<fk::WarpType::Perspective, CV_8UC3>(in, matrices, img.size(), usedPlanes, cv::Scalar());
        auto fk_outputs = cvGS::gpuMat2RawPtr2D_arr<uchar3>(out);
        return std::make_tuple(warpFunc, fk::Cast<float3, uchar3>::build(), fk::PerThreadWrite<fk::_2D, uchar3>::build(fk_outputs));
    };
    std::apply([&](const auto&... iops) { cvGS::executeOperations(stream, iops...); }, chain(d_imgs, d_results));
    std::apply([&](const auto&... iops) { run_oracle(iops...); }, chain(hv_imgs, hv_results));
    stream.waitForCompletion();
    bool same = true;
    for (size_t i = 0; i < NUM_IMGS; ++i) {
        cv::Mat result;
        d_results[i].download(result);
        bool zero = true;
        for (int y = 0; y < img.rows; ++y) {
            same = same && bit_equal(result.ptr<uchar>(y), h_refs[i].template ptr<uchar>(y), (size_t)img.cols * 3);
            for (int x = 0; x < img.cols * 3 && zero; ++x) zero = result.ptr<uchar>(y)[x] == 0;
        }
        CHECK(zero == ((int)i >= usedPlanes), "plane " << i << (zero ? " is empty" : " is not empty"));
    }
    CHECK(same, "perspective batch of " << NUM_IMGS << " (" << usedPlanes << " used) bit-exact vs oracle");
    return same;
}

// the reference's parameter helpers under their own names (include/cvGPUSpeedup.cuh:269-284, 311-377): host arithmetic only
static void testParameterHelpers() {
    const cv::Size sz(300, 200);
    const double inv_a[6] = {1, 0, -50, 0, 1, -100};
    const auto pa = cvGS::internal::warp_getWarpingAffineParameters(inv_a, sz);
    bool ok = pa.dstSize.width == 300 && pa.dstSize.height == 200;
    for (int i = 0; i < 6; ++i) ok = ok && pa.transformMatrix[i / 3][i % 3] == (float)inv_a[i];
    const double inv_p[9] = {1.25, 0.5, -3, 0.125, 2, 7, 0.001, 0.002, 1};
    const auto pp = cvGS::internal::warp_getWarpingPerspectiveParameters(inv_p, sz);
    for (int i = 0; i < 9; ++i) ok = ok && pp.transformMatrix[i / 3][i % 3] == (float)inv_p[i];
    CHECK(ok, "warp_getWarping{Affine,Perspective}Parameters narrow the inverted transform to float, row-major");
    // the batch helpers take FORWARD transforms and must agree with the single-image path (inversion in double, then float)
    std::array<cv::Mat, 3> fwd;
    std::array<cv::Size, 3> sizes;
    for (size_t i = 0; i < 3; ++i) {
        fwd[i] = cv::getPerspectiveTransform(kSrc[i], kDst[i]);
        sizes[i] = cv::Size(100 + 10 * (int)i, 90);
    }
    const auto all = cvGS::internal::warp_batchParameters<fk::WarpType::Perspective>(fwd, sizes);
    const auto two = cvGS::internal::warp_batchParameters<fk::WarpType::Perspective>(fwd, sizes, 2);
    bool same = true;
    for (size_t i = 0; i < 3; ++i) {
        const cv::Mat inv(fwd[i].inv());
        const auto one = cvGS::internal::warp_getWarpingPerspectiveParameters(inv.ptr<double>(), sizes[i]);
        for (int k = 0; k < 9; ++k) {
            same = same && all[i].transformMatrix[k / 3][k % 3] == one.transformMatrix[k / 3][k % 3];
            if (i < 2) same = same && two[i].transformMatrix[k / 3][k % 3] == one.transformMatrix[k / 3][k % 3];
        }
        same = same && (int)all[i].dstSize.width == sizes[i].width;
    }
    CHECK(same, "warp_batchParameters == the single-image parameters, plane by plane");
    std::array<cv::Mat, 2> aff = {(cv::Mat_<double>(2, 3) << 1, 0, 50, 0, 1, 100), (cv::Mat_<double>(2, 3) << 2, 0, 0, 0, 4, 8)};
    std::array<cv::Size, 2> asz = {sz, sz};
    const auto ab = cvGS::internal::warp_batchParameters<fk::WarpType::Affine>(aff, asz);
    CHECK(ab[0].transformMatrix[0][2] == -50.f && ab[0].transformMatrix[1][2] == -100.f && ab[1].transformMatrix[0][0] == 0.5f &&
              ab[1].transformMatrix[1][1] == 0.25f && ab[1].transformMatrix[1][2] == -2.f,
          "warp_batchParameters<Affine> inverts the forward transforms");
}

int main() {
    testParameterHelpers();
    testPerspective();
    testAffine();
    testPerspectiveBatch<5>(5);
    testPerspectiveBatch<10>(3);
    return report("test_warping");
}
