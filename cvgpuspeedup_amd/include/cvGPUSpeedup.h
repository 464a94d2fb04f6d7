// cvGPUSpeedup.h -- the cvGS:: facade for the hot path, source-compatible with the reference's
// include/cvGPUSpeedup.cuh call shapes, on the MI355X engine.
//
//   cvGS::executeOperations(stream, cvGS::resize<CV_8UC3, cv::INTER_LINEAR, N>(crops, size, used),
//                           cvGS::cvtColor<cv::COLOR_RGB2BGR, CV_32FC3>(), cvGS::multiply<CV_32FC3>(a),
//                           cvGS::subtract<CV_32FC3>(s), cvGS::divide<CV_32FC3>(d),
//                           cvGS::split<CV_32FC3>(tensor, size));
//
// Each builder returns a typed IOp value (fk_compat.h); executeOperations checks the type chain at compile time,
// lowers the list to ONE cvgs_chain_desc and calls cvgs_execute() (include/cvgs_hip.h) -> ONE HIP kernel.  All
// calls are asynchronous on the given stream and never synchronise; device memory stays caller-owned.
// Not provided (outside the hot path): fk::StaticLoop.
#pragma once

#include <array>
#include <cassert>
#include <vector>

#include "cvGPUSpeedupHelpers.h"

namespace cvGS {

enum AspectRatio { PRESERVE_AR = 0, IGNORE_AR = 1, PRESERVE_AR_RN_EVEN = 2, PRESERVE_AR_LEFT = 3 };

// ---- GpuMat -> fk pointer adapters ------------------------------------------------------------------------
template <typename T>
inline fk::Ptr2D<T> gpuMat2Ptr2D(const cv::cuda::GpuMat& m) {
    return fk::Ptr2D<T>(reinterpret_cast<T*>(m.data), (uint)m.cols, (uint)m.rows, (uint)m.step);
}
template <typename T>
inline fk::RawPtr<fk::_2D, T> gpuMat2RawPtr2D(const cv::cuda::GpuMat& m) {
    fk::RawPtr<fk::_2D, T> p;
    p.data = reinterpret_cast<T*>(m.data);
    p.dims = {(uint)m.cols, (uint)m.rows, (uint)m.step};
    return p;
}
// reference include/cvGPUSpeedup.cuh:46-52
template <typename T, int Batch>
inline std::array<fk::Ptr2D<T>, Batch> gpuMat2Ptr2D_arr(const std::array<cv::cuda::GpuMat, Batch>& src) {
    std::array<fk::Ptr2D<T>, Batch> out;
    for (int i = 0; i < Batch; ++i) out[(size_t)i] = gpuMat2Ptr2D<T>(src[(size_t)i]);
    return out;
}
template <typename T, size_t N>
inline std::array<fk::RawPtr<fk::_2D, T>, N> gpuMat2RawPtr2D_arr(const std::array<cv::cuda::GpuMat, N>& src, int used = (int)N) {
    std::array<fk::RawPtr<fk::_2D, T>, N> out{};
    for (int i = 0; i < used && i < (int)N; ++i) out[(size_t)i] = gpuMat2RawPtr2D<T>(src[(size_t)i]);
    return out;
}
template <typename T>
inline fk::Tensor<T> gpuMat2Tensor(const cv::cuda::GpuMat& m, const cv::Size& plane, int colorPlanes) {
    return fk::Tensor<T>(reinterpret_cast<T*>(m.data), (uint)plane.width, (uint)plane.height, (uint)m.rows, (uint)colorPlanes);
}

// ---- pointwise builders ------------------------------------------------------------------------------------
template <int I, int O>
inline auto convertTo() {
    static_assert(CV_MAT_CN(I) == CV_MAT_CN(O), "convertTo does not support changing the number of channels, neither in cvGS nor in OpenCV. Please, use cvGS::cvtColor instead.");
    return fk::Unary<fk::SaturateCast<CUDA_T(I), CUDA_T(O)>>{};
}

namespace internal {
// integral outputs go through float: cast -> mul -> (add) -> saturate; float outputs: cast -> mul -> (add)
template <int I, int O>
inline auto convertToScaled(float alpha, const float* beta) {
    static_assert(CV_MAT_CN(I) == CV_MAT_CN(O), "convertTo does not support changing the number of channels, neither in cvGS nor in OpenCV. Please, use cvGS::cvtColor instead.");
    constexpr bool integral = CV_MAT_DEPTH(O) <= CV_32S || CV_MAT_DEPTH(O) == CV_16F; // CV_16F: storage only, rounded once
    constexpr int mid = integral ? CV_32F : CV_MAT_DEPTH(O);
    fk::PointwiseSeq<CUDA_T(I), CUDA_T(O)> seq;
    fk::ChainBuilder b;
    b.op(CVGS_OP_CAST, mid);
    const float a[4] = {alpha, alpha, alpha, alpha};
    b.op(CVGS_OP_MUL, 0, a); // operand_d follows as (double)alpha: fk::make_set<FloatType>(alpha) on CV_64F outputs
    if (beta) {
        const float bb[4] = {*beta, *beta, *beta, *beta};
        b.op(CVGS_OP_ADD, 0, bb);
    }
    if (integral) b.op(CVGS_OP_CAST, CV_MAT_DEPTH(O));
    seq.ops.assign(b.d.ops, b.d.ops + b.d.n_ops);
    return seq;
}
} // namespace internal

template <int I, int O> inline auto convertTo(float alpha) { return internal::convertToScaled<I, O>(alpha, nullptr); }
template <int I, int O> inline auto convertTo(float alpha, float beta) { return internal::convertToScaled<I, O>(alpha, &beta); }

template <int I> inline auto multiply(const cv::Scalar& s) { return fk::Binary<fk::Mul<CUDA_T(I)>>{cvScalar2CUDAV<I>::get(s)}; }
template <int I> inline auto subtract(const cv::Scalar& s) { return fk::Binary<fk::Sub<CUDA_T(I)>>{cvScalar2CUDAV<I>::get(s)}; }
// (the reference README's example spells it `substract`, README.md:127: the snippet users copy compiles)
template <int I> inline auto substract(const cv::Scalar& s) { return subtract<I>(s); }
template <int I> inline auto divide(const cv::Scalar& s) { return fk::Binary<fk::Div<CUDA_T(I)>>{cvScalar2CUDAV<I>::get(s)}; }
template <int I> inline auto add(const cv::Scalar& s) { return fk::Binary<fk::Add<CUDA_T(I)>>{cvScalar2CUDAV<I>::get(s)}; }

template <cv::ColorConversionCodes CODE, int I, int O = I>
inline auto cvtColor() {
    static_assert((CV_MAT_DEPTH(I) == CV_8U || CV_MAT_DEPTH(I) == CV_16U || CV_MAT_DEPTH(I) == CV_32F) &&
                  (CV_MAT_DEPTH(O) == CV_8U || CV_MAT_DEPTH(O) == CV_16U || CV_MAT_DEPTH(O) == CV_32F),
                  "Wrong CV_TYPE_DEPTH, it has to be CV_8U, or CV_16U or CV_32F");
    static_assert(isSupportedColorConversion<CODE>, "Color conversion type not supported yet.");
    return fk::Unary<fk::ColorConversion<(fk::ColorConversionCodes)CODE, CUDA_T(I), CUDA_T(O)>>{};
}

// ---- write builders ------------------------------------------------------------------------------------------
template <int O>
inline auto split(const std::vector<cv::cuda::GpuMat>& output) {
    std::vector<fk::Ptr2D<BASE_CUDA_T(O)>> planes;
    for (const auto& m : output) planes.push_back(gpuMat2Ptr2D<BASE_CUDA_T(O)>(m));
    return fk::SplitWrite<fk::_2D, CUDA_T(O)>::build(planes);
}
template <int O, size_t N>
inline auto split(const std::array<std::vector<cv::cuda::GpuMat>, N>& output) {
    std::array<std::vector<fk::Ptr2D<BASE_CUDA_T(O)>>, N> planes{};
    for (size_t i = 0; i < N; ++i)
        for (const auto& m : output[i]) planes[i].push_back(gpuMat2Ptr2D<BASE_CUDA_T(O)>(m));
    return fk::SplitWrite<fk::_2D, CUDA_T(O)>::build(planes);
}
// NCHW tensor living in a GpuMat with one image per row
template <int O>
inline auto split(const cv::cuda::GpuMat& output, const cv::Size& plane) {
    assert(output.cols % (plane.width * plane.height) == 0 && output.cols / (plane.width * plane.height) == CV_MAT_CN(O) &&
           "Each row of the GpuMat should contain as many planes as width / (planeDims.width * planeDims.height)");
    return fk::Write<fk::TensorSplit<CUDA_T(O)>>{gpuMat2Tensor<BASE_CUDA_T(O)>(output, plane, CV_MAT_CN(O)).ptr()};
}
template <int O>
inline auto split(const fk::RawPtr<fk::_3D, typename fk::VectorTraits<CUDA_T(O)>::base>& output) {
    return fk::Write<fk::TensorSplit<CUDA_T(O)>>{output};
}
template <int O>
inline auto splitT(const fk::RawPtr<fk::T3D, typename fk::VectorTraits<CUDA_T(O)>::base>& output) {
    return fk::Write<fk::TensorTSplit<CUDA_T(O)>>{output};
}
template <int O>
inline auto write(const cv::cuda::GpuMat& output) {
    return fk::Write<fk::PerThreadWrite<fk::_2D, CUDA_T(O)>>{gpuMat2RawPtr2D<CUDA_T(O)>(output)};
}
template <int O>
inline auto write(const cv::cuda::GpuMat& output, const cv::Size& plane) {
    return fk::Write<fk::PerThreadWrite<fk::_3D, CUDA_T(O)>>{gpuMat2Tensor<CUDA_T(O)>(output, plane, 1).ptr()};
}
template <typename T>
inline auto write(const fk::Tensor<T>& output) {
    return fk::Write<fk::PerThreadWrite<fk::_3D, T>>{output.ptr()};
}

// ---- read builders ---------------------------------------------------------------------------------------------
// resize<INTER>(dsize): a resize still waiting for its source -- complete it with readIOp.then(...)
template <int INTER_F>
inline auto resize(const cv::Size& dsize) {
    static_assert(isSupportedInterpolation<INTER_F>, "Interpolation type not supported yet.");
    return fk::Resize<(fk::InterpolationType)INTER_F>::build(fk::Size(dsize.width, dsize.height));
}
// single image: resize<T, INTER>(GpuMat, dsize, fx, fy); the output type is CV_32F of the same channels
template <int T, int INTER_F>
inline auto resize(const cv::cuda::GpuMat& input, const cv::Size& dsize, double fx, double fy) {
    static_assert(isSupportedInterpolation<INTER_F>, "Interpolation type not supported yet.");
    return fk::Resize<(fk::InterpolationType)INTER_F>::build(gpuMat2RawPtr2D<CUDA_T(T)>(input), fk::Size(dsize.width, dsize.height), fx, fy);
}

// batch: N crops -> dsize, planes >= usedPlanes (and the AR padding) take backgroundValue
template <int T, int INTER_F, int NPtr, AspectRatio AR_ = IGNORE_AR>
inline auto resize(const std::array<cv::cuda::GpuMat, NPtr>& input, const cv::Size& dsize, const int& usedPlanes,
                   const cv::Scalar& backgroundValue_ = cvScalar_set<CV_MAKETYPE(CV_32F, CV_MAT_CN(T))>(0)) {
    static_assert(isSupportedInterpolation<INTER_F>, "Interpolation type not supported yet.");
    fk::BatchResizeRead<CUDA_T(T)> rd;
    rd.planes.resize((size_t)NPtr, cvgs_image2d{nullptr, 0, 0, 0, 0});
    for (int i = 0; i < usedPlanes && i < NPtr; ++i) rd.planes[(size_t)i] = fk::image2d(gpuMat2RawPtr2D<CUDA_T(T)>(input[(size_t)i]));
    rd.used = usedPlanes;
    rd.dsize = fk::Size(dsize.width, dsize.height);
    rd.ar = (int)AR_;
    for (int c = 0; c < CV_MAT_CN(T); ++c) rd.background[c] = static_cast<float>(backgroundValue_[c]);
    return rd;
}

// ---- NV12 sources (NEW: the reference has no cvGS:: wrapper for its fk::ReadYUV path) --------------
// cvtColorNV12<cv::COLOR_YUV2BGR_NV12 | RGB | BGRA | RGBA>(nv12): reads a decoder surface (CV_8UC1 GpuMat with
// H luma rows followed by H/2 interleaved UV rows, i.e. rows = H*3/2) as float BGR/RGB[A] pixels -- the first IOp of
// a chain.  resize<INTER>(thatIOp, dsize) fuses it as the read-back of the bilinear resize (one kernel).
namespace internal {
template <fk::PixelFormat PF, cv::ColorConversionCodes CODE, fk::ColorRange CR, fk::ColorPrimitives CP>
inline auto yuv_surface_read(const cv::cuda::GpuMat& surf, const char* what) {
    static_assert(CODE == cv::COLOR_YUV2RGB_NV12 || CODE == cv::COLOR_YUV2BGR_NV12 || CODE == cv::COLOR_YUV2RGBA_NV12 ||
                  CODE == cv::COLOR_YUV2BGRA_NV12, "Color conversion type not supported yet.");
    if (surf.type() != (PF == fk::P010 ? CV_16UC1 : CV_8UC1) || surf.rows % 3 != 0)
        throw std::runtime_error(std::string(what) + " needs a single-channel surface (CV_8UC1; CV_16UC1 for P010) with rows = H * 3 / 2");
    constexpr bool alpha = CODE == cv::COLOR_YUV2RGBA_NV12 || CODE == cv::COLOR_YUV2BGRA_NV12;
    constexpr bool swap = CODE == cv::COLOR_YUV2BGR_NV12 || CODE == cv::COLOR_YUV2BGRA_NV12;
    using O = std::conditional_t<alpha, float4, float3>;
    fk::RawPtr<fk::_2D, fk::YuvSample<PF>> luma;
    luma.data = (fk::YuvSample<PF>*)surf.data;
    luma.dims = {(uint)surf.cols, (uint)(surf.rows / 3 * 2), (uint)surf.step};
    return fk::YuvRead<PF, CR, CP, alpha, O, swap>{luma};
}
template <typename Read, size_t N>
inline void yuv_surface_crops(Read& rd, const cv::cuda::GpuMat& surf, const std::array<cv::Rect, N>& crops) {
    const int H = surf.rows / 3 * 2, step = (int)surf.step, esz = (int)surf.elemSize();
    for (const cv::Rect& r : crops) {
        if ((r.x | r.y | r.width | r.height) & 1) throw std::runtime_error("NV12 crops need even x, y, width and height");
        if (r.x < 0 || r.y < 0 || r.width < 2 || r.height < 2 || r.x + r.width > surf.cols || r.y + r.height > H)
            throw std::runtime_error("NV12 crop outside the surface");
        rd.crops.push_back(cvgs_image2d{surf.data + (size_t)r.y * step + (size_t)r.x * esz, r.width, r.height, step, (H - r.y + r.y / 2) * step});
    }
}
} // namespace internal
template <cv::ColorConversionCodes CODE, fk::ColorRange CR = fk::Full, fk::ColorPrimitives CP = fk::bt709>
inline auto cvtColorNV12(const cv::cuda::GpuMat& nv12) {
    return internal::yuv_surface_read<fk::NV12, CODE, CR, CP>(nv12, "cvtColorNV12");
}
// cvtColorNV12<CODE>(nv12, crops): N crops of the surface (even x, y, width, height) as a batch of N planes -- with
// resize<INTER>(thatIOp, dsize) the decode-side version of the headline path: N detections of a decoder surface ->
// colour conversion -> bilinear resize -> normalize -> NCHW tensor, ONE kernel, no intermediate BGR frame.
template <cv::ColorConversionCodes CODE, fk::ColorRange CR = fk::Full, fk::ColorPrimitives CP = fk::bt709, size_t N>
inline auto cvtColorNV12(const cv::cuda::GpuMat& nv12, const std::array<cv::Rect, N>& crops) {
    auto rd = cvtColorNV12<CODE, CR, CP>(nv12);
    internal::yuv_surface_crops(rd, nv12, crops);
    return rd;
}
// cvtColorP010<CODE[, range, primaries]>(p010[, crops]): the same for a 10-bit decoder surface (CV_16UC1, rows = H * 3 / 2,
// 10-bit codes in the high bits of every sample); R, G, B[, A] arrive on the 0..1023 scale (A = 1023).  The codes reuse
// OpenCV's *_NV12 names for the channel order.
template <cv::ColorConversionCodes CODE, fk::ColorRange CR = fk::Limited, fk::ColorPrimitives CP = fk::bt2020>
inline auto cvtColorP010(const cv::cuda::GpuMat& p010) {
    return internal::yuv_surface_read<fk::P010, CODE, CR, CP>(p010, "cvtColorP010");
}
template <cv::ColorConversionCodes CODE, fk::ColorRange CR = fk::Limited, fk::ColorPrimitives CP = fk::bt2020, size_t N>
inline auto cvtColorP010(const cv::cuda::GpuMat& p010, const std::array<cv::Rect, N>& crops) {
    auto rd = cvtColorP010<CODE, CR, CP>(p010);
    internal::yuv_surface_crops(rd, p010, crops);
    return rd;
}
template <int INTER_F, fk::PixelFormat PF, fk::ColorRange CR, fk::ColorPrimitives CP, bool ALPHA, typename O, bool SW>
inline auto resize(const fk::YuvRead<PF, CR, CP, ALPHA, O, SW>& nv12Read, const cv::Size& dsize) {
    static_assert(isSupportedInterpolation<INTER_F>, "Interpolation type not supported yet.");
    return fk::Resize<(fk::InterpolationType)INTER_F>::build(nv12Read, fk::Size(dsize.width, dsize.height));
}
// resize<INTER, AR>(thatIOp, dsize, backgroundValue): the decoder surface letterboxed into dsize (the AspectRatio modes of the
// batched resize, reference :218-245, on the NV12 read-back); the padding takes backgroundValue (in the IOp's channel order) and
// runs through the rest of the chain like a pixel.  One K4 launch: surface -> detector input tensor.
template <int INTER_F, AspectRatio AR, fk::PixelFormat PF, fk::ColorRange CR, fk::ColorPrimitives CP, bool ALPHA, typename O, bool SW>
inline auto resize(const fk::YuvRead<PF, CR, CP, ALPHA, O, SW>& nv12Read, const cv::Size& dsize, const cv::Scalar& backgroundValue = cv::Scalar()) {
    static_assert(isSupportedInterpolation<INTER_F>, "Interpolation type not supported yet.");
    const float bg[4] = {(float)backgroundValue[0], (float)backgroundValue[1], (float)backgroundValue[2], (float)backgroundValue[3]};
    return fk::Resize<(fk::InterpolationType)INTER_F, (fk::AspectRatio)AR>::build(nv12Read, fk::Size(dsize.width, dsize.height), bg);
}

// ---- warp (reference include/cvGPUSpeedup.cuh:267-442) --------------------------------------------------------------
// warp<WT, InputType[, BATCH]>(input(s), FORWARD transform(s) as CV_64FC1 cv::Mat, dstSize(s)[, usedPlanes, defaultValue]):
// the first IOp of a chain; its output is CV_32F of the input's channels.  The matrices are inverted on the host in
// double (cv::invertAffineTransform / cv::Mat::inv) and narrowed to float, like the reference.
namespace internal {
template <fk::WarpType WT>
inline fk::WarpingParameters<WT> warp_parameters(const cv::Mat& transform_matrix, const cv::Size& dstSize) {
    if (transform_matrix.type() != CV_64FC1) throw std::runtime_error("Transform matrix type should be CV_64FC1.");
    fk::WarpingParameters<WT> p;
    p.dstSize = fk::Size(dstSize.width, dstSize.height);
    if constexpr (WT == fk::WarpType::Affine) {
        cv::Mat inverse_transform_matrix;
        cv::invertAffineTransform(transform_matrix, inverse_transform_matrix);
        for (int y = 0; y < 2; ++y)
            for (int x = 0; x < 3; ++x) p.transformMatrix[y][x] = static_cast<float>(inverse_transform_matrix.ptr<double>(y)[x]);
    } else {
        const cv::Mat inverse_transform_matrix(transform_matrix.inv());
        for (int y = 0; y < 3; ++y)
            for (int x = 0; x < 3; ++x) p.transformMatrix[y][x] = static_cast<float>(inverse_transform_matrix.ptr<double>(y)[x]);
    }
    return p;
}
// the reference's public spellings (include/cvGPUSpeedup.cuh:269-284): the 6 / 9 doubles of an ALREADY INVERTED transform, row-major,
// narrowed to float, plus the target size
inline fk::WarpingParameters<fk::WarpType::Affine> warp_getWarpingAffineParameters(const double* const tm_raw, const cv::Size& dstSize) {
    fk::WarpingParameters<fk::WarpType::Affine> p;
    for (int i = 0; i < 6; ++i) p.transformMatrix[i / 3][i % 3] = static_cast<float>(tm_raw[i]);
    p.dstSize = fk::Size(dstSize.width, dstSize.height);
    return p;
}
inline fk::WarpingParameters<fk::WarpType::Perspective> warp_getWarpingPerspectiveParameters(const double* const tm_raw, const cv::Size& dstSize) {
    fk::WarpingParameters<fk::WarpType::Perspective> p;
    for (int i = 0; i < 9; ++i) p.transformMatrix[i / 3][i % 3] = static_cast<float>(tm_raw[i]);
    p.dstSize = fk::Size(dstSize.width, dstSize.height);
    return p;
}
// the batch spellings (include/cvGPUSpeedup.cuh:311-377): FORWARD transforms in, the device's (inverted, float) parameters out; entries at
// and beyond usedPlanes stay value-initialised
template <fk::WarpType WT, size_t BATCH>
inline std::array<fk::WarpingParameters<WT>, BATCH> warp_batchParameters(const std::array<cv::Mat, BATCH>& transform_matrices,
                                                                         const std::array<cv::Size, BATCH>& dstSize, const int& usedPlanes = BATCH) {
    std::array<fk::WarpingParameters<WT>, BATCH> out{};
    for (int i = 0; i < usedPlanes && i < (int)BATCH; ++i) out[(size_t)i] = warp_parameters<WT>(transform_matrices[(size_t)i], dstSize[(size_t)i]);
    return out;
}
} // namespace internal

template <fk::WarpType WT, int InputType = CV_8UC3>
inline auto warp(const cv::cuda::GpuMat& input, const cv::Mat& transform_matrix, const cv::Size& dstSize) {
    if (InputType != input.type()) throw std::runtime_error("Input type does not match the input type of the operation.");
    fk::WarpRead<WT, CUDA_T(InputType)> rd;
    rd.planes.assign(1, fk::image2d(gpuMat2RawPtr2D<CUDA_T(InputType)>(input)));
    rd.params.assign(1, internal::warp_parameters<WT>(transform_matrix, dstSize));
    rd.used = 1;
    return rd;
}
template <fk::WarpType WT, int InputType, size_t BATCH>
inline auto warp(const std::array<cv::cuda::GpuMat, BATCH>& inputs, const std::array<cv::Mat, BATCH>& transform_matrices,
                 const std::array<cv::Size, BATCH>& dstSize, const int& usedPlanes, const cv::Scalar& defaultValue) {
    fk::WarpRead<WT, CUDA_T(InputType)> rd;
    rd.planes.resize(BATCH, cvgs_image2d{nullptr, 0, 0, 0, 0});
    rd.params.resize(BATCH);
    for (size_t i = 0; i < BATCH; ++i) rd.params[i].dstSize = fk::Size(dstSize[i].width, dstSize[i].height); // also of unused planes
    for (int i = 0; i < usedPlanes && i < (int)BATCH; ++i) {
        if (InputType != inputs[(size_t)i].type()) throw std::runtime_error("Input type does not match the input type of the operation.");
        rd.planes[(size_t)i] = fk::image2d(gpuMat2RawPtr2D<CUDA_T(InputType)>(inputs[(size_t)i]));
        rd.params[(size_t)i] = internal::warp_parameters<WT>(transform_matrices[(size_t)i], dstSize[(size_t)i]);
    }
    rd.used = usedPlanes;
    for (int c = 0; c < CV_MAT_CN(InputType); ++c) rd.background[c] = static_cast<float>(defaultValue[c]); // Scalar() = zeros
    return rd;
}
template <fk::WarpType WT, int InputType, size_t BATCH>
inline auto warp(const std::array<cv::cuda::GpuMat, BATCH>& inputs, const std::array<cv::Mat, BATCH>& transform_matrices,
                 const std::array<cv::Size, BATCH>& dstSize) {
    return warp<WT, InputType>(inputs, transform_matrices, dstSize, (int)BATCH, cv::Scalar());
}
template <fk::WarpType WT, int InputType, size_t BATCH>
inline auto warp(const std::array<cv::cuda::GpuMat, BATCH>& inputs, const std::array<cv::Mat, BATCH>& transform_matrices,
                 const cv::Size& dstSize) {
    std::array<cv::Size, BATCH> sizes;
    sizes.fill(dstSize);
    return warp<WT, InputType>(inputs, transform_matrices, sizes, (int)BATCH, cv::Scalar());
}
template <fk::WarpType WT, int InputType, size_t BATCH>
inline auto warp(const std::array<cv::cuda::GpuMat, BATCH>& inputs, const std::array<cv::Mat, BATCH>& transform_matrices,
                 const cv::Size& dstSize, const int& usedPlanes, const cv::Scalar& defaultValue) {
    std::array<cv::Size, BATCH> sizes;
    sizes.fill(dstSize);
    return warp<WT, InputType>(inputs, transform_matrices, sizes, usedPlanes, defaultValue);
}

// crop == an ROI view (zero cost): same pointer arithmetic as GpuMat::operator()(Rect); Rect2d doubles truncate
// (reference :247-265,444-447: uint x/y, int width/height)
namespace internal {
inline fk::Rect fk_rect(const cv::Rect2d& r) { return fk::Rect(static_cast<uint>(r.x), static_cast<uint>(r.y), static_cast<int>(r.width), static_cast<int>(r.height)); }
} // namespace internal
inline auto crop(const cv::Rect2d& rect) { return fk::CropSpec{internal::fk_rect(rect)}; }
template <size_t BATCH>
inline auto crop(const std::array<cv::Rect2d, BATCH>& rects) {
    fk::BatchCropSpec<BATCH> spec;
    for (size_t i = 0; i < BATCH; ++i) spec.rects[i] = internal::fk_rect(rects[i]);
    return spec;
}
template <typename BackIOp> inline auto crop(const BackIOp& backIOp, const cv::Rect2d& rect) { return backIOp.then(crop(rect)); }
template <typename BackIOp, size_t BATCH>
inline auto crop(const BackIOp& backIOp, const std::array<cv::Rect2d, BATCH>& rects) { return backIOp.then(crop(rects)); }
inline cv::cuda::GpuMat crop(const cv::cuda::GpuMat& input, const cv::Rect2d& rect) { return input(cv::Rect(rect)); }
template <size_t BATCH>
inline std::array<cv::cuda::GpuMat, BATCH> crop(const cv::cuda::GpuMat& input, const std::array<cv::Rect2d, BATCH>& rects) {
    std::array<cv::cuda::GpuMat, BATCH> out;
    for (size_t i = 0; i < BATCH; ++i) out[i] = input(cv::Rect(rects[i]));
    return out;
}

// ---- executeOperations ---------------------------------------------------------------------------------------------
template <bool ENABLE_THREAD_FUSION, typename... IOpTypes>
inline void executeOperations(const cv::cuda::Stream& stream, const IOpTypes&... iops) {
    fk::executeOperations<ENABLE_THREAD_FUSION>(cv::cuda::StreamAccessor::getStream(stream), iops...);
}
template <typename... IOpTypes>
inline void executeOperations(const cv::cuda::Stream& stream, const IOpTypes&... iops) {
    executeOperations<true>(stream, iops...);
}

namespace internal {
template <typename First, typename... Rest> struct first_of { using type = First; };
template <typename... T> using first_input_t = typename first_of<T...>::type::InputType;
template <typename... T> struct last_of;
template <typename T> struct last_of<T> { using type = T; };
template <typename T, typename... R> struct last_of<T, R...> : last_of<R...> {};
template <typename... T> using last_output_t = typename last_of<T...>::type::OutputType;

template <typename T, size_t N>
inline fk::BatchPixelRead<T> batchRead(const std::array<cv::cuda::GpuMat, N>& in, size_t active, const cv::Scalar& def) {
    fk::BatchPixelRead<T> rd;
    rd.planes.resize(N, cvgs_image2d{nullptr, 0, 0, 0, 0});
    for (size_t i = 0; i < active && i < N; ++i) rd.planes[i] = fk::image2d(gpuMat2RawPtr2D<T>(in[i]));
    rd.used = (int)active;
    const T d = cvScalar2CUDAV_t<T>::get(def); // narrowed to the INPUT type, as the reference does
    fk::vec_to_floats(d, rd.background);
    return rd;
}
} // namespace internal

// input GpuMat given: a per-pixel read of it is prepended
template <bool TF, typename... IOpTypes>
inline void executeOperations(const cv::cuda::GpuMat& input, const cv::cuda::Stream& stream, const IOpTypes&... iops) {
    using In = internal::first_input_t<IOpTypes...>;
    const fk::Read<fk::PerThreadRead<fk::_2D, In>> read{gpuMat2RawPtr2D<In>(input)};
    fk::executeOperations<TF>(cv::cuda::StreamAccessor::getStream(stream), read, iops...);
}
template <typename... IOpTypes>
inline void executeOperations(const cv::cuda::GpuMat& input, const cv::cuda::Stream& stream, const IOpTypes&... iops) {
    executeOperations<true>(input, stream, iops...);
}
// input and output GpuMat given: read prepended, per-pixel write appended
template <bool TF, typename... IOpTypes>
inline void executeOperations(const cv::cuda::GpuMat& input, cv::cuda::GpuMat& output, cv::cuda::Stream& stream, const IOpTypes&... iops) {
    using In = internal::first_input_t<IOpTypes...>;
    using Out = internal::last_output_t<IOpTypes...>;
    const fk::Read<fk::PerThreadRead<fk::_2D, In>> read{gpuMat2RawPtr2D<In>(input)};
    const fk::Write<fk::PerThreadWrite<fk::_2D, Out>> wr{gpuMat2RawPtr2D<Out>(output)};
    fk::executeOperations<TF>(cv::cuda::StreamAccessor::getStream(stream), read, iops..., wr);
}
template <typename... IOpTypes>
inline void executeOperations(const cv::cuda::GpuMat& input, cv::cuda::GpuMat& output, cv::cuda::Stream& stream, const IOpTypes&... iops) {
    executeOperations<true>(input, output, stream, iops...);
}
// batch reads
template <bool TF, size_t Batch, typename... IOpTypes>
inline void executeOperations(const std::array<cv::cuda::GpuMat, Batch>& input, const size_t& activeBatch, const cv::Scalar& defaultValue,
                              const cv::cuda::Stream& stream, const IOpTypes&... iops) {
    using In = internal::first_input_t<IOpTypes...>;
    fk::executeOperations<TF>(cv::cuda::StreamAccessor::getStream(stream), internal::batchRead<In>(input, activeBatch, defaultValue), iops...);
}
template <bool TF, size_t Batch, typename... IOpTypes>
inline void executeOperations(const std::array<cv::cuda::GpuMat, Batch>& input, const cv::cuda::Stream& stream, const IOpTypes&... iops) {
    using In = internal::first_input_t<IOpTypes...>;
    fk::executeOperations<TF>(cv::cuda::StreamAccessor::getStream(stream), internal::batchRead<In>(input, Batch, cv::Scalar()), iops...);
}
template <size_t Batch, typename... IOpTypes>
inline void executeOperations(const std::array<cv::cuda::GpuMat, Batch>& input, const size_t& activeBatch, const cv::Scalar& defaultValue,
                              const cv::cuda::Stream& stream, const IOpTypes&... iops) {
    executeOperations<true>(input, activeBatch, defaultValue, stream, iops...);
}
template <size_t Batch, typename... IOpTypes>
inline void executeOperations(const std::array<cv::cuda::GpuMat, Batch>& input, const cv::cuda::Stream& stream, const IOpTypes&... iops) {
    executeOperations<true>(input, stream, iops...);
}
// batch reads with a tensor output (one packed image per row of `output`)
template <bool TF, size_t Batch, typename... IOpTypes>
inline void executeOperations(const std::array<cv::cuda::GpuMat, Batch>& input, const size_t& activeBatch, const cv::Scalar& defaultValue,
                              const cv::cuda::GpuMat& output, const cv::Size& outputPlane, const cv::cuda::Stream& stream, const IOpTypes&... iops) {
    using In = internal::first_input_t<IOpTypes...>;
    using Out = internal::last_output_t<IOpTypes...>;
    const fk::Write<fk::PerThreadWrite<fk::_3D, Out>> wr{gpuMat2Tensor<Out>(output, outputPlane, 1).ptr()};
    fk::executeOperations<TF>(cv::cuda::StreamAccessor::getStream(stream), internal::batchRead<In>(input, activeBatch, defaultValue), iops..., wr);
}
template <bool TF, size_t Batch, typename... IOpTypes>
inline void executeOperations(const std::array<cv::cuda::GpuMat, Batch>& input, const cv::cuda::GpuMat& output, const cv::Size& outputPlane,
                              const cv::cuda::Stream& stream, const IOpTypes&... iops) {
    executeOperations<TF>(input, Batch, cv::Scalar(), output, outputPlane, stream, iops...);
}
template <size_t Batch, typename... IOpTypes>
inline void executeOperations(const std::array<cv::cuda::GpuMat, Batch>& input, const size_t& activeBatch, const cv::Scalar& defaultValue,
                              const cv::cuda::GpuMat& output, const cv::Size& outputPlane, const cv::cuda::Stream& stream, const IOpTypes&... iops) {
    executeOperations<true>(input, activeBatch, defaultValue, output, outputPlane, stream, iops...);
}
template <size_t Batch, typename... IOpTypes>
inline void executeOperations(const std::array<cv::cuda::GpuMat, Batch>& input, const cv::cuda::GpuMat& output, const cv::Size& outputPlane,
                              const cv::cuda::Stream& stream, const IOpTypes&... iops) {
    executeOperations<true>(input, output, outputPlane, stream, iops...);
}

// ---- CircularTensor --------------------------------------------------------------------------------------------------
// MIRRORED = true selects the opt-in mirrored-ring layout (engine extension; see include/cvgs_hip.h): data() then moves
// with every update instead of being stable, and an update costs one pass over the new frame only.  CAPTURABLE = true makes
// update() capturable into a HIP graph (engine extension: the update count lives on the device).
template <int I, int O, int COLOR_PLANES, int BATCH, fk::CircularTensorOrder CT_ORDER, fk::ColorPlanes CP_MODE = fk::ColorPlanes::Standard,
          bool MIRRORED = false, bool CAPTURABLE = false>
class CircularTensor : public fk::CircularTensor<CUDA_T(O), COLOR_PLANES, BATCH, CT_ORDER, CP_MODE, MIRRORED, CAPTURABLE> {
    using Base = fk::CircularTensor<CUDA_T(O), COLOR_PLANES, BATCH, CT_ORDER, CP_MODE, MIRRORED, CAPTURABLE>;
public:
    CircularTensor() = default;
    CircularTensor(const uint& width_, const uint& height_, const int& deviceID_ = 0) : Base(width_, height_, deviceID_) {}

    // new frame read per pixel from a GpuMat of type I
    template <typename... IOpTypes>
    void update(const cv::cuda::Stream& stream, const cv::cuda::GpuMat& input, const IOpTypes&... iops) {
        const fk::Read<fk::PerThreadRead<fk::_2D, CUDA_T(I)>> read{gpuMat2RawPtr2D<CUDA_T(I)>(input)};
        Base::update(cv::cuda::StreamAccessor::getStream(stream), read, iops...);
    }
    // first IOp is already a read (e.g. cvGS::resize(...)): "push with resize+normalize"
    template <typename... IOpTypes>
    void update(const cv::cuda::Stream& stream, const IOpTypes&... iops) {
        Base::update(cv::cuda::StreamAccessor::getStream(stream), iops...);
    }
    CUDA_T(O)* data() { return this->ptr_a.data; }
};

// ---- launch batching (engine extension, see fk::ChainBatch): several independent chains, ONE kernel launch -------------
class ChainBatch {
public:
    template <typename... IOps> void add(const IOps&... iops) { batch_.add(iops...); }
    void execute(const cv::cuda::Stream& stream) { batch_.execute(cv::cuda::StreamAccessor::getStream(stream)); }
    void clear() { batch_.clear(); }
    size_t size() const { return batch_.size(); }
private:
    fk::ChainBatch batch_;
};

// ---- device-side descriptor queue (engine extension, see fk::Queue): executeOperations without a launch per call -------
class Queue {
public:
    explicit Queue(int device = -1 /* the current device */, int depth = 0, double idle_us = 0.0) : q_(device, depth, idle_us) {}
    template <typename... IOps> uint64_t submit(const IOps&... iops) { return q_.submit(iops...); }
    void wait(uint64_t ticket, double timeout_s = 10.0) { q_.wait(ticket, timeout_s); }
    void wait(uint64_t ticket, const cv::cuda::Stream& consumer) { q_.wait(ticket, cv::cuda::StreamAccessor::getStream(consumer)); }
    uint64_t recover() { return q_.recover(); }
    fk::Queue& fk() { return q_; }
private:
    fk::Queue q_;
};
// The reference's own call shape on the queue: after attachQueue(stream, queue) every cvGS::executeOperations(stream, iops...) whose
// chain the server takes is submitted stream-ordered (behind the stream's earlier work, in front of its later work; no host
// synchronisation -- include/cvGPUSpeedup.cuh:464-473's contract), everything else is the ordinary launch.  The latency policy decides per
// call: one gate kernel per call is the price of stream order, so the server takes ChainBatch::execute(stream) ticks of >= 8 chains (2.5 us
// per 50-crop frame at 16 frames per tick against 8.9 us as launches) and single calls stay launches -- an attached stream is never slower
// than an unattached one (minGroup overrides the 8; 1 = always the server).  deferWait: the stream is not held on each batch;
// cvGS::fence(stream) orders the consumer (several batches of one stream then overlap on the device).
inline void attachQueue(const cv::cuda::Stream& stream, Queue& queue, bool deferWait = false, int minGroup = 0) {
    queue.fk().attach(cv::cuda::StreamAccessor::getStream(stream), deferWait, minGroup);
}
// Recorded ticks: the reference's loop UNCHANGED -- one cvGS::executeOperations(stream, ...) per camera, one stream.waitForCompletion() (or
// cvGS::fence(stream)) per tick -- on a stream attached with attachQueueTicks: the calls are recorded (0.14 us each) and go to the queue's
// server `tick` at a time behind one gate; waitForCompletion / fence submit what is pending first.  Sources and tensors of recorded calls
// are in flight until then (the deferred-wait contract).
inline void attachQueueTicks(const cv::cuda::Stream& stream, Queue& queue, int tick = 16) {
    queue.fk().attachTicks(cv::cuda::StreamAccessor::getStream(stream), tick);
}
// The same with NO queue: the recorded calls are launched `tick` at a time as ONE multi-chain kernel (cvgs_execute_many), strictly
// stream-ordered, nothing resident on the GPU between ticks.  stopRecording (or detachQueue) launches what is pending and ends it.
inline void recordTicks(const cv::cuda::Stream& stream, int tick = 16) { fk::recordTicks(cv::cuda::StreamAccessor::getStream(stream), tick); }
inline void stopRecording(const cv::cuda::Stream& stream) { fk::stopRecording(cv::cuda::StreamAccessor::getStream(stream)); }
inline void detachQueue(const cv::cuda::Stream& stream) { fk::Queue::detach(cv::cuda::StreamAccessor::getStream(stream)); }
inline void fence(const cv::cuda::Stream& stream) { fk::Queue::fence(cv::cuda::StreamAccessor::getStream(stream)); }
inline bool lastTicket(const cv::cuda::Stream& stream, uint64_t* ticket) { return fk::Queue::lastTicket(cv::cuda::StreamAccessor::getStream(stream), ticket); }
// cvGS::executeOperations(queue, iops...): the stream form's IOps, a ticket instead of a stream position
template <typename... IOpTypes>
inline uint64_t executeOperations(Queue& queue, const IOpTypes&... iops) {
    return queue.submit(iops...);
}

} // namespace cvGS
