// cv2cuda_types.h -- OpenCV type code -> device vector type shims (the role of the reference's
// include/cv2cuda_types.cuh:34-96), on HIP vector types.  The mapping is computed from the type code's
// depth and channel count instead of being listed case by case.
#pragma once

#include <hip/hip_vector_types.h>

#include <type_traits>

#include "cvgs/cv_shim.h"

namespace cvGS {

namespace detail {
template <int DEPTH> struct depth_base;
template <> struct depth_base<CV_8U> { using type = uchar; };
template <> struct depth_base<CV_8S> { using type = char; };
template <> struct depth_base<CV_16U> { using type = ushort; };
template <> struct depth_base<CV_16S> { using type = short; };
template <> struct depth_base<CV_32S> { using type = int; };
template <> struct depth_base<CV_32F> { using type = float; };
template <> struct depth_base<CV_64F> { using type = double; };
template <> struct depth_base<CV_16F> { using type = cvgs::half_t; }; // engine extension: half-precision hand-off

// scalar for one channel, HIP_vector_type<base, N> otherwise (uchar3, float4, ...)
template <typename B, int CN> struct vec_of { using type = HIP_vector_type<B, CN>; };
template <typename B> struct vec_of<B, 1> { using type = B; };
} // namespace detail

template <int CV_TYPE>
struct cv2cuda_t {
    static_assert(CV_MAT_CN(CV_TYPE) >= 1 && CV_MAT_CN(CV_TYPE) <= 4 && CV_MAT_DEPTH(CV_TYPE) <= CV_16F,
                  "unsupported OpenCV type code");
    using base = typename detail::depth_base<CV_MAT_DEPTH(CV_TYPE)>::type;
    using type = typename detail::vec_of<base, CV_MAT_CN(CV_TYPE)>::type;
};

// --- reverse direction: device vector type -> (base, channels, CV type code) -----------------------
template <typename T> struct vector_traits {
    using base = T;
    static constexpr int cn = 1;
};
template <typename B, unsigned N> struct vector_traits<HIP_vector_type<B, N>> {
    using base = B;
    static constexpr int cn = (int)N;
};

template <typename B> struct base_depth;
template <> struct base_depth<uchar> { static constexpr int value = CV_8U; };
template <> struct base_depth<char> { static constexpr int value = CV_8S; };
template <> struct base_depth<schar> { static constexpr int value = CV_8S; };
template <> struct base_depth<ushort> { static constexpr int value = CV_16U; };
template <> struct base_depth<short> { static constexpr int value = CV_16S; };
template <> struct base_depth<int> { static constexpr int value = CV_32S; };
template <> struct base_depth<uint> { static constexpr int value = CV_32S; };
template <> struct base_depth<float> { static constexpr int value = CV_32F; };
template <> struct base_depth<double> { static constexpr int value = CV_64F; };
template <> struct base_depth<cvgs::half_t> { static constexpr int value = CV_16F; };

template <typename T>
constexpr int cv_type_of = CV_MAKETYPE(base_depth<typename vector_traits<T>::base>::value, vector_traits<T>::cn);

// --- supported-code lists (reference include/cv2cuda_types.cuh:63-92) -------------------------------
template <int... CODES> struct CodesList {};

template <int CODE, typename LIST> struct one_of_c;
template <int CODE, int... CODES> struct one_of_c<CODE, CodesList<CODES...>> : std::bool_constant<((CODE == CODES) || ...)> {};

using SupportedColorConversions =
    CodesList<cv::COLOR_BGR2BGRA, cv::COLOR_BGRA2BGR, cv::COLOR_BGR2RGBA, cv::COLOR_BGRA2RGB, cv::COLOR_BGR2RGB,
              cv::COLOR_BGRA2RGBA, cv::COLOR_RGB2GRAY, cv::COLOR_RGBA2GRAY, cv::COLOR_BGR2GRAY, cv::COLOR_BGRA2GRAY>;
using SupportedInterpolations = CodesList<cv::INTER_LINEAR>;

template <int CODE> constexpr bool isSupportedColorConversion = one_of_c<CODE, SupportedColorConversions>::value;
template <int CODE> constexpr bool isSupportedInterpolation = one_of_c<CODE, SupportedInterpolations>::value;

} // namespace cvGS

#define CUDA_T(CV_TYPE) typename cvGS::cv2cuda_t<CV_TYPE>::type
#define BASE_CUDA_T(CV_TYPE) typename cvGS::cv2cuda_t<CV_MAT_DEPTH(CV_TYPE)>::type
