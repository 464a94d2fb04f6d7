// cvGPUSpeedupHelpers.h -- cv::Scalar <-> device vector helpers (the role of the reference's
// include/cvGPUSpeedupHelpers.cuh:23-69).
#pragma once

#include "cv2cuda_types.h"
#include "cvgs/fk_compat.h"

namespace cvGS {

// cvScalar_set<T>(v): a Scalar with the first CV_MAT_CN(T) entries set to v
template <int T>
inline cv::Scalar cvScalar_set(const BASE_CUDA_T(T)& value) {
    cv::Scalar s;
    for (int c = 0; c < CV_MAT_CN(T); ++c) s[c] = (double)value;
    return s;
}

// double -> base type narrowing per channel (static_cast, as the reference does)
template <typename V>
struct cvScalar2CUDAV_t {
    static V get(const cv::Scalar& val) {
        using B = fk::VBase<V>;
        if constexpr (fk::cn<V> == 1) return static_cast<B>(val[0]);
        else if constexpr (fk::cn<V> == 2) return V{static_cast<B>(val[0]), static_cast<B>(val[1])};
        else if constexpr (fk::cn<V> == 3) return V{static_cast<B>(val[0]), static_cast<B>(val[1]), static_cast<B>(val[2])};
        else return V{static_cast<B>(val[0]), static_cast<B>(val[1]), static_cast<B>(val[2]), static_cast<B>(val[3])};
    }
};

template <int I>
struct cvScalar2CUDAV {
    static CUDA_T(I) get(const cv::Scalar& val) { return cvScalar2CUDAV_t<CUDA_T(I)>::get(val); }
};

} // namespace cvGS
