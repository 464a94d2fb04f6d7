// cv_shim.h -- the slice of OpenCV's core/cuda API the cvGS facade and its tests touch, implemented on the
// HIP runtime so the facade builds where OpenCV does not exist (this image).  Same names, same numeric
// constants as OpenCV 4.x, so code written against <opencv2/core/cuda.hpp> compiles unchanged for the hot path.
// Define CVGS_USE_OPENCV to use a real OpenCV instead (the facade only needs the names below).
//
// Covered: cv::Scalar, Size, Point2d, Rect2d, Rect, Mat (host, dense), cuda::GpuMat (refcounted, pitched,
// ROI views, upload/download/setTo/convertTo-free), cuda::Stream + StreamAccessor, CV_* type macros,
// INTER_LINEAR, ColorConversionCodes.  Everything here is host-side plumbing; no pixel arithmetic of the hot
// path lives in this file (setTo fills on the host and uploads: test scaffolding, like the reference's
// GpuMat(rows, cols, type, Scalar) constructor use).
#pragma once

#ifdef CVGS_USE_OPENCV
#include <opencv2/core.hpp>
#include <opencv2/core/cuda.hpp>
#include <opencv2/core/cuda_stream_accessor.hpp>
#include <opencv2/imgproc.hpp>
#else

#include <hip/hip_runtime_api.h>

#include "half.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <utility>
#include <string>
#include <vector>

typedef unsigned char uchar;
typedef signed char schar;
typedef unsigned short ushort;
typedef unsigned int uint;

#define CV_CN_SHIFT 3
#define CV_DEPTH_MAX (1 << CV_CN_SHIFT)
#define CV_8U 0
#define CV_8S 1
#define CV_16U 2
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6
#define CV_16F 7 /* half-precision hand-off type (engine extension; OpenCV 4's numeric value) */
#define CV_MAT_DEPTH_MASK (CV_DEPTH_MAX - 1)
#define CV_MAT_DEPTH(flags) ((flags) & CV_MAT_DEPTH_MASK)
#define CV_MAKETYPE(depth, cn) (CV_MAT_DEPTH(depth) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_MAKE_TYPE CV_MAKETYPE
#define CV_MAT_CN(flags) ((((flags) >> CV_CN_SHIFT) & 63) + 1)
#define CVGS_DECL_TYPES(D)                                                                                      \
    constexpr int CV_##D##C1 = CV_MAKETYPE(CV_##D, 1), CV_##D##C2 = CV_MAKETYPE(CV_##D, 2),                    \
                  CV_##D##C3 = CV_MAKETYPE(CV_##D, 3), CV_##D##C4 = CV_MAKETYPE(CV_##D, 4);
CVGS_DECL_TYPES(8U) CVGS_DECL_TYPES(8S) CVGS_DECL_TYPES(16U) CVGS_DECL_TYPES(16S) CVGS_DECL_TYPES(32S)
CVGS_DECL_TYPES(32F) CVGS_DECL_TYPES(64F) CVGS_DECL_TYPES(16F)
#undef CVGS_DECL_TYPES

namespace cv {

enum InterpolationFlags { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2, INTER_AREA = 3 };

enum ColorConversionCodes {
    COLOR_BGR2BGRA = 0, COLOR_RGB2RGBA = COLOR_BGR2BGRA,
    COLOR_BGRA2BGR = 1, COLOR_RGBA2RGB = COLOR_BGRA2BGR,
    COLOR_BGR2RGBA = 2, COLOR_RGB2BGRA = COLOR_BGR2RGBA,
    COLOR_RGBA2BGR = 3, COLOR_BGRA2RGB = COLOR_RGBA2BGR,
    COLOR_BGR2RGB = 4, COLOR_RGB2BGR = COLOR_BGR2RGB,
    COLOR_BGRA2RGBA = 5, COLOR_RGBA2BGRA = COLOR_BGRA2RGBA,
    COLOR_BGR2GRAY = 6, COLOR_RGB2GRAY = 7, COLOR_GRAY2BGR = 8, COLOR_GRAY2RGB = COLOR_GRAY2BGR,
    COLOR_GRAY2BGRA = 9, COLOR_GRAY2RGBA = COLOR_GRAY2BGRA, COLOR_BGRA2GRAY = 10, COLOR_RGBA2GRAY = 11,
    COLOR_YUV2RGB_NV12 = 90, COLOR_YUV2BGR_NV12 = 91, COLOR_YUV2RGBA_NV12 = 94, COLOR_YUV2BGRA_NV12 = 95
};

struct Scalar {
    double val[4];
    Scalar() : val{0, 0, 0, 0} {}
    Scalar(double v0) : val{v0, 0, 0, 0} {}
    Scalar(double v0, double v1, double v2 = 0, double v3 = 0) : val{v0, v1, v2, v3} {}
    double& operator[](int i) { return val[i]; }
    const double& operator[](int i) const { return val[i]; }
    bool operator==(const Scalar& o) const { return !std::memcmp(val, o.val, sizeof(val)); }
    static Scalar all(double v) { return Scalar(v, v, v, v); }
};

struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size& o) const { return width == o.width && height == o.height; }
};

struct Point2d {
    double x = 0, y = 0;
    Point2d() = default;
    Point2d(double x_, double y_) : x(x_), y(y_) {}
};

struct Point2f {
    float x = 0, y = 0;
    Point2f() = default;
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

struct Rect2d {
    double x = 0, y = 0, width = 0, height = 0;
    Rect2d() = default;
    Rect2d(double x_, double y_, double w, double h) : x(x_), y(y_), width(w), height(h) {}
    Rect2d(const Point2d& a, const Point2d& b)
        : x(std::fmin(a.x, b.x)), y(std::fmin(a.y, b.y)), width(std::fabs(b.x - a.x)), height(std::fabs(b.y - a.y)) {}
};

struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() = default;
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
    Rect(const Rect2d& r) : x((int)r.x), y((int)r.y), width((int)r.width), height((int)r.height) {}
};

inline size_t cvgs_elem_size(int type) {
    static const int bytes[8] = {1, 1, 2, 2, 4, 4, 8, 2};
    return (size_t)bytes[CV_MAT_DEPTH(type)] * CV_MAT_CN(type);
}

inline void cvgs_hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// Host matrix: dense rows, owns its storage (or wraps user memory).
class Mat {
public:
    int rows = 0, cols = 0;
    uchar* data = nullptr;
    size_t step = 0;
    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type) { create(s.height, s.width, type); }
    Mat(int r, int c, int type, const Scalar& v) { create(r, c, type); setTo(v); }
    Mat(Size s, int type, const Scalar& v) { create(s.height, s.width, type); setTo(v); }
    Mat(int r, int c, int type, void* user, size_t step_ = 0)
        : rows(r), cols(c), data((uchar*)user), step(step_ ? step_ : c * cvgs_elem_size(type)), type_(type) {}
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type; step = c * cvgs_elem_size(type);
        store_ = std::make_shared<std::vector<uchar>>((size_t)r * step);
        data = store_->data();
    }
    int type() const { return type_; }
    int depth() const { return CV_MAT_DEPTH(type_); }
    int channels() const { return CV_MAT_CN(type_); }
    size_t elemSize() const { return cvgs_elem_size(type_); }
    // cv::Mat::inv() for the one case the warp builders need: a 3x3 CV_64FC1 matrix, OpenCV's closed form
    // (adjugate / determinant; a singular matrix gives zeros)
    Mat inv() const {
        if (rows != 3 || cols != 3 || type_ != CV_MAKETYPE(CV_64F, 1)) throw std::runtime_error("Mat::inv: 3x3 CV_64FC1 only");
        const double* r0 = ptr<double>(0); const double* r1 = ptr<double>(1); const double* r2 = ptr<double>(2);
        const double det = r0[0] * (r1[1] * r2[2] - r1[2] * r2[1]) - r0[1] * (r1[0] * r2[2] - r1[2] * r2[0]) +
                           r0[2] * (r1[0] * r2[1] - r1[1] * r2[0]);
        Mat out(3, 3, type_);
        double* o = out.ptr<double>(0);
        if (det == 0.0) { for (int i = 0; i < 9; ++i) o[i] = 0.0; return out; }
        const double d = 1.0 / det;
        o[0] = (r1[1] * r2[2] - r1[2] * r2[1]) * d; o[1] = (r0[2] * r2[1] - r0[1] * r2[2]) * d; o[2] = (r0[1] * r1[2] - r0[2] * r1[1]) * d;
        o[3] = (r1[2] * r2[0] - r1[0] * r2[2]) * d; o[4] = (r0[0] * r2[2] - r0[2] * r2[0]) * d; o[5] = (r0[2] * r1[0] - r0[0] * r1[2]) * d;
        o[6] = (r1[0] * r2[1] - r1[1] * r2[0]) * d; o[7] = (r0[1] * r2[0] - r0[0] * r2[1]) * d; o[8] = (r0[0] * r1[1] - r0[1] * r1[0]) * d;
        return out;
    }
    bool empty() const { return !data; }
    Size size() const { return Size(cols, rows); }
    template <typename T> T* ptr(int y = 0) { return (T*)(data + (size_t)y * step); }
    template <typename T> const T* ptr(int y = 0) const { return (const T*)(data + (size_t)y * step); }
    template <typename T> T& at(int y, int x) { return ptr<T>(y)[x]; }
    template <typename T> const T& at(int y, int x) const { return ptr<T>(y)[x]; }
    Mat row(int y) const { Mat m = *this; m.rows = 1; m.data = data + (size_t)y * step; return m; }
    // saturating per-channel fill, like cv::Mat::setTo / the Scalar constructors
    void setTo(const Scalar& v) {
        const int cn = channels(), d = depth();
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x)
                for (int c = 0; c < cn; ++c) store(y, x * cn + c, d, v[c]);
    }

private:
    static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
    void store(int y, int e, int d, double v) {
        uchar* row = data + (size_t)y * step;
        switch (d) {
        case CV_8U: row[e] = (uchar)std::nearbyint(clampd(v, 0, 255)); break;
        case CV_8S: ((schar*)row)[e] = (schar)std::nearbyint(clampd(v, -128, 127)); break;
        case CV_16U: ((ushort*)row)[e] = (ushort)std::nearbyint(clampd(v, 0, 65535)); break;
        case CV_16S: ((short*)row)[e] = (short)std::nearbyint(clampd(v, -32768, 32767)); break;
        case CV_32S: ((int*)row)[e] = (int)std::nearbyint(clampd(v, -2147483648.0, 2147483647.0)); break;
        case CV_32F: ((float*)row)[e] = (float)v; break;
        case CV_16F: ((cvgs::half_t*)row)[e] = (cvgs::half_t)v; break;
        default: ((double*)row)[e] = v; break;
        }
    }
    int type_ = 0;
    std::shared_ptr<std::vector<uchar>> store_;
};

// cv::Mat_<T>(rows, cols) << a, b, c, ...  (the comma initialiser the reference's warp test builds its matrices with)
template <typename T> struct cvgs_depth_of;
template <> struct cvgs_depth_of<uchar> { static constexpr int value = CV_8U; };
template <> struct cvgs_depth_of<short> { static constexpr int value = CV_16S; };
template <> struct cvgs_depth_of<ushort> { static constexpr int value = CV_16U; };
template <> struct cvgs_depth_of<int> { static constexpr int value = CV_32S; };
template <> struct cvgs_depth_of<float> { static constexpr int value = CV_32F; };
template <> struct cvgs_depth_of<double> { static constexpr int value = CV_64F; };

template <typename T> class Mat_;
template <typename T> class MatCommaInitializer_ {
public:
    MatCommaInitializer_(Mat_<T>* m, T first) : m_(m), i_(0) { put(first); }
    template <typename V> MatCommaInitializer_& operator,(V v) { put((T)v); return *this; }
    operator Mat() const;
    operator Mat_<T>() const;
private:
    void put(T v);
    Mat_<T>* m_;
    size_t i_;
};
template <typename T> class Mat_ : public Mat {
public:
    Mat_() = default;
    Mat_(int r, int c) : Mat(r, c, CV_MAKETYPE(cvgs_depth_of<T>::value, 1)) {}
    template <typename V> MatCommaInitializer_<T> operator<<(V v) { return MatCommaInitializer_<T>(this, (T)v); }
    T& operator()(int y, int x) { return this->template ptr<T>(y)[x]; }
};
template <typename T> void MatCommaInitializer_<T>::put(T v) {
    if (i_ >= (size_t)m_->rows * m_->cols) throw std::runtime_error("Mat_ comma initialiser: too many values");
    m_->template ptr<T>((int)(i_ / m_->cols))[i_ % m_->cols] = v;
    ++i_;
}
template <typename T> MatCommaInitializer_<T>::operator Mat() const { return *m_; }
template <typename T> MatCommaInitializer_<T>::operator Mat_<T>() const { return *m_; }

// cv::invertAffineTransform (2x3, CV_64F or CV_32F), the arithmetic of OpenCV's imgwarp.cpp in double
inline void invertAffineTransform(const Mat& M, Mat& iM) {
    if (M.rows != 2 || M.cols != 3 || (M.type() != CV_MAKETYPE(CV_64F, 1) && M.type() != CV_MAKETYPE(CV_32F, 1)))
        throw std::runtime_error("invertAffineTransform: 2x3 CV_32FC1 / CV_64FC1 only");
    double m[6];
    for (int y = 0; y < 2; ++y)
        for (int x = 0; x < 3; ++x) m[y * 3 + x] = M.type() == CV_MAKETYPE(CV_64F, 1) ? M.ptr<double>(y)[x] : (double)M.ptr<float>(y)[x];
    double D = m[0] * m[4] - m[1] * m[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = m[4] * D, A22 = m[0] * D, A12 = -m[1] * D, A21 = -m[3] * D;
    const double b1 = -A11 * m[2] - A12 * m[5], b2 = -A21 * m[2] - A22 * m[5];
    iM.create(2, 3, M.type());
    const double o[6] = {A11, A12, b1, A21, A22, b2};
    for (int y = 0; y < 2; ++y)
        for (int x = 0; x < 3; ++x) {
            if (M.type() == CV_MAKETYPE(CV_64F, 1)) iM.ptr<double>(y)[x] = o[y * 3 + x];
            else iM.ptr<float>(y)[x] = (float)o[y * 3 + x];
        }
}

// cv::getPerspectiveTransform: the 3x3 CV_64FC1 homography through four point pairs (h22 = 1), solved by Gaussian
// elimination with partial pivoting in double
inline Mat getPerspectiveTransform(const Point2f src[4], const Point2f dst[4]) {
    double a[8][9];
    for (int i = 0; i < 4; ++i) {
        const double x = src[i].x, y = src[i].y, u = dst[i].x, v = dst[i].y;
        const double r0[9] = {x, y, 1, 0, 0, 0, -x * u, -y * u, u};
        const double r1[9] = {0, 0, 0, x, y, 1, -x * v, -y * v, v};
        for (int k = 0; k < 9; ++k) { a[i][k] = r0[k]; a[i + 4][k] = r1[k]; }
    }
    for (int c = 0; c < 8; ++c) {
        int piv = c;
        for (int r = c + 1; r < 8; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) throw std::runtime_error("getPerspectiveTransform: degenerate point set");
        for (int k = 0; k < 9; ++k) std::swap(a[c][k], a[piv][k]);
        for (int r = c + 1; r < 8; ++r) {
            const double f = a[r][c] / a[c][c];
            for (int k = c; k < 9; ++k) a[r][k] -= f * a[c][k];
        }
    }
    double h[9];
    for (int r = 7; r >= 0; --r) {
        double acc = a[r][8];
        for (int k = r + 1; k < 8; ++k) acc -= a[r][k] * h[k];
        h[r] = acc / a[r][r];
    }
    h[8] = 1.0;
    Mat out(3, 3, CV_MAKETYPE(CV_64F, 1));
    for (int i = 0; i < 9; ++i) out.ptr<double>(i / 3)[i % 3] = h[i];
    return out;
}

namespace cuda {

// hook of the facade's attached streams (fk_compat.h: streams attached to a descriptor queue with recorded ticks flush and fence before the
// host waits) -- null unless a stream has been attached
inline void (*&cvgs_stream_sync_hook())(hipStream_t) {
    static void (*hook)(hipStream_t) = nullptr;
    return hook;
}
// ... and the one a stream's destruction calls first: an attachment must not outlive its stream (a later stream may get the same handle)
inline void (*&cvgs_stream_destroy_hook())(hipStream_t) {
    static void (*hook)(hipStream_t) = nullptr;
    return hook;
}

class Stream {
public:
    Stream() : impl_(std::make_shared<Impl>(true)) {}
    static Stream& Null() { static Stream s{nullptr, 0}; return s; }
    void waitForCompletion() const {
        if (auto hook = cvgs_stream_sync_hook()) hook(impl_->s);
        cvgs_hip_check(hipStreamSynchronize(impl_->s), "hipStreamSynchronize");
    }
    hipStream_t raw() const { return impl_->s; }
    static Stream wrap(hipStream_t s) { return Stream(s, 0); }

private:
    struct Impl {
        hipStream_t s = nullptr;
        bool own = false;
        explicit Impl(bool create) : own(create) { if (create) cvgs_hip_check(hipStreamCreate(&s), "hipStreamCreate"); }
        Impl(hipStream_t user) : s(user), own(false) {}
        ~Impl() {
            if (!own || !s) return;
            if (auto hook = cvgs_stream_destroy_hook()) {
                try { hook(s); } catch (...) {} // (recorded calls are submitted, the attachment goes)
            }
            (void)hipStreamDestroy(s);
        }
    };
    Stream(hipStream_t s, int) : impl_(std::make_shared<Impl>(s)) {}
    std::shared_ptr<Impl> impl_;
};

struct StreamAccessor {
    static hipStream_t getStream(const Stream& s) { return s.raw(); }
    static Stream wrapStream(hipStream_t s) { return Stream::wrap(s); }
};

// Device matrix: pitched allocation shared between views (ROI / row), like cv::cuda::GpuMat.
class GpuMat {
public:
    int flags = 0, rows = 0, cols = 0;
    size_t step = 0;
    uchar* data = nullptr;

    GpuMat() = default;
    GpuMat(int r, int c, int type) { create(r, c, type); }
    GpuMat(Size s, int type) { create(s.height, s.width, type); }
    GpuMat(int r, int c, int type, const Scalar& v) { create(r, c, type); setTo(v); }
    GpuMat(Size s, int type, const Scalar& v) { create(s.height, s.width, type); setTo(v); }
    // user-allocated memory (device, or host memory when the descriptor is handed to a CPU checker)
    GpuMat(int r, int c, int type, void* user, size_t step_ = 0)
        : flags(type), rows(r), cols(c), step(step_ ? step_ : c * cvgs_elem_size(type)), data((uchar*)user) {}
    GpuMat(Size s, int type, void* user, size_t step_ = 0) : GpuMat(s.height, s.width, type, user, step_) {}
    explicit GpuMat(const Mat& m) { upload(m); }

    void create(int r, int c, int type) {
        if (data && r == rows && c == cols && type == flags && store_) return;
        flags = type; rows = r; cols = c;
        size_t pitch = 0;
        void* p = nullptr;
        // pitched like cudaMallocPitch: rows start on 512-byte boundaries unless the matrix is one row
        cvgs_hip_check(hipMallocPitch(&p, &pitch, (size_t)c * cvgs_elem_size(type), (size_t)r), "hipMallocPitch");
        store_ = std::shared_ptr<void>(p, [](void* q) { (void)hipFree(q); });
        data = (uchar*)p;
        step = r == 1 ? (size_t)c * cvgs_elem_size(type) : pitch;
    }
    void create(Size s, int type) { create(s.height, s.width, type); }
    void release() { store_.reset(); data = nullptr; rows = cols = 0; step = 0; }

    int type() const { return flags; }
    int depth() const { return CV_MAT_DEPTH(flags); }
    int channels() const { return CV_MAT_CN(flags); }
    size_t elemSize() const { return cvgs_elem_size(flags); }
    bool empty() const { return !data; }
    Size size() const { return Size(cols, rows); }

    GpuMat operator()(const Rect& r) const {
        if (r.x < 0 || r.y < 0 || r.width < 0 || r.height < 0 || r.x + r.width > cols || r.y + r.height > rows)
            throw std::runtime_error("GpuMat ROI outside the matrix");
        GpuMat m = *this;
        m.data = data + (size_t)r.y * step + (size_t)r.x * elemSize();
        m.cols = r.width;
        m.rows = r.height;
        return m;
    }
    GpuMat operator()(const Rect2d& r) const { return (*this)(Rect(r)); }
    GpuMat row(int y) const { return (*this)(Rect(0, y, cols, 1)); }
    GpuMat reshape(int cn, int new_rows) const {
        GpuMat m = *this;
        const size_t total = (size_t)rows * cols * channels();
        m.flags = CV_MAKETYPE(depth(), cn);
        m.rows = new_rows;
        m.cols = (int)(total / ((size_t)new_rows * cn));
        m.step = (size_t)m.cols * m.elemSize();
        return m;
    }

    void upload(const Mat& m) {
        create(m.rows, m.cols, m.type());
        cvgs_hip_check(hipMemcpy2D(data, step, m.data, m.step, (size_t)cols * elemSize(), rows, hipMemcpyHostToDevice),
                       "hipMemcpy2D(upload)");
    }
    void upload(const Mat& m, const Stream& s) {
        create(m.rows, m.cols, m.type());
        cvgs_hip_check(hipMemcpy2DAsync(data, step, m.data, m.step, (size_t)cols * elemSize(), rows, hipMemcpyHostToDevice,
                                        s.raw()), "hipMemcpy2DAsync(upload)");
    }
    void download(Mat& m) const {
        if (m.rows != rows || m.cols != cols || m.type() != type() || m.empty()) m.create(rows, cols, type());
        cvgs_hip_check(hipMemcpy2D(m.data, m.step, data, step, (size_t)cols * elemSize(), rows, hipMemcpyDeviceToHost),
                       "hipMemcpy2D(download)");
    }
    void download(Mat& m, const Stream& s) const {
        if (m.rows != rows || m.cols != cols || m.type() != type() || m.empty()) m.create(rows, cols, type());
        cvgs_hip_check(hipMemcpy2DAsync(m.data, m.step, data, step, (size_t)cols * elemSize(), rows, hipMemcpyDeviceToHost,
                                        s.raw()), "hipMemcpy2DAsync(download)");
    }
    GpuMat& setTo(const Scalar& v) {
        Mat h(rows, cols, type(), v);
        cvgs_hip_check(hipMemcpy2D(data, step, h.data, h.step, (size_t)cols * elemSize(), rows, hipMemcpyHostToDevice),
                       "hipMemcpy2D(setTo)");
        return *this;
    }
    GpuMat& setTo(const Scalar& v, const Stream& s) {
        s.waitForCompletion();
        return setTo(v);
    }

private:
    std::shared_ptr<void> store_;
};

} // namespace cuda
} // namespace cv

#endif // CVGS_USE_OPENCV
