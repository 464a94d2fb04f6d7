// fk_compat.h -- the small slice of the fk:: (FusedKernelLibrary) vocabulary that the reference's hot-path
// call sites spell directly: RawPtr / Ptr2D / Tensor / TensorT, the Read / Unary / Binary /
// Write instantiable-operation wrappers, the operation tags of the hot chains, fk::executeOperations and
// fk::CircularTensor.  Nothing here computes pixels: every IOp only knows how to append itself to ONE
// cvgs_chain_desc (include/cvgs_hip.h); executeOperations() type-checks the list at compile time (the output
// type of IOp k must be the input type of IOp k+1, as in the reference) and hands the descriptor to the
// HIP library through the C-ABI -> one kernel launch.
#pragma once

#include <array>
#include <atomic>
#include <mutex>
#include <cstring>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../../include/cvgs_hip_ext.h" // cvgs_hip.h + the descriptor queue behind cvGS::Queue / attachQueue
#include "../cv2cuda_types.h"

// The reference's sequence selectors are spelled with CUDA function qualifiers (tests/batchread/test_circularbatchread_x_write3D.cu:89-93:
// `constexpr static __device__ __forceinline__ uint at(...)`; tests/resize/test_fused_resize.cu:22-26: FK_HOST_DEVICE_FUSE).  This facade is
// host C++: the selector is evaluated on the HOST (fk::executeDivergentBatch), so in a translation unit the HIP compiler's language mode
// has not defined them the qualifiers mean nothing.
#if !defined(__HIP__) && !defined(__CUDACC__)
#ifndef __device__
#define __device__
#endif
#ifndef __host__
#define __host__
#endif
#ifndef __forceinline__
#define __forceinline__ inline
#endif
#endif
#ifndef FK_HOST_DEVICE_FUSE
#define FK_HOST_DEVICE_FUSE static constexpr inline
#define FK_DEVICE_FUSE static constexpr inline
#define FK_HOST_FUSE static inline
#endif

namespace fk {

// ---- small vocabulary types ---------------------------------------------------------------------------
template <typename T> constexpr int cn = cvGS::vector_traits<T>::cn;
template <typename T> using VBase = typename cvGS::vector_traits<T>::base;
template <typename T> struct VectorTraits { using base = VBase<T>; };
template <typename B, int N> using VectorType_t = typename cvGS::detail::vec_of<B, N>::type;

template <typename V> inline V make_set(VBase<V> v) {
    V r{};
    if constexpr (cn<V> == 1) r = v;
    else {
        r.x = v; r.y = v;
        if constexpr (cn<V> >= 3) r.z = v;
        if constexpr (cn<V> >= 4) r.w = v;
    }
    return r;
}
template <typename V, typename... A> inline V make_(A... a) { return V{static_cast<VBase<V>>(a)...}; }

template <typename V> inline void vec_to_doubles(const V& v, double out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0.0;
    if constexpr (cn<V> == 1) out[0] = (double)v;
    else {
        out[0] = (double)v.x; out[1] = (double)v.y;
        if constexpr (cn<V> >= 3) out[2] = (double)v.z;
        if constexpr (cn<V> >= 4) out[3] = (double)v.w;
    }
}

template <typename V> inline void vec_to_floats(const V& v, float out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0.f;
    if constexpr (cn<V> == 1) out[0] = (float)v;
    else {
        out[0] = (float)v.x; out[1] = (float)v.y;
        if constexpr (cn<V> >= 3) out[2] = (float)v.z;
        if constexpr (cn<V> >= 4) out[3] = (float)v.w;
    }
}

struct Size {
    int width = 0, height = 0;
    constexpr Size() = default;
    constexpr Size(int w, int h) : width(w), height(h) {}
    constexpr bool operator==(const Size& o) const { return width == o.width && height == o.height; }
};
struct Rect {
    uint x = 0, y = 0;
    int width = 0, height = 0;
    constexpr Rect() = default;
    constexpr Rect(uint x_, uint y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};

enum ND { _2D = 2, _3D = 3, T3D = 4 };
enum InterpolationType { INTER_LINEAR = 1 };
enum AspectRatio { PRESERVE_AR = 0, IGNORE_AR = 1, PRESERVE_AR_RN_EVEN = 2, PRESERVE_AR_LEFT = 3 };
enum class CircularTensorOrder { NewestFirst = 0, OldestFirst = 1 };
enum class ColorPlanes { Standard = 0, Transposed = 1 };
// fk::PixelFormat: the reference's tests instantiate NV12 only; NV21 / I420 / YV12 / P010 (10-bit codes in 16-bit samples, the
// result on the 0..1023 scale) are this engine's further 4:2:0 readers (numeric values = cvgs_yuv_layout)
enum PixelFormat { NV12 = 0, NV21 = 1, I420 = 2, YV12 = 3, P010 = 4 };
enum ColorRange { Full = 0, Limited = 1 };
enum ColorPrimitives { bt601 = 0, bt709 = 1, bt2020 = 2 };
template <PixelFormat PF> using YuvSample = std::conditional_t<PF == P010, unsigned short, unsigned char>;
enum ColorConversionCodes {
    COLOR_BGR2BGRA = 0, COLOR_RGB2RGBA = 0, COLOR_BGRA2BGR = 1, COLOR_RGBA2RGB = 1, COLOR_BGR2RGBA = 2,
    COLOR_RGB2BGRA = 2, COLOR_RGBA2BGR = 3, COLOR_BGRA2RGB = 3, COLOR_BGR2RGB = 4, COLOR_RGB2BGR = 4,
    COLOR_BGRA2RGBA = 5, COLOR_RGBA2BGRA = 5, COLOR_BGR2GRAY = 6, COLOR_RGB2GRAY = 7, COLOR_BGRA2GRAY = 10,
    COLOR_RGBA2GRAY = 11
};

struct Dims2D { uint width = 0, height = 0, pitch = 0; };
struct Dims3D { uint width = 0, height = 0, planes = 0, color_planes = 1, pitch = 0, plane_pitch = 0; };

template <ND D, typename T> struct RawPtr;
template <ND D, typename T> class TensorBase;
template <typename T> struct RawPtr<_2D, T> { T* data = nullptr; Dims2D dims; using type = T; };
// (3D: also constructible from the owning tensor, so that the reference's `Write<PerThreadWrite<_3D, T>> w{ {tensor} }` compiles:
//  tests/batchread/test_circularbatchread_x_write3D.cu:64)
template <typename T> struct RawPtr<_3D, T> {
    T* data = nullptr; Dims3D dims; using type = T;
    RawPtr() = default;
    RawPtr(T* d, const Dims3D& dm) : data(d), dims(dm) {}
    RawPtr(const TensorBase<_3D, T>& t);
};
template <typename T> struct RawPtr<T3D, T> {
    T* data = nullptr; Dims3D dims; using type = T;
    RawPtr() = default;
    RawPtr(T* d, const Dims3D& dm) : data(d), dims(dm) {}
    RawPtr(const TensorBase<T3D, T>& t);
};

inline void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// Pitched device image, refcounted (fk::Ptr2D).
template <typename T> class Ptr2D {
public:
    Ptr2D() = default;
    Ptr2D(uint w, uint h) {
        void* p = nullptr; size_t pitch = 0;
        hip_check(hipMallocPitch(&p, &pitch, (size_t)w * sizeof(T), h), "hipMallocPitch");
        store_ = std::shared_ptr<void>(p, [](void* q) { (void)hipFree(q); });
        raw_.data = (T*)p; raw_.dims = {w, h, (uint)pitch};
    }
    Ptr2D(T* data, uint w, uint h, uint pitch) { raw_.data = data; raw_.dims = {w, h, pitch}; }
    RawPtr<_2D, T> ptr() const { return raw_; }
    Dims2D dims() const { return raw_.dims; }
    operator RawPtr<_2D, T>() const { return raw_; }
private:
    RawPtr<_2D, T> raw_;
    std::shared_ptr<void> store_;
};

// Dense device tensor [planes][color_planes][H][W] (fk::Tensor) or [color_planes][planes][H][W] (fk::TensorT).
template <ND D, typename T> class TensorBase {
public:
    TensorBase() = default;
    TensorBase(T* data, uint w, uint h, uint planes, uint color_planes = 1) { set(data, w, h, planes, color_planes); }
    TensorBase(uint w, uint h, uint planes, uint color_planes = 1) { allocTensor(w, h, planes, color_planes); }
    void allocTensor(uint w, uint h, uint planes, uint color_planes = 1) {
        void* p = nullptr;
        hip_check(hipMalloc(&p, (size_t)w * h * planes * color_planes * sizeof(T)), "hipMalloc(tensor)");
        store_ = std::shared_ptr<void>(p, [](void* q) { (void)hipFree(q); });
        set((T*)p, w, h, planes, color_planes);
    }
    RawPtr<D, T> ptr() const { return raw_; }
    Dims3D dims() const { return raw_.dims; }
    size_t sizeInBytes() const { return (size_t)raw_.dims.plane_pitch * raw_.dims.planes * raw_.dims.color_planes; }
    operator RawPtr<D, T>() const { return raw_; }
protected:
    void set(T* data, uint w, uint h, uint planes, uint cp) {
        raw_.data = data;
        raw_.dims = {w, h, planes, cp, (uint)(w * sizeof(T)), (uint)(w * sizeof(T) * h)};
    }
    RawPtr<D, T> raw_;
    std::shared_ptr<void> store_;
};
template <typename T> using Tensor = TensorBase<_3D, T>;
template <typename T> using TensorT = TensorBase<T3D, T>;
template <typename T> inline RawPtr<_3D, T>::RawPtr(const TensorBase<_3D, T>& t) : data(t.ptr().data), dims(t.ptr().dims) {}
template <typename T> inline RawPtr<T3D, T>::RawPtr(const TensorBase<T3D, T>& t) : data(t.ptr().data), dims(t.ptr().dims) {}

// ---- small-buffer array for the builders --------------------------------------------------------------------------
// The chain builders hold plane tables / matrices / op lists of at most a few dozen entries; a std::vector there costs
// one malloc + free per member per executeOperations call on the host path that feeds a ~2 us kernel.  SmallVec keeps up
// to N trivially-copyable elements inline (no heap at all for the reference's batches, <= 64 planes) and only spills to
// the heap beyond that.  It offers the subset of std::vector the IOps use.
namespace detail {
template <typename T, size_t N> class SmallVec {
    static_assert(std::is_trivially_copyable_v<T>, "SmallVec holds plain descriptors only");
public:
    SmallVec() = default;
    SmallVec(const SmallVec& o) { assign(o.begin(), o.end()); }
    SmallVec(SmallVec&& o) noexcept { take(o); }
    SmallVec& operator=(const SmallVec& o) { if (this != &o) assign(o.begin(), o.end()); return *this; }
    SmallVec& operator=(SmallVec&& o) noexcept { if (this != &o) { release(); take(o); } return *this; }
    ~SmallVec() { release(); }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T* data() { return p_; }
    const T* data() const { return p_; }
    T* begin() { return p_; }
    T* end() { return p_ + n_; }
    const T* begin() const { return p_; }
    const T* end() const { return p_ + n_; }
    T& operator[](size_t i) { return p_[i]; }
    const T& operator[](size_t i) const { return p_[i]; }
    T& back() { return p_[n_ - 1]; }
    void clear() { n_ = 0; }
    void reserve(size_t c) {
        if (c <= cap_) return;
        size_t nc = cap_ * 2 > c ? cap_ * 2 : c;
        T* q = static_cast<T*>(::operator new(nc * sizeof(T)));
        if (n_) std::memcpy(static_cast<void*>(q), p_, n_ * sizeof(T));
        if (p_ != inline_ptr()) ::operator delete(p_);
        p_ = q; cap_ = nc;
    }
    void push_back(const T& v) { T tmp = v; reserve(n_ + 1); p_[n_++] = tmp; } // v may alias an element
    void resize(size_t n, const T& v = T{}) { T tmp = v; reserve(n); for (size_t i = n_; i < n; ++i) p_[i] = tmp; n_ = n; }
    void assign(size_t n, const T& v) { T tmp = v; n_ = 0; reserve(n); for (size_t i = 0; i < n; ++i) p_[i] = tmp; n_ = n; }
    template <typename It, typename = std::enable_if_t<!std::is_integral_v<It>>> void assign(It first, It last) {
        n_ = 0;
        for (; first != last; ++first) push_back(*first);
    }
private:
    T* inline_ptr() { return reinterpret_cast<T*>(inline_); }
    void release() { if (p_ != inline_ptr()) ::operator delete(p_); p_ = inline_ptr(); n_ = 0; cap_ = N; }
    void take(SmallVec& o) {
        if (o.p_ != o.inline_ptr()) { p_ = o.p_; n_ = o.n_; cap_ = o.cap_; o.p_ = o.inline_ptr(); o.n_ = 0; o.cap_ = N; }
        else { p_ = inline_ptr(); cap_ = N; n_ = o.n_; if (n_) std::memcpy(static_cast<void*>(p_), o.p_, n_ * sizeof(T)); o.n_ = 0; }
    }
    alignas(T) unsigned char inline_[N * sizeof(T)];
    T* p_ = inline_ptr();
    size_t n_ = 0, cap_ = N;
};
constexpr size_t kInlinePlanes = 64; // the reference's largest batch in its tests and benchmarks is 50..60 crops
} // namespace detail

// ---- chain builder ----------------------------------------------------------------------------------------
struct ChainBuilder {
    cvgs_chain_desc d;
    detail::SmallVec<cvgs_image2d, detail::kInlinePlanes> src;
    detail::SmallVec<cvgs_image2d, 3 * detail::kInlinePlanes> dst; // SplitWrite<_2D>: up to 4 planes per batch element
    detail::SmallVec<float, 9 * detail::kInlinePlanes> warp; // WARP reads: batch x 9 floats
    detail::SmallVec<int32_t, 2 * detail::kInlinePlanes> warp_sizes; // WARP reads with per-plane destination sizes: batch x 2
    ChainBuilder() { std::memset(&d, 0, sizeof(d)); d.struct_size = sizeof(d); }
    // finish() points d.read.src / d.write.planes2d / the warp tables at this object's INLINE buffers: a copy or a move would
    // leave them pointing into the old object.  Builders are used in place (stack) or behind unique_ptr (ChainBatch).
    ChainBuilder(const ChainBuilder&) = delete;
    ChainBuilder& operator=(const ChainBuilder&) = delete;
    ChainBuilder(ChainBuilder&&) = delete;
    ChainBuilder& operator=(ChainBuilder&&) = delete;
    void op(int opcode, int aux, const float* operand = nullptr, const double* operand_d = nullptr) {
        if (d.n_ops >= CVGS_MAX_OPS) throw std::runtime_error("cvGS: too many pointwise operations in one chain");
        cvgs_op& o = d.ops[d.n_ops++];
        o.opcode = opcode; o.aux = aux;
        for (int i = 0; i < 4; ++i) {
            o.operand[i] = operand ? operand[i] : 0.f;
            o.operand_d[i] = operand_d ? operand_d[i] : (operand ? (double)operand[i] : 0.0);
        }
    }
    void finish() {
        if (!src.empty() && !(d.read.flags & CVGS_READ_FLAG_TABLE_ON_DEVICE)) d.read.src = src.data();
        if (!dst.empty()) d.write.planes2d = dst.data();
        if (!warp.empty()) d.read.warp_matrices = warp.data();
        if (!warp_sizes.empty()) d.read.warp_dst_sizes = warp_sizes.data();
    }
};

template <typename T> inline cvgs_image2d image2d(const RawPtr<_2D, T>& p) {
    return cvgs_image2d{p.data, (int32_t)p.dims.width, (int32_t)p.dims.height, (int32_t)p.dims.pitch, 0};
}

enum class Stage { Read, Pointwise, Write };

// ---- read-back specifications completed by ReadIOp::then(...) (reference include/cvGPUSpeedup.cuh:204-207,247-265,
// 444-447: cvGS::resize<INTER>(Size), cvGS::crop(Rect2d | array<Rect2d,N>)) ----------------------------------------
struct CropSpec { Rect rect; };
template <size_t N> struct BatchCropSpec { std::array<Rect, N> rects; };
template <InterpolationType IT> struct IncompleteResize { Size dsize; };
template <typename T> struct ResizeRead;
template <typename T> struct BatchResizeRead;
template <typename T> struct BatchPixelRead;
template <ND D, typename T> struct PerThreadRead;

template <typename T> inline RawPtr<_2D, T> crop_view(const RawPtr<_2D, T>& p, const Rect& r) {
    if (r.width < 0 || r.height < 0 || r.x + (uint)r.width > p.dims.width || r.y + (uint)r.height > p.dims.height)
        throw std::runtime_error("cvGS::crop: rectangle outside the image");
    RawPtr<_2D, T> v = p;
    v.data = (T*)((unsigned char*)p.data + (size_t)r.y * p.dims.pitch) + r.x; // fk::Crop: back.exec(thread + rect.xy)
    v.dims.width = (uint)r.width;
    v.dims.height = (uint)r.height;
    return v;
}

// ---- instantiable-operation wrappers ---------------------------------------------------------------------------
template <typename Op> struct Read {
    typename Op::ParamsType params;
    using Operation = Op;
    using OutputType = typename Op::OutputType;
    static constexpr Stage stage = Stage::Read;
    void lower(ChainBuilder& b) const { Op::lower(params, b); }
    // readIOp.then(crop / crops / resize): a crop is a view of the same memory, a resize turns the read into its back-op
    auto then(const CropSpec& c) const {
        Read r = *this;
        r.params = crop_view(params, c.rect);
        return r;
    }
    template <size_t N> auto then(const BatchCropSpec<N>& c) const;
    template <InterpolationType IT> auto then(const IncompleteResize<IT>& rs) const;
};
template <typename Op> using ReadInstantiableOperation = Read<Op>;

template <typename Op> struct Unary {
    using Operation = Op;
    using InputType = typename Op::InputType;
    using OutputType = typename Op::OutputType;
    static constexpr Stage stage = Stage::Pointwise;
    void lower(ChainBuilder& b) const { Op::lower(b); }
};

template <typename Op> struct Binary {
    typename Op::ParamsType params;
    using Operation = Op;
    using InputType = typename Op::InputType;
    using OutputType = typename Op::OutputType;
    static constexpr Stage stage = Stage::Pointwise;
    void lower(ChainBuilder& b) const { Op::lower(params, b); }
};

template <typename Op> struct Write {
    typename Op::ParamsType params;
    using Operation = Op;
    using InputType = typename Op::InputType;
    static constexpr Stage stage = Stage::Write;
    void lower(ChainBuilder& b) const { Op::lower(params, b); }
};
template <typename Op> using WriteInstantiableOperation = Write<Op>;

// A run of pointwise stages with known end types (what cvGS::convertTo(alpha[,beta]) returns).
template <typename I, typename O> struct PointwiseSeq {
    using InputType = I;
    using OutputType = O;
    static constexpr Stage stage = Stage::Pointwise;
    detail::SmallVec<cvgs_op, CVGS_MAX_OPS> ops;
    void lower(ChainBuilder& b) const { for (const auto& o : ops) b.op(o.opcode, o.aux, o.operand, o.operand_d); }
    template <typename Next> auto then(const Next& n) const {
        static_assert(std::is_same_v<O, typename Next::InputType>, "then(): types do not chain");
        PointwiseSeq<I, typename Next::OutputType> r;
        ChainBuilder tmp;
        lower(tmp); n.lower(tmp);
        r.ops.assign(tmp.d.ops, tmp.d.ops + tmp.d.n_ops);
        return r;
    }
};

// ---- operation tags ------------------------------------------------------------------------------------------------
template <ND D, typename T> struct PerThreadRead;
template <typename T> struct PerThreadRead<_2D, T> {
    using ParamsType = RawPtr<_2D, T>;
    using OutputType = T;
    static void lower(const ParamsType& p, ChainBuilder& b) {
        b.d.read.kind = CVGS_READ_PIXEL;
        b.d.read.src_type = cvGS::cv_type_of<T>;
        b.d.read.batch = 1; b.d.read.used_planes = 1;
        b.src.assign(1, image2d(p));
    }
};

// one pitched image per batch element (PerThreadWrite<_2D,T>::build(std::array<RawPtr<_2D,T>,N>), as the reference's
// batched warp test writes its results, tests/warping/test_warping_opencv.cu:173-176)
template <typename T> struct BatchPixelWrite {
    detail::SmallVec<cvgs_image2d, detail::kInlinePlanes> planes;
    using InputType = T;
    static constexpr Stage stage = Stage::Write;
    void lower(ChainBuilder& b) const {
        cvgs_write_desc& w = b.d.write;
        w.kind = CVGS_WRITE_PIXEL_2D_BATCH; w.dst_type = cvGS::cv_type_of<T>;
        b.dst.assign(planes.begin(), planes.end());
        if (!planes.empty()) { w.width = planes[0].width; w.height = planes[0].height; }
        w.planes = (int)planes.size();
    }
};

template <ND D, typename T> struct PerThreadWrite;
template <typename T> struct PerThreadWrite<_2D, T> {
    using ParamsType = RawPtr<_2D, T>;
    using InputType = T;
    static void lower(const ParamsType& p, ChainBuilder& b) {
        cvgs_write_desc& w = b.d.write;
        w.kind = CVGS_WRITE_PIXEL_2D; w.dst_type = cvGS::cv_type_of<T>; w.data = p.data;
        w.width = p.dims.width; w.height = p.dims.height; w.step = p.dims.pitch; w.planes = 1;
    }
    template <size_t N> static BatchPixelWrite<T> build(const std::array<RawPtr<_2D, T>, N>& out) {
        BatchPixelWrite<T> w;
        for (const auto& o : out) w.planes.push_back(image2d(o));
        return w;
    }
};
template <typename T> struct PerThreadWrite<_3D, T> {
    using ParamsType = RawPtr<_3D, T>;
    using InputType = T;
    static void lower(const ParamsType& p, ChainBuilder& b) {
        cvgs_write_desc& w = b.d.write;
        w.kind = CVGS_WRITE_PIXEL_3D; w.dst_type = cvGS::cv_type_of<T>; w.data = p.data;
        w.width = p.dims.width; w.height = p.dims.height; w.planes = p.dims.planes;
    }
};
template <typename T> using TensorWrite = PerThreadWrite<_3D, T>;

template <typename T> struct TensorSplit {
    using ParamsType = RawPtr<_3D, VBase<T>>;
    using InputType = T;
    static void lower(const ParamsType& p, ChainBuilder& b) {
        cvgs_write_desc& w = b.d.write;
        w.kind = CVGS_WRITE_TENSOR_SPLIT; w.dst_type = cvGS::cv_type_of<T>; w.data = p.data;
        w.width = p.dims.width; w.height = p.dims.height; w.planes = p.dims.planes;
    }
};
template <typename T> struct TensorTSplit {
    using ParamsType = RawPtr<T3D, VBase<T>>;
    using InputType = T;
    static void lower(const ParamsType& p, ChainBuilder& b) {
        cvgs_write_desc& w = b.d.write;
        w.kind = CVGS_WRITE_TENSOR_T_SPLIT; w.dst_type = cvGS::cv_type_of<T>; w.data = p.data;
        w.width = p.dims.width; w.height = p.dims.height; w.planes = p.dims.planes;
    }
};

// SplitWrite<_2D,T>: C pitched planes per batch element
template <ND D, typename T> struct SplitWrite {
    struct ParamsType { detail::SmallVec<RawPtr<_2D, VBase<T>>, 3 * detail::kInlinePlanes> planes; int batch = 1; };
    using InputType = T;
    static void lower(const ParamsType& p, ChainBuilder& b) {
        cvgs_write_desc& w = b.d.write;
        w.kind = CVGS_WRITE_SPLIT_2D; w.dst_type = cvGS::cv_type_of<T>;
        b.dst.clear();
        for (const auto& pl : p.planes) b.dst.push_back(image2d(pl));
        if (!p.planes.empty()) { w.width = p.planes[0].dims.width; w.height = p.planes[0].dims.height; }
        w.planes = p.batch;
    }
    static Write<SplitWrite> build(const std::vector<Ptr2D<VBase<T>>>& out) {
        static_assert(cn<T> >= 2, "Split operations can only be used with types of 2, 3 or 4 channels.");
        Write<SplitWrite> w;
        for (const auto& o : out) w.params.planes.push_back(o.ptr());
        return w;
    }
    template <size_t N> static Write<SplitWrite> build(const std::array<std::vector<Ptr2D<VBase<T>>>, N>& out) {
        static_assert(cn<T> >= 2, "Split operations can only be used with types of 2, 3 or 4 channels.");
        Write<SplitWrite> w;
        w.params.batch = (int)N;
        for (const auto& img : out) for (const auto& o : img) w.params.planes.push_back(o.ptr());
        return w;
    }
};

template <typename I, typename O> struct SaturateCast {
    using InputType = I;
    using OutputType = O;
    static_assert(cn<I> == cn<O>, "SaturateCast cannot change the number of channels");
    static void lower(ChainBuilder& b) { b.op(CVGS_OP_CAST, cvGS::base_depth<VBase<O>>::value); }
};

// fk::Cast<I,O>: static_cast per channel (float -> integer truncates), as the reference's warp tests spell it
// (tests/warping/test_warping_opencv.cu:63)
template <typename I, typename O> struct Cast {
    using InputType = I;
    using OutputType = O;
    static_assert(cn<I> == cn<O>, "Cast cannot change the number of channels");
    static void lower(ChainBuilder& b) { b.op(CVGS_OP_CAST_TRUNC, cvGS::base_depth<VBase<O>>::value); }
    static Unary<Cast> build() { return {}; }
};

// Integer pixel types T: the scalar arrives as T (cvScalar2CUDAV<I> truncated it), the stage runs in integer arithmetic and
// saturates back to T (include/cvgs_hip.h CVGS_OP_MUL...; FKL's own definition is not in the reference tree).
#define CVGS_FK_BINARY(NAME, OPC)                                                           \
    template <typename T> struct NAME {                                                     \
        using InputType = T; using OutputType = T; using ParamsType = T;                    \
        static void lower(const T& v, ChainBuilder& b) { float f[4]; double d[4]; vec_to_floats(v, f); vec_to_doubles(v, d); b.op(OPC, 0, f, d); } \
    };
CVGS_FK_BINARY(Mul, CVGS_OP_MUL)
CVGS_FK_BINARY(Add, CVGS_OP_ADD)
CVGS_FK_BINARY(Sub, CVGS_OP_SUB)
CVGS_FK_BINARY(Div, CVGS_OP_DIV)
#undef CVGS_FK_BINARY

template <typename T, int... IDX> struct VectorReorder {
    using InputType = T; using OutputType = T;
    static_assert(sizeof...(IDX) == cn<T>, "VectorReorder needs one index per channel");
    static void lower(ChainBuilder& b) {
        int aux = 0, k = 0;
        ((aux |= (IDX & 3) << (2 * k++)), ...);
        b.op(CVGS_OP_REORDER, aux);
    }
};

namespace detail {
constexpr int kSwap3 = 2 | (1 << 2) | (0 << 4), kId3 = 0 | (1 << 2) | (2 << 4);
template <typename B> constexpr float alpha_max() {
    if constexpr (std::is_same_v<B, uchar>) return 255.f;
    else if constexpr (std::is_same_v<B, ushort>) return 65535.f;
    else return 1.f;
}
} // namespace detail

template <ColorConversionCodes CODE, typename I, typename O = I> struct ColorConversion {
    using InputType = I; using OutputType = O;
    static void lower(ChainBuilder& b) {
        constexpr int c = (int)CODE;
        const float a[4] = {detail::alpha_max<VBase<I>>(), 0, 0, 0};
        if constexpr (c == 0 || c == 2) {
            static_assert(cn<I> == 3 && cn<O> == 4, "colour code needs 3 -> 4 channels");
            b.op(CVGS_OP_ADD_ALPHA, c == 0 ? detail::kId3 : detail::kSwap3, a);
        } else if constexpr (c == 1 || c == 3) {
            static_assert(cn<I> == 4 && cn<O> == 3, "colour code needs 4 -> 3 channels");
            b.op(CVGS_OP_DROP_ALPHA, c == 1 ? detail::kId3 : detail::kSwap3);
        } else if constexpr (c == 4) {
            static_assert(cn<I> == 3 && cn<O> == 3, "colour code needs 3 -> 3 channels");
            b.op(CVGS_OP_REORDER, detail::kSwap3);
        } else if constexpr (c == 5) {
            static_assert(cn<I> == 4 && cn<O> == 4, "colour code needs 4 -> 4 channels");
            b.op(CVGS_OP_REORDER, detail::kSwap3 | (3 << 6));
        } else {
            static_assert(c == 6 || c == 7 || c == 10 || c == 11, "Color conversion type not supported yet.");
            static_assert(cn<O> == 1 && cn<I> == ((c == 6 || c == 7) ? 3 : 4), "gray conversion channel counts");
            b.op(CVGS_OP_GRAY, (c == 6 || c == 10) ? detail::kSwap3 : detail::kId3);
        }
    }
};

// ---- NV12 read-back ------------------------------------------------------------------------------------------------
template <PixelFormat PF> struct ReadYUV {
    using ParamsType = RawPtr<_2D, YuvSample<PF>>; // luma view; the chroma plane(s) follow it (height/2 rows)
    using OutputType = VectorType_t<YuvSample<PF>, 3>; // (Y, U, V) -- only meaningful fused with ConvertYUVToRGB
    static void lower(const ParamsType&, ChainBuilder&) {
        throw std::runtime_error("ReadYUV must be fused with ConvertYUVToRGB (fk::fuse) on this engine");
    }
};
template <PixelFormat PF, ColorRange CR, ColorPrimitives CP, bool ALPHA, typename O = VectorType_t<YuvSample<PF>, ALPHA ? 4 : 3>>
struct ConvertYUVToRGB {
    using InputType = VectorType_t<YuvSample<PF>, 3>;
    using OutputType = O;
};

// fuse(Read<ReadYUV>, Unary<ConvertYUVToRGB>) -> one read IOp producing RGB(A)
template <PixelFormat PF, ColorRange CR, ColorPrimitives CP, bool ALPHA, typename O, bool SWAP_RB = false> struct YuvRead {
    RawPtr<_2D, YuvSample<PF>> params;
    // optional: N crops of the surface (even x, y, w, h), each a view with its own luma -> chroma offset; the read is
    // then a batch of N planes in ONE launch (engine extension: crops straight from a decoder surface)
    detail::SmallVec<cvgs_image2d, detail::kInlinePlanes> crops;
    using OutputType = O;
    static constexpr Stage stage = Stage::Read;
    static constexpr bool float_out = std::is_same_v<VBase<O>, float>;
    static constexpr bool swap_rb = SWAP_RB; // deliver B,G,R[,A] instead of R,G,B[,A]
    static void lower_swap(ChainBuilder& b) {
        if constexpr (SWAP_RB) b.op(CVGS_OP_REORDER, ALPHA ? (detail::kSwap3 | (3 << 6)) : detail::kSwap3);
    }
    void lower_read(ChainBuilder& b, int kind) const {
        cvgs_read_desc& r = b.d.read;
        r.kind = kind; r.src_type = PF == P010 ? CV_16UC1 : CV_8UC1;
        r.yuv_range = (int)CR; r.yuv_primaries = (int)CP; r.yuv_alpha = ALPHA ? 1 : 0;
        r.yuv_layout = (int)PF;
        if (crops.empty()) b.src.assign(1, image2d(params));
        else b.src = crops;
        r.batch = r.used_planes = (int)b.src.size();
    }
    void lower(ChainBuilder& b) const {
        lower_read(b, CVGS_READ_NV12);
        lower_swap(b);
        if constexpr (!float_out) b.op(CVGS_OP_CAST, cvGS::base_depth<VBase<O>>::value);
    }
};
template <PixelFormat PF, ColorRange CR, ColorPrimitives CP, bool ALPHA, typename O>
inline auto fuse(const Read<ReadYUV<PF>>& r, const Unary<ConvertYUVToRGB<PF, CR, CP, ALPHA, O>>&) {
    return YuvRead<PF, CR, CP, ALPHA, O>{r.params};
}

// ---- Resize ------------------------------------------------------------------------------------------------
template <typename T> struct ResizeRead {          // single image, pixel source
    RawPtr<_2D, T> src;
    Size dsize;
    using OutputType = VectorType_t<float, cn<T>>;
    static constexpr Stage stage = Stage::Read;
    void lower(ChainBuilder& b) const {
        cvgs_read_desc& r = b.d.read;
        r.kind = CVGS_READ_RESIZE_LINEAR; r.src_type = cvGS::cv_type_of<T>; r.batch = 1; r.used_planes = 1;
        r.dst_width = dsize.width; r.dst_height = dsize.height; r.aspect_ratio = CVGS_IGNORE_AR;
        b.src.assign(1, image2d(src));
    }
};
template <typename Yuv> struct ResizeYuvRead {     // single image, NV12 read-back source
    Yuv back;
    Size dsize;
    int ar = CVGS_IGNORE_AR;            // cvGS::AspectRatio: PRESERVE_AR* letterboxes the surface into dsize
    float bg[4] = {0.f, 0.f, 0.f, 0.f}; // the padding value, in the read's output channel order
    using OutputType = VectorType_t<float, cn<typename Yuv::OutputType>>;
    static constexpr Stage stage = Stage::Read;
    static_assert(Yuv::float_out, "Resize over an NV12 read-back interpolates in float: use ConvertYUVToRGB<..., floatN>");
    void lower(ChainBuilder& b) const {
        back.lower_read(b, CVGS_READ_NV12_RESIZE_LINEAR);
        b.d.read.dst_width = dsize.width; b.d.read.dst_height = dsize.height; b.d.read.aspect_ratio = ar;
        // the background enters BEFORE the R <-> B swap the BGR codes append (it runs through the program like a pixel)
        constexpr bool kSwap = Yuv::swap_rb;
        b.d.read.background[0] = kSwap ? bg[2] : bg[0];
        b.d.read.background[1] = bg[1];
        b.d.read.background[2] = kSwap ? bg[0] : bg[2];
        b.d.read.background[3] = bg[3];
        Yuv::lower_swap(b); // a channel permutation commutes with the per-channel interpolation
    }
};

// N pitched sources -> resize -> default value for unused planes (BatchRead<N, CONDITIONAL_WITH_DEFAULT>)
template <typename T> struct BatchResizeRead {
    detail::SmallVec<cvgs_image2d, detail::kInlinePlanes> planes;
    int used = 0;
    Size dsize;
    int ar = CVGS_IGNORE_AR;
    float background[4] = {0, 0, 0, 0};
    using OutputType = VectorType_t<float, cn<T>>;
    static constexpr Stage stage = Stage::Read;
    void lower(ChainBuilder& b) const {
        cvgs_read_desc& r = b.d.read;
        r.kind = CVGS_READ_RESIZE_LINEAR; r.src_type = cvGS::cv_type_of<T>;
        r.batch = (int)planes.size(); r.used_planes = used;
        r.dst_width = dsize.width; r.dst_height = dsize.height; r.aspect_ratio = ar;
        for (int i = 0; i < 4; ++i) r.background[i] = background[i];
        b.src = planes;
    }
};

// N pitched sources read per pixel (the batch executeOperations overloads)
template <typename T> struct BatchPixelRead {
    detail::SmallVec<cvgs_image2d, detail::kInlinePlanes> planes;
    int used = 0;
    float background[4] = {0, 0, 0, 0};
    using OutputType = T;
    static constexpr Stage stage = Stage::Read;
    // crops.then(resize<INTER>(size)): every plane is stretched to dsize (IGNORE_AR), the K1 read
    template <InterpolationType IT> BatchResizeRead<T> then(const IncompleteResize<IT>& rs) const {
        static_assert(IT == INTER_LINEAR, "Interpolation type not supported yet.");
        BatchResizeRead<T> rd;
        rd.planes = planes;
        rd.used = used;
        rd.dsize = rs.dsize;
        for (int i = 0; i < 4; ++i) rd.background[i] = background[i];
        return rd;
    }
    void lower(ChainBuilder& b) const {
        cvgs_read_desc& r = b.d.read;
        r.kind = CVGS_READ_PIXEL; r.src_type = cvGS::cv_type_of<T>;
        r.batch = (int)planes.size(); r.used_planes = used;
        for (int i = 0; i < 4; ++i) r.background[i] = background[i];
        b.src = planes;
    }
};

// ---- fk::CircularBatchRead (reference tests/batchread/test_circularbatchread_x_write3D.cu:59-66, 72-83) -----------------------------
//     fk::Read<fk::CircularBatchRead<fk::Ascendent, fk::PerThreadRead<fk::_2D, uchar3>, BATCH>> r;
//     r.params.first = FIRST;  r.params.opData[i].params = input[i];
// Plane z of the read is input plane (z + first) mod BATCH (Ascendent; the reference's check: out[z] == in[z + FIRST wrapped]) or
// (first - z) mod BATCH (Descendent, [FKL-mem]: the direction CircularTensor's NewestFirst order reads its ring in).  The rotation
// happens where the descriptor is built: the kernel reads an ordinary batch of planes.
enum CircularDirection { Ascendent = 0, Descendent = 1 };
template <CircularDirection DIR, typename ReadOp, int BATCH> struct CircularBatchRead;
template <CircularDirection DIR, typename T, int BATCH> struct CircularBatchRead<DIR, PerThreadRead<_2D, T>, BATCH> {
    static_assert(BATCH >= 1, "CircularBatchRead needs at least one plane");
    struct OpData { RawPtr<_2D, T> params; };
    struct ParamsType {
        uint first = 0;
        OpData opData[BATCH];
    };
    using OutputType = T;
    static void lower(const ParamsType& p, ChainBuilder& b) {
        cvgs_read_desc& r = b.d.read;
        r.kind = CVGS_READ_PIXEL; r.src_type = cvGS::cv_type_of<T>;
        r.batch = BATCH; r.used_planes = BATCH;
        b.src.clear();
        const uint first = p.first % (uint)BATCH;
        for (uint z = 0; z < (uint)BATCH; ++z) {
            const uint from = DIR == Ascendent ? (z + first) % (uint)BATCH : (first + (uint)BATCH - z) % (uint)BATCH;
            b.src.push_back(image2d(p.opData[from].params));
        }
    }
};

// ---- Warping (cvGS::warp) ------------------------------------------------------------------------------------
enum class WarpType { Affine = 0, Perspective = 1 };
// the INVERSE (destination -> source) transform narrowed to float, plus the target size (reference
// include/cvGPUSpeedup.cuh:269-284); affine matrices leave the last row at (0, 0, 1)
template <WarpType WT> struct WarpingParameters {
    float transformMatrix[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    Size dstSize;
};
// N pitched sources -> warp -> default value for unused planes
template <WarpType WT, typename T> struct WarpRead {
    detail::SmallVec<cvgs_image2d, detail::kInlinePlanes> planes;
    detail::SmallVec<WarpingParameters<WT>, detail::kInlinePlanes> params;
    int used = 0;
    float background[4] = {0, 0, 0, 0};
    using OutputType = VectorType_t<float, cn<T>>;
    static constexpr Stage stage = Stage::Read;
    void lower(ChainBuilder& b) const {
        cvgs_read_desc& r = b.d.read;
        r.kind = WT == WarpType::Affine ? CVGS_READ_WARP_AFFINE : CVGS_READ_WARP_PERSPECTIVE;
        r.src_type = cvGS::cv_type_of<T>;
        r.batch = (int)planes.size(); r.used_planes = used;
        if (params.empty()) throw std::runtime_error("cvGS::warp: no transform given");
        r.dst_width = params[0].dstSize.width; r.dst_height = params[0].dstSize.height;
        for (int i = 0; i < 4; ++i) r.background[i] = background[i];
        b.src = planes;
        b.warp.clear();
        b.warp_sizes.clear();
        bool differ = false;
        for (const auto& p : params) {
            differ = differ || !(p.dstSize == params[0].dstSize);
            r.dst_width = std::max(r.dst_width, (int32_t)p.dstSize.width);
            r.dst_height = std::max(r.dst_height, (int32_t)p.dstSize.height);
            for (int y = 0; y < 3; ++y) for (int x = 0; x < 3; ++x) b.warp.push_back(p.transformMatrix[y][x]);
        }
        if (differ) // per-plane destination sizes (reference include/cvGPUSpeedup.cuh:381-401): needs one output image per plane
            for (const auto& p : params) { b.warp_sizes.push_back(p.dstSize.width); b.warp_sizes.push_back(p.dstSize.height); }
    }
};

template <typename Op> template <size_t N> inline auto Read<Op>::then(const BatchCropSpec<N>& c) const {
    using T = typename Op::OutputType;
    BatchPixelRead<T> rd;
    for (const Rect& r : c.rects) rd.planes.push_back(image2d(crop_view(params, r)));
    rd.used = (int)N;
    return rd;
}
template <typename Op> template <InterpolationType IT> inline auto Read<Op>::then(const IncompleteResize<IT>& rs) const {
    static_assert(IT == INTER_LINEAR, "Interpolation type not supported yet.");
    return ResizeRead<typename Op::OutputType>{params, rs.dsize};
}

template <InterpolationType IT, AspectRatio AR = IGNORE_AR> struct Resize {
    static_assert(IT == INTER_LINEAR, "Interpolation type not supported yet.");
    static IncompleteResize<IT> build(const Size& dsize) { return {dsize}; }
    template <typename T> static auto build(const RawPtr<_2D, T>& in, const Size& dsize, double fx = 0., double fy = 0.) {
        Size d = dsize;
        if (d.width == 0 || d.height == 0) {
            d.width = (int)std::nearbyint(in.dims.width * fx);
            d.height = (int)std::nearbyint(in.dims.height * fy);
        }
        return ResizeRead<T>{in, d};
    }
    template <PixelFormat PF, ColorRange CR, ColorPrimitives CP, bool ALPHA, typename O, bool SW>
    static auto build(const YuvRead<PF, CR, CP, ALPHA, O, SW>& back, const Size& dsize, const float* background = nullptr) {
        ResizeYuvRead<YuvRead<PF, CR, CP, ALPHA, O, SW>> r{back, dsize};
        r.ar = (int)AR; // same numeric values as cvgs_aspect_ratio
        if (background)
            for (int c = 0; c < 4; ++c) r.bg[c] = background[c];
        return r;
    }
};

// ---- execution ------------------------------------------------------------------------------------------------
namespace detail {
// The reference README's own example (README.md:123-130) spells `cvGS::convertTo<CV_8UC3, CV_32FC3>()` directly behind the batched
// resize, whose output is ALREADY float3 (include/cvGPUSpeedup.cuh:227) -- the snippet users copy.  That one spelling is accepted: a
// SaturateCast<T, floatN> directly behind a resize read of T sources is the identity and is not lowered at all (the chain keeps its
// compile-time program and its kernel).  Every other type mismatch is still a compile-time error.
template <typename R> struct resize_source { using type = void; };
template <typename T> struct resize_source<ResizeRead<T>> { using type = T; };
template <typename T> struct resize_source<BatchResizeRead<T>> { using type = T; };
template <typename A, typename B> struct redundant_cast : std::false_type {};
template <typename A, typename I, typename O> struct redundant_cast<A, Unary<SaturateCast<I, O>>>
    : std::bool_constant<!std::is_void_v<typename resize_source<A>::type> && std::is_same_v<typename resize_source<A>::type, I> &&
                         std::is_same_v<typename A::OutputType, O>> {};
template <typename A, typename B> constexpr bool chains = std::is_same_v<typename A::OutputType, typename B::InputType> || redundant_cast<A, B>::value;

template <typename Tuple, size_t... I> constexpr bool types_chain(std::index_sequence<I...>) {
    return (chains<std::tuple_element_t<I, Tuple>, std::tuple_element_t<I + 1, Tuple>> && ...);
}
template <typename Tuple, size_t... I> constexpr bool middle_pointwise(std::index_sequence<I...>) {
    return ((std::tuple_element_t<I + 1, Tuple>::stage == Stage::Pointwise) && ...);
}
inline void check_status(int rc) {
    if (rc != CVGS_OK) throw std::runtime_error(std::string("cvGS: ") + cvgs_last_error());
}
} // namespace detail

// Lower a full IOp list (Read, pointwise..., Write) into a chain descriptor.  `b` must outlive the use of b.d.
template <typename... IOps> inline void lowerChain(ChainBuilder& b, const IOps&... iops) {
    using Tuple = std::tuple<IOps...>;
    constexpr size_t N = sizeof...(IOps);
    static_assert(N >= 2, "a chain needs at least a read and a write operation");
    static_assert(std::tuple_element_t<0, Tuple>::stage == Stage::Read, "the first operation must be a Read/ReadBack");
    static_assert(std::tuple_element_t<N - 1, Tuple>::stage == Stage::Write, "the last operation must be a Write");
    static_assert(detail::middle_pointwise<Tuple>(std::make_index_sequence<N - 2>{}),
                  "only Unary/Binary operations may sit between the read and the write");
    static_assert(detail::types_chain<Tuple>(std::make_index_sequence<N - 1>{}),
                  "the output type of each operation must be the input type of the next one");
    using First = std::tuple_element_t<0, Tuple>;
    const auto lower_one = [&b](const auto& iop, bool skip) { if (!skip) iop.lower(b); };
    size_t k = 0;
    (lower_one(iops, k++ == 1 && detail::redundant_cast<First, std::decay_t<decltype(iops)>>::value), ...);
    b.finish();
}

// fk::executeOperations<TF>(stream, iops...): ONE kernel, asynchronous on `stream`.
// ---- streams attached to a descriptor queue (engine extension) ----------------------------------------------------
// The reference's call shape is executeOperations(stream, iops...), asynchronous on that stream (include/cvGPUSpeedup.cuh:464-473).
// A stream attached to a fk::Queue (cvGS::attachQueue) keeps exactly that contract and routes the chains the server takes through
// cvgs_queue_submit_on: ordered behind the stream's earlier work and in front of its later work, no host synchronisation; anything
// else -- and a batch nothing in flight could overlap with (CVGS_QUEUE_SUBMIT_HYBRID) -- is the ordinary launch on the stream.
namespace detail {
struct StreamAttachment {
    hipStream_t stream;
    cvgs_queue_t queue;
    uint32_t flags;       // CVGS_QUEUE_SUBMIT_* (HYBRID always set)
    uint64_t last_ticket; // DEFER_WAIT streams: what fence() orders the stream behind
    bool has_ticket;
    // recorded ticks (attachTicks): executeOperations calls are RECORDED and go behind one gate when `tick` of them are pending, at
    // fence() / cv::cuda::Stream::waitForCompletion() / detach
    int tick = 0;
    std::vector<std::unique_ptr<ChainBuilder>> pending;
};
struct StreamAttachments {
    std::atomic<bool> any{false};
    std::mutex mu;
    std::vector<StreamAttachment> list;
};
inline StreamAttachments& stream_attachments() {
    // leaked on purpose: a namespace-scope cv::cuda::Stream is destroyed AFTER function-local statics of other translation
    // units may be gone, and its destroy hook looks in here
    static StreamAttachments* a = new StreamAttachments;
    return *a;
}
// submit what a recording stream has pending: ONE cvgs_queue_submit_many_on per <= 64 chains (the pending list is taken under the lock, the
// submit runs outside it)
inline void flush_attached(hipStream_t stream) {
    StreamAttachments& A = stream_attachments();
    std::vector<std::unique_ptr<ChainBuilder>> take;
    cvgs_queue_t q = nullptr;
    uint32_t flags = 0;
    {
        std::lock_guard<std::mutex> lock(A.mu);
        for (auto& a : A.list)
            if (a.stream == stream) { take.swap(a.pending); q = a.queue; flags = a.flags; break; }
    }
    if (take.empty()) return;
    // A chunk that fails as a whole (one chain the lowering rejects fails the group) is retried chain by chain, in order, as plain launches:
    // the other recorded calls are NOT dropped and the error names the offending chain.  Every chunk is attempted; the first
    // error is thrown once all of them have been.
    std::string first_error;
    auto one_by_one = [&](size_t base, size_t cnt) {
        for (size_t i = base; i < base + cnt; ++i)
            if (cvgs_execute(&take[i]->d, stream) != CVGS_OK && first_error.empty())
                first_error = std::string("cvGS (recorded call ") + std::to_string(i) + " of the tick): " + cvgs_last_error();
    };
    if (!q) { // recording without a queue (fk::recordTicks): ONE cvgs_execute_many launch per <= CVGS_MAX_CHAINS chains, strictly ordered
        std::vector<cvgs_chain_desc> flat(take.size());
        for (size_t i = 0; i < take.size(); ++i) flat[i] = take[i]->d;
        for (size_t base = 0; base < flat.size(); base += CVGS_MAX_CHAINS) {
            const size_t cnt = flat.size() - base < (size_t)CVGS_MAX_CHAINS ? flat.size() - base : (size_t)CVGS_MAX_CHAINS;
            if (cvgs_execute_many(flat.data() + base, (int32_t)cnt, stream) != CVGS_OK) one_by_one(base, cnt);
        }
        if (!first_error.empty()) throw std::runtime_error(first_error);
        return;
    }
    std::vector<const cvgs_chain_desc*> ptrs(take.size());
    for (size_t i = 0; i < take.size(); ++i) ptrs[i] = &take[i]->d;
    uint64_t last = CVGS_QUEUE_TICKET_DIRECT, newest = CVGS_QUEUE_TICKET_DIRECT;
    for (size_t base = 0; base < ptrs.size(); base += CVGS_QUEUE_MAX_GROUP) {
        const size_t cnt = ptrs.size() - base < (size_t)CVGS_QUEUE_MAX_GROUP ? ptrs.size() - base : (size_t)CVGS_QUEUE_MAX_GROUP;
        const int rc = cvgs_queue_submit_many_on(q, ptrs.data() + base, (int32_t)cnt, stream, flags, &last);
        if (rc == CVGS_ERR_UNSUPPORTED || rc == CVGS_ERR_INVALID) { // refused BEFORE anything was published: the chunk runs as plain launches
            // (with DEFER_WAIT the launches are ordered behind the stream, not behind earlier groups still on the server: order them first)
            if (newest != CVGS_QUEUE_TICKET_DIRECT) (void)cvgs_queue_stream_wait(q, newest, stream);
            one_by_one(base, cnt);
            continue;
        }
        if (rc != CVGS_OK) { // a HIP / server error: part of the group may already be on the server -- running it again would run chains twice
            if (first_error.empty()) first_error = std::string("cvGS (recorded tick, chunk at call ") + std::to_string(base) + "): " + cvgs_last_error();
            continue;
        }
        if (last != CVGS_QUEUE_TICKET_DIRECT) newest = last;
    }
    if (newest != CVGS_QUEUE_TICKET_DIRECT) {
        std::lock_guard<std::mutex> lock(A.mu);
        for (auto& a : A.list)
            if (a.stream == stream) { a.last_ticket = newest; a.has_ticket = true; break; }
    }
    if (!first_error.empty()) throw std::runtime_error(first_error);
}
// a recording stream takes the chain (true) -- and flushes when the tick is full
inline bool record_attached(hipStream_t stream, std::unique_ptr<ChainBuilder>& b) {
    StreamAttachments& A = stream_attachments();
    bool full = false;
    {
        std::lock_guard<std::mutex> lock(A.mu);
        StreamAttachment* at = nullptr;
        for (auto& a : A.list)
            if (a.stream == stream) { at = &a; break; }
        if (!at || at->tick <= 0) return false;
        at->pending.push_back(std::move(b));
        full = (int)at->pending.size() >= at->tick;
    }
    if (full) flush_attached(stream);
    return true;
}
inline bool stream_records(hipStream_t stream) {
    StreamAttachments& A = stream_attachments();
    std::lock_guard<std::mutex> lock(A.mu);
    for (const auto& a : A.list)
        if (a.stream == stream) return a.tick > 0;
    return false;
}
inline bool submit_attached(hipStream_t stream, const cvgs_chain_desc* d) {
    StreamAttachments& A = stream_attachments();
    cvgs_queue_t q = nullptr;
    uint32_t flags = 0;
    {
        std::lock_guard<std::mutex> lock(A.mu);
        for (const auto& a : A.list)
            if (a.stream == stream) { q = a.queue; flags = a.flags; break; }
    }
    if (!q) return false;
    uint64_t ticket = 0;
    check_status(cvgs_queue_submit_on(q, d, stream, flags, &ticket));
    if ((flags & CVGS_QUEUE_SUBMIT_DEFER_WAIT) && ticket != CVGS_QUEUE_TICKET_DIRECT) {
        std::lock_guard<std::mutex> lock(A.mu);
        for (auto& a : A.list)
            if (a.stream == stream) { a.last_ticket = ticket; a.has_ticket = true; break; }
    }
    return true;
}
} // namespace detail

template <bool THREAD_FUSION = true, typename... IOps>
inline void executeOperations(hipStream_t stream, const IOps&... iops) {
    if (detail::stream_attachments().any.load(std::memory_order_acquire) && detail::stream_records(stream)) {
        std::unique_ptr<ChainBuilder> rec(new ChainBuilder); // recorded: the builder outlives this call (it owns the descriptor's arrays)
        lowerChain(*rec, iops...);
        detail::check_status(cvgs_validate(&rec->d)); // an invalid chain throws HERE, from the call that spelled it, not from a later flush
        if (detail::record_attached(stream, rec)) return;
        detail::check_status(cvgs_execute(&rec->d, stream)); // (detached in between)
        return;
    }
    ChainBuilder b;
    lowerChain(b, iops...);
    // THREAD_FUSION = false is a tuning hint of the reference (its fused-thread path does not cover every type; the reference's
    // own tests spell executeOperations<false> for 3-channel cvtColor and batched reads).  This engine's multi-pixel kernels give
    // the same bits as its one-pixel kernel for every chain they accept, so the hint is NOT forwarded: honouring it would only
    // select the slower kernel (4K BGR -> RGB: 11 us vs 50 us).  Define CVGS_HONOUR_THREAD_FUSION_HINT to forward it (debugging).
#ifdef CVGS_HONOUR_THREAD_FUSION_HINT
    if (!THREAD_FUSION) b.d.flags |= CVGS_CHAIN_NO_THREAD_FUSION;
#endif
    if (detail::stream_attachments().any.load(std::memory_order_acquire) && detail::submit_attached(stream, &b.d)) return;
    detail::check_status(cvgs_execute(&b.d, stream));
}

// ---- launch batching (engine extension: cvgs_execute_many) -----------------------------------------------------------
// A 50-crop chain is ~1 us of HBM time behind a ~1.8 us launch floor; a host with several frames in hand (multi-camera
// serving) records one chain per frame and submits them together: chains of the same K1 shape run as ONE kernel launch,
// bit-identical to one executeOperations per chain.  The reference's closest spelling is its batch sweep
// (tests/batchresize/test_batchresize_x_split3D.cu:384-392).
//     fk::ChainBatch batch;
//     for (auto& cam : cameras) batch.add(cvGS::resize<...>(cam.crops, size, n), ..., cvGS::split<CV_32FC3>(cam.tensor, size));
//     batch.execute(stream);          // then batch.clear() and record the next frames
class ChainBatch {
public:
    template <typename... IOps> void add(const IOps&... iops) {
        if (builders_.size() >= (size_t)CVGS_MAX_CHAINS) throw std::runtime_error("cvGS: more than CVGS_MAX_CHAINS chains in one batch");
        builders_.emplace_back(new ChainBuilder);
        lowerChain(*builders_.back(), iops...);
    }
    size_t size() const { return builders_.size(); }
    void clear() { builders_.clear(); }
    void execute(hipStream_t stream) {
        if (builders_.empty()) return;
        descs_.resize(builders_.size());
        for (size_t i = 0; i < builders_.size(); ++i) descs_[i] = builders_[i]->d; // POD copy; the builders keep the arrays alive
        // a stream attached to a queue: the tick's chains go behind ONE gate on the stream (cvgs_queue_submit_many_on), 64 at a time
        if (detail::stream_attachments().any.load(std::memory_order_acquire)) {
            detail::flush_attached(stream); // calls recorded on this stream before the batch go first
            cvgs_queue_t q = nullptr;
            uint32_t flags = 0;
            detail::StreamAttachments& A = detail::stream_attachments();
            {
                std::lock_guard<std::mutex> lock(A.mu);
                for (const auto& a : A.list)
                    if (a.stream == stream) { q = a.queue; flags = a.flags; break; }
            }
            if (q) {
                std::vector<const cvgs_chain_desc*> ptrs(descs_.size());
                for (size_t i = 0; i < descs_.size(); ++i) ptrs[i] = &descs_[i];
                uint64_t last = CVGS_QUEUE_TICKET_DIRECT;
                for (size_t base = 0; base < ptrs.size(); base += CVGS_QUEUE_MAX_GROUP) {
                    const size_t cnt = ptrs.size() - base < (size_t)CVGS_QUEUE_MAX_GROUP ? ptrs.size() - base : (size_t)CVGS_QUEUE_MAX_GROUP;
                    detail::check_status(cvgs_queue_submit_many_on(q, ptrs.data() + base, (int32_t)cnt, stream, flags, &last));
                }
                if ((flags & CVGS_QUEUE_SUBMIT_DEFER_WAIT) && last != CVGS_QUEUE_TICKET_DIRECT) {
                    std::lock_guard<std::mutex> lock(A.mu);
                    for (auto& a : A.list)
                        if (a.stream == stream) { a.last_ticket = last; a.has_ticket = true; break; }
                }
                return;
            }
        }
        detail::check_status(cvgs_execute_many(descs_.data(), (int32_t)descs_.size(), stream));
    }
private:
    std::vector<std::unique_ptr<ChainBuilder>> builders_;
    std::vector<cvgs_chain_desc> descs_;
};

// ---- fk::buildOperationSequence + the divergent-batch launch (reference tests/batchread/test_circularbatchread_x_write3D.cu:147-156,
// tests/resize/test_fused_resize.cu:73-92) --------------------------------------------------------------------------------------------
// The reference runs DIFFERENT operation sequences on the planes of one launch: grid z = plane, plane z executes sequence
// SequenceSelector::at(z) (1-based) with z as its thread's z index,
//     fk::launchDivergentBatchTransformDPP_Kernel<fk::ParArch::GPU_NVIDIA, Selector><<<grid(.., .., BATCH), block, 0, stream>>>(seq1, seq2);
// A raw kernel launch has no host-C++ spelling; its counterpart here is
//     fk::executeDivergentBatch<Selector>(stream, BATCH, seq1, seq2);
// Each plane's sequence is lowered to its own chain with "z as the thread's z index" resolved on the host -- a batched read keeps its
// plane z, a tensor write starts at its plane z, 2D reads / writes ignore z -- and all planes go to ONE cvgs_execute_many call: chains of
// one hot shape (crops -> resize -> ... -> tensor) are ONE launch with grid z = plane, anything else is launched plane by plane in z
// order on the stream (same results: the planes are independent by construction).
template <typename... IOps> struct OperationSequence {
    std::tuple<IOps...> iops;
};
template <typename... IOps> inline OperationSequence<IOps...> buildOperationSequence(const IOps&... iops) {
    using Tuple = std::tuple<IOps...>;
    constexpr size_t N = sizeof...(IOps);
    static_assert(N >= 2, "an operation sequence needs at least a read and a write operation");
    static_assert(std::tuple_element_t<0, Tuple>::stage == Stage::Read, "the first operation must be a Read/ReadBack");
    static_assert(std::tuple_element_t<N - 1, Tuple>::stage == Stage::Write, "the last operation must be a Write");
    return OperationSequence<IOps...>{Tuple(iops...)};
}
enum class ParArch { GPU_NVIDIA = 0, GPU_AMD = 1 };
namespace detail {
// "z is the thread's z index" for one lowered chain
inline void select_plane(ChainBuilder& b, uint z) {
    cvgs_read_desc& r = b.d.read;
    if (r.kind == CVGS_READ_WARP_AFFINE || r.kind == CVGS_READ_WARP_PERSPECTIVE || (r.flags & CVGS_READ_FLAG_TABLE_ON_DEVICE))
        throw std::runtime_error("cvGS: executeDivergentBatch takes pixel / resize / 4:2:0 reads with host descriptors");
    if (r.batch > 1) { // a batched read: plane z of it
        if (z >= (uint)r.batch) throw std::runtime_error("cvGS: executeDivergentBatch: plane index beyond the sequence's batched read");
        if (z != 0) b.src[0] = b.src[z];
        b.src.resize(1);
        r.used_planes = (int)z < r.used_planes ? 1 : 0;
        r.batch = 1;
    }
    cvgs_write_desc& w = b.d.write;
    static const size_t kDepthBytes[8] = {1, 1, 2, 2, 4, 4, 8, 2}; // CV_8U .. CV_64F, CV_16F
    const size_t base = kDepthBytes[CVGS_TYPE_DEPTH(w.dst_type) & 7]; // bytes of one channel element
    const size_t plane = (size_t)w.width * (size_t)w.height;
    switch (w.kind) {
    case CVGS_WRITE_PIXEL_3D: w.data = (uint8_t*)w.data + (size_t)z * plane * base * (size_t)CVGS_TYPE_CN(w.dst_type); break;
    case CVGS_WRITE_TENSOR_SPLIT: w.data = (uint8_t*)w.data + (size_t)z * plane * base * (size_t)CVGS_TYPE_CN(w.dst_type); break;
    case CVGS_WRITE_TENSOR_T_SPLIT: w.data = (uint8_t*)w.data + (size_t)z * plane * base; break; // (the channel stride stays the tensor's N)
    case CVGS_WRITE_PIXEL_2D: break; // a 2D write ignores z
    default: throw std::runtime_error("cvGS: executeDivergentBatch takes tensor / image writes");
    }
    if ((w.kind == CVGS_WRITE_PIXEL_3D || w.kind == CVGS_WRITE_TENSOR_SPLIT || w.kind == CVGS_WRITE_TENSOR_T_SPLIT) && z >= (uint)w.planes)
        throw std::runtime_error("cvGS: executeDivergentBatch: plane index beyond the tensor");
}
template <typename Seq, size_t... I> inline void lower_sequence(ChainBuilder& b, const Seq& seq, std::index_sequence<I...>) {
    lowerChain(b, std::get<I>(seq.iops)...);
}
template <size_t K, typename... Seqs> inline void lower_selected(ChainBuilder& b, uint which, const std::tuple<const Seqs&...>& seqs) {
    if constexpr (K < sizeof...(Seqs)) {
        if (which == K + 1) {
            const auto& seq = std::get<K>(seqs);
            lower_sequence(b, seq, std::make_index_sequence<std::tuple_size_v<decltype(seq.iops)>>{});
        } else lower_selected<K + 1>(b, which, seqs);
    }
}
} // namespace detail
template <typename SequenceSelector, typename... Seqs>
inline void executeDivergentBatch(hipStream_t stream, uint batch, const Seqs&... seqs) {
    static_assert(sizeof...(Seqs) >= 1, "executeDivergentBatch needs at least one operation sequence");
    if (batch > (uint)CVGS_MAX_CHAINS) throw std::runtime_error("cvGS: more than CVGS_MAX_CHAINS planes in one divergent batch");
    std::vector<std::unique_ptr<ChainBuilder>> builders;
    std::vector<cvgs_chain_desc> descs;
    const std::tuple<const Seqs&...> all(seqs...);
    for (uint z = 0; z < batch; ++z) {
        const uint which = (uint)SequenceSelector::at(z); // 1-based; 0 or beyond the list: the plane runs nothing (the reference's divergent_operate falls through)
        if (which < 1 || which > sizeof...(Seqs)) continue;
        builders.emplace_back(new ChainBuilder);
        detail::lower_selected<0>(*builders.back(), which, all);
        detail::select_plane(*builders.back(), z);
        builders.back()->finish();
        descs.push_back(builders.back()->d); // POD copy; the builder keeps the arrays alive
    }
    if (descs.empty()) return;
    if (detail::stream_attachments().any.load(std::memory_order_acquire)) detail::flush_attached(stream); // recorded calls of this stream go first
    detail::check_status(cvgs_execute_many(descs.data(), (int32_t)descs.size(), stream));
}

// ---- device-side descriptor queue (engine extension: cvgs_queue_*) ---------------------------------------------------
// executeOperations' call shape -- one call per frame, the same IOps -- without a kernel launch per call: a resident server
// grid takes the batch from a ring and consecutive batches overlap on the device (a 50-crop batch every ~2.6 us instead of
// 4.4 us with one launch each; include/cvgs_hip.h "device-side descriptor queue").  Taken: K1's hot shape (batched
// 8UC3 / 8UC4 bilinear resize -> [cvtColor swap] -> multiply -> subtract -> divide [-> convertTo CV_16F] -> split into an fp32 / fp16 tensor; any batch size, 74 crops per ring slot)
// or the same behind crops of NV12 / NV21 decoder surfaces (resize(cvtColorNV12<...>(surface), size) / Resize over ReadYUV);
// a queue serves the kind of its first call.  Anything else throws, exactly as an unsupported chain does elsewhere in the
// facade; those chains belong to the stream form.
//     fk::Queue q;                                             // once
//     auto t = fk::executeOperations(q, resize, cvt, mul, sub, div, split);   // per frame, asynchronous
//     q.wait(t);   or   q.wait(t, consumerStream);             // host wait, or order a consumer stream behind the ticket
// A wait covers the ticket's batch AND every batch submitted before it (the device completes batches in any order).
// The sources must be complete when the call is made (the server is not ordered behind any stream).
class Queue {
public:
    explicit Queue(int device = -1 /* the current device */, int depth = 0, double idle_us = 0.0) { detail::check_status(cvgs_queue_create(&q_, device, depth, idle_us, 0u)); }
    Queue(const Queue&) = delete;
    Queue& operator=(const Queue&) = delete;
    ~Queue() {
        if (!q_) return;
        {
            detail::StreamAttachments& A = detail::stream_attachments();
            std::vector<hipStream_t> mine;
            {
                std::lock_guard<std::mutex> lock(A.mu);
                for (const auto& a : A.list)
                    if (a.queue == q_ && !a.pending.empty()) mine.push_back(a.stream);
            }
            for (hipStream_t s : mine) { // recorded calls are not dropped with the queue
                try { detail::flush_attached(s); } catch (...) {}
            }
            std::lock_guard<std::mutex> lock(A.mu);
            for (size_t i = A.list.size(); i-- > 0;)
                if (A.list[i].queue == q_) A.list.erase(A.list.begin() + (long)i);
            A.any.store(!A.list.empty(), std::memory_order_release);
        }
        (void)cvgs_queue_destroy(q_);
    }
    template <typename... IOps> uint64_t submit(const IOps&... iops) {
        ChainBuilder b;
        lowerChain(b, iops...);
        uint64_t ticket = 0;
        detail::check_status(cvgs_queue_submit(q_, &b.d, &ticket));
        return ticket;
    }
    // the stream-ordered form (cvgs_queue_submit_on): behind everything already on `stream`, in front of everything after it
    template <typename... IOps> uint64_t submitOn(hipStream_t stream, uint32_t flags, const IOps&... iops) {
        ChainBuilder b;
        lowerChain(b, iops...);
        uint64_t ticket = 0;
        detail::check_status(cvgs_queue_submit_on(q_, &b.d, stream, flags, &ticket));
        return ticket;
    }
    void wait(uint64_t ticket, double timeout_s = 10.0) { detail::check_status(cvgs_queue_wait(q_, ticket, timeout_s)); }
    void wait(uint64_t ticket, hipStream_t consumer) { detail::check_status(cvgs_queue_stream_wait(q_, ticket, consumer)); }
    // after a wait / submit has thrown because the server's watchdog fired: reset the queue; returns the number of lost batches
    uint64_t recover() {
        uint64_t lost = 0;
        detail::check_status(cvgs_queue_recover(q_, &lost));
        return lost;
    }
    // executeOperations(stream, ...) on `stream` goes through this queue from now on (deferWait: the caller orders consumers with fence())
    // minGroup: the smallest number of chains behind one gate the server takes (0 = the engine's default, 8; 1 = always the server)
    void attach(hipStream_t stream, bool deferWait = false, int minGroup = 0) {
        detail::StreamAttachments& A = detail::stream_attachments();
        if (A.any.load(std::memory_order_acquire)) detail::flush_attached(stream); // (calls a previous attachment recorded)
        cv::cuda::cvgs_stream_destroy_hook() = &Queue::stream_dying; // a cv::cuda::Stream that dies takes its attachment with it
        std::lock_guard<std::mutex> lock(A.mu);
        const uint32_t f = CVGS_QUEUE_SUBMIT_HYBRID | (deferWait ? CVGS_QUEUE_SUBMIT_DEFER_WAIT : 0u) | CVGS_QUEUE_SUBMIT_MIN_GROUP(minGroup);
        for (auto& a : A.list)
            if (a.stream == stream) { a = detail::StreamAttachment{stream, q_, f, 0, false}; return; }
        A.list.push_back(detail::StreamAttachment{stream, q_, f, 0, false});
        A.any.store(true, std::memory_order_release);
    }
    // RECORDED TICKS: executeOperations(stream, ...) calls on `stream` are recorded and submitted `tick` at a time behind ONE gate (deferred
    // waits); fence(stream) -- which cv::cuda::Stream::waitForCompletion() calls first -- submits what is pending and orders the stream
    // behind it.  The reference's multi-camera loop (one call per camera, one synchronisation per tick) gets the queue's speed unchanged.
    // Contract (as deferWait): sources and tensors of recorded calls are in flight until the fence; do not rewrite / read them before.
    void attachTicks(hipStream_t stream, int tick = 16) {
        attach(stream, /*deferWait=*/true, /*minGroup=*/0);
        detail::StreamAttachments& A = detail::stream_attachments();
        std::lock_guard<std::mutex> lock(A.mu);
        for (auto& a : A.list)
            if (a.stream == stream) a.tick = tick < 1 ? 1 : (tick > 4 * CVGS_QUEUE_MAX_GROUP ? 4 * CVGS_QUEUE_MAX_GROUP : tick);
        cv::cuda::cvgs_stream_sync_hook() = &Queue::fence;
        cv::cuda::cvgs_stream_destroy_hook() = &Queue::stream_dying;
    }
    // what a dying cv::cuda::Stream calls (cv_shim.h: cvgs_stream_destroy_hook, installed when this header is loaded): recorded calls are
    // submitted and the attachment goes, then the engine retires what it keeps for the stream HANDLE (cvgs_stream_release: the table ring of
    // cvgs_execute_many -- a handle the runtime hands out again must not continue it)
    static void stream_dying(hipStream_t stream) {
        detach(stream);
        (void)cvgs_stream_release(stream);
    }
    static void detach(hipStream_t stream) {
        detail::flush_attached(stream);
        detail::StreamAttachments& A = detail::stream_attachments();
        std::lock_guard<std::mutex> lock(A.mu);
        for (size_t i = 0; i < A.list.size(); ++i)
            if (A.list[i].stream == stream) { A.list.erase(A.list.begin() + (long)i); break; }
        A.any.store(!A.list.empty(), std::memory_order_release);
    }
    // deferWait streams: the ticket of the newest batch the stream has submitted to the server (false: none since the last fence);
    // wait(ticket, stream) later orders a consumer behind it and everything before it -- a pipeline of any depth
    static bool lastTicket(hipStream_t stream, uint64_t* ticket) {
        detail::StreamAttachments& A = detail::stream_attachments();
        std::lock_guard<std::mutex> lock(A.mu);
        for (const auto& a : A.list)
            if (a.stream == stream && a.has_ticket) { *ticket = a.last_ticket; return true; }
        return false;
    }
    // deferWait streams: order everything enqueued on `stream` from here on behind the batches it has submitted so far
    static void fence(hipStream_t stream) {
        detail::StreamAttachments& A = detail::stream_attachments();
        if (!A.any.load(std::memory_order_acquire)) return;
        detail::flush_attached(stream);
        cvgs_queue_t q = nullptr;
        uint64_t t = 0;
        {
            std::lock_guard<std::mutex> lock(A.mu);
            for (auto& a : A.list)
                if (a.stream == stream && a.has_ticket) { q = a.queue; t = a.last_ticket; a.has_ticket = false; break; }
        }
        if (q) detail::check_status(cvgs_queue_stream_wait(q, t, stream));
    }
    cvgs_queue_t handle() const { return q_; }
private:
    cvgs_queue_t q_ = nullptr;
};
template <bool THREAD_FUSION = true, typename... IOps>
inline uint64_t executeOperations(Queue& queue, const IOps&... iops) {
    return queue.submit(iops...);
}
// RECORDED TICKS WITHOUT A QUEUE: executeOperations(stream, ...) calls on `stream` are recorded and launched `tick` at a time as ONE
// multi-chain kernel (cvgs_execute_many) -- strictly stream-ordered, no resident server; Queue::fence(stream) /
// cv::cuda::Stream::waitForCompletion() / stopRecording launch what is pending.  Recorded calls are in flight until then (a consumer
// enqueued on the stream before the fence is not ordered behind them).
inline void recordTicks(hipStream_t stream, int tick = 16) {
    detail::StreamAttachments& A = detail::stream_attachments();
    detail::flush_attached(stream);
    std::lock_guard<std::mutex> lock(A.mu);
    detail::StreamAttachment at{stream, nullptr, 0u, 0, false};
    at.tick = tick < 1 ? 1 : (tick > CVGS_MAX_CHAINS ? CVGS_MAX_CHAINS : tick);
    bool found = false;
    for (auto& a : A.list)
        if (a.stream == stream) { a = std::move(at); found = true; break; }
    if (!found) A.list.push_back(std::move(at));
    A.any.store(true, std::memory_order_release);
    cv::cuda::cvgs_stream_sync_hook() = &Queue::fence;
    cv::cuda::cvgs_stream_destroy_hook() = &Queue::stream_dying;
}
inline void stopRecording(hipStream_t stream) { Queue::detach(stream); }
// installed when this header is loaded: every owning cv::cuda::Stream that dies releases what the engine keeps for its handle
namespace detail {
inline const bool stream_hook_installed = (cv::cuda::cvgs_stream_destroy_hook() = &Queue::stream_dying, true);
}

// ---- CircularTensor ------------------------------------------------------------------------------------------------
// MIRRORED (engine extension, default off = the reference's behaviour): the opt-in mirrored-ring layout of
// cvgs_circular_create_ex -- no shift traffic per update, but ptr()/data() MOVE with every update.
// CAPTURABLE (engine extension, default off): update() may be captured into a HIP graph and replayed (CVGS_CIRCULAR_CAPTURABLE:
// the update count lives on the device; N captured updates replay as the NEXT N updates).  For MIRRORED + CAPTURABLE tensors
// ptr() asks the device where the window stands (it synchronises): call it outside captures.
template <typename T, int COLOR_PLANES, int BATCH, CircularTensorOrder ORDER, ColorPlanes MODE = ColorPlanes::Standard,
          bool MIRRORED = false, bool CAPTURABLE = false>
class CircularTensor {
    static constexpr ND kND = MODE == ColorPlanes::Transposed ? T3D : _3D;
    static_assert(!MIRRORED || MODE == ColorPlanes::Standard, "mirrored CircularTensors exist in the Standard plane order only");
public:
    CircularTensor() = default;
    CircularTensor(uint w, uint h, int device = 0) { Alloc(w, h, device); }
    CircularTensor(const CircularTensor&) = delete;
    CircularTensor& operator=(const CircularTensor&) = delete;
    ~CircularTensor() { if (handle_) (void)cvgs_circular_destroy(handle_); }

    void Alloc(uint w, uint h, int device = 0) {
        detail::check_status(cvgs_circular_create_ex(&handle_, (int)w, (int)h, cvGS::cv_type_of<T>, COLOR_PLANES, BATCH,
                                                     (int)ORDER, (int)MODE, device,
                                                     (MIRRORED ? CVGS_CIRCULAR_MIRRORED : 0u) | (CAPTURABLE ? CVGS_CIRCULAR_CAPTURABLE : 0u)));
        ptr_a.data = (T*)cvgs_circular_data(handle_);
        ptr_a.dims = {w, h, (uint)BATCH, (uint)COLOR_PLANES, (uint)(w * sizeof(T)), (uint)(w * sizeof(T) * h)};
    }
    // update(stream, readIOp, ops..., writeIOp): the new frame = ops(read) becomes slot 0 (NewestFirst) or
    // BATCH-1 (OldestFirst), every older frame moves one slot, the tensor at ptr()/data() is rewritten.
    template <typename... IOps> void update(hipStream_t stream, const IOps&... iops) {
        ChainBuilder b;
        lowerChain(b, iops...);
        constexpr bool tsplit_needed = MODE == ColorPlanes::Transposed;
        if (tsplit_needed != (b.d.write.kind == CVGS_WRITE_TENSOR_T_SPLIT))
            throw std::runtime_error("Need to use TensorTSplit as write function exactly when CP_MODE = Transposed");
        if (detail::stream_attachments().any.load(std::memory_order_acquire)) detail::flush_attached(stream); // recorded calls of this stream go first
        detail::check_status(cvgs_circular_update(handle_, &b.d, stream));
        if constexpr (MIRRORED && !CAPTURABLE) ptr_a.data = (T*)cvgs_circular_data(handle_); // the window moved
    }
    RawPtr<kND, T> ptr() const {
        if constexpr (MIRRORED && CAPTURABLE) {
            RawPtr<kND, T> p = ptr_a;
            p.data = (T*)cvgs_circular_data(handle_);
            return p;
        } else {
            return ptr_a;
        }
    }
    Dims3D dims() const { return ptr_a.dims; }
    size_t sizeInBytes() const { return cvgs_circular_bytes(handle_); }

protected:
    RawPtr<kND, T> ptr_a;
    cvgs_circular_t handle_ = nullptr;
};

} // namespace fk
