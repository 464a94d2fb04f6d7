// k_generic.hip -- the interpreted fused kernel: one thread = one output element of plane z,
// Read -> pointwise program -> Write, everything selected by wave-uniform kernel arguments.
// It accepts every chain the C-ABI can express and is the semantic reference for the specialised
// kernels (tests check they agree bit for bit).  Replaces FKL's launchTransformDPP_Kernel as
// launched by fk::executeOperations (reference include/cvGPUSpeedup.cuh:464-583).
#include "k_common.hpp"

namespace cvgs {

__device__ __forceinline__ void read_tap(const ReadArgs& r, const PlaneParams& P, const YuvK& yk, int x, int y,
                                         Px& p) {
    if (r.kind == CVGS_READ_NV12 || r.kind == CVGS_READ_NV12_RESIZE_LINEAR) {
        nv12_px(P, x, y, yk, p);
    } else {
        load_px(P.data + (size_t)y * (size_t)P.step, r.depth, r.cn, x, p);
#pragma unroll
        for (int c = 0; c < 4; ++c) p.v[c] = tap_f(p.v[c], r.depth);
    }
}

// 64 x 4 threads: a wave is one 64-pixel row segment, so planar stores are 256-byte coalesced.
template <int NPL>
__global__ __launch_bounds__(256) void k_generic(const KernArgs<NPL> a, const MirrorArgs mirrors) {
    const ChainArgs& c = a.c;
    const ReadArgs& r = c.read;
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int z = blockIdx.z;
    if (x >= r.dst_w || y >= r.dst_h) return;

    Px p;
    p.v[0] = p.v[1] = p.v[2] = p.v[3] = 0.f;
    int depth = r.is_resize || r.kind == CVGS_READ_NV12 ? CVGS_DEPTH_32F : r.depth;
    int cn = r.out_cn;

    if (z >= r.used) {
        // fk::BatchRead<N,CONDITIONAL_WITH_DEFAULT>: default value, then the whole chain
#pragma unroll
        for (int k = 0; k < 4; ++k)
            p.v[k] = depth == CVGS_DEPTH_32S ? from_int((int)r.bg[k]) : (depth == CVGS_DEPTH_16F ? round_half(r.bg[k]) : r.bg[k]);
    } else {
        PlaneParams P;
        if constexpr (NPL == 0) P = r.table[z];
        else P = a.planes[z];
        YuvK yk = yuv_matrix(r.yuv_range, r.yuv_primaries, r.yuv_layout);
        if (!r.is_resize) {
            if (r.kind == CVGS_READ_NV12) nv12_px(P, x, y, yk, p);
            else load_px(P.data + (size_t)y * (size_t)P.step, r.depth, r.cn, x, p);
        } else if (x >= P.x1 && x <= P.x2 && y >= P.y1 && y <= P.y2) {
            // fk::Resize + fk::Interpolate<INTER_LINEAR>: see oracle/cvgs_oracle.c interpolate_linear
            const float sx = (float)(x - P.x1) * P.fx;
            const float sy = (float)(y - P.y1) * P.fy;
            const int x1 = (int)floorf(sx), y1 = (int)floorf(sy);
            const int x2 = x1 + 1, y2 = y1 + 1;
            const int x2r = min(x2, P.w - 1), y2r = min(y2, P.h - 1);
            Px p00, p10, p01, p11;
            read_tap(r, P, yk, x1, y1, p00);
            read_tap(r, P, yk, x2r, y1, p10);
            read_tap(r, P, yk, x1, y2r, p01);
            read_tap(r, P, yk, x2r, y2r, p11);
            const float w00 = ((float)x2 - sx) * ((float)y2 - sy);
            const float w10 = (sx - (float)x1) * ((float)y2 - sy);
            const float w01 = ((float)x2 - sx) * (sy - (float)y1);
            const float w11 = (sx - (float)x1) * (sy - (float)y1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float acc = p00.v[k] * w00;
                acc = acc + p10.v[k] * w10;
                acc = acc + p01.v[k] * w01;
                acc = acc + p11.v[k] * w11;
                p.v[k] = acc;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) p.v[k] = r.bg[k];
        }
    }

    InterpProgInt::run(c.prog, p, depth, cn);

    const DstPlane* dst = c.write.table ? c.write.table : c.dst_inline;
    write_px(c.write, dst, x, y, z, p, depth, cn);
    // cvgs_write_desc.mirrors (tensor kinds): the same value at the same offsets of every further tensor
    for (int m = 0; m < mirrors.n; ++m) {
        WriteArgs w = c.write;
        w.data = mirrors.p[m];
        w.data2 = nullptr;
        write_px(w, dst, x, y, z, p, depth, cn);
    }
}

template <int NPL>
static int launch_generic_t(const ChainArgs& c, const PlaneParams* inline_planes, int n_inline, const MirrorArgs& mirrors, hipStream_t stream) {
    KernArgs<NPL> a;
    a.c = c;
    if constexpr (NPL > 0) {
        for (int i = 0; i < n_inline; ++i) a.planes[i] = inline_planes[i];
        for (int i = n_inline; i < NPL; ++i) a.planes[i] = PlaneParams{};
    } else {
        a.planes[0] = PlaneParams{};
    }
    const dim3 block(64, 4, 1);
    const dim3 grid((c.read.dst_w + 63) / 64, (c.read.dst_h + 3) / 4, c.read.batch);
    hipLaunchKernelGGL(k_generic<NPL>, grid, block, 0, stream, a, mirrors);
    return (int)hipGetLastError();
}

int launch_generic(const ChainArgs& c, const PlaneParams* inline_planes, int n_inline, const MirrorArgs& mirrors, void* stream, bool dry_run,
                   LaunchInfo* info) {
    if (info) info->kernel = c.read.table ? "generic_table" : (n_inline <= 8 ? "generic_inline8" : "generic_inline64");
    if (dry_run) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (c.read.table) e = (hipError_t)launch_generic_t<0>(c, nullptr, 0, mirrors, s);
    else if (n_inline <= 8) e = (hipError_t)launch_generic_t<8>(c, inline_planes, n_inline, mirrors, s);
    else e = (hipError_t)launch_generic_t<CVGS_KERNARG_PLANES>(c, inline_planes, n_inline, mirrors, s);
    return e == hipSuccess ? 0 : -(int)e - 1000;
}

} // namespace cvgs
