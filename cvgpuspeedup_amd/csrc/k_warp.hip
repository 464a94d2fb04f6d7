// k_warp.hip -- cvGS::warp: affine / perspective warping fused in front of a pointwise chain and a write stage
// (reference include/cvGPUSpeedup.cuh:288-442 -> fk::Warping<WT, Read<PerThreadRead<_2D,T>>>, batched with
// usedPlanes / default value like the resize; tests/warping/test_warping_opencv.cu).  SURVEY.md 8(f)4.
// One thread = one output pixel of plane z: destination (x,y) -> source position through the inverse transform
// (strict fp32, products and sums in the written order, IEEE division), INTER_LINEAR taps as in the resize, zero
// outside the source.  Interpreted program and generic write stage: this is not a hot-path kernel.
#include "k_common.hpp"

namespace cvgs {

template <int NPL>
struct WarpKernArgs {
    ChainArgs c;
    WarpPlane planes[NPL > 0 ? NPL : 1];
};

template <int NPL>
__global__ __launch_bounds__(256) void k_warp(const WarpKernArgs<NPL> a, const WarpPlane* __restrict__ table) {
    const ChainArgs& c = a.c;
    const ReadArgs& r = c.read;
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int z = blockIdx.z;
    if (x >= r.dst_w || y >= r.dst_h) return;

    Px p;
    p.v[0] = p.v[1] = p.v[2] = p.v[3] = 0.f;
    int depth = CVGS_DEPTH_32F, cn = r.out_cn;
    if (z >= r.used) {
#pragma unroll
        for (int k = 0; k < 4; ++k) p.v[k] = r.bg[k]; // fk::BatchRead default value, then the whole chain
    } else {
        WarpPlane P;
        if constexpr (NPL == 0) P = table[z];
        else P = a.planes[z];
        const float fx = (float)x, fy = (float)y;
        float sx = (P.m[0] * fx + P.m[1] * fy) + P.m[2];
        float sy = (P.m[3] * fx + P.m[4] * fy) + P.m[5];
        if (r.kind == CVGS_READ_WARP_PERSPECTIVE) {
            const float w = (P.m[6] * fx + P.m[7] * fy) + P.m[8];
            sx = sx / w;
            sy = sy / w;
        }
        if (sx >= 0.f && sx < (float)P.w && sy >= 0.f && sy < (float)P.h) {
            const int x1 = (int)floorf(sx), y1 = (int)floorf(sy);
            const int x2 = x1 + 1, y2 = y1 + 1;
            const int x2r = min(x2, P.w - 1), y2r = min(y2, P.h - 1);
            Px p00, p10, p01, p11;
            const uint8_t* ra = P.data + (size_t)y1 * (size_t)P.step;
            const uint8_t* rb = P.data + (size_t)y2r * (size_t)P.step;
            load_px(ra, r.depth, r.cn, x1, p00);
            load_px(ra, r.depth, r.cn, x2r, p10);
            load_px(rb, r.depth, r.cn, x1, p01);
            load_px(rb, r.depth, r.cn, x2r, p11);
            const float w00 = ((float)x2 - sx) * ((float)y2 - sy);
            const float w10 = (sx - (float)x1) * ((float)y2 - sy);
            const float w01 = ((float)x2 - sx) * (sy - (float)y1);
            const float w11 = (sx - (float)x1) * (sy - (float)y1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < r.cn) {
                    float acc = tap_f(p00.v[k], r.depth) * w00;
                    acc = acc + tap_f(p10.v[k], r.depth) * w10;
                    acc = acc + tap_f(p01.v[k], r.depth) * w01;
                    acc = acc + tap_f(p11.v[k], r.depth) * w11;
                    p.v[k] = acc;
                }
            }
        }
    }
    InterpProg::run(c.prog, p, depth, cn);
    const DstPlane* dst = c.write.table ? c.write.table : c.dst_inline;
    write_px(c.write, dst, x, y, z, p, depth, cn);
}

template <int NPL>
static hipError_t launch_warp_t(const ChainArgs& c, const WarpPlane* planes, int n, const WarpPlane* table, hipStream_t s) {
    WarpKernArgs<NPL> a;
    a.c = c;
    for (int i = 0; i < (NPL > 0 ? NPL : 1); ++i) a.planes[i] = (NPL > 0 && i < n) ? planes[i] : WarpPlane{};
    const dim3 block(64, 4, 1);
    const dim3 grid((c.read.dst_w + 63) / 64, (c.read.dst_h + 3) / 4, c.read.batch);
    hipLaunchKernelGGL(k_warp<NPL>, grid, block, 0, s, a, table);
    return hipGetLastError();
}

int launch_warp(const ChainArgs& c, const WarpPlane* planes, int n, const WarpPlane* dev_table, void* stream, bool dry_run,
                LaunchInfo* info) {
    const bool persp = c.read.kind == CVGS_READ_WARP_PERSPECTIVE;
    if (info) info->kernel = persp ? "warp_perspective_interp" : "warp_affine_interp";
    if (dry_run) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (dev_table) e = launch_warp_t<0>(c, nullptr, 0, dev_table, s);
    else if (n <= 4) e = launch_warp_t<4>(c, planes, n, nullptr, s);
    else e = launch_warp_t<kInlineWarp>(c, planes, n, nullptr, s);
    return e == hipSuccess ? 0 : -(int)e - 1000;
}

} // namespace cvgs
