// k_k1.hip -- K1, the north-star kernel: N crops (u8 C3/C4 pitched views of a frame) ->
// bilinear resize -> pointwise program -> planar fp32 tensor (NCHW TensorSplit / CNHW TensorTSplit).
// Replaces the FKL instantiation
//   BatchRead<N,CONDITIONAL_WITH_DEFAULT>[Resize<LINEAR,AR,Read<PerThreadRead<_2D,uchar3>>>]
//     -> ColorConversion -> Mul -> Sub -> Div -> Write<TensorSplit<float3>>
// launched at reference tests/batchresize/test_batchresize_x_split3D.cu:311-314.
//
// CDNA4 mapping (this path is bandwidth/latency bound: no MFMA, no LDS round trip needed):
//  * lane = output column, wave = RPW consecutive output rows of one crop.  Everything that depends
//    only on the column (x1, the two x weights, the byte window) is computed once per lane and
//    reused for every row; everything that depends only on the row (y1, y weights, the two source
//    row pointers) is wave-uniform and lives in SGPRs.
//  * each lane fetches BOTH horizontal taps of a source row with ONE unaligned 8-byte load (a
//    u8c3 pixel pair is 6 bytes, a u8c4 pair 8): 2 loads per output pixel instead of 12-16 byte
//    loads.  The window is clamped to the crop row, so no byte outside the ROI is ever read.
//  * the three/four planar stores of a wave are full 256-byte rows.
//  * workgroup ids are remapped so the tiles of one crop run on one XCD and share its L2.
#include <initializer_list>
#include <type_traits>

#include "k_common.hpp"

namespace cvgs {

// the two chains the reference's tests spell, as compile-time programs; anything else is interpreted
using ProgReorderMulSubDiv = StaticProg<CVGS_OP_REORDER, CVGS_OP_MUL, CVGS_OP_SUB, CVGS_OP_DIV>;
using ProgMulSubDiv = StaticProg<CVGS_OP_MUL, CVGS_OP_SUB, CVGS_OP_DIV>;

struct K1Geom {
    uint32_t col_tiles;       // ceil(dst_w / 64)
    uint32_t tiles_per_plane; // col_tiles * row_tiles
    uint32_t total_tiles;     // tiles_per_plane * batch
    uint32_t padded_tiles;    // total rounded up to a multiple of 8 (XCD remap is a bijection on it)
    int64_t img_stride;       // output elements between images
    int64_t ch_stride;        // output elements between channel planes
};

__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8); // gfx950 global loads need no alignment: one global_load_dwordx2
    return v;
}

// crops narrower than 8 bytes per row (1-2 pixels): gather the pair byte by byte, never past the row
__device__ __forceinline__ uint64_t load_pair_bytes(const uint8_t* row, int o, int n, int row_bytes) {
    uint64_t v = 0;
    for (int k = 0; k < n; ++k)
        if (o + k < row_bytes) v |= (uint64_t)row[o + k] << (8 * k);
    return v;
}

template <int CN>
__device__ __forceinline__ void unpack_pair(uint64_t v, bool edge, float* a, float* b) {
    const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    a[0] = (float)(lo & 0xffu);
    a[1] = (float)((lo >> 8) & 0xffu);
    a[2] = (float)((lo >> 16) & 0xffu);
    if constexpr (CN == 3) {
        b[0] = (float)(lo >> 24);
        b[1] = (float)(hi & 0xffu);
        b[2] = (float)((hi >> 8) & 0xffu);
    } else {
        a[3] = (float)(lo >> 24);
        b[0] = (float)(hi & 0xffu);
        b[1] = (float)((hi >> 8) & 0xffu);
        b[2] = (float)((hi >> 16) & 0xffu);
        b[3] = (float)(hi >> 24);
    }
#pragma unroll
    for (int c = 0; c < CN; ++c) b[c] = edge ? a[c] : b[c];
}

template <int CN, int NPL, int RPW, class Prog>
__global__ __launch_bounds__(256) void k1_direct(const KernArgs<NPL> a, const K1Geom g) {
    const ChainArgs& c = a.c;
    const ReadArgs& r = c.read;
    const uint32_t bid = xcd_remap(blockIdx.x, g.padded_tiles);
    if (bid >= g.total_tiles) return;
    const int z = (int)(bid / g.tiles_per_plane);
    const uint32_t t = bid - (uint32_t)z * g.tiles_per_plane;
    const int col_tile = (int)(t % g.col_tiles);
    const int row_tile = (int)(t / g.col_tiles);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int x = col_tile * 64 + lane;
    const int row0 = (row_tile * 4 + wave) * RPW;
    if (row0 >= r.dst_h) return;
    const bool x_ok = x < r.dst_w;

    float* const out = (float*)c.write.data + (int64_t)z * g.img_stride + x;
    const int W = c.write.width;

    // background value pushed through the whole chain: planes >= usedPlanes and AR padding
    Px bgp;
    int bdepth = CVGS_DEPTH_32F, bcn = CN;
#pragma unroll
    for (int k = 0; k < 4; ++k) bgp.v[k] = r.bg[k];
    Prog::run(c.prog, bgp, bdepth, bcn);

    if (z >= r.used) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int y = row0 + j;
            if (y < r.dst_h && x_ok) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < bcn) out[(int64_t)k * g.ch_stride + (int64_t)y * W] = bgp.v[k];
            }
        }
        return;
    }

    PlaneParams P;
    if constexpr (NPL == 0) P = r.table[z];
    else P = a.planes[z];

    // ---- per-lane column geometry (reused for every row) ----
    const bool in_x = x >= P.x1 && x <= P.x2;
    const int xr = in_x ? x - P.x1 : 0;
    const float sx = (float)xr * P.fx;
    const int x1 = (int)floorf(sx);
    const int x2 = x1 + 1;
    const float wxa = (float)x2 - sx;
    const float wxb = sx - (float)x1;
    const bool edge = x2 > P.w - 1;
    const int row_bytes = P.w * CN;
    const int o = x1 * CN;
    const bool tiny = row_bytes < 8; // wave-uniform
    const int ol = tiny ? o : min(o, row_bytes - 8);
    const int sh = (o - ol) * 8;

    uint64_t va[RPW], vb[RPW];
    float wya[RPW], wyb[RPW];
    bool in_y[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = min(row0 + j, r.dst_h - 1);
        in_y[j] = y >= P.y1 && y <= P.y2;
        const int yr = in_y[j] ? y - P.y1 : 0;
        const float sy = (float)yr * P.fy;
        const int y1 = (int)floorf(sy);
        const int y2 = y1 + 1;
        const int y2r = min(y2, P.h - 1);
        wya[j] = (float)y2 - sy;
        wyb[j] = sy - (float)y1;
        const uint8_t* ra = P.data + (size_t)y1 * (size_t)P.step;
        const uint8_t* rb = P.data + (size_t)y2r * (size_t)P.step;
        if (!tiny) {
            va[j] = load_u64_unaligned(ra + ol);
            vb[j] = load_u64_unaligned(rb + ol);
        } else {
            va[j] = load_pair_bytes(ra, o, 2 * CN, row_bytes);
            vb[j] = load_pair_bytes(rb, o, 2 * CN, row_bytes);
        }
    }

#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = row0 + j;
        float p00[4], p10[4], p01[4], p11[4];
        unpack_pair<CN>(va[j] >> sh, edge, p00, p10);
        unpack_pair<CN>(vb[j] >> sh, edge, p01, p11);
        const float w00 = wxa * wya[j];
        const float w10 = wxb * wya[j];
        const float w01 = wxa * wyb[j];
        const float w11 = wxb * wyb[j];
        Px p;
        p.v[3] = 0.f;
#pragma unroll
        for (int k = 0; k < CN; ++k) {
            float acc = p00[k] * w00;
            acc = acc + p10[k] * w10;
            acc = acc + p01[k] * w01;
            acc = acc + p11[k] * w11;
            p.v[k] = acc;
        }
        int depth = CVGS_DEPTH_32F, cn = CN;
        Prog::run(c.prog, p, depth, cn);
        const bool inside = in_x && in_y[j];
        if (y < r.dst_h && x_ok) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < cn) out[(int64_t)k * g.ch_stride + (int64_t)y * W] = inside ? p.v[k] : bgp.v[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
static K1Geom make_geom(const ChainArgs& c, int rows_per_wg, int out_cn) {
    K1Geom g;
    g.col_tiles = (uint32_t)((c.read.dst_w + 63) / 64);
    const uint32_t row_tiles = (uint32_t)((c.read.dst_h + rows_per_wg - 1) / rows_per_wg);
    g.tiles_per_plane = g.col_tiles * row_tiles;
    g.total_tiles = g.tiles_per_plane * (uint32_t)c.read.batch;
    g.padded_tiles = (g.total_tiles + 7u) / 8u * 8u;
    const int64_t plane = (int64_t)c.write.width * c.write.height;
    if (c.write.kind == CVGS_WRITE_TENSOR_SPLIT) {
        g.img_stride = plane * out_cn;
        g.ch_stride = plane;
    } else {
        g.img_stride = plane;
        g.ch_stride = plane * c.write.planes;
    }
    return g;
}

template <int CN, int NPL, int RPW, class Prog>
static hipError_t launch_t(const ChainArgs& c, const PlaneParams* inline_planes, int n_inline, int out_cn,
                           hipStream_t stream) {
    KernArgs<NPL> a;
    a.c = c;
    if constexpr (NPL > 0) {
        for (int i = 0; i < n_inline; ++i) a.planes[i] = inline_planes[i];
        for (int i = n_inline; i < NPL; ++i) a.planes[i] = PlaneParams{};
    } else {
        a.planes[0] = PlaneParams{};
    }
    const K1Geom g = make_geom(c, 4 * RPW, out_cn);
    hipLaunchKernelGGL((k1_direct<CN, NPL, RPW, Prog>), dim3(g.padded_tiles), dim3(256), 0, stream, a, g);
    return hipGetLastError();
}

template <int CN, int NPL, class Prog>
static hipError_t launch_rpw(int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, int out_cn, hipStream_t s) {
    // the interpreted program keeps its opcode loop rolled; more than one row per wave only bloats it
    if constexpr (std::is_same_v<Prog, InterpProg>) return launch_t<CN, NPL, 1, Prog>(c, ip, ni, out_cn, s);
    else
    switch (rpw) {
    case 1: return launch_t<CN, NPL, 1, Prog>(c, ip, ni, out_cn, s);
    case 2: return launch_t<CN, NPL, 2, Prog>(c, ip, ni, out_cn, s);
    default: return launch_t<CN, NPL, 4, Prog>(c, ip, ni, out_cn, s);
    }
}

template <int CN, class Prog>
static hipError_t launch_npl(bool table, int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, int out_cn,
                             hipStream_t s) {
    if (table) return launch_rpw<CN, 0, Prog>(rpw, c, ip, ni, out_cn, s);
    return launch_rpw<CN, CVGS_KERNARG_PLANES, Prog>(rpw, c, ip, ni, out_cn, s);
}

static bool prog_is(const ProgArgs& p, std::initializer_list<int> ops) {
    if (p.n != (int)ops.size()) return false;
    int k = 0;
    for (int o : ops)
        if (p.opcode[k++] != o) return false;
    return true;
}

int launch_k1(const ChainArgs& c, const PlaneParams* inline_planes, int n_inline, uint32_t chain_flags,
              void* stream, bool dry_run, LaunchInfo* info) {
    const ReadArgs& r = c.read;
    // eligibility: u8 C3/C4 resize read, fp32 planar tensor write
    if (r.kind != CVGS_READ_RESIZE_LINEAR || r.depth != CVGS_DEPTH_8U || (r.cn != 3 && r.cn != 4)) return 0;
    if (c.write.kind != CVGS_WRITE_TENSOR_SPLIT && c.write.kind != CVGS_WRITE_TENSOR_T_SPLIT) return 0;
    if (c.write.depth != CVGS_DEPTH_32F) return 0;
    for (int k = 0; k < c.prog.n; ++k) // value must stay fp32 through the program
        if (c.prog.opcode[k] == CVGS_OP_CAST) return 0;

    // rows per wave: small launches are latency bound -> maximum parallelism (1 row per wave);
    // large ones amortise the column geometry and the hoisted division set-up over more rows.
    const int64_t wave_rows = (int64_t)r.batch * r.dst_h * ((r.dst_w + 63) / 64);
    const int rpw = wave_rows <= 8192 ? 1 : (wave_rows <= 32768 ? 2 : 4);

    const bool table = r.table != nullptr;
    int prog_id = 2;
    if (prog_is(c.prog, {CVGS_OP_REORDER, CVGS_OP_MUL, CVGS_OP_SUB, CVGS_OP_DIV})) prog_id = 0;
    else if (prog_is(c.prog, {CVGS_OP_MUL, CVGS_OP_SUB, CVGS_OP_DIV})) prog_id = 1;

    if (info) {
        static const char* names[2][3] = {{"k1_u8c3_direct_reorder_mul_sub_div", "k1_u8c3_direct_mul_sub_div",
                                           "k1_u8c3_direct_interp"},
                                          {"k1_u8c4_direct_reorder_mul_sub_div", "k1_u8c4_direct_mul_sub_div",
                                           "k1_u8c4_direct_interp"}};
        info->kernel = names[r.cn == 4][prog_id];
    }
    if (dry_run) return 1;
    (void)chain_flags;

    hipStream_t s = (hipStream_t)stream;
    const int out_cn = c.write.cn;
    hipError_t e;
    if (r.cn == 3) {
        if (prog_id == 0) e = launch_npl<3, ProgReorderMulSubDiv>(table, rpw, c, inline_planes, n_inline, out_cn, s);
        else if (prog_id == 1) e = launch_npl<3, ProgMulSubDiv>(table, rpw, c, inline_planes, n_inline, out_cn, s);
        else e = launch_npl<3, InterpProg>(table, rpw, c, inline_planes, n_inline, out_cn, s);
    } else {
        if (prog_id == 0) e = launch_npl<4, ProgReorderMulSubDiv>(table, rpw, c, inline_planes, n_inline, out_cn, s);
        else if (prog_id == 1) e = launch_npl<4, ProgMulSubDiv>(table, rpw, c, inline_planes, n_inline, out_cn, s);
        else e = launch_npl<4, InterpProg>(table, rpw, c, inline_planes, n_inline, out_cn, s);
    }
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

} // namespace cvgs
