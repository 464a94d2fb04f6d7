// k_k1.hip -- K1, the north-star kernel: N crops (u8 C3/C4 pitched views of a frame) ->
// bilinear resize -> pointwise program -> planar fp32 (or, as the half-precision hand-off option, fp16) tensor
// (NCHW TensorSplit / CNHW TensorTSplit).
// Replaces the FKL instantiation
//   BatchRead<N,CONDITIONAL_WITH_DEFAULT>[Resize<LINEAR,AR,Read<PerThreadRead<_2D,uchar3>>>]
//     -> ColorConversion -> Mul -> Sub -> Div -> Write<TensorSplit<float3>>
// launched at reference tests/batchresize/test_batchresize_x_split3D.cu:311-314.
//
// CDNA4 mapping.  Measured (profiles/r02_a_*: round 2's A/B of experimental variants, since removed): a 50-crop launch is latency bound (an EMPTY 1600-workgroup
// launch already costs 1.76 us of the ~5 us), a 3200-crop launch is HBM bound (mixed read/write traffic at ~4.9 TB/s
// real); the VALU work hides under both.  Hence:
//  * lane = output column, wave = RPW consecutive output rows of one crop, workgroup = kK1Waves such waves, blockIdx.y = crop
//    (no integer division).
//    Column geometry (x1, the two x weights, the byte window) is computed once per lane and reused for every row;
//    row geometry is wave-uniform (SGPRs).
//  * every kernel-argument field and the crop's PlaneParams are fetched in ONE batch of scalar loads before the
//    first branch, so a wave pays one scalar-memory round trip, not one per early-exit test.
//  * each lane fetches BOTH horizontal taps of a source row with ONE unaligned 8-byte global load (a u8c3 pixel
//    pair is 6 bytes, u8c4 8 bytes) from a uniform row base + 32-bit lane offset: 2 loads per output pixel.  The
//    window is clamped into the crop row, so no byte outside the ROI is ever read; rows narrower than 8 bytes
//    gather their bytes one by one.
//  * planar stores are full 256-byte rows per wave and non-temporal: the output is written once and never re-read
//    by this kernel, so it should not wait in L2 for the end-of-kernel write-back.
//  * small launches use 1 row per wave (maximum parallelism, shortest critical path), large ones 4 (amortises the
//    column geometry).
#include "k_k1_impl.hpp"

namespace cvgs {

// the planar-tensor variants of 3- / 4-channel sources live in k_k1_c3.hip / k_k1_c4.hip (parallel compilation)
hipError_t k1_launch_planar_c3(int src, bool f16, int prog_id, bool table, int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, int out_cn, LaunchCtx& s);
hipError_t k1_launch_planar_c4(int src, bool f16, int prog_id, bool table, int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, int out_cn, LaunchCtx& s);

int launch_k1(const ChainArgs& c_in, const PlaneParams* inline_planes, int n_inline, LaunchCtx& ctx, bool dry_run, LaunchInfo* info, uint32_t chain_flags) {
    const MirrorArgs& mirrors = ctx.mirrors;
    const ManySeg* const segs = ctx.segs;
    const int n_segs = ctx.n_segs;
    void* const stream = ctx.stream;
    const ReadArgs& r = c_in.read;
    // eligibility: 8U / 16U / 16S C3/C4 resize read, fp32 planar tensor write -- or, for 8U sources, an fp16 planar
    // tensor whose conversion is the chain's LAST stage (the half-precision hand-off option)
    if (r.kind != CVGS_READ_RESIZE_LINEAR || r.cn < 1 || r.cn > 4) return 0;
    if (r.depth != CVGS_DEPTH_8U && r.depth != CVGS_DEPTH_16U && r.depth != CVGS_DEPTH_16S && r.depth != CVGS_DEPTH_32F) return 0;
    const bool few = r.cn < 3; // 1 / 2 channels: planar fp32, or packed fp32 / u8
    if (few && (c_in.write.kind == CVGS_WRITE_SPLIT_2D || c_in.write.depth == CVGS_DEPTH_16F)) return 0;
    const int wk = c_in.write.kind;
    const bool planar = wk == CVGS_WRITE_TENSOR_SPLIT || wk == CVGS_WRITE_TENSOR_T_SPLIT;
    const bool packed = wk == CVGS_WRITE_PIXEL_2D || wk == CVGS_WRITE_PIXEL_3D;
    const bool split2d = wk == CVGS_WRITE_SPLIT_2D;
    if (!planar && !packed && !split2d) return 0;
    if (r.batch > 65535) return 0;
    if ((mirrors.n > 0 || segs) && (!planar || c_in.write.data2)) return 0; // extra targets / fused chains: planar tensors only
    // mirrors: u8 C3/C4 -> fp32 planar with the planes in the kernel arguments (cfg #5: 64 crops per GPU); the rest is
    // the interpreted kernel's business
    if (mirrors.n > 0 && (r.depth != CVGS_DEPTH_8U || r.cn < 3 || r.table || segs || (c_in.write.depth != CVGS_DEPTH_32F && c_in.write.depth != CVGS_DEPTH_16F)))
        return 0;
    if (segs && (n_segs < 1 || n_segs > CVGS_MAX_CHAINS)) return 0;
    // fused chains without a table: segments AND planes in the kernel arguments (KernArgsManyInline; u8 sources, 3 / 4 channels)
    const bool inline_many = segs && !r.table;
    if (inline_many && (!inline_planes || n_inline < 1 || n_inline > kManyInlineLarge || r.depth != CVGS_DEPTH_8U || few)) return 0;
    // more than CVGS_KERNARG_PLANES descriptors in the kernel arguments: the 3- / 4-channel planar-tensor kernels only
    if (!r.table && !inline_many && n_inline > CVGS_KERNARG_PLANES && (n_inline > kKernargPlanesBig || !planar || few)) return 0;
    const bool f16 = c_in.write.depth == CVGS_DEPTH_16F;
    const bool u8out = c_in.write.depth == CVGS_DEPTH_8U;
    const bool i16out = c_in.write.depth == CVGS_DEPTH_16U || c_in.write.depth == CVGS_DEPTH_16S;
    if (!f16 && !u8out && !i16out && c_in.write.depth != CVGS_DEPTH_32F) return 0;
    if ((u8out || i16out) && !packed) return 0;
    if (i16out && r.depth != c_in.write.depth) return 0; // 16U -> 16U, 16S -> 16S
    if (split2d && c_in.write.depth != CVGS_DEPTH_32F) return 0;
    // non-planar targets of non-u8 sources: packed pixels of the source's own type, or (16-bit, 3 / 4 channels) separate fp32 planes
    const bool same_type_packed = packed && r.depth != CVGS_DEPTH_8U && c_in.write.depth == r.depth && r.cn != 2;
    const bool planes_16 = split2d && (r.depth == CVGS_DEPTH_16U || r.depth == CVGS_DEPTH_16S) && !few;
    if (!planar && r.depth != CVGS_DEPTH_8U && !same_type_packed && !planes_16) return 0;
    if (same_type_packed && (mirrors.n > 0 || segs)) return 0;
    int n_prog = c_in.prog.n;
    if (f16 || u8out || i16out) {
        if (((f16 || u8out) && r.depth != CVGS_DEPTH_8U) || n_prog < 1 || c_in.prog.opcode[n_prog - 1] != CVGS_OP_CAST) return 0;
        --n_prog; // the trailing CAST(CV_16F / CV_8U / CV_16U / CV_16S) happens in the store
    }
    for (int k = 0; k < n_prog; ++k) // value must stay fp32 through the program
        if (c_in.prog.opcode[k] == CVGS_OP_CAST || c_in.prog.opcode[k] == CVGS_OP_CAST_TRUNC) return 0;
    ChainArgs c_mut = c_in;
    c_mut.prog.n = n_prog;
    c_mut.prog.fast_div = 0;
    for (int k = 0; k < 4; ++k) c_mut.prog.rdiv[k] = 0.f;
    const ChainArgs& c = c_mut;

    // rows per wave: small launches are latency bound -> maximum parallelism (1 row per wave);
    // large ones amortise the column geometry over more rows (measured: profiles/r02_*).
    int64_t planes_total = r.batch;
    if (segs) {
        planes_total = 0;
        for (int i = 0; i < n_segs; ++i) planes_total += segs[i].batch;
    }
    const int64_t wave_rows = planes_total * r.dst_h * ((r.dst_w + 63) / 64);
    // round 2 sweep (profiles/r02_*): planar u8 output -- 1 row per wave up to 16 Ki wave-rows (100 crops of 64x128: 7.7 vs
    // 7.9 us; 200 crops and a 1080p whole-frame output tie or prefer 2), 2 beyond (4 ties at 3200
    // crops of ONE frame and loses 3-5 % when the crops come from many frames: 16 x 50 crops 38.3 vs 39.2 us); the packed /
    // separate-plane modes keep round 1's whole-frame tuning (4 rows from 64 Ki wave-rows)
    int rpw = wave_rows <= 16384 ? 1 : ((planar && wave_rows > 65536 && r.depth == CVGS_DEPTH_8U) || wave_rows <= 65536 ? 2 : 4);
    // separate pitched planes (the K2 chain): 4 rows per wave already from 16 Ki wave-rows (session-5 sweep: 4K -> 1080p into 3 planes
    // 14.9 -> 13.3 us, 6K -> 720p 8.9 -> 8.5 us; the packed modes tie or lose there)
    if (split2d && wave_rows > 16384) rpw = 4;
    static const char* rpw_env = getenv("CVGS_K1_RPW"); // tuning hook (benchmarks only): force 1 / 2 / 4 rows per wave
    if (rpw_env) rpw = atoi(rpw_env) >= 4 ? 4 : (atoi(rpw_env) == 2 ? 2 : 1);

    const bool table = r.table != nullptr;
    const int prog_id = k1_classify_program(c.prog, r.cn);
    if (prog_id < 2 && r.depth != CVGS_DEPTH_32F) fast_div_setup(c_mut.prog, prog_id == 0 ? 3 : 2, prog_id == 0 ? 1 : 0, r.cn, r.bg);
    // u8 C3 / C4 into a planar tensor: a chain of the canonical arithmetic shape ([swap] {mul|add|sub} x 0..2 [div] {mul|add|sub} x 0..2) is rewritten
    // into the straight-line K1CanonProg (k_taps.hpp); other programs run interpreted (arithmetic-only ones: InterpProgT<true>::run_arith, k_common.hpp)
    // (planar tensors of 1-4 channels and packed fp32 / fp16 / u8 pixels; not the separate-plane mode, not the mirrored launches)
    int planar_prog = prog_id;
    bool canon_packed = false;
    const bool canon_src = r.depth == CVGS_DEPTH_8U || (planar && !few); // (16-bit and fp32 sources: the 3- / 4-channel planar-tensor kernels)
    if (prog_id == 2 && n_prog > 0 && canon_src && mirrors.n == 0 && (planar || packed) && !same_type_packed) {
        ProgArgs canon;
        if (k1_canonicalise(c_mut.prog, r.cn, canon)) {
            c_mut.prog = canon;
            if (planar) planar_prog = 3;
            else canon_packed = true;
        }
    } else if (prog_id == 2 && n_prog == 0 && planar && !few && mirrors.n == 0) {
        ProgArgs canon; // the EMPTY program into a planar tensor (resize -> split): the canonical pipeline of four identities beats the interpreted kernel's one row per wave
        if (k1_canonicalise(c_mut.prog, r.cn, canon)) {
            c_mut.prog = canon;
            planar_prog = 3;
        }
    }
    if (prog_id == 2 && planar_prog == 2 && !canon_packed) interp_arith_setup(c_mut.prog, r.cn);

    const int src = r.depth == CVGS_DEPTH_8U ? SRC_U8 : (r.depth == CVGS_DEPTH_16U ? SRC_U16 : (r.depth == CVGS_DEPTH_16S ? SRC_S16 : SRC_F32));
    // whole-frame resize -> cast -> packed pixels of the SOURCE's type with nothing in between (the reference's
    // tests/resize/test_resize_write.cu chain): several output pixels per lane (k_k1_x4.hip) once the launch is in the
    // throughput regime.  CVGS_CHAIN_NO_THREAD_FUSION keeps the one-pixel kernel.
    if (packed && c.write.depth == r.depth && (u8out || same_type_packed) && n_prog == 0 && !table && !segs && mirrors.n == 0 &&
        !(chain_flags & CVGS_CHAIN_NO_THREAD_FUSION)) {
        const char* x4_env = getenv("CVGS_K1_X4"); // tuning / test hook: 0 = never, 1 = whenever eligible
        const bool force = x4_env && x4_env[0] == '1';
        if (force || (!x4_env && wave_rows >= kX4MinWaveRows)) {
            const int rc = launch_k1_packed_x4(c, inline_planes, n_inline, stream, dry_run, force);
            if (rc != 0) {
                static const char* names_x4[4][4] = {
                    {"k1_u8c1_packed_u8_x4", "k1_u8c2_packed_u8_x4", "k1_u8c3_packed_u8_x4", "k1_u8c4_packed_u8_x4"},
                    {"k1_u16c1_packed_u16_x2", "", "k1_u16c3_packed_u16_x2", "k1_u16c4_packed_u16_x2"},
                    {"k1_s16c1_packed_s16_x2", "", "k1_s16c3_packed_s16_x2", "k1_s16c4_packed_s16_x2"},
                    {"k1_f32c1_packed_f32_x4", "", "", ""}};
                if (info) info->kernel = names_x4[src][r.cn - 1];
                return rc;
            }
        }
    }
    if (info) {
        static const char* names[4][2][3] = {
            {{"k1_u8c3_swap_mul_sub_div", "k1_u8c3_mul_sub_div", "k1_u8c3_interp"},
             {"k1_u8c4_swap_mul_sub_div", "k1_u8c4_mul_sub_div", "k1_u8c4_interp"}},
            {{"k1_u16c3_swap_mul_sub_div", "k1_u16c3_mul_sub_div", "k1_u16c3_interp"},
             {"k1_u16c4_swap_mul_sub_div", "k1_u16c4_mul_sub_div", "k1_u16c4_interp"}},
            {{"k1_s16c3_swap_mul_sub_div", "k1_s16c3_mul_sub_div", "k1_s16c3_interp"},
             {"k1_s16c4_swap_mul_sub_div", "k1_s16c4_mul_sub_div", "k1_s16c4_interp"}},
            {{"k1_f32c3_swap_mul_sub_div", "k1_f32c3_mul_sub_div", "k1_f32c3_interp"},
             {"k1_f32c4_swap_mul_sub_div", "k1_f32c4_mul_sub_div", "k1_f32c4_interp"}}};
        static const char* names16[2][3] = {{"k1_u8c3_swap_mul_sub_div_f16", "k1_u8c3_mul_sub_div_f16", "k1_u8c3_interp_f16"},
                                            {"k1_u8c4_swap_mul_sub_div_f16", "k1_u8c4_mul_sub_div_f16", "k1_u8c4_interp_f16"}};
        static const char* names_other[2][4] = {{"k1_u8c3_packed_f32", "k1_u8c3_packed_f16", "k1_u8c3_packed_u8", "k1_u8c3_planes2d_f32"},
                                                {"k1_u8c4_packed_f32", "k1_u8c4_packed_f16", "k1_u8c4_packed_u8", "k1_u8c4_planes2d_f32"}};
        static const char* names_few[4][2][2] = {{{"k1_u8c1_mul_sub_div", "k1_u8c1_interp"}, {"k1_u8c2_mul_sub_div", "k1_u8c2_interp"}},
                                                 {{"k1_u16c1_mul_sub_div", "k1_u16c1_interp"}, {"k1_u16c2_mul_sub_div", "k1_u16c2_interp"}},
                                                 {{"k1_s16c1_mul_sub_div", "k1_s16c1_interp"}, {"k1_s16c2_mul_sub_div", "k1_s16c2_interp"}},
                                                 {{"k1_f32c1_mul_sub_div", "k1_f32c1_interp"}, {"k1_f32c2_mul_sub_div", "k1_f32c2_interp"}}};
        static const char* names_few_packed[2][2] = {{"k1_u8c1_packed_f32", "k1_u8c1_packed_u8"}, {"k1_u8c2_packed_f32", "k1_u8c2_packed_u8"}};
        static const char* names_same[4][4] = {{"", "", "", ""},
                                               {"k1_u16c1_packed_u16", "", "k1_u16c3_packed_u16", "k1_u16c4_packed_u16"},
                                               {"k1_s16c1_packed_s16", "", "k1_s16c3_packed_s16", "k1_s16c4_packed_s16"},
                                               {"k1_f32c1_packed_f32", "", "k1_f32c3_packed_f32", "k1_f32c4_packed_f32"}};
        static const char* names_planes16[2][2] = {{"k1_u16c3_planes2d_f32", "k1_u16c4_planes2d_f32"}, {"k1_s16c3_planes2d_f32", "k1_s16c4_planes2d_f32"}};
        if (same_type_packed) info->kernel = names_same[src][r.cn - 1];
        else if (planes_16) info->kernel = names_planes16[src == SRC_S16][r.cn == 4];
        else if (few && planar && planar_prog == 3) info->kernel = r.cn == 1 ? "k1_u8c1_arith" : "k1_u8c2_arith";
        else if (few) info->kernel = planar ? names_few[src][r.cn - 1][prog_id - 1] : (canon_packed ? (u8out ? (r.cn == 1 ? "k1_u8c1_packed_u8_arith" : "k1_u8c2_packed_u8_arith") : (r.cn == 1 ? "k1_u8c1_packed_f32_arith" : "k1_u8c2_packed_f32_arith")) : names_few_packed[r.cn - 1][u8out]);
        else if (planar && planar_prog == 3) {
            static const char* names_arith[4][2] = {{"k1_u8c3_arith", "k1_u8c4_arith"}, {"k1_u16c3_arith", "k1_u16c4_arith"}, {"k1_s16c3_arith", "k1_s16c4_arith"}, {"k1_f32c3_arith", "k1_f32c4_arith"}};
            info->kernel = f16 ? (r.cn == 4 ? "k1_u8c4_arith_f16" : "k1_u8c3_arith_f16") : names_arith[src][r.cn == 4];
        }
        else if (planar) info->kernel = f16 ? names16[r.cn == 4][prog_id] : names[src][r.cn == 4][prog_id];
        else if (canon_packed) {
            static const char* names_canon[2][3] = {{"k1_u8c3_packed_f32_arith", "k1_u8c3_packed_f16_arith", "k1_u8c3_packed_u8_arith"},
                                                    {"k1_u8c4_packed_f32_arith", "k1_u8c4_packed_f16_arith", "k1_u8c4_packed_u8_arith"}};
            info->kernel = names_canon[r.cn == 4][u8out ? 2 : (f16 ? 1 : 0)];
        } else info->kernel = names_other[r.cn == 4][split2d ? 3 : (u8out ? 2 : (f16 ? 1 : 0))];
    }
    if (dry_run) return 1;
    LaunchCtx& s = ctx;
    const int out_cn = c.write.cn;
    hipError_t e;
    if (mirrors.n > 0) {
        // one row per wave (a 64-crop launch is in the latency regime), planes in the kernel arguments
        // fp16 tensors (the half-precision hand-off) halve the bytes every xGMI link has to carry
        auto mir_t = [&](auto prog_tag, auto ot_tag) {
            using Pg = decltype(prog_tag);
            using OT = decltype(ot_tag);
            if (n_inline > CVGS_KERNARG_PLANES) // a shard of up to 320 crops per GPU: the 16 KB argument block
                return r.cn == 3 ? launch_t<3, kKernargPlanesBig, 1, Pg, SRC_U8, OT, WM_PLANAR, true>(c, inline_planes, n_inline, out_cn, s)
                                 : launch_t<4, kKernargPlanesBig, 1, Pg, SRC_U8, OT, WM_PLANAR, true>(c, inline_planes, n_inline, out_cn, s);
            return r.cn == 3 ? launch_t<3, CVGS_KERNARG_PLANES, 1, Pg, SRC_U8, OT, WM_PLANAR, true>(c, inline_planes, n_inline, out_cn, s)
                             : launch_t<4, CVGS_KERNARG_PLANES, 1, Pg, SRC_U8, OT, WM_PLANAR, true>(c, inline_planes, n_inline, out_cn, s);
        };
        auto mir = [&](auto prog_tag) { return f16 ? mir_t(prog_tag, _Float16{}) : mir_t(prog_tag, float{}); };
        e = prog_id == 0 ? mir(ProgSwapMulSubDiv{}) : (prog_id == 1 ? mir(ProgMulSubDiv{}) : mir(InterpProg{}));
    } else if (same_type_packed) {
        e = r.cn == 1 ? launch_same_type_packed<1>(src, table, rpw, c, inline_planes, n_inline, s)
            : r.cn == 3 ? launch_same_type_packed<3>(src, table, rpw, c, inline_planes, n_inline, s)
                        : launch_same_type_packed<4>(src, table, rpw, c, inline_planes, n_inline, s);
    } else if (planes_16) {
        e = r.cn == 3 ? launch_split2d_16<3>(src, table, rpw, c, inline_planes, n_inline, s)
                      : launch_split2d_16<4>(src, table, rpw, c, inline_planes, n_inline, s);
    } else if (few) {
        e = r.cn == 1 ? launch_few<1>(src, planar, u8out, planar_prog, table, rpw, c, inline_planes, n_inline, s, canon_packed)
                      : launch_few<2>(src, planar, u8out, planar_prog, table, rpw, c, inline_planes, n_inline, s, canon_packed);
    } else if (!planar) {
        if (split2d) e = r.cn == 3 ? launch_split2d<3>(prog_id, table, rpw, c, inline_planes, n_inline, s)
                                   : launch_split2d<4>(prog_id, table, rpw, c, inline_planes, n_inline, s);
        else if (u8out) e = r.cn == 3 ? launch_other_np<3, uint8_t, WM_PACKED>(n_prog == 0, table, rpw, c, inline_planes, n_inline, s, canon_packed)
                                      : launch_other_np<4, uint8_t, WM_PACKED>(n_prog == 0, table, rpw, c, inline_planes, n_inline, s, canon_packed);
        else if (f16) e = r.cn == 3 ? launch_other_np<3, _Float16, WM_PACKED>(false, table, rpw, c, inline_planes, n_inline, s, canon_packed)
                                    : launch_other_np<4, _Float16, WM_PACKED>(false, table, rpw, c, inline_planes, n_inline, s, canon_packed);
        else e = r.cn == 3 ? launch_other_np<3, float, WM_PACKED>(n_prog == 0, table, rpw, c, inline_planes, n_inline, s, canon_packed)
                           : launch_other_np<4, float, WM_PACKED>(n_prog == 0, table, rpw, c, inline_planes, n_inline, s, canon_packed);
    } else if (r.cn == 3) {
        e = k1_launch_planar_c3(src, f16, planar_prog, table, rpw, c, inline_planes, n_inline, out_cn, s);
    } else {
        e = k1_launch_planar_c4(src, f16, planar_prog, table, rpw, c, inline_planes, n_inline, out_cn, s);
    }
    return e == hipSuccess ? 1 : -(int)e - 1000;
}

} // namespace cvgs
