// k_k1_c4.hip -- K1's planar-tensor instantiations for 4-channel sources (see k_k1_impl.hpp / k_k1.hip).
#include "k_k1_impl.hpp"

namespace cvgs {

hipError_t k1_launch_planar_c4(int src, bool f16, int prog_id, bool table, int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, int out_cn, LaunchCtx& s) {
    if (f16) return launch_prog<4, SRC_U8, _Float16>(prog_id, table, rpw, c, ip, ni, out_cn, s);
    return src == SRC_U8    ? launch_prog<4, SRC_U8>(prog_id, table, rpw, c, ip, ni, out_cn, s)
           : src == SRC_U16 ? launch_prog<4, SRC_U16>(prog_id, table, rpw, c, ip, ni, out_cn, s)
           : src == SRC_S16 ? launch_prog<4, SRC_S16>(prog_id, table, rpw, c, ip, ni, out_cn, s)
                            : launch_prog<4, SRC_F32>(prog_id, table, rpw, c, ip, ni, out_cn, s);
}

} // namespace cvgs
