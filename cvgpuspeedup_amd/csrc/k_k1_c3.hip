// k_k1_c3.hip -- K1's planar-tensor instantiations for 3-channel sources (see k_k1_impl.hpp / k_k1.hip).
#include "k_k1_impl.hpp"

namespace cvgs {

hipError_t k1_launch_planar_c3(int src, bool f16, int prog_id, bool table, int rpw, const ChainArgs& c, const PlaneParams* ip, int ni, int out_cn, LaunchCtx& s) {
    if (f16) return launch_prog<3, SRC_U8, _Float16>(prog_id, table, rpw, c, ip, ni, out_cn, s);
    return src == SRC_U8    ? launch_prog<3, SRC_U8>(prog_id, table, rpw, c, ip, ni, out_cn, s)
           : src == SRC_U16 ? launch_prog<3, SRC_U16>(prog_id, table, rpw, c, ip, ni, out_cn, s)
           : src == SRC_S16 ? launch_prog<3, SRC_S16>(prog_id, table, rpw, c, ip, ni, out_cn, s)
                            : launch_prog<3, SRC_F32>(prog_id, table, rpw, c, ip, ni, out_cn, s);
}

} // namespace cvgs

#if (CVGS_K1_ABLATE & 32)
// probe only (tools/probes/xcd_worklist_probe.py; never in the product build): the work lists of the NEXT launch
extern "C" void cvgs_probe_set_worklist(const uint32_t* base, uint32_t slots) {
    cvgs::probe_worklist().base = base;
    cvgs::probe_worklist().slots = slots;
}
#endif
