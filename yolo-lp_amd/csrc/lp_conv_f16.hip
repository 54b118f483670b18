// Instantiates the implicit-GEMM conv kernels (lp_conv_kernel.inc) for f16 activations.
#include "lp_conv_kernel.inc"
#include "lp_conv1x1_stream.inc"
#include "lp_head_rows.inc"
#include "lp_head_box.inc"
#include "lp_conv3x3_pipe.inc"
#include "lp_conv3x3_pipe16.inc"
#include "lp_conv3x3_pipe16v.inc"
#include "lp_conv3x3_s2p16.inc"
#include "lp_stem_planar.inc"
#include "lp_stem2_fused.inc"
#include "lp_pw_s2_fused.inc"
#include "lp_bifusion_fused.inc"

namespace lp {
int conv_launch_f16(int cfg, int mode, int ksize, int stride, int nbuf, const ConvArgs& a, hipStream_t st) {
    return launch_dtype<f16>(cfg, mode, ksize, stride, nbuf, a, st);
}
int conv_stream_launch_f16(int wc, const ConvArgs& a, int cb_pack, int lds, hipStream_t st) {
    return stream_launch_dtype<f16>(wc, a, cb_pack, lds, st);
}
int head_rows_launch_f16(const ConvArgs& a, int cb_pack, hipStream_t st) { return head_rows_launch_dtype<f16>(a, cb_pack, st); }
int head_box_det_launch_f16(const ConvArgs& a, int cb_pack, hipStream_t st) { return head_box_det_launch_dtype<f16>(a, cb_pack, st); }
int conv_pipe_launch_f16(int pcfg, const ConvArgs& a, int ncu, hipStream_t st) {
    if (pcfg == PIPE_FUSED2) return stem2_fused_launch<f16>(a, ncu, st);
    if (pcfg == PIPE_FUSED_PW) return pw_s2_fused_launch<f16>(a, ncu, st);
    if (pcfg == PIPE_FUSED_BF) return bifusion_launch_dtype<f16>(a, ncu, st);
    if (pipe_is_16s2(pcfg)) return s2p16_launch_dtype<f16>(pcfg, a, ncu, st);
    if (pipe_is_16v(pcfg)) return pipe16v_launch_dtype<f16>(pcfg, a, ncu, st);
    if (pipe_is_16(pcfg)) return pipe16_launch_dtype<f16>(pcfg, a, ncu, st);
    return pcfg == PIPE_P ? stem_planar_launch<f16>(a, ncu, st) : pipe_launch_dtype<f16>(pcfg, a, ncu, st);
}
}  // namespace lp
