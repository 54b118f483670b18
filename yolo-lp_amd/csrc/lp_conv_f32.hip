// Instantiates the implicit-GEMM conv kernels (lp_conv_kernel.inc) for f32 activations.
#include "lp_conv_kernel.inc"

namespace lp {
int conv_launch_f32(int cfg, int mode, int ksize, int stride, int nbuf, const ConvArgs& a, hipStream_t st) {
    return launch_dtype<float>(cfg, mode, ksize, stride, nbuf, a, st);
}
}  // namespace lp
