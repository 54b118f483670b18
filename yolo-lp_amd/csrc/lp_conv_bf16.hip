// Instantiates the implicit-GEMM conv kernels (lp_conv_kernel.inc) for bf16 activations.
#include "lp_conv_kernel.inc"

namespace lp {
int conv_launch_bf16(int cfg, int mode, int ksize, int stride, int nbuf, const ConvArgs& a, hipStream_t st) {
    return launch_dtype<bf16>(cfg, mode, ksize, stride, nbuf, a, st);
}
}  // namespace lp
