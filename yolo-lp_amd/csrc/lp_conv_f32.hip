// Instantiates the implicit-GEMM conv kernels (lp_conv_kernel.inc) for f32 activations.
#include "lp_conv_kernel.inc"
#include "lp_conv1x1_stream.inc"
#include "lp_head_rows.inc"
#include "lp_head_box.inc"

namespace lp {
int conv_launch_f32(int cfg, int mode, int ksize, int stride, int nbuf, const ConvArgs& a, hipStream_t st) {
    return launch_dtype<float>(cfg, mode, ksize, stride, nbuf, a, st);
}
int conv_stream_launch_f32(int wc, const ConvArgs& a, int cb_pack, int lds, hipStream_t st) {
    return stream_launch_dtype<float>(wc, a, cb_pack, lds, st);
}
int head_rows_launch_f32(const ConvArgs& a, int cb_pack, hipStream_t st) { return head_rows_launch_dtype<float>(a, cb_pack, st); }
int head_box_det_launch_f32(const ConvArgs& a, int cb_pack, hipStream_t st) { return head_box_det_launch_dtype<float>(a, cb_pack, st); }

// Every pair of neighbouring fp32 values a < b (all 2^32 - 1 of them, NaNs skipped): is sigmoid_fast(a) <= sigmoid_fast(b)?
// The detections-only head relies on it (lp_head_rows.inc: largest sigmoid of a head = sigmoid of its largest logit).
__global__ __launch_bounds__(256) void sigmoid_monotone_kernel(unsigned long long* __restrict__ violations) {
    auto value_of = [](unsigned key) {                 // key -> the key-th smallest fp32 bit pattern (negative NaNs first, -inf, ..., -0, +0, ..., +inf, NaNs)
        return __uint_as_float((key & 0x80000000u) ? key ^ 0x80000000u : ~key);
    };
    unsigned long long bad = 0;
    for (unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x; k < 0xffffffffull; k += (unsigned long long)gridDim.x * 256) {
        const float x0 = value_of((unsigned)k), x1 = value_of((unsigned)k + 1u);
        if (x0 != x0 || x1 != x1) continue;
        if (!(sigmoid_fast(x0) <= sigmoid_fast(x1))) ++bad;
    }
    if (bad) atomicAdd(violations, bad);
}
}  // namespace lp

extern "C" int lp_check_sigmoid_monotone(unsigned long long* dev_violations, void* stream) {
    if (!dev_violations) return lp::fail(LP_ERR_ARG, "lp_check_sigmoid_monotone: null pointer");
    hipStream_t st = (hipStream_t)stream;
    LP_HIP_CHECK(hipMemsetAsync(dev_violations, 0, 8, st));
    hipLaunchKernelGGL(lp::sigmoid_monotone_kernel, dim3(256 * 64), dim3(256), 0, st, dev_violations);
    LP_HIP_CHECK(hipGetLastError());
    return LP_OK;
}
