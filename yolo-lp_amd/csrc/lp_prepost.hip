// Callers either side of the hot path ("next rows"):
//   lp_preprocess_letterbox -- Inferer.precess_image (yolov6/core/inferer.py:191-201) + letterbox
//                              (yolov6/data/data_augment.py:30-61): uint8 HWC BGR frame -> resized (bilinear), padded
//                              (114), RGB, CHW, /255 tensor of the engine's input dtype, in one kernel.
//   lp_rescale_round        -- Inferer.rescale (inferer.py:203-228) followed by .round() (:100) on the 12 coordinates
//                              of every detection row.
// The bilinear resize is the fixed-point scheme of OpenCV's INTER_LINEAR for 8-bit images (11-bit coefficients,
// horizontal then vertical pass, the (>>4, >>16, +2, >>2) rounding of VResizeLinear) -- restated from the published
// algorithm; OpenCV is not installed in this image, so against cv2 itself this is "parity unpinned".  The host
// mirror (yolov6/data/data_augment.py) implements the same integer arithmetic in numpy and the two are bit-exact.
#include "lp_internal.h"

namespace lp {

__device__ __forceinline__ void resize_coef(int d, double scale, int src, int* s0, int* a0, int* a1) {
    // cv::resize: fx = (dx + 0.5) * scale - 0.5 ; sx = floor(fx) ; fx -= sx ; clamps at the borders
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
    *s0 = s;
    const float c0 = (1.f - f) * 2048.f, c1 = f * 2048.f;
    *a0 = (int)rintf(c0);   // saturate_cast<short>(cvRound(v * INTER_RESIZE_COEF_SCALE))
    *a1 = (int)rintf(c1);
}

template <typename TO>
__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* __restrict__ img, int h0, int w0, TO* __restrict__ out,
                                                        int H, int W, int rh, int rw, int top, int left, double sy, double sx,
                                                        int resize) {
    const long long total = (long long)H * W;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        const int ry = y - top, rx = x - left;
        int bgr[3] = {114, 114, 114};
        if (ry >= 0 && ry < rh && rx >= 0 && rx < rw) {
            if (!resize) {
                const unsigned char* p = img + ((long long)ry * w0 + rx) * 3;
                bgr[0] = p[0]; bgr[1] = p[1]; bgr[2] = p[2];
            } else {
                int y0, b0, b1, x0, a0, a1;
                resize_coef(ry, sy, h0, &y0, &b0, &b1);
                resize_coef(rx, sx, w0, &x0, &a0, &a1);
                const int y1 = y0 + 1 < h0 ? y0 + 1 : h0 - 1, x1 = x0 + 1 < w0 ? x0 + 1 : w0 - 1;
                const unsigned char* r0 = img + (long long)y0 * w0 * 3;
                const unsigned char* r1 = img + (long long)y1 * w0 * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int h0v = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;   // HResizeLinear (scaled by 2048)
                    const int h1v = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
                    bgr[c] = (((b0 * (h0v >> 4)) >> 16) + ((b1 * (h1v >> 4)) >> 16) + 2) >> 2;   // VResizeLinear
                }
            }
        }
        // HWC BGR -> CHW RGB, uint8 -> float / 255 (inferer.py:195-199); the division is done in the output dtype's
        // arithmetic like `image.half(); image /= 255` does (fp16 / bf16 path: one rounding of the quotient)
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(long long)c * total + i] = (TO)((float)bgr[2 - c] / 255.f);
    }
}

__global__ __launch_bounds__(256) void rescale_round_kernel(float* __restrict__ det, int n, float ratio, float padx, float pady,
                                                           float wmax, float hmax) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 12) return;
    const int row = i / 12, c = i - row * 12;
    float* p = det + (long long)row * LP_DET_COLS + c;
    float v = *p;
    v = v - ((c & 1) ? pady : padx);
    v = v / ratio;
    const float hi = (c & 1) ? hmax : wmax;
    v = v < 0.f ? 0.f : v;
    v = v > hi ? hi : v;
    *p = rintf(v);   // torch.round: half to even
}

}  // namespace lp

using namespace lp;

extern "C" int lp_preprocess_letterbox(const unsigned char* img, int h0, int w0, void* out, int out_dtype, int H, int W, int rh,
                                       int rw, int top, int left, void* stream) {
    if (!img || !out || h0 < 1 || w0 < 1 || H < 1 || W < 1 || rh < 1 || rw < 1 || top < 0 || left < 0 || top + rh > H ||
        left + rw > W)
        return fail(LP_ERR_ARG, "lp_preprocess_letterbox: bad geometry");
    const long long total = (long long)H * W;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const int resize = !(rh == h0 && rw == w0);
    const double sy = (double)h0 / rh, sx = (double)w0 / rw;
    hipStream_t st = (hipStream_t)stream;
    switch (out_dtype) {
        case LP_F16: hipLaunchKernelGGL(preprocess_kernel<f16>, dim3((unsigned)blocks), dim3(256), 0, st, img, h0, w0, (f16*)out, H, W, rh, rw, top, left, sy, sx, resize); break;
        case LP_BF16: hipLaunchKernelGGL(preprocess_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, st, img, h0, w0, (bf16*)out, H, W, rh, rw, top, left, sy, sx, resize); break;
        case LP_F32: hipLaunchKernelGGL(preprocess_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, img, h0, w0, (float*)out, H, W, rh, rw, top, left, sy, sx, resize); break;
        default: return fail(LP_ERR_ARG, "lp_preprocess_letterbox: dtype");
    }
    LP_HIP_CHECK(hipGetLastError());
    return LP_OK;
}

extern "C" int lp_rescale_round(float* det, int n, double ratio, double padx, double pady, int img_w, int img_h, void* stream) {
    if (n == 0) return LP_OK;
    if (!det || n < 0 || !(ratio > 0.0)) return fail(LP_ERR_ARG, "lp_rescale_round: bad argument");
    hipLaunchKernelGGL(rescale_round_kernel, dim3((unsigned)((n * 12 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, det, n,
                       (float)ratio, (float)padx, (float)pady, (float)img_w, (float)img_h);
    LP_HIP_CHECK(hipGetLastError());
    return LP_OK;
}
