// What does s_memtime (clock64) count, and what is the shader clock under an MFMA load?
// Kernel A spins on s_memtime for a fixed number of ticks (light load); kernel B runs a fixed number of dependent MFMAs per
// wave on every SIMD (heavy load) and reports ticks per MFMA.  Wall time comes from hipEvents.
//   hipcc -O3 --offload-arch=gfx950 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void spin_kernel(long long ticks, long long* out) {
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    while (clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = clock64() - t0; out[1] = wall_clock64() - w0; }
}
__global__ __launch_bounds__(256) void mfma_kernel(int iters, long long* out, float* sink) {
    f16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(threadIdx.x * 0.001f); b[k] = (_Float16)(k * 0.01f); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) *sink = 1.f;
}
int main() {
    long long* out; float* sink; long long h[2];
    CK(hipMalloc(&out, 16)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, 0, 20000000LL, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        printf("spin (1 wave):    %lld clock64 ticks, %lld wall_clock64 ticks in %.3f ms -> clock64 %.1f MHz, wall_clock64 %.1f MHz\n", h[0], h[1], ms, h[0] / ms / 1e3, h[1] / ms / 1e3);
    }
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = 200000;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_kernel, dim3(256 * 2), dim3(256), 0, 0, iters, out, sink);      // 8 waves per CU, 2 per SIMD
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
        const double flops = 512.0 * 4 * iters * 4.0 * 32768.0;
        printf("MFMA (all SIMDs): %lld clock64 ticks, %lld wall ticks in %.3f ms -> clock64 %.1f MHz; %.1f ticks per MFMA per SIMD; %.0f TFLOP/s\n", h[0], h[1], ms,
               h[0] / ms / 1e3, (double)h[0] / (iters * 4.0 * 2.0), flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
