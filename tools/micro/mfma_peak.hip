// Micro-benchmark: MFMA 32x32x16 f16 issue rate on this MI355X, bare and fed by ds_read_b128 (1 read per MFMA,
// the ratio of the conv kernel's 2x2 wave tile), for 1/2/4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(const f16x8* src, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += blockDim.x) ((f16x8*)smem)[i] = src[i];
    __syncthreads();
    f32x16 acc[4] = {};
    f16x8 a = src[lane], b = src[64 + lane];
    const int base = ((tid >> 6) * 2048 + (lane & 31) * 32 + ((lane >> 5) ^ ((lane >> 3) & 1)) * 16) & (64 * 1024 - 1);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                f16x8 a0 = *(const f16x8*)(smem + ((base + u * 4096) & 65535));
                f16x8 a1 = *(const f16x8*)(smem + ((base + u * 4096 + 1024) & 65535));
                f16x8 b0 = *(const f16x8*)(smem + ((base + u * 4096 + 2048) & 65535));
                f16x8 b1 = *(const f16x8*)(smem + ((base + u * 4096 + 3072) & 65535));
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[3], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[3], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 123.456f) out[tid] = s;
}

int main() {
    std::vector<_Float16> h(4096 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 2001 - 1000) / 1000.0f);
    f16x8* d; float* o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, 4096 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 256 * wps, blocks = 256;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, d, o, iters);
                else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, d, o, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 2) {
                    const double fl = (double)blocks * threads / 64 * iters * 36 * 32768.0;
                    printf("%s  %d waves/SIMD: %.1f TFLOP/s (%.3f ms)\n", mode ? "ds_read-fed" : "bare       ", wps, fl / ms / 1e9, ms);
                }
            }
        }
    return 0;
}
