// Micro-benchmark: how fast can a wave feed v_mfma_f32_32x32x16_f16 from LDS with ds_read_b128?
// Variants: wave tile (WC x WP fragments), serialized vs register-double-buffered reads, linear (conflict-free by
// construction) vs the conv kernel's swizzled 32-byte-row addressing.  256 workgroups, 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int WC, int WP, int PIPE, int SWZ>
__global__ __launch_bounds__(1024) void k(const f16x8* src, float* out, int iters) {
    __shared__ __attribute__((aligned(1024))) char smem[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) ((f16x8*)smem)[i] = src[i];
    __syncthreads();
    f32x16 acc[WC][WP] = {};
    // per-lane base: linear = lane*16 (one KiB per wave-read); swizzled = conv kernel rows of 32 B
    const int r = lane & 31, h = lane >> 5;
    const int lbase = SWZ ? r * 32 + ((h ^ ((r >> 3) & 1)) * 16) : lane * 16;
    const int wbase = (wave * 4096) & 65535;
    auto rd = [&](int step, int f) { return *(const f16x8*)(smem + ((wbase + step * 8192 + f * (SWZ ? 1024 : 1024) + lbase) & 65535)); };
    for (int it = 0; it < iters; ++it) {
        if (PIPE) {
            f16x8 fa[2][WC], fb[2][WP];
#pragma unroll
            for (int c = 0; c < WC; ++c) fa[0][c] = rd(0, c);
#pragma unroll
            for (int p = 0; p < WP; ++p) fb[0][p] = rd(0, WC + p);
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                if (s + 1 < 9) {
#pragma unroll
                    for (int c = 0; c < WC; ++c) fa[(s + 1) & 1][c] = rd(s + 1, c);
#pragma unroll
                    for (int p = 0; p < WP; ++p) fb[(s + 1) & 1][p] = rd(s + 1, WC + p);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < WC; ++c)
#pragma unroll
                    for (int p = 0; p < WP; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s & 1][c], fb[s & 1][p], acc[c][p], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                f16x8 fa[WC], fb[WP];
#pragma unroll
                for (int c = 0; c < WC; ++c) fa[c] = rd(s, c);
#pragma unroll
                for (int p = 0; p < WP; ++p) fb[p] = rd(s, WC + p);
#pragma unroll
                for (int c = 0; c < WC; ++c)
#pragma unroll
                    for (int p = 0; p < WP; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[c], fb[p], acc[c][p], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int c = 0; c < WC; ++c) for (int p = 0; p < WP; ++p) for (int j = 0; j < 16; ++j) s += acc[c][p][j];
    if (s == 123.456f) out[tid] = s;
}

template <int WC, int WP, int PIPE, int SWZ>
void run(const f16x8* d, float* o, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        if (WC * WP * 16 > 128 && wps > 2) continue;
        const int threads = 256 * wps, blocks = 256;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<WC, WP, PIPE, SWZ>), dim3(blocks), dim3(threads), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double fl = (double)blocks * threads / 64 * iters * 9 * WC * WP * 32768.0;
        printf("%-34s %d waves/SIMD: %7.1f TFLOP/s  (reads/MFMA %.2f)\n", name, wps, fl / ms / 1e9, (double)(WC + WP) / (WC * WP));
    }
}

int main() {
    std::vector<_Float16> h(4096 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(((int)(i * 2654435761u >> 20) % 2001 - 1000) / 1000.0f);
    f16x8* d; float* o;
    (void)hipMalloc(&d, h.size() * 2); (void)hipMalloc(&o, 4096 * 4);
    (void)hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<2, 2, 0, 0>(d, o, "2x2 serialized linear");
    run<2, 2, 1, 0>(d, o, "2x2 pipelined  linear");
    run<2, 2, 1, 1>(d, o, "2x2 pipelined  swizzled-32B-rows");
    run<2, 4, 1, 0>(d, o, "2x4 pipelined  linear");
    run<4, 2, 1, 1>(d, o, "4x2 pipelined  swizzled-32B-rows");
    run<4, 4, 1, 0>(d, o, "4x4 pipelined  linear");
    return 0;
}
