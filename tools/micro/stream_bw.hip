// HBM streaming microbenchmark: what read / copy bandwidth does a kernel STRUCTURE reach on this GPU?
//   hipcc -O3 --offload-arch=gfx950 -o stream_bw stream_bw.hip && ./stream_bw [MiB]
// Variants: loads per lane in flight (U x 16 B), waves per CU (grid size, persistent grid-stride loop), and the order in
// which a wave walks memory (contiguous 1 KiB per instruction, or one 128-B row per lane pair like an MFMA B operand).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// each wave handles chunks of U KiB; chunk c of the buffer: [c*U*1024, (c+1)*U*1024)
template <int U, int PATTERN, bool WRITE>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ src, char* __restrict__ dst, long long nchunks, unsigned* sink) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    u32x4 accv = {0, 0, 0, 0};
    for (long long c = wave; c < nchunks; c += nwaves) {
        const char* p = src + c * (U * 1024);
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned off;
            if (PATTERN == 0) off = u * 1024 + lane * 16;                                    // 1 KiB contiguous per instruction
            else off = ((lane & 31) * 128 + (u % 4) * 32 + (lane >> 5) * 16) + (u / 4) * 4096;   // MFMA B-operand order (U multiple of 4)
            v[u] = *(const u32x4*)(p + off);
        }
        if (WRITE) {
            char* q = dst + c * (U * 1024);
#pragma unroll
            for (int u = 0; u < U; ++u) *(u32x4*)(q + u * 1024 + lane * 16) = v[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) accv += v[u];
        }
    }
    if (!WRITE && (accv[0] ^ accv[1] ^ accv[2] ^ accv[3]) == 0x12345u) *sink = 1;
}

template <int U, int PATTERN, bool WRITE>
static void run(const char* name, const char* src, char* dst, size_t bytes, int blocks, unsigned* sink) {
    const long long nchunks = bytes / (U * 1024);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<U, PATTERN, WRITE>), dim3(blocks), dim3(256), 0, 0, src, dst, nchunks, sink);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<U, PATTERN, WRITE>), dim3(blocks), dim3(256), 0, 0, src, dst, nchunks, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%-28s U=%d blocks=%6d : %7.1f us  %5.2f TB/s%s\n", name, U, blocks, us, bytes * (WRITE ? 2.0 : 1.0) / us / 1e6, WRITE ? " (read+write)" : " (read)");
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? atoi(argv[1]) : 100;
    const size_t bytes = mib << 20;
    char *src, *dst; unsigned* sink;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
    printf("buffer %zu MiB\n", mib);
    const int grids[] = {256, 512, 1024, 2048, 4096, 0};
    for (int gi = 0; gi < 6; ++gi) {
        const int g = grids[gi] ? grids[gi] : (int)(bytes / (8 * 1024) / 4);   // 0: one 8 KiB chunk per wave (non-persistent)
        run<8, 0, false>("read  contiguous", src, dst, bytes, g, sink);
        run<8, 1, false>("read  B-operand order", src, dst, bytes, g, sink);
        run<4, 0, false>("read  contiguous", src, dst, bytes, g, sink);
        run<16, 0, false>("read  contiguous", src, dst, bytes, g, sink);
        run<8, 0, true>("copy  contiguous", src, dst, bytes, g, sink);
        run<8, 1, true>("copy  B-order reads", src, dst, bytes, g, sink);
    }
    return 0;
}
