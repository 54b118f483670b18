// Per-CU throughput of the two ways of fetching operands: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction)
// and plain global_load_dwordx4 into registers, from an L2-resident buffer.  Answers: how many bytes per clock can a CU
// stage through the texture-address path, i.e. what is the floor of the conv kernel's "DMA issue" phase?
//   hipcc -O3 --offload-arch=gfx950 -o dma_lds_bw dma_lds_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool DMA, int PER_ROUND>
__global__ __launch_bounds__(256) void fetch_kernel(const char* __restrict__ src, int rounds, int region, unsigned* sink, long long* cycles) {
    __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + (size_t)(blockIdx.x % 64) * region;          // a 64 x region working set: L2 resident
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int k = 0; k < PER_ROUND; ++k) {
            const int off = ((r * PER_ROUND + k) * 4 + wave) * 1024 % region;
            if (DMA) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + off + lane * 16), (lds_ptr_t)(lds + (k * 4 + wave) * 1024), 16, 0, 0);
            else acc += *(const u32x4*)(base + off + lane * 16);
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (!DMA && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x1234567u) *sink = 1;
    if (DMA && lds[threadIdx.x] == 77 && rounds < 0) *sink = 2;
}

template <bool DMA, int PER_ROUND>
static void run(const char* name, const char* src, int blocks_per_cu, unsigned* sink, long long* cyc) {
    const int rounds = 400, region = 256 * 1024, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((fetch_kernel<DMA, PER_ROUND>), dim3(blocks), dim3(256), 0, 0, src, rounds, region, sink, cyc);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((fetch_kernel<DMA, PER_ROUND>), dim3(blocks), dim3(256), 0, 0, src, rounds, region, sink, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)blocks * rounds * PER_ROUND * 4096.0;
    printf("%-22s %2d instr/round/wave, %d workgroup(s)/CU: %7.1f us  %6.2f TB/s aggregate  %5.1f B/clk/CU at 2.4 GHz\n", name, PER_ROUND,
           blocks_per_cu, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / 256.0 / (ms * 1e-3 * 2.4e9));
}

int main() {
    char* src; unsigned* sink; long long* cyc;
    CK(hipMalloc(&src, 64 * 256 * 1024)); CK(hipMemset(src, 1, 64 * 256 * 1024));
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&cyc, 8 * 4096));
    for (int bpc = 1; bpc <= 2; ++bpc) {      // 64 KiB of LDS per workgroup: at most 2 per CU
        run<true, 4>("LDS-DMA", src, bpc, sink, cyc);
        run<true, 8>("LDS-DMA", src, bpc, sink, cyc);
        run<true, 16>("LDS-DMA", src, bpc, sink, cyc);
        run<false, 8>("loads to registers", src, bpc, sink, cyc);
        run<false, 16>("loads to registers", src, bpc, sink, cyc);
    }
    return 0;
}
