// How fast can the [M][290] fp32 prediction rows be written?  (a) the current head_cls pattern: one workgroup per
// (256 pixels, 32 columns), 4-byte stores, a wave instruction covers 2 rows x 128 B;  (b) whole rows as a flat stream of
// 16-byte stores.  M = 32 x 6400 (level 0 of yololps at 640x640, batch 32): 238 MB.
//   hipcc -O3 --offload-arch=gfx950 -o pred_store_bw pred_store_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void store_tiles(float* out, int M) {   // grid: (ceil(M/256) * 9)
    const int ct = blockIdx.x % 9, pt = blockIdx.x / 9;
    const int p0 = pt * 256;
    for (int idx = threadIdx.x; idx < 256 * 32; idx += 256) {
        const int pl = idx >> 5, cl = idx & 31;
        const int c = ct * 32 + cl;
        if (p0 + pl < M && c < 277) out[(long long)(p0 + pl) * 290 + 13 + c] = 1.0f / (1.0f + __expf(-(float)(idx & 7)));
    }
}
__global__ __launch_bounds__(256) void store_rows_flat(float* out, int M, int rows_per_block) {   // whole rows, 16-B stores
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    float4* base = (float4*)(out + r0 * 290);       // rows_per_block even -> 16-B aligned
    const int n4 = rows_per_block * 290 / 4;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float v = 1.0f / (1.0f + __expf(-(float)(i & 7)));
        base[i] = make_float4(v, v, v, v);
    }
}
__global__ __launch_bounds__(256) void store_rows_cls(float* out, int M, int rows_per_block) {   // columns 13.. only, dwordx4 at 4-B alignment
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    for (int i = threadIdx.x; i < rows_per_block * 70; i += 256) {
        const int r = i / 70, q = i - r * 70;
        float* p = out + (r0 + r) * 290 + 13 + 4 * q;
        const float v = 1.0f / (1.0f + __expf(-(float)(i & 7)));
        if (q < 69) { typedef float f4 __attribute__((ext_vector_type(4), aligned(4))); *(f4*)p = f4{v, v, v, v}; }
        else *p = v;
    }
}
int main() {
    const int M = 32 * 6400;
    float* out; CK(hipMalloc(&out, (size_t)M * 290 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = -3; rep < 20; ++rep) {
            if (rep == 0) CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(store_tiles, dim3((M + 255) / 256 * 9), dim3(256), 0, 0, out, M);
            if (mode == 1) hipLaunchKernelGGL(store_rows_flat, dim3(M / 32), dim3(256), 0, 0, out, M, 32);
            if (mode == 2) hipLaunchKernelGGL(store_rows_cls, dim3(M / 32), dim3(256), 0, 0, out, M, 32);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = mode == 1 ? (double)M * 1160 : (double)M * 1108;
        printf("%-44s %7.1f us  %.2f TB/s\n", mode == 0 ? "tiles 256px x 32 cols, 4-B stores (current)" : mode == 1 ? "whole rows, flat 16-B stores" : "cols 13.., 16-B stores at 4-B alignment", ms * 1e3 / 20, bytes / (ms * 1e-3 / 20) / 1e12);
    }
    return 0;
}
