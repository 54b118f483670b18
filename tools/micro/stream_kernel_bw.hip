// Runs the library's streaming 1x1 kernel on plain hipMalloc buffers (64 -> 64 channels, f16), next to a bare copy kernel of
// the same access pattern, to separate "kernel structure" from "where the tensors live".
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I yolo-lp_amd/csrc -o stream_kernel_bw tools/micro/stream_kernel_bw.hip
#include "lp_conv1x1_stream.inc"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
namespace lp {
void set_error(const std::string&) {}
int fail(int code, const std::string& msg) { printf("fail: %s\n", msg.c_str()); return code; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_kernel(const char* __restrict__ src, char* __restrict__ dst, long long nchunks) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long c = wave; c < nchunks; c += nwaves) {
        const char* p = src + c * 8192;
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const u32x4*)(p + ((lane & 31) * 128 + (u % 4) * 32 + (lane >> 5) * 16) + (u / 4) * 4096);
        char* q = dst + c * 8192;
#pragma unroll
        for (int u = 0; u < 8; ++u) *(u32x4*)(q + u * 1024 + lane * 16) = v[u];
    }
}
int main(int argc, char** argv) {
    const int B = 32, H = 160, W = 160, C = 64;
    const size_t bytes = (size_t)B * H * W * C * 2;
    const size_t gap = argc > 1 ? (size_t)atol(argv[1]) : 0;      // extra bytes between src and dst
    char* buf; CK(hipMalloc(&buf, 2 * bytes + gap + (1 << 20)));
    char* src = buf; char* dst = buf + bytes + gap;
    std::vector<unsigned short> hsrc(bytes / 2);
    for (size_t i = 0; i < hsrc.size(); ++i) hsrc[i] = (unsigned short)(0x3000 + (rand() & 0x3ff));   // random small f16 values
    CK(hipMemcpy(src, hsrc.data(), bytes, hipMemcpyHostToDevice));
    char* wts; CK(hipMalloc(&wts, 1 << 20)); CK(hipMemset(wts, 0, 1 << 20));
    if (argc > 2 && atoi(argv[2])) {     // non-zero weights: outputs are not all zero
        std::vector<unsigned short> hw(64 * 64);
        for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned short)(0x2c00 + (rand() & 0x3ff));
        CK(hipMemcpy(wts + 4096, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    }
    lp::ConvArgs a; memset(&a, 0, sizeof(a));
    a.nsrc = 1; a.src[0].ptr = src; a.src[0].cs = C;
    for (int i = 1; i <= LP_MAX_SRC; ++i) a.chunk_begin[i] = 1;
    a.w = wts + 4096; a.bias = (const float*)(wts + 65536); a.zero = wts; a.trash = wts + 128;
    a.out = dst; a.B = B; a.H = H; a.W = W; a.Ho = H; a.Wo = W; a.nct = 1; a.out_c = C; a.out_pix_stride = C;
    a.out_img_stride = (long long)H * W * C; a.out_scale = 1; a.nphase = 1; a.act = LP_ACT_RELU;
    const int lds = 1 * 64 * 128 + 64 * 4 + 4 * 64 * (64 * 2 + 16);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 3; ++i) {
            if (pass == 0) hipLaunchKernelGGL(copy_kernel, dim3(512), dim3(256), 0, 0, src, dst, (long long)(bytes / 8192));
            else lp::stream_launch_one<lp::f16, 2, 1, false>(a, 64, lds, 0);
        }
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) {
            if (pass == 0) hipLaunchKernelGGL(copy_kernel, dim3(512), dim3(256), 0, 0, src, dst, (long long)(bytes / 8192));
            else lp::stream_launch_one<lp::f16, 2, 1, false>(a, 64, lds, 0);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-22s gap %zu: %6.1f us  %.2f TB/s\n", pass ? "conv1x1_stream_kernel" : "copy_kernel", gap, ms * 1e3 / 20, 2.0 * bytes / (ms * 1e-3 / 20) / 1e12);
    }
    return 0;
}
