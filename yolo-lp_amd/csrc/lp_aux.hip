// Bandwidth-bound helper kernels of the engine: network input re-layout and the SPPF pool chain.
#include "lp_internal.h"

namespace lp {

// ---- input: caller's NCHW [B,3,H,W] (fp32 / fp16 / bf16) -> NHWC with 8 stored channels (3 real + 5 zero)
// of the activation dtype.  One thread per pixel: three coalesced plane reads, one 16-/32-byte row write.
// Replaces the implicit layout of the first conv's input (yolov6/models/yolo.py:34).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void input_kernel(const TI* __restrict__ x, TO* __restrict__ dst, int B, long long HW) {
    const long long total = (long long)B * HW;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long b = i / HW, p = i - b * HW;
        const TI* xp = x + b * 3 * HW + p;
        TO row[8] __attribute__((aligned(16)));
        row[0] = (TO)(float)xp[0];
        row[1] = (TO)(float)xp[HW];
        row[2] = (TO)(float)xp[2 * HW];
#pragma unroll
        for (int k = 3; k < 8; ++k) row[k] = (TO)0.f;
        uint4* o = (uint4*)(dst + i * 8);
        const uint4* rv = (const uint4*)row;
#pragma unroll
        for (int k = 0; k < (int)(8 * sizeof(TO) / 16); ++k) o[k] = rv[k];
    }
}

template <typename TI>
static int input_launch_to(const void* x, void* dst, int dtype, int B, int H, int W, hipStream_t st) {
    const long long HW = (long long)H * W;
    long long blocks = ((long long)B * HW + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    dim3 grid((unsigned)blocks);
    switch (dtype) {
        case LP_F16: hipLaunchKernelGGL((input_kernel<TI, f16>), grid, dim3(256), 0, st, (const TI*)x, (f16*)dst, B, HW); break;
        case LP_BF16: hipLaunchKernelGGL((input_kernel<TI, bf16>), grid, dim3(256), 0, st, (const TI*)x, (bf16*)dst, B, HW); break;
        case LP_F32: hipLaunchKernelGGL((input_kernel<TI, float>), grid, dim3(256), 0, st, (const TI*)x, (float*)dst, B, HW); break;
        default: return fail(LP_ERR_ARG, "input: dtype");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LP_ERR_HIP, std::string("input launch: ") + hipGetErrorString(e));
    return LP_OK;
}

int input_launch(const void* x, int x_dtype, void* dst, int dtype, int B, int H, int W, hipStream_t st) {
    switch (x_dtype) {
        case LP_F16: return input_launch_to<f16>(x, dst, dtype, B, H, W, st);
        case LP_BF16: return input_launch_to<bf16>(x, dst, dtype, B, H, W, st);
        case LP_F32: return input_launch_to<float>(x, dst, dtype, B, H, W, st);
    }
    return fail(LP_ERR_ARG, "input: x dtype");
}

// ---- input, space-to-depth form: NCHW [B,3,H,W] -> NHWC [B,H/2,W/2,16] with channel c*4 + py*2 + px holding
// x[c][2Y+py][2X+px] (12 real + 4 zero channels).  A 3x3 stride-2 conv of the image is then a stride-1 conv on this
// tensor (taps (dy,py): (0,1)->ky 0, (1,0)->ky 1, (1,1)->ky 2; the third tap row/column has zero weights), which
// runs on the stride-1 path of the MFMA kernel with a quarter of the halo.  One thread per output pixel: six
// coalesced 2-element reads, one 32-byte (fp32: 64-byte) row write.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void input_s2d_kernel(const TI* __restrict__ x, TO* __restrict__ dst, int B, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)B * Ho * Wo, HW = (long long)H * W;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int X = (int)(i % Wo);
        const long long t = i / Wo;
        const int Y = (int)(t % Ho);
        const long long b = t / Ho;
        TO row[16] __attribute__((aligned(16)));
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const TI* p = x + (b * 3 + c) * HW + (long long)(2 * Y + py) * W + 2 * X;
                row[c * 4 + py * 2 + 0] = (TO)(float)p[0];
                row[c * 4 + py * 2 + 1] = (TO)(float)p[1];
            }
#pragma unroll
        for (int k = 12; k < 16; ++k) row[k] = (TO)0.f;
        uint4* o = (uint4*)(dst + i * 16);
        const uint4* rv = (const uint4*)row;
#pragma unroll
        for (int k = 0; k < (int)(16 * sizeof(TO) / 16); ++k) o[k] = rv[k];
    }
}

template <typename TI>
static int input_s2d_launch_to(const void* x, void* dst, int dtype, int B, int H, int W, hipStream_t st) {
    long long blocks = ((long long)B * (H / 2) * (W / 2) + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    dim3 grid((unsigned)blocks);
    switch (dtype) {
        case LP_F16: hipLaunchKernelGGL((input_s2d_kernel<TI, f16>), grid, dim3(256), 0, st, (const TI*)x, (f16*)dst, B, H, W); break;
        case LP_BF16: hipLaunchKernelGGL((input_s2d_kernel<TI, bf16>), grid, dim3(256), 0, st, (const TI*)x, (bf16*)dst, B, H, W); break;
        case LP_F32: hipLaunchKernelGGL((input_s2d_kernel<TI, float>), grid, dim3(256), 0, st, (const TI*)x, (float*)dst, B, H, W); break;
        default: return fail(LP_ERR_ARG, "input: dtype");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LP_ERR_HIP, std::string("input launch: ") + hipGetErrorString(e));
    return LP_OK;
}

int input_s2d_launch(const void* x, int x_dtype, void* dst, int dtype, int B, int H, int W, hipStream_t st) {
    switch (x_dtype) {
        case LP_F16: return input_s2d_launch_to<f16>(x, dst, dtype, B, H, W, st);
        case LP_BF16: return input_s2d_launch_to<bf16>(x, dst, dtype, B, H, W, st);
        case LP_F32: return input_s2d_launch_to<float>(x, dst, dtype, B, H, W, st);
    }
    return fail(LP_ERR_ARG, "input: x dtype");
}

// ---- SPPF pool chain: y1 = m(x), y2 = m(y1), y3 = m(y2) with m = 5x5 stride-1 pad-2 max pool
// (yolov6/layers/common.py:144-146).  One block owns (image, 8-channel group): the whole h x w plane of the
// group lives in LDS and each pool is a separable row pass + column pass (max is exact in every dtype, so
// separability and the activation dtype do not change results).  Out-of-image taps are skipped, which is
// what -inf padding does.
// Element-wise max of 8 channels (one 16-B granule; fp32: 32 B).  fp16 uses the packed max (4 instructions, no
// conversions); max of finite values is exact in every type.
template <typename T> struct Max8 {
    typedef T V8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ V8 f(V8 a, V8 c) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = (float)c[k] > (float)a[k] ? c[k] : a[k];
        return a;
    }
};
template <> struct Max8<f16> {
    typedef f16 V8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ V8 f(V8 a, V8 c) { return __builtin_elementwise_max(a, c); }
};

// One workgroup per (image, GP granules = 8*GP channels): the plane sits in LDS, three chained separable 5x5 passes.
// A work item is (pixel, granule), so neighbouring lanes move neighbouring 16-B pieces of a pixel's channel run.
template <typename T, int NT>
__global__ __launch_bounds__(NT) void pool_chain_kernel(const T* __restrict__ src, T* __restrict__ d1, T* __restrict__ d2,
                                                        T* __restrict__ d3, int h, int w, int cs, int gp_log2, unsigned w_magic) {
    typedef T V8 __attribute__((ext_vector_type(8)));          // the 8 channels of one granule: 16 B (fp32: 32 B)
    extern __shared__ __attribute__((aligned(16))) char pool_smem[];
    const int GP = 1 << gp_log2;
    const int hw = h * w, items = hw * GP;
    // p / w as a multiply-high (w_magic = floor(2^32 / w) + 1, exact for p < 2^16): an integer division per work item and
    // pass was most of this kernel's instruction count
    // Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for vmcnt(0), i.e. for the round's global
    // stores, three store round trips per workgroup that nothing depends on
    auto lds_barrier = []() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto row_of = [&](int p) { return w == 1 ? p : (int)__umulhi((unsigned)p, w_magic); };   // (2^32 / 1 does not fit the magic)
    V8* cur = (V8*)pool_smem;     // [hw][GP]
    V8* tmp = cur + items;        // [hw][GP]
    const int groups = cs / (8 * GP);
    const int b = blockIdx.x / groups, cg = blockIdx.x - b * groups;
    const long long base = (long long)b * hw * cs + cg * 8 * GP;
    for (int i0 = threadIdx.x; i0 < items; i0 += NT * 8) {      // eight loads in flight per thread, then the LDS writes
        V8 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * NT;
            const int ic = i < items ? i : items - 1;
            const int p = ic >> gp_log2, g = ic & (GP - 1);
            v[k] = *(const V8*)(src + base + (long long)p * cs + g * 8);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i0 + k * NT < items) cur[i0 + k * NT] = v[k];
    }
    __syncthreads();
    for (int round = 0; round < 3; ++round) {
        T* const out = round == 0 ? d1 : round == 1 ? d2 : d3;      // (an indexed pointer array would live in scratch memory)
        for (int i = threadIdx.x; i < items; i += NT) {           // row pass
            const int p = i >> gp_log2, g = i & (GP - 1);
            const int y = row_of(p), x = p - y * w;
            const int x0 = x - 2 < 0 ? 0 : x - 2, x1 = x + 2 >= w ? w - 1 : x + 2;
            V8 m = cur[(y * w + x0) * GP + g];
            for (int xx = x0 + 1; xx <= x1; ++xx) m = Max8<T>::f(m, cur[(y * w + xx) * GP + g]);
            tmp[i] = m;
        }
        lds_barrier();
        for (int i = threadIdx.x; i < items; i += NT) {           // column pass
            const int p = i >> gp_log2, g = i & (GP - 1);
            const int y = row_of(p), x = p - y * w;
            const int y0 = y - 2 < 0 ? 0 : y - 2, y1 = y + 2 >= h ? h - 1 : y + 2;
            V8 m = tmp[(y0 * w + x) * GP + g];
            for (int yy = y0 + 1; yy <= y1; ++yy) m = Max8<T>::f(m, tmp[(yy * w + x) * GP + g]);
            *(V8*)(out + base + (long long)p * cs + g * 8) = m;
            cur[i] = m;   // only read again after the barrier below
        }
        lds_barrier();
    }
}

template <typename T>
static int pool_launch_t(const void* src, void* d1, void* d2, void* d3, int B, int h, int w, int cs, hipStream_t st) {
    // granules per workgroup: the widest channel run (up to 64 channels = one 128-B line in 16-bit types) that still
    // gives every CU two workgroups and fits the LDS (256 ch, 20x20, batch 32: 26 / 21 / 26 / 41 us for 1 / 2 / 4 / 8)
    int gp = 1;
    for (int cand = 8; cand >= 1; cand >>= 1) {
        const size_t need = (size_t)h * w * cand * 8 * sizeof(T) * 2;
        if ((cs / 8) % cand == 0 && need <= 64 * 1024 && (long long)B * (cs / 8 / cand) >= 512) { gp = cand; break; }
    }
    const size_t lds = (size_t)h * w * gp * 8 * sizeof(T) * 2;
    if (lds > POOL_MAX_LDS || h * w >= 65536) return fail(LP_ERR_UNSUPPORTED, "pool: feature map too large for the LDS-resident kernel");
    int gp_log2 = 0;
    while ((1 << gp_log2) < gp) ++gp_log2;
    const unsigned w_magic = (unsigned)(0x100000000ULL / (unsigned)w) + 1u;
    // few workgroups with big planes (yolov6m 1280x1280 bs=8: 384 workgroups of 1 600 work items, 50 us): 1 024 threads each
    const unsigned nwg = (unsigned)(B * (cs / 8 / gp));
    const bool wide = nwg < 512 && h * w * gp >= 1024;
    if (lds > 48 * 1024) {
        static std::atomic<unsigned long long> attr_done{0}, attr_done_w{0};   // one bit per device
        if (int rc = wide ? set_max_lds_once(pool_chain_kernel<T, 1024>, (int)POOL_MAX_LDS, attr_done_w, "pool")
                          : set_max_lds_once(pool_chain_kernel<T, 256>, (int)POOL_MAX_LDS, attr_done, "pool")) return rc;
    }
    if (wide) hipLaunchKernelGGL((pool_chain_kernel<T, 1024>), dim3(nwg), dim3(1024), lds, st, (const T*)src, (T*)d1, (T*)d2, (T*)d3, h, w, cs, gp_log2, w_magic);
    else hipLaunchKernelGGL((pool_chain_kernel<T, 256>), dim3(nwg), dim3(256), lds, st, (const T*)src, (T*)d1, (T*)d2, (T*)d3, h, w, cs, gp_log2, w_magic);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LP_ERR_HIP, std::string("pool launch: ") + hipGetErrorString(e));
    return LP_OK;
}

size_t pool_min_lds_bytes(int dtype, int h, int w) { return (size_t)h * w * 8 * dtype_size(dtype) * 2; }

int pool_launch(const void* src, void* d1, void* d2, void* d3, int dtype, int B, int h, int w, int cs, hipStream_t st) {
    switch (dtype) {
        case LP_F16: return pool_launch_t<f16>(src, d1, d2, d3, B, h, w, cs, st);
        case LP_BF16: return pool_launch_t<bf16>(src, d1, d2, d3, B, h, w, cs, st);
        case LP_F32: return pool_launch_t<float>(src, d1, d2, d3, B, h, w, cs, st);
    }
    return fail(LP_ERR_ARG, "pool: dtype");
}

}  // namespace lp

// ---- test hook: poison the LDS of every CU ----
// LDS keeps its contents between kernels, and two launches of one kernel on the same data leave in every ring slot exactly the
// bytes the next launch is about to fetch: a fragment read that runs ahead of its LDS-DMA (a missed or mis-counted wait) then
// reads stale but CORRECT bytes and the race stays invisible.  This kernel fills all 160 KiB of every CU with 0xFFFF (a NaN in
// fp16 and bf16) so that any such read poisons the output deterministically (tests/test_hip_kernels.py).
namespace lp {
__global__ __launch_bounds__(1024) void poison_lds_kernel(unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_words[];
    constexpr int NWORDS = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < NWORDS; i += 1024) lds_words[i] = 0xFFFFFFFFu;
    __syncthreads();
    // keep the stores alive and the workgroup resident for a moment, so that the workgroups of the grid spread over all CUs
    unsigned acc = 0;
    for (int i = threadIdx.x; i < NWORDS; i += 1024) acc |= lds_words[i];
    __builtin_amdgcn_s_sleep(64);
    if (acc != 0xFFFFFFFFu && sink) sink[0] = acc;
}
}  // namespace lp

extern "C" int lp_debug_poison_lds(void* stream) {
    using namespace lp;
    static std::atomic<unsigned long long> attr_done{0};
    if (int rc = set_max_lds_once(poison_lds_kernel, 160 * 1024, attr_done, "poison_lds")) return rc;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    // a workgroup takes a whole CU's LDS, so the dispatcher can place at most one per CU at a time: three rounds of the grid
    hipLaunchKernelGGL(poison_lds_kernel, dim3((unsigned)(3 * ncu)), dim3(1024), 160 * 1024, (hipStream_t)stream, (unsigned*)nullptr);
    LP_HIP_CHECK(hipGetLastError());
    return LP_OK;
}
