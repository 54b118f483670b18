// LP accuracy metric: per-image matching of detections to labels and the counters the metric is made of.
//
// Reference: Evaler.eval (yolov6/core/evaler.py:153-243) -- a python triple loop with a device sync per scalar --
// and box_iou (yolov6/utils/general.py:93-115).  One workgroup per image: for every label the best detection (IoU
// max over detections, first index on ties like torch's CPU max; a NaN IoU wins and stops the scan), then the
// 0.05-wide IoU bins, the corner test (mean |d| of the 8 corner coordinates < 0.1 sqrt(label area)) and the class
// test (all 8 characters equal after int() truncation).  float32 arithmetic in the reference's operation order
// (-ffp-contract=off, IEEE divide / sqrt), so the counters are identical, not close.
//
// Not reproduced: a matched label whose IoU fits no bin (IoU >= 1.0f: identical boxes) re-uses the bin of the previous
// matched label in the reference's sequential loop (stale variable, UnboundLocalError if there is none).  Such labels
// are skipped here and counted in counts[LP_EVAL_UNBINNED] so that the caller can see that the case occurred.
#include "lp_internal.h"

namespace lp {

struct EvalBins { float lo[10], hi[10]; };

__device__ __forceinline__ float eval_iou(const float* a, const float* b) {
    const float a1 = (a[2] - a[0]) * (a[3] - a[1]);
    const float a2 = (b[2] - b[0]) * (b[3] - b[1]);
    float dx = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
    float dy = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
    dx = dx > 0.f ? dx : 0.f;          // clamp(0): NaN stays NaN in torch; boxes are finite here
    dy = dy > 0.f ? dy : 0.f;
    const float inter = dx * dy;
    return inter / ((a1 + a2) - inter);
}

// (v, i) better than (bv, bi) in the order of a sequential scan with "!(v <= best)": larger wins, NaN wins over
// numbers, equal values / several NaNs keep the lower index
__device__ __forceinline__ bool eval_better(float v, int i, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || i < bi);
    return v > bv || (v == bv && i < bi);
}

__device__ __forceinline__ int eval_bin(float t, const EvalBins& bins) {
    for (int n = 0; n < 10; ++n)
        if (t >= bins.lo[n] && t < bins.hi[n]) return n;
    return -1;
}

__global__ __launch_bounds__(256) void eval_counts_kernel(const float* __restrict__ det, const int* __restrict__ det_count, int max_det,
                                                          const float* __restrict__ tgt, const int* __restrict__ tgt_count, int max_t,
                                                          const EvalBins bins, unsigned long long* __restrict__ counts) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    __shared__ unsigned s_cnt[LP_EVAL_NCOUNTS];
    const int b = blockIdx.x, tid = threadIdx.x;
    int n = det_count[b], m = tgt_count[b];
    n = n < max_det ? n : max_det;
    m = m < max_t ? m : max_t;
    if (tid < LP_EVAL_NCOUNTS) s_cnt[tid] = 0;
    __syncthreads();
    if (tid == 0) s_cnt[LP_EVAL_TRUE] = (unsigned)m;
    const float* D = det + (long long)b * max_det * 28;
    const float* T = tgt + (long long)b * max_t * 20;
    if (n > 0)
        for (int k = 0; k < m; ++k) {
            const float* g = T + k * 20;
            float bv = 0.f;
            int bi = -1;
            for (int i = tid; i < n; i += 256) {
                const float v = eval_iou(D + i * 28, g + 8);
                if (bi < 0 || eval_better(v, i, bv, bi)) { bv = v; bi = i; }
            }
            s_v[tid] = bv;
            s_i[tid] = bi;
            __syncthreads();
            for (int s = 128; s > 0; s >>= 1) {
                if (tid < s && s_i[tid + s] >= 0 && (s_i[tid] < 0 || eval_better(s_v[tid + s], s_i[tid + s], s_v[tid], s_i[tid]))) {
                    s_v[tid] = s_v[tid + s];
                    s_i[tid] = s_i[tid + s];
                }
                __syncthreads();
            }
            if (tid == 0) {
                const float t = s_v[0];
                if (!(t < 0.5f)) {                       // evaler.py:199 ``if t_iou < 0.5: continue`` (NaN does not skip)
                    if (t >= 0.7f) ++s_cnt[LP_EVAL_PRED];
                    const int bn = eval_bin(t, bins);
                    if (bn < 0) {
                        ++s_cnt[LP_EVAL_UNBINNED];
                    } else {
                        ++s_cnt[LP_EVAL_PRED_BINS + bn];   // the reference's second pass (:234-243) counts the same labels
                        const float* p = D + s_i[0] * 28;
                        const float area = (g[10] - g[8]) * (g[11] - g[9]);
                        float sum = 0.f;
                        for (int q = 0; q < 8; ++q) sum = sum + fabsf(p[4 + q] - g[12 + q]);
                        const bool is_cor = sum / 8.0f < 0.1f * sqrtf(area);
                        bool is_cls = true;
                        for (int q = 0; q < 8; ++q) is_cls = is_cls && ((int)p[20 + q] == (int)g[q]);
                        if (is_cor) ++s_cnt[LP_EVAL_COR + bn];
                        if (is_cls) ++s_cnt[LP_EVAL_CLS + bn];
                        if (is_cor && is_cls) ++s_cnt[LP_EVAL_RIGHT + bn];
                    }
                }
            }
            __syncthreads();
        }
    __syncthreads();
    if (tid < LP_EVAL_NCOUNTS && s_cnt[tid]) atomicAdd(counts + tid, (unsigned long long)s_cnt[tid]);
}

}  // namespace lp

using namespace lp;

extern "C" int lp_eval_counts(const float* det, const int* det_count, int max_det, const float* tgt, const int* tgt_count,
                              int max_t, int B, long long* counts, void* stream) {
    if (B < 0 || max_det < 0 || max_t < 0) return fail(LP_ERR_ARG, "lp_eval_counts: negative size");
    if (!counts) return fail(LP_ERR_ARG, "lp_eval_counts: counts is null");
    if (B == 0) return LP_OK;
    if (!det_count || !tgt_count || (max_det > 0 && !det) || (max_t > 0 && !tgt)) return fail(LP_ERR_ARG, "lp_eval_counts: null pointer");
    EvalBins bins;
    for (int i = 0; i < 10; ++i) {   // the reference's python doubles (evaler.py:159,203), rounded like torch's scalar promotion
        const double lo = 0.5 + i * 0.05;
        bins.lo[i] = (float)lo;
        bins.hi[i] = (float)(lo + 0.05);
    }
    hipLaunchKernelGGL(eval_counts_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, det, det_count, max_det, tgt, tgt_count,
                       max_t, bins, (unsigned long long*)counts);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LP_ERR_HIP, std::string("lp_eval_counts launch: ") + hipGetErrorString(e));
    return LP_OK;
}
