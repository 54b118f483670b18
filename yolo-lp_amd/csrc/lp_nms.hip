// Post-processing of the LP head output on gfx950: score / arg-max wavefront reductions, candidate
// collection, stable descending sort and greedy IoU suppression, for the whole batch in four launches and
// without any device->host synchronisation.
//
// Reference semantics (bit-exact selection is the bar): yolov6/utils/nms.py:68-125 and the CPU kernel of
// torchvision.ops.nms it calls (:121).  Everything is fp32, evaluated op by op in the reference's order
// (this file is compiled with -ffp-contract=off; `/` is the correctly rounded division):
//   x[:,13:] *= x[:,4:5]                              in place (written back only where obj != 1)
//   box = (cx - w/2, cy - h/2, cx + w/2, cy + h/2)
//   8 x (max, FIRST arg-max) over column groups [13,44) [44,68) [68,105) ... [253,290)
//   keep-mask: (pro+alp+ad0+ad1+ad2+ad3+ad4+ad4)/8 >= fp32(conf)     (ad4 twice, ad5 omitted: reference quirk)
//   score:     (pro+alp+ad0+ad1+ad2+ad3+ad4+ad5)/8                   (left-to-right sums)
//   order: descending score, ties by ascending anchor index (stable sort); at most 30000 candidates
//   greedy: keep i, suppress later j with inter/(area_i+area_j-inter) > iou (double compare), until max_det
#include "lp_internal.h"
#include "lp_score.inc"

namespace lp {

static constexpr int NCOL = LP_PRED_COLS;
static constexpr int NDET = LP_DET_COLS;
static constexpr int MAX_NMS = 30000;
static constexpr int SORT_LDS_KEYS = 8192;   // keys sorted inside LDS (64 KiB); larger lists are sorted in global memory
static int next_pow2(int n) {
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

NmsWs nms_carve(void* base, int B, int N) {
    NmsWs w;
    w.NP = next_pow2(N < 64 ? 64 : N);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) / 256 * 256;
        return (char*)base + o;
    };
    w.cnt = (int32_t*)take((size_t)B * 4);
    w.keys = (unsigned long long*)take((size_t)B * w.NP * 8);
    w.rows = (float*)take((size_t)B * N * NDET * 4);
    w.sbox = (float4*)take((size_t)B * N * 16);
    w.kept = (int32_t*)take((size_t)B * N * 4);
    w.bytes = off;
    return w;
}

// 16 lanes per anchor row, 4 rows per wave, 16 rows per block iteration.  Lane j of a row group owns columns
// j, j+16, ...: it folds its own columns of each head in ascending order (first maximum wins), then four
// rotate steps (1,2,4,8) of a (value, index) max-combine leave every lane of the group with the head's result
// (the combine is commutative, associative and idempotent, so a rotate all-reduce is exact).
// The rows may be one pyramid level of the anchors only (detections-only forward, levels whose class predictors still go
// through a prediction scratch): RPI prediction rows per image = anchors anchor0 .. anchor0 + RPI of the N per image;
// write_box false leaves columns 0..11 of the candidate rows to the decode kernel that has written them already.
__global__ __launch_bounds__(256) void score_kernel(float* __restrict__ pred, int B, int RPI, int anchor0, int N, float conf_f,
                                                   float* __restrict__ rows, unsigned long long* __restrict__ keys,
                                                   int32_t* __restrict__ cnt, int NP, int write_box) {
    constexpr int NK = (NCOL + 15) / 16;   // 19 column slots per lane
    const int j = threadIdx.x & 15;
    const long long nrows = (long long)B * RPI;
    const long long gstride = (long long)gridDim.x * 16;
    for (long long base = (long long)blockIdx.x * 16; base < nrows; base += gstride) {   // block-uniform trip count
        const long long row0 = base + (threadIdx.x >> 4);
        const bool live = row0 < nrows;                 // whole 16-lane groups are live or not
        const long long row = live ? row0 : nrows - 1;  // dead groups shadow the last row and write nothing
        float* x = pred + row * NCOL;
        float v[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int c = j + 16 * k;
            v[k] = c < NCOL ? x[c] : 0.f;
        }
        const float obj = write_box ? __shfl(v[0], 4, 16) : 1.0f;   // (detections-only levels: the decode kernel does not fill the scratch; obj is 1)
        // conf = obj_conf * cls_conf, in place (nms.py:76); obj == 1 leaves the bits unchanged: no store needed
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int c = j + 16 * k;
            if (c >= 13 && c < NCOL) {
                v[k] = v[k] * obj;
                if (obj != 1.0f && live) x[c] = v[k];
            }
        }
        float cf[8];
        int ci[8];
        score_heads<0, NK>(v, j, cf, ci);
        const float m = score_mask_value(cf);
        const bool pass = live && (m >= conf_f);        // uniform inside a 16-lane group
        const float sc = score_value(cf);

        // all lanes take part in the shuffles; only passing groups store
        const float cx = __shfl(v[0], 0, 16), cy = __shfl(v[0], 1, 16), bw = __shfl(v[0], 2, 16), bh = __shfl(v[0], 3, 16);
        const float corner = __shfl(v[0], (j + 1) & 15, 16);      // lanes 4..11 pick columns 5..12
        if (pass) {
            float o0, o1;
            if (j == 0) o0 = cx - bw / 2;
            else if (j == 1) o0 = cy - bh / 2;
            else if (j == 2) o0 = cx + bw / 2;
            else if (j == 3) o0 = cy + bh / 2;
            else if (j < 12) o0 = corner;
            else { o0 = cf[0]; if (j == 13) o0 = cf[1]; if (j == 14) o0 = cf[2]; if (j == 15) o0 = cf[3]; }
            // second element: detection column 16 + j  (conf 4..7 for j < 4, indices 0..7 for 4 <= j < 12)
            o1 = cf[4];
            if (j == 1) o1 = cf[5];
            if (j == 2) o1 = cf[6];
            if (j == 3) o1 = cf[7];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (j == 4 + k) o1 = (float)ci[k];
            const int bimg = (int)(row / RPI), n = anchor0 + (int)(row - (long long)bimg * RPI);
            float* out = rows + ((long long)bimg * N + n) * NDET;
            if (write_box || j >= 12) out[j] = o0;
            if (j < 12) out[16 + j] = o1;
            if (j == 0) {
                const int pos = atomicAdd(&cnt[bimg], 1);
                keys[(long long)bimg * NP + pos] = score_key(sc, n);
            }
        }
    }
}

// One block per image: ascending bitonic sort of its candidate keys (all keys are distinct, so the result is
// unique whatever order the atomics appended them in).
__global__ __launch_bounds__(1024) void sort_kernel(unsigned long long* __restrict__ keys, const int32_t* __restrict__ cnt, int NP) {
    __shared__ unsigned long long skeys[SORT_LDS_KEYS];
    const int b = blockIdx.x;
    const int nc = cnt[b];
    if (nc <= 1) return;
    int n = 64;
    while (n < nc) n <<= 1;
    unsigned long long* g = keys + (long long)b * NP;
    const bool in_lds = n <= SORT_LDS_KEYS;
    unsigned long long* d = in_lds ? skeys : g;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const unsigned long long k = i < nc ? g[i] : ~0ull;
        d[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = d[i], c = d[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { d[i] = c; d[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    if (in_lds)
        for (int i = threadIdx.x; i < nc; i += 1024) g[i] = d[i];
}

__device__ __forceinline__ bool iou_gt(float ix1, float iy1, float ix2, float iy2, float iarea, float jx1, float jy1,
                                      float jx2, float jy2, float thr_f) {
    const float xx1 = ix1 > jx1 ? ix1 : jx1;
    const float yy1 = iy1 > jy1 ? iy1 : jy1;
    const float xx2 = ix2 < jx2 ? ix2 : jx2;
    const float yy2 = iy2 < jy2 ? iy2 : jy2;
    float w = xx2 - xx1;
    if (!(w > 0.f)) w = 0.f;
    float h = yy2 - yy1;
    if (!(h > 0.f)) h = 0.f;
    const float inter = w * h;
    const float jarea = (jx2 - jx1) * (jy2 - jy1);
    const float ovr = inter / (iarea + jarea - inter);
    return ovr > thr_f;   // thr_f = largest fp32 <= the double threshold  <=>  (double)ovr > iou_thres
}

// One block per image: greedy suppression over the sorted candidates, 64 at a time.  Wave 0 resolves the
// dependencies inside a chunk with ballots; then every thread strikes out the later candidates that overlap
// one of the chunk's survivors.  Ends with the gather of the kept rows into det.
static constexpr int NMS_T = 512;
__global__ __launch_bounds__(NMS_T) void greedy_kernel(const unsigned long long* __restrict__ keys, const float* __restrict__ rows,
                                                      float4* __restrict__ sbox, int32_t* __restrict__ kept,
                                                      const int32_t* __restrict__ cnt, int N, int NP, float thr_f, int max_det,
                                                      float* __restrict__ det, int32_t* __restrict__ count,
                                                      int32_t* __restrict__ keep_out) {
    __shared__ unsigned sup[(MAX_NMS + 31) / 32 + 1];
    __shared__ float ck[64][5];
    __shared__ int s_nk, s_total;
    const int b = blockIdx.x, tid = threadIdx.x;
    int nc = cnt[b];
    if (nc > MAX_NMS) nc = MAX_NMS;
    const unsigned long long* kb = keys + (long long)b * NP;
    const float* rb = rows + (long long)b * N * NDET;
    float4* sb = sbox + (long long)b * N;
    int32_t* kp = kept + (long long)b * N;
    for (int i = tid; i < (nc + 31) / 32 + 1; i += NMS_T) sup[i] = 0;
    for (int i = tid; i < nc; i += NMS_T) {
        const int idx = (int)(kb[i] & 0xffffffffu);
        sb[i] = *(const float4*)(rb + (long long)idx * NDET);   // rows are 112 B apart: 16-B aligned
    }
    if (tid == 0) s_total = 0;
    __syncthreads();

    const int nchunk = (nc + 63) / 64;
    for (int c = 0; c < nchunk; ++c) {
        if (tid < 64) {
            const int i = c * 64 + tid;
            const bool valid = i < nc;
            float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) bx = sb[i];
            const float area = (bx.z - bx.x) * (bx.w - bx.y);
            bool alive = valid && !((sup[i >> 5] >> (i & 31)) & 1u);
            unsigned long long todo = __ballot(alive);
            unsigned long long keptmask = 0;
            while (todo) {
                const int k = __ffsll((long long)todo) - 1;
                keptmask |= 1ull << k;
                todo &= ~(1ull << k);
                const float kx1 = __shfl(bx.x, k), ky1 = __shfl(bx.y, k), kx2 = __shfl(bx.z, k), ky2 = __shfl(bx.w, k);
                const float karea = __shfl(area, k);
                if (alive && tid > k && iou_gt(kx1, ky1, kx2, ky2, karea, bx.x, bx.y, bx.z, bx.w, thr_f)) alive = false;
                todo &= __ballot(alive);
            }
            const int total = s_total;
            if ((keptmask >> tid) & 1ull) {
                const int rank = __popcll(keptmask & ((1ull << tid) - 1ull));
                ck[rank][0] = bx.x; ck[rank][1] = bx.y; ck[rank][2] = bx.z; ck[rank][3] = bx.w; ck[rank][4] = area;
                kp[total + rank] = (int)(kb[i] & 0xffffffffu);
            }
            if (tid == 0) { s_nk = __popcll(keptmask); }
        }
        __syncthreads();
        const int nk = s_nk;
        const int total_after = s_total + nk;
        __syncthreads();                       // everyone has read s_total before it is updated
        if (tid == 0) s_total = total_after;
        if (total_after >= max_det) break;     // block-uniform
        for (int jj = (c + 1) * 64 + tid; jj < nc; jj += NMS_T) {
            if ((sup[jj >> 5] >> (jj & 31)) & 1u) continue;
            const float4 bj = sb[jj];
            for (int k = 0; k < nk; ++k) {
                if (iou_gt(ck[k][0], ck[k][1], ck[k][2], ck[k][3], ck[k][4], bj.x, bj.y, bj.z, bj.w, thr_f)) {
                    atomicOr(&sup[jj >> 5], 1u << (jj & 31));
                    break;
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    int total = s_total;
    if (total > max_det) total = max_det;
    if (tid == 0) count[b] = total;
    float* db = det + (long long)b * max_det * NDET;
    for (int i = tid; i < max_det * NDET; i += NMS_T) {
        const int k = i / NDET, col = i - k * NDET;
        db[i] = k < total ? rb[(long long)kp[k] * NDET + col] : 0.f;
    }
    if (keep_out)
        for (int k = tid; k < max_det; k += NMS_T) keep_out[(long long)b * max_det + k] = k < total ? kp[k] : -1;
}

}  // namespace lp

using namespace lp;

namespace lp {
int nms_score_launch(float* pred, int B, int rows_per_img, int anchor0, int N, float conf_f, const NmsWs& w, bool write_box, hipStream_t st) {
    const long long nrows = (long long)B * rows_per_img;
    long long blocks = (nrows + 15) / 16;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(score_kernel, dim3((unsigned)blocks), dim3(256), 0, st, pred, B, rows_per_img, anchor0, N, conf_f, w.rows, w.keys,
                       w.cnt, w.NP, write_box ? 1 : 0);
    LP_HIP_CHECK(hipGetLastError());
    return LP_OK;
}
}  // namespace lp

extern "C" size_t lp_nms_workspace_bytes(int B, int N) {
    if (B < 1 || N < 1) return 256;
    return nms_carve(nullptr, B, N).bytes;
}

extern "C" int lp_nms(float* pred, int B, int N, double conf_thres, double iou_thres, int max_det, float* det,
                      int32_t* count, int32_t* keep, void* workspace, size_t workspace_bytes, void* stream) {
    if (!pred || !det || !count || !workspace) return fail(LP_ERR_ARG, "lp_nms: null pointer");
    if (B < 1 || N < 1 || max_det < 1) return fail(LP_ERR_ARG, "lp_nms: B, N and max_det must be positive");
    if (!(conf_thres >= 0.0 && conf_thres <= 1.0) || !(iou_thres >= 0.0 && iou_thres <= 1.0))
        return fail(LP_ERR_ARG, "lp_nms: thresholds must be in [0, 1]");
    if (((uintptr_t)workspace & 255) != 0) return fail(LP_ERR_ARG, "lp_nms: workspace must be 256-byte aligned");
    NmsWs w = nms_carve(workspace, B, N);
    if (workspace_bytes < w.bytes) return fail(LP_ERR_ARG, "lp_nms: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const float conf_f = (float)conf_thres;
    float thr_f = (float)iou_thres;                       // largest fp32 not above the double threshold
    if ((double)thr_f > iou_thres) thr_f = nextafterf(thr_f, -INFINITY);

    LP_HIP_CHECK(hipMemsetAsync(w.cnt, 0, (size_t)B * 4, st));
    if (int rc = nms_score_launch(pred, B, N, 0, N, conf_f, w, true, st)) return rc;
    hipLaunchKernelGGL(sort_kernel, dim3((unsigned)B), dim3(1024), 0, st, w.keys, w.cnt, w.NP);
    hipLaunchKernelGGL(greedy_kernel, dim3((unsigned)B), dim3(NMS_T), 0, st, w.keys, w.rows, w.sbox, w.kept, w.cnt, N, w.NP,
                       thr_f, max_det, det, count, keep);
    LP_HIP_CHECK(hipGetLastError());
    return LP_OK;
}

extern "C" int lp_nms_candidates(int B, int N, double iou_thres, int max_det, float* det, int32_t* count, int32_t* keep,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!det || !count || !workspace) return fail(LP_ERR_ARG, "lp_nms_candidates: null pointer");
    if (B < 1 || N < 1 || max_det < 1) return fail(LP_ERR_ARG, "lp_nms_candidates: B, N and max_det must be positive");
    if (!(iou_thres >= 0.0 && iou_thres <= 1.0)) return fail(LP_ERR_ARG, "lp_nms_candidates: threshold must be in [0, 1]");
    if (((uintptr_t)workspace & 255) != 0) return fail(LP_ERR_ARG, "lp_nms_candidates: workspace must be 256-byte aligned");
    NmsWs w = nms_carve(workspace, B, N);
    if (workspace_bytes < w.bytes) return fail(LP_ERR_ARG, "lp_nms_candidates: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float thr_f = (float)iou_thres;                       // largest fp32 not above the double threshold
    if ((double)thr_f > iou_thres) thr_f = nextafterf(thr_f, -INFINITY);
    hipLaunchKernelGGL(sort_kernel, dim3((unsigned)B), dim3(1024), 0, st, w.keys, w.cnt, w.NP);
    hipLaunchKernelGGL(greedy_kernel, dim3((unsigned)B), dim3(NMS_T), 0, st, w.keys, w.rows, w.sbox, w.kept, w.cnt, N, w.NP,
                       thr_f, max_det, det, count, keep);
    LP_HIP_CHECK(hipGetLastError());
    return LP_OK;
}
