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
static constexpr int SORT_LDS_KEYS = 16384;  // keys per LDS block of the sort (128 KiB)
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

// 16 lanes per anchor row, 4 rows per wave at a time, runs of 16 consecutive rows per wave.  Lane j of a row group owns
// columns j, j+16, ...: it folds its own columns of each head in ascending order (first maximum wins), then four
// rotate steps (1,2,4,8) of a (value, index) max-combine leave every lane of the group with the head's result
// (the combine is commutative, associative and idempotent, so a rotate all-reduce is exact).
// The candidates of a run are appended to the image's key list with ONE atomic per wave (a returning atomic on one
// address costs ~50 ns and they serialise: 8400 per image at a 100 % pass rate was 0.45 ms of a 0.7 ms launch); the order
// of the list is irrelevant (all keys are distinct and sort_kernel orders them).
// The rows may be one pyramid level of the anchors only (detections-only forward, levels whose class predictors still go
// through a prediction scratch): RPI prediction rows per image = anchors anchor0 .. anchor0 + RPI of the N per image;
// write_box false leaves columns 0..11 of the candidate rows to the decode kernel that has written them already.
__global__ __launch_bounds__(256) void score_kernel(float* __restrict__ pred, int B, int RPI, int anchor0, int N, float conf_f,
                                                   float* __restrict__ rows, unsigned long long* __restrict__ keys,
                                                   int32_t* __restrict__ cnt, int NP, int write_box) {
    constexpr int NK = (NCOL + 15) / 16;   // 19 column slots per lane
    constexpr int RUN = 16;                // rows per wave and append
    const int j = threadIdx.x & 15, lane = threadIdx.x & 63;
    const int grp = lane >> 4, wv = threadIdx.x >> 6;
    const long long nrows = (long long)B * RPI;
    const long long nruns = (nrows + RUN - 1) / RUN;
    for (long long run = (long long)blockIdx.x * 4 + wv; run < nruns; run += (long long)gridDim.x * 4) {   // wave-uniform trip count
        unsigned long long mykey[RUN / 4];
        bool mypass[RUN / 4];
        int myimg[RUN / 4];
#pragma unroll
        for (int it = 0; it < RUN / 4; ++it) {
            const long long row0 = run * RUN + it * 4 + grp;
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
            const int bimg = (int)(row / RPI), n = anchor0 + (int)(row - (long long)bimg * RPI);
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
                float* out = rows + ((long long)bimg * N + n) * NDET;
                if (write_box || j >= 12) out[j] = o0;
                if (j < 12) out[16 + j] = o1;
            }
            mypass[it] = pass && j == 0;
            mykey[it] = score_key(sc, n);
            myimg[it] = bimg;
        }
        // ---- append the run's candidates ----
        const long long rlast = run * RUN + RUN - 1 < nrows ? run * RUN + RUN - 1 : nrows - 1;
        const int img0 = (int)(run * RUN / RPI), img1 = (int)(rlast / RPI);      // wave-uniform
        if (img0 == img1) {
            unsigned long long mk[RUN / 4];
            int total = 0;
#pragma unroll
            for (int it = 0; it < RUN / 4; ++it) {
                mk[it] = __ballot(mypass[it]);
                total += __popcll(mk[it]);
            }
            if (total) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&cnt[img0], total);
                base = __builtin_amdgcn_readfirstlane(base);
                int before = 0;
#pragma unroll
                for (int it = 0; it < RUN / 4; ++it) {
                    if (mypass[it]) keys[(long long)img0 * NP + base + before + __popcll(mk[it] & ((1ull << lane) - 1ull))] = mykey[it];
                    before += __popcll(mk[it]);
                }
            }
        } else {                                              // a run that straddles two images (RPI not a multiple of 16)
#pragma unroll
            for (int it = 0; it < RUN / 4; ++it)
                if (mypass[it]) {
                    const int pos = atomicAdd(&cnt[myimg[it]], 1);
                    keys[(long long)myimg[it] * NP + pos] = mykey[it];
                }
        }
    }
}

// One block per image: ascending bitonic sort of its candidate keys (all keys are distinct, so the result is
// unique whatever order the atomics appended them in).  Lists of up to SORT_LDS_KEYS keys are sorted inside LDS; longer
// ones (1280^2 inputs: up to 33 600 candidates) run the same network with LDS-sized blocks: every stage whose partner
// distance is below the block size runs on a block held in LDS, only the few steps with a longer distance touch global
// memory (a 65 536-key list: 6 of its 136 steps).
static constexpr int SORT_T = 1024;
template <typename P>
__device__ __forceinline__ void bitonic_step(P d, int count, int g0, int k, int j) {   // one compare-exchange step over d[0 .. count)
    for (int p = threadIdx.x; p < count / 2; p += SORT_T) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), ixj = i | j;
        const unsigned long long a = d[i], c = d[ixj];
        const bool up = ((g0 + i) & k) == 0;
        if ((a > c) == up) { d[i] = c; d[ixj] = a; }
    }
}
__global__ __launch_bounds__(SORT_T) void sort_kernel(unsigned long long* __restrict__ keys, const int32_t* __restrict__ cnt, int NP) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];
    const int b = blockIdx.x;
    const int nc = cnt[b];
    if (nc <= 1) return;
    int n = 64;
    while (n < nc) n <<= 1;
    unsigned long long* g = keys + (long long)b * NP;
    const int L = n < SORT_LDS_KEYS ? n : SORT_LDS_KEYS;
    // every stage up to k = L, block by block
    for (int blk = 0; blk < n; blk += L) {
        for (int i = threadIdx.x; i < L; i += SORT_T) skeys[i] = blk + i < nc ? g[blk + i] : ~0ull;
        __syncthreads();
        for (int k = 2; k <= L; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                bitonic_step(skeys, L, blk, k, j);
                __syncthreads();
            }
        const int lim = n > L ? L : nc;                        // a one-block list: the padding is not written back
        for (int i = threadIdx.x; i < lim; i += SORT_T) g[blk + i] = skeys[i];
        __syncthreads();
    }
    // the remaining stages: steps with partner distance >= L in global memory, the rest block by block in LDS
    for (int k = 2 * L; k <= n; k <<= 1) {
        for (int j = k >> 1; j >= L; j >>= 1) {
            bitonic_step(g, n, 0, k, j);
            __syncthreads();
        }
        for (int blk = 0; blk < n; blk += L) {
            for (int i = threadIdx.x; i < L; i += SORT_T) skeys[i] = g[blk + i];
            __syncthreads();
            for (int j = L >> 1; j > 0; j >>= 1) {
                bitonic_step(skeys, L, blk, k, j);
                __syncthreads();
            }
            for (int i = threadIdx.x; i < L; i += SORT_T) g[blk + i] = skeys[i];
            __syncthreads();
        }
    }
}

// torchvision's predicate `inter / (area_i + area_j - inter) > iou_threshold`, fp32 op by op, without the division for
// (nearly) every pair.  With U = fl(fl(area_i + area_j) - inter), t = thr_f (the largest fp32 <= the double threshold, so that
// (double)ovr > iou_thres <=> ovr > t) and ovr = fl(inter / U) (round to nearest): ovr > t  <=>  inter / U >= m (resp. > m),
// m = the midpoint of t and the next fp32 above it, i.e. m = t (1 + e) with 0 < e <= 2^-24.  Let p = fl(t U) = t U (1 + d),
// |d| <= 2^-24, p and t normal and positive (then U > 0):
//   inter > fl(p (1 + 2^-20))  =>  inter > t U (1 - 2^-24)^2 (1 + 2^-20) > t U (1 + 2^-21) > m U   =>  ovr > t;
//   inter < fl(p (1 - 2^-20))  =>  inter < t U (1 + 2^-24)^2 (1 - 2^-20) < t U < m U               =>  not.
// Only inside that 2^-19-wide band -- and for U <= 0, NaN, infinities, a zero threshold or a subnormal product, where every
// comparison below is false -- the IEEE division decides, as before.  An fp32 division is ~11 dependent VALU instructions, the
// two products and compares are 5.  lp_check_iou_predicate runs both forms side by side (tests/test_hip_kernels.py).
template <bool EXACT_ONLY = false>
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
    // disjoint boxes (most pairs): inter = 0 * h or w * 0 is 0 (or NaN for an infinite side), the quotient 0, -0 or NaN, and
    // none of them is > thr_f >= 0 -- the same answer without the division
    if (w == 0.f || h == 0.f) return false;
    const float inter = w * h;
    const float jarea = (jx2 - jx1) * (jy2 - jy1);
    const float u = iarea + jarea - inter;
    if (!EXACT_ONLY) {
        const float p = thr_f * u;
        const bool normal = p >= 1.17549435e-38f && thr_f >= 1.17549435e-38f;   // FLT_MIN: the bounds on p and on m need normal numbers
        if (normal && inter > p * 1.00000095367431640625f) return true;      // 1 + 2^-20
        if (normal && inter < p * 0.99999904632568359375f) return false;     // 1 - 2^-20
    }
    const float ovr = inter / u;
    return ovr > thr_f;   // thr_f = largest fp32 <= the double threshold  <=>  (double)ovr > iou_thres
}

// Verification hook: both forms of the predicate on n box pairs (pairs [n][8] = box i xyxy, box j xyxy); out[k] bit 0 = the
// product form the NMS kernels use, bit 1 = the plain division.
__global__ __launch_bounds__(256) void iou_predicate_kernel(const float* __restrict__ pairs, long long n, float thr_f, unsigned char* __restrict__ out) {
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256) {
        const float* q = pairs + k * 8;
        const float ia = (q[2] - q[0]) * (q[3] - q[1]);
        const bool a = iou_gt<false>(q[0], q[1], q[2], q[3], ia, q[4], q[5], q[6], q[7], thr_f);
        const bool b = iou_gt<true>(q[0], q[1], q[2], q[3], ia, q[4], q[5], q[6], q[7], thr_f);
        out[k] = (unsigned char)((a ? 1 : 0) | (b ? 2 : 0));
    }
}

// One block per image: greedy suppression over the sorted candidates, 64 at a time, with LAZY strikes: a chunk of
// candidates is tested against the list of boxes kept so far (<= max_det + 63 of them, cached in LDS) when its turn comes,
// instead of every chunk's survivors being applied to all later candidates at once (which walks the whole candidate list
// in global memory once per chunk).  The only serial part -- which candidates of a chunk survive each other -- is reduced
// to bit operations: every IoU test it needs has been made a chunk earlier, in parallel.  Per chunk c, between two barriers:
//   waves 1..  take chunk c + 1 (one candidate per lane; the box it is tested against is wave-uniform):
//              against the kept list as it stood at the barrier (a share of the list per wave, one broadcast ds_read_b128
//              per test)                                          -> dead[(c + 1) % 3]   bit j: candidate j is suppressed
//              against the 64 candidates of chunk c (v_readlane)  -> xt[(c + 1) & 1][j]  bit k: candidate k of chunk c overlaps j
//              against the earlier candidates of its own chunk    -> mt[(c + 1) & 1][j]  bit k: candidate k < j overlaps j
//   wave 0     takes chunk c: alive = not dead, and no survivor of chunk c - 1 in xt; then in candidate order: the first
//              alive candidate survives and clears the alive bit of every j whose mt has its bit (a shift, a compare and a
//              ballot per survivor); appends the survivors to the kept list.
// Nothing in the chunk loop waits for global memory: the sorted candidates (box + anchor number) are staged in LDS in
// batches of 1024 = 16 chunks, one candidate per thread, fetched a whole batch ahead; the kept list (box + anchor) lives in
// LDS up to kcap entries; the chunk barrier orders LDS only.  (With a global load per chunk the compiler's vmcnt(0) at the
// loop edge and the one inside __syncthreads put two memory round trips on the critical path of every chunk: 7.5 us per
// chunk, 0.86 ms for the 114 chunks of a 1280^2 image.)  Only a kept list longer than kcap (max_det > 4032) spills: survivor k
// goes to sb[k] / kept[k], and full barriers are used from then on.  Ends with the gather of the kept rows into det.
static constexpr int NMS_T = 1024, NMS_KCAP = 4096, NMS_BATCH = NMS_T / 64;     // kept boxes cached in LDS (20 B each); chunks per staged batch
typedef float box4 __attribute__((ext_vector_type(4)));   // (a native vector: selects and copies of HIP's float4 struct go through scratch)
__global__ __launch_bounds__(NMS_T) void greedy_kernel(const unsigned long long* __restrict__ keys, const float* __restrict__ rows,
                                                      float4* sbox, int32_t* kept,
                                                      const int32_t* __restrict__ cnt, int N, int NP, float thr_f, int max_det, int kcap,
                                                      float* __restrict__ det, int32_t* __restrict__ count,
                                                      int32_t* __restrict__ keep_out) {
    extern __shared__ __attribute__((aligned(16))) box4 kbox[];      // [kcap] the first kept boxes, then [kcap] their anchors
    __shared__ __attribute__((aligned(16))) box4 cbox[2][NMS_T];     // two staged batches of candidates
    __shared__ int canchor[2][NMS_T];
    __shared__ unsigned long long dead[3], xt[2][64], mt[2][64];     // dead: chunk c's mask lives in dead[c % 3] (read by everyone during iteration c)
    __shared__ int s_nk[2];
    int* const kanchor = (int*)(kbox + kcap);
    constexpr int NSW = NMS_T / 64 - 1;                // striking waves
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nc = cnt[b];
    if (nc > MAX_NMS) nc = MAX_NMS;
    const unsigned long long* kb = keys + (long long)b * NP;
    const float* rb = rows + (long long)b * N * NDET;
    box4* sb = (box4*)sbox + (long long)b * N;
    int32_t* kp = kept + (long long)b * N;
    const box4 none = {0.f, 0.f, 0.f, 0.f};
    const int nchunk = (nc + 63) / 64;

    // candidate tid of batch g: anchor number from its key, box from its row (rows are 112 B apart: 16-B aligned)
    // (two dependent loads, issued seven chunks apart in the loop so that neither is waited for where it is issued)
    auto fetch_anchor = [&](int g) { const int i = g * NMS_T + tid; return i < nc ? (int)(kb[i] & 0xffffffffu) : -1; };
    auto fetch_box = [&](int a_) { box4 v = none; if (a_ >= 0) v = *(const box4*)(rb + (long long)a_ * NDET); return v; };
    int fa = fetch_anchor(0);
    box4 fb = fetch_box(fa);
    cbox[0][tid] = fb; canchor[0][tid] = fa;
    if (tid < 3) dead[tid] = 0;
    if (tid < 128) { xt[tid >> 6][tid & 63] = 0; mt[tid >> 6][tid & 63] = 0; }
    __syncthreads();

    auto area_of = [](const box4& q) { return (q.z - q.x) * (q.w - q.y); };
    // (__builtin_bit_cast of a vector ELEMENT expression reads element 0 whatever the element: go through named floats)
    auto lane_f = [](float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); };
    auto bcast = [&](const box4& q, int k) {          // the box of lane k (k wave-uniform)
        const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
        const box4 o = {lane_f(qx, k), lane_f(qy, k), lane_f(qz, k), lane_f(qw, k)};
        return o;
    };
    // bit k of the result: candidate k of `from` (k in this wave's share; below `lane` if `earlier_only`) overlaps this lane's box
    // Candidates of `from` that the kept list has already suppressed (`gone`) can never survive, so what they overlap does not matter.
    auto pair_mask = [&](const box4& from, const box4& mine, bool earlier_only, unsigned long long gone) {
        unsigned long long m = 0;
        for (int k = wave - 1; k < 64; k += NSW) {
            if ((gone >> k) & 1ull) continue;                     // wave-uniform
            const box4 q = bcast(from, k);
            if ((!earlier_only || lane > k) && iou_gt(q.x, q.y, q.z, q.w, area_of(q), mine.x, mine.y, mine.z, mine.w, thr_f)) m |= 1ull << k;
        }
        return m;
    };
    auto chunk_box = [&](int c) { return cbox[(c / NMS_BATCH) & 1][(c % NMS_BATCH) * 64 + lane]; };   // (staged chunks only)

    box4 bx = chunk_box(0);                            // (waves 1..: the chunk before theirs)
    if (wave != 0) {                                   // chunk 0 against itself
        const unsigned long long m = pair_mask(bx, bx, true, 0ull);
        if (m) atomicOr(&mt[0][lane], m);
    }
    __syncthreads();
    int T = 0;                                         // kept through chunk c - 1
    unsigned long long kept_prev = 0;                  // wave 0: the survivors of chunk c - 1 (a mask over its candidates)
    bool spilled = false;                              // the kept list has outgrown the LDS cache: global memory carries part of it
    for (int c = 0; c < nchunk; ++c) {
        const int p = c & 1;
        if (c % NMS_BATCH == 0) fa = fetch_anchor(c / NMS_BATCH + 1);      // the next batch, a piece at a time
        if (c % NMS_BATCH == NMS_BATCH / 2 - 1) fb = fetch_box(fa);
        if (wave == 0) {
            bx = chunk_box(c);
            const int anchor = canchor[(c / NMS_BATCH) & 1][(c % NMS_BATCH) * 64 + lane];
            const int i = c * 64 + lane;
            const bool alive = i < nc && !((dead[c % 3] >> lane) & 1ull) && (xt[p][lane] & kept_prev) == 0;
            const unsigned long long mrow = mt[p][lane];
            if (lane == 0) dead[(c + 2) % 3] = 0;      // chunk c - 1's mask: nobody reads it any more; next written (for chunk c + 2) behind the barrier
            xt[p][lane] = 0;
            mt[p][lane] = 0;
            unsigned long long todo = __ballot(alive);
            unsigned long long keptmask = 0;
            while (todo) {
                const int k = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
                keptmask |= 1ull << k;
                todo &= ~(1ull << k);
                todo &= ~__ballot((mrow >> k) & 1ull);
            }
            if ((keptmask >> lane) & 1ull) {
                const int pos = T + __popcll(keptmask & ((1ull << lane) - 1ull));
                if (pos < kcap) { kbox[pos] = bx; kanchor[pos] = anchor; }
                else { sb[pos] = bx; kp[pos] = anchor; }
            }
            kept_prev = keptmask;
            if (lane == 0) s_nk[p] = __popcll(keptmask);
        } else if (c + 1 < nchunk) {
            const box4 pbx = bx;
            bx = chunk_box(c + 1);
            // a dead candidate needs no further test: every eight tests the wave publishes its findings, picks up those of
            // the other waves, and stops once none of its 64 candidates is left
            bool dd = (c + 1) * 64 + lane >= nc;
            int k = wave - 1;
            const int Tl = T < kcap ? T : kcap;
            auto test = [&](const box4& q) { return iou_gt(q.x, q.y, q.z, q.w, area_of(q), bx.x, bx.y, bx.z, bx.w, thr_f); };
            auto share = [&]() {                       // false: all 64 are dead
                const unsigned long long m = __ballot(dd);
                if (lane == 0 && m) atomicOr(&dead[(c + 1) % 3], m);
                dd = (dead[(c + 1) % 3] >> lane) & 1ull;
                return __ballot(!dd) != 0;
            };
            bool more = true;
            for (; more && k + 7 * NSW < Tl; k += 8 * NSW) {
                box4 q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = kbox[k + u * NSW];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (!dd) dd = test(q[u]);
                more = share();
            }
            if (more) {
                for (; k < Tl; k += NSW)
                    if (!dd) dd = test(kbox[k]);
                for (; k < T; k += NSW)
                    if (!dd) dd = test(sb[k]);
                const unsigned long long m = __ballot(dd);
                if (lane == 0 && m) atomicOr(&dead[(c + 1) % 3], m);
            }
            // pair tests only against candidates that can still survive: chunk c's mask is complete, chunk c + 1's as far as
            // the waves have published it (any bit set is final); this lane's own tests are skipped once it is gone itself
            const unsigned long long gone_c = dead[c % 3], gone_n = dead[(c + 1) % 3];
            const bool me_gone = (gone_n >> lane) & 1ull;
            unsigned long long xm = 0, mm = 0;
            if (!__builtin_expect(__ballot(!me_gone) == 0ull, 0)) {
                xm = pair_mask(pbx, bx, false, gone_c);
                mm = pair_mask(bx, bx, true, gone_n);
            }
            if (xm && !me_gone) atomicOr(&xt[p ^ 1][lane], xm);
            if (mm && !me_gone) atomicOr(&mt[p ^ 1][lane], mm);
        }
        if (c % NMS_BATCH == NMS_BATCH - 2) {           // -> the buffer whose last chunk was read a chunk ago
            const int g = c / NMS_BATCH + 1;
            cbox[g & 1][tid] = fb; canchor[g & 1][tid] = fa;
        }
        spilled = spilled || T + 128 > kcap;           // block-uniform
        if (spilled) __syncthreads();
        else {                                         // LDS only
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        T += s_nk[p];
        if (T >= max_det) break;                       // block-uniform
    }
    __syncthreads();
    const int total = T > max_det ? max_det : T;
    if (tid == 0) count[b] = total;
    auto anchor_of = [&](int k) { return k < kcap ? kanchor[k] : kp[k]; };
    box4* db = (box4*)(det + (long long)b * max_det * NDET);              // rows of 28 floats = 7 x 16 bytes
    constexpr int Q = NDET / 4;
    for (int i0 = tid; i0 < max_det * Q; i0 += 4 * NMS_T) {             // four independent row loads in flight
        box4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NMS_T, k = i / Q, q = i - k * Q;
            v[u] = none;
            if (k < total) v[u] = *(const box4*)(rb + (long long)anchor_of(k) * NDET + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * NMS_T < max_det * Q) db[i0 + u * NMS_T] = v[u];
    }
    if (keep_out)
        for (int k = tid; k < max_det; k += NMS_T) keep_out[(long long)b * max_det + k] = k < total ? anchor_of(k) : -1;
}

static int sort_greedy_launch(const NmsWs& w, int B, int N, float thr_f, int max_det, float* det, int32_t* count, int32_t* keep, hipStream_t st) {
    static std::atomic<unsigned long long> sort_attr{0}, greedy_attr{0};
    const int lds = (w.NP < SORT_LDS_KEYS ? w.NP : SORT_LDS_KEYS) * 8;
    if (int rc = set_max_lds_once(sort_kernel, SORT_LDS_KEYS * 8, sort_attr, "nms sort")) return rc;
    if (int rc = set_max_lds_once(greedy_kernel, NMS_KCAP * 20, greedy_attr, "nms greedy")) return rc;
    long long kcap = ((long long)max_det + 64 + 63) / 64 * 64;      // the kept list ends below max_det + 64
    if (kcap > NMS_KCAP) kcap = NMS_KCAP;
    hipLaunchKernelGGL(sort_kernel, dim3((unsigned)B), dim3(SORT_T), (size_t)lds, st, w.keys, w.cnt, w.NP);
    hipLaunchKernelGGL(greedy_kernel, dim3((unsigned)B), dim3(NMS_T), (size_t)kcap * 20, st, w.keys, w.rows, w.sbox, w.kept, w.cnt, N, w.NP,
                       thr_f, max_det, (int)kcap, det, count, keep);
    LP_HIP_CHECK(hipGetLastError());
    return LP_OK;
}

}  // namespace lp

using namespace lp;

namespace lp {
int nms_score_launch(float* pred, int B, int rows_per_img, int anchor0, int N, float conf_f, const NmsWs& w, bool write_box, hipStream_t st) {
    const long long nrows = (long long)B * rows_per_img;
    long long blocks = ((nrows + 15) / 16 + 3) / 4;       // a wave per run of 16 rows, four waves per block
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

extern "C" const int32_t* lp_nms_candidate_counts(const void* workspace, int B, int N) {
    if (!workspace || B < 1 || N < 1) return nullptr;
    return nms_carve(const_cast<void*>(workspace), B, N).cnt;
}

extern "C" int lp_check_iou_predicate(const float* dev_pairs, long long n, double iou_thres, unsigned char* dev_out, void* stream) {
    if (!dev_pairs || !dev_out || n < 1) return fail(LP_ERR_ARG, "lp_check_iou_predicate: bad argument");
    if (!(iou_thres >= 0.0 && iou_thres <= 1.0)) return fail(LP_ERR_ARG, "lp_check_iou_predicate: threshold must be in [0, 1]");
    float thr_f = (float)iou_thres;
    if ((double)thr_f > iou_thres) thr_f = nextafterf(thr_f, -INFINITY);
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(iou_predicate_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dev_pairs, n, thr_f, dev_out);
    LP_HIP_CHECK(hipGetLastError());
    return LP_OK;
}

// Zeroes the per-image candidate counters.  A kernel of our own, not hipMemsetAsync: the runtime's fill is a blit kernel with
// barrier packets around it (the device sat idle 16-38 us in front of it at every step boundary, tools/micro/fwd_idle.py).
namespace lp {
__global__ void zero_counts_kernel(int* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}
int zero_counts_launch(int* p, int n, hipStream_t st) {
    hipLaunchKernelGGL(zero_counts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n);
    LP_HIP_CHECK(hipGetLastError());
    return LP_OK;
}
}  // namespace lp

extern "C" int lp_nms(float* pred, int B, int N, double conf_thres, double iou_thres, int max_det, float* det,
                      int32_t* count, int32_t* keep, void* workspace, size_t workspace_bytes, void* stream) {
    if (!pred || !det || !count || !workspace) return fail(LP_ERR_ARG, "lp_nms: null pointer");
    if (B < 1 || N < 1 || max_det < 1) return fail(LP_ERR_ARG, "lp_nms: B, N and max_det must be positive");
    if (!(conf_thres >= 0.0 && conf_thres <= 1.0) || !(iou_thres >= 0.0 && iou_thres <= 1.0))
        return fail(LP_ERR_ARG, "lp_nms: thresholds must be in [0, 1]");
    if (((uintptr_t)workspace & 255) != 0) return fail(LP_ERR_ARG, "lp_nms: workspace must be 256-byte aligned");
    if (((uintptr_t)det & 15) != 0) return fail(LP_ERR_ARG, "lp_nms: det must be 16-byte aligned");
    NmsWs w = nms_carve(workspace, B, N);
    if (workspace_bytes < w.bytes) return fail(LP_ERR_ARG, "lp_nms: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const float conf_f = (float)conf_thres;
    float thr_f = (float)iou_thres;                       // largest fp32 not above the double threshold
    if ((double)thr_f > iou_thres) thr_f = nextafterf(thr_f, -INFINITY);

    if (int rc = zero_counts_launch(w.cnt, B, st)) return rc;
    if (int rc = nms_score_launch(pred, B, N, 0, N, conf_f, w, true, st)) return rc;
    return sort_greedy_launch(w, B, N, thr_f, max_det, det, count, keep, st);
}

extern "C" int lp_nms_candidates(int B, int N, double iou_thres, int max_det, float* det, int32_t* count, int32_t* keep,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!det || !count || !workspace) return fail(LP_ERR_ARG, "lp_nms_candidates: null pointer");
    if (B < 1 || N < 1 || max_det < 1) return fail(LP_ERR_ARG, "lp_nms_candidates: B, N and max_det must be positive");
    if (!(iou_thres >= 0.0 && iou_thres <= 1.0)) return fail(LP_ERR_ARG, "lp_nms_candidates: threshold must be in [0, 1]");
    if (((uintptr_t)workspace & 255) != 0) return fail(LP_ERR_ARG, "lp_nms_candidates: workspace must be 256-byte aligned");
    if (((uintptr_t)det & 15) != 0) return fail(LP_ERR_ARG, "lp_nms_candidates: det must be 16-byte aligned");
    NmsWs w = nms_carve(workspace, B, N);
    if (workspace_bytes < w.bytes) return fail(LP_ERR_ARG, "lp_nms_candidates: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float thr_f = (float)iou_thres;                       // largest fp32 not above the double threshold
    if ((double)thr_f > iou_thres) thr_f = nextafterf(thr_f, -INFINITY);
    return sort_greedy_launch(w, B, N, thr_f, max_det, det, count, keep, st);
}
