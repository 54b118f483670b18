/* ORACLE (test infrastructure, not product code).
 *
 * Plain-C restatement of YOLO-LP's post-processing for ONE image:
 *   yolov6/utils/nms.py:68-125  (non_max_suppression body)  and
 *   torchvision.ops.nms         (called at nms.py:121; third-party, torchvision>=0.9.0
 *                                per requirements.txt:5, NOT present under /root/reference and not
 *                                installed in this image).
 *
 * Parity status:
 *   - rows / columns / mask / score / max_det / ordering (nms.py:76-125): PINNED by golden vectors produced
 *     by running the reference's own non_max_suppression (tests/golden/make_golden.py).
 *   - the greedy IoU selection itself (torchvision CPU kernel nms_kernel_impl): "parity unpinned" -- it is
 *     restated here from the published algorithm: stable descending sort of the scores, areas =
 *     (x2-x1)*(y2-y1), for each unsuppressed i in order suppress every later j with
 *     inter/(area_i+area_j-inter) > iou_threshold (strict, compared in double since iou_threshold is a
 *     double there), inter = max(0,min(x2)-max(x1))*max(0,min(y2)-max(y1)), all fp32.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile). fp32 arithmetic
 * is done op by op exactly as the torch expressions do (no FMA contraction, left-to-right sums).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define LP_DET_COLS 28
static const int SEG[9] = {13, 44, 68, 105, 142, 179, 216, 253, 290};  /* nms.py:81-88 */

typedef struct { float score; int idx; } cand_t;

/* descending score, ties by ascending original index (= stable sort, descending) */
static int cand_cmp(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/* torch.max(x[:, a:b], 1): value and FIRST maximal index; a NaN wins and sticks (ATen max kernel). */
static void seg_max(const float* row, int a, int b, float* val, int* idx) {
    float m = row[a]; int mi = 0;
    for (int c = a + 1; c < b; ++c) {
        float v = row[c];
        if (!(m != m) && (v > m || v != v)) { m = v; mi = c - a; }
    }
    *val = m; *idx = mi;
}

/* pred: [n, ncols] fp32 (ncols = 290 for the LP head); MUTATED in place like nms.py:76.
 * rows_out: [max_det, 28]; keep_out: [max_det] original anchor indices of the kept rows (may be NULL).
 * Returns the number of detections written (<= max_det), or -1 on bad arguments. */
int lp_post_ref_image(float* pred, int n, int ncols, double conf_thres, double iou_thres, int max_det,
                      int max_nms, float* rows_out, int* keep_out) {
    if (ncols != SEG[8] || n < 0 || max_det < 0) return -1;
    const float conf_f = (float)conf_thres;          /* torch casts the python scalar to the tensor dtype */
    float* det = (float*)malloc((size_t)(n > 0 ? n : 1) * LP_DET_COLS * sizeof(float));
    cand_t* cand = (cand_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(cand_t));
    int nc = 0;
    for (int i = 0; i < n; ++i) {
        float* r = pred + (size_t)i * ncols;
        const float obj = r[4];
        for (int c = 13; c < ncols; ++c) r[c] = r[c] * obj;                  /* nms.py:76 */
        float* d = det + (size_t)i * LP_DET_COLS;
        d[0] = r[0] - r[2] / 2; d[1] = r[1] - r[3] / 2;                      /* xywh2xyxy, nms.py:21-28 */
        d[2] = r[0] + r[2] / 2; d[3] = r[1] + r[3] / 2;
        for (int c = 0; c < 8; ++c) d[4 + c] = r[5 + c];                     /* corners, nms.py:94 */
        float cf[8]; int ci[8];
        for (int s = 0; s < 8; ++s) seg_max(r, SEG[s], SEG[s + 1], &cf[s], &ci[s]);
        for (int s = 0; s < 8; ++s) { d[12 + s] = cf[s]; d[20 + s] = (float)ci[s]; }
        /* nms.py:90-91 -- ad4 twice, ad5 omitted (reference quirk, reproduced) */
        float m = cf[0] + cf[1]; m = m + cf[2]; m = m + cf[3]; m = m + cf[4]; m = m + cf[5];
        m = m + cf[6]; m = m + cf[6]; m = m / 8.0f;
        if (m >= conf_f) {
            /* nms.py:120 -- score uses all eight, left to right */
            float s = cf[0] + cf[1]; s = s + cf[2]; s = s + cf[3]; s = s + cf[4]; s = s + cf[5];
            s = s + cf[6]; s = s + cf[7]; s = s / 8.0f;
            cand[nc].score = s; cand[nc].idx = i; ++nc;
        }
    }
    /* stable descending order; nms.py:115-116 keeps the best max_nms first (its argsort is unstable, so only
     * tie-free inputs are comparable there) */
    qsort(cand, (size_t)nc, sizeof(cand_t), cand_cmp);
    if (nc > max_nms) nc = max_nms;
    unsigned char* sup = (unsigned char*)calloc((size_t)(nc > 0 ? nc : 1), 1);
    int kept = 0;
    for (int a = 0; a < nc && kept < max_det; ++a) {                          /* nms.py:122-123: keep[:max_det] */
        if (sup[a]) continue;
        const float* bi = det + (size_t)cand[a].idx * LP_DET_COLS;
        memcpy(rows_out + (size_t)kept * LP_DET_COLS, bi, LP_DET_COLS * sizeof(float));
        if (keep_out) keep_out[kept] = cand[a].idx;
        ++kept;
        const float iarea = (bi[2] - bi[0]) * (bi[3] - bi[1]);
        for (int b = a + 1; b < nc; ++b) {
            if (sup[b]) continue;
            const float* bj = det + (size_t)cand[b].idx * LP_DET_COLS;
            const float xx1 = bi[0] > bj[0] ? bi[0] : bj[0];
            const float yy1 = bi[1] > bj[1] ? bi[1] : bj[1];
            const float xx2 = bi[2] < bj[2] ? bi[2] : bj[2];
            const float yy2 = bi[3] < bj[3] ? bi[3] : bj[3];
            float w = xx2 - xx1; if (!(w > 0.0f)) w = 0.0f;
            float h = yy2 - yy1; if (!(h > 0.0f)) h = 0.0f;
            const float inter = w * h;
            const float jarea = (bj[2] - bj[0]) * (bj[3] - bj[1]);
            const float ovr = inter / (iarea + jarea - inter);
            if ((double)ovr > iou_thres) sup[b] = 1;
        }
    }
    free(sup); free(cand); free(det);
    return kept;
}

/* batch wrapper: pred [B, n, ncols]; rows_out [B, max_det, 28]; counts [B]; keep_out [B, max_det] or NULL */
int lp_post_ref(float* pred, int B, int n, int ncols, double conf_thres, double iou_thres, int max_det,
                float* rows_out, int* counts, int* keep_out) {
    for (int b = 0; b < B; ++b) {
        int k = lp_post_ref_image(pred + (size_t)b * n * ncols, n, ncols, conf_thres, iou_thres, max_det, 30000,
                                  rows_out + (size_t)b * max_det * LP_DET_COLS,
                                  keep_out ? keep_out + (size_t)b * max_det : 0);
        if (k < 0) return k;
        counts[b] = k;
    }
    return 0;
}
