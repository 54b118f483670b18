/* lp_hip.h -- C ABI of libyololp_hip.so: the MI355X (gfx950) implementation of YOLO-LP's detection
 * forward + decode + NMS hot path.
 *
 * The reference (KyleHuang9/YOLO-LP) is pure Python/PyTorch and has no FFI of its own; its boundary for this
 * path is the Python module API.  This header is the C-ABI a binding for that API
 * sits on: plain pointers and sizes, no torch types.  Each entry point names the reference interface it
 * replaces (file:line relative to the reference repository).
 *
 * Conventions
 *   - every function returns LP_OK (0) or a negative lp_status; lp_last_error() gives the message of the
 *     last failure on the calling thread.  Nothing here falls back to a CPU path.
 *   - all device pointers are caller-owned (the Python host allocates them as torch tensors); the library
 *     allocates no device memory.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - activations inside an engine are NHWC with the channel count padded to a multiple of 8; the element
 *     type is the engine's activation dtype (lp_dtype).  Accumulation is always fp32 (MFMA).
 */
#ifndef LP_HIP_H
#define LP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { LP_OK = 0, LP_ERR_ARG = -1, LP_ERR_STATE = -2, LP_ERR_HIP = -3, LP_ERR_UNSUPPORTED = -4 } lp_status;
typedef enum { LP_F16 = 0, LP_BF16 = 1, LP_F32 = 2 } lp_dtype;
typedef enum { LP_ACT_NONE = 0, LP_ACT_RELU = 1, LP_ACT_SILU = 2 } lp_act;

#define LP_PRED_COLS 290   /* 4 xywh + 1 obj + 8 corners + 31 + 24 + 6*37 (effidehead.py:283-301) */
#define LP_DET_COLS 28     /* xyxy + 8 corners + 8 conf + 8 idx (nms.py:94-96) */
#define LP_MAX_SRC 4       /* inputs of one conv that are read as a channel concat without materialising it */

typedef struct lp_engine lp_engine;

const char* lp_version(void);
const char* lp_last_error(void);

/* ---------------------------------------------------------------------------------------------------
 * Engine: a static graph of fused conv kernels over NHWC tensors, built once per model by the host from the
 * model's folded weights, then run per batch.  Replaces Model.forward = backbone -> neck -> detect
 * (yolov6/models/yolo.py:32-40) and everything below it (layers/common.py, models/efficientrep.py,
 * models/reppan.py, models/effidehead.py:214-301, assigners/anchor_generator.py:11-31,
 * utils/general.py:29-66).
 * ------------------------------------------------------------------------------------------------- */
int lp_engine_create(lp_engine** out, int act_dtype /* lp_dtype */);
void lp_engine_destroy(lp_engine* e);

/* Declare an activation tensor with `channels` channels at 1/2^stride_log2 of the input resolution.
 * Returns its id (>= 0) or a negative lp_status. */
int lp_engine_tensor(lp_engine* e, int channels, int stride_log2);

/* Execution lane (0..4) of the ops added from now on.  Lane 0 is the caller's stream; lanes 1..4 are side
 * streams of the engine.  Ops on different lanes may overlap; the engine derives every cross-lane dependency from
 * the tensors the ops read and write and forks / joins the side streams inside lp_engine_forward, so callers still
 * see one in-order forward on `stream`.  Used for the independent branches of BiFusion (common.py:523-527) and the
 * per-level towers of the head (effidehead.py:228-245), which a builder may add right behind the neck layer that feeds them so
 * that they run under the rest of the neck. */
int lp_engine_set_lane(lp_engine* e, int lane);

/* The network input: caller's NCHW image batch [B,3,H,W] -> tensor `dst` (declared with 3 channels,
 * stride_log2 0).  Must be the first op. */
int lp_engine_add_input(lp_engine* e, int dst);

typedef struct lp_conv_desc {
    int n_src;              /* 1..LP_MAX_SRC; the sources are concatenated along channels in this order */
    int src[LP_MAX_SRC];
    int dst;
    int ksize;              /* 1 or 3 (padding ksize/2) */
    int stride;             /* 1 or 2 */
    int act;                /* lp_act, applied after bias */
    int res;                /* tensor id added after the activation (out = act(conv+b) + res_alpha*res), or -1 */
    float res_alpha;
    const float* weight;    /* host, fp32, [Cout][Cin_total][k][k]  (torch Conv2d.weight, BN already folded) */
    const float* bias;      /* host, fp32, [Cout] */
    int dst2;               /* <= 0 (tensor 0 is the network input, never a destination): none; else a second destination: TWO sibling layers of the reference that read the same input with the same
                             * kernel size, stride and activation (the class and box towers of a head level, effidehead.py:232-244; cv1 /
                             * cv2 of SimCSPSPPF and BepC3, common.py:139-141,497-500) run as ONE launch: weight / bias hold the rows of the
                             * first layer (dst's channels, a multiple of 8) followed by those of the second (dst2's).  No residual. */
} lp_conv_desc;

/* act(conv(cat(src...)) + bias) [+ alpha*res].  Replaces RepVGGBlock deploy forward (common.py:258-259),
 * Conv/SimConv/Conv_C3.forward_fuse (common.py:41-42,65-66,475-476), the torch.cat feeding them
 * (common.py:146-147,499,527; reppan.py:227-232) and BottleRep's residual (common.py:455). */
int lp_engine_add_conv(lp_engine* e, const lp_conv_desc* d);

/* 2x2 stride-2 transposed conv with bias: Transpose.forward (common.py:186-187).
 * weight: host fp32 [Cin][Cout][2][2] (torch ConvTranspose2d.weight), bias [Cout]. */
int lp_engine_add_deconv2x2(lp_engine* e, int src, int dst, const float* weight, const float* bias);

/* Three chained 5x5 stride-1 pad-2 max pools (windows 5/9/13): the `self.m` calls of
 * SimCSPSPPF/SimSPPF.forward (common.py:101-103,144-146).  dst1 = m(src), dst2 = m(dst1), dst3 = m(dst2). */
int lp_engine_add_pool5_chain(lp_engine* e, int src, int dst1, int dst2, int dst3);

/* Classification predictors of one pyramid level: the eight 1x1 convs + sigmoid of Detect.forward
 * (effidehead.py:235-242,251-258) as ONE contraction; writes columns [13,13+n_cls) of the level's rows of
 * pred.  weight: host fp32 [n_cls][C] (the eight predictor weights stacked in head order), bias [n_cls]. */
int lp_engine_add_head_cls(lp_engine* e, int src, int level, int n_cls, const float* weight, const float* bias);

/* Box + corner predictors of one level and the anchor-free decode: reg_preds / cor_preds 1x1 convs
 * (effidehead.py:244-245), optional DFL softmax-projection (:247-249, reg_bins = reg_max+1 = 17, proj =
 * proj_conv.weight[17]), generate_anchors (anchor_generator.py:11-31), dist2bbox 'xywh' and dist2cor
 * (general.py:29-40,51-66), * stride (effidehead.py:285-286) and the objectness column of ones (:290).
 * Writes columns [0,13) of the level's rows.  weight: host fp32 [4*reg_bins + 8][C], bias likewise.
 * reg_bins = 1 means plain ltrb distances (use_dfl False). */
int lp_engine_add_head_box(lp_engine* e, int src, int level, int reg_bins, const float* weight, const float* bias,
                           const float* proj);

/* Freeze the graph and pack the weights for the MFMA kernels (host side). */
int lp_engine_finalize(lp_engine* e, int n_levels);
size_t lp_engine_weight_bytes(const lp_engine* e);
/* Copy the packed weights into caller-owned device memory (>= lp_engine_weight_bytes, 256-B aligned).  The buffer must
 * stay alive and writable while the engine is used: its first 256 bytes are the kernels' zero page and a 16-byte scratch
 * granule (target of stores that must not land anywhere else). */
int lp_engine_upload(lp_engine* e, void* dev_weights, void* stream);

/* Activation arena for a given input shape (H, W multiples of 32). */
size_t lp_engine_arena_bytes(const lp_engine* e, int B, int H, int W);
int lp_engine_bind(lp_engine* e, void* dev_arena, size_t bytes, int B, int H, int W);
/* Placement of tensor `id` inside the bound arena: byte offset, logical channels, stored channels (padded
 * to 8), height, width.  Layout is [B][h][w][c_stored] of the activation dtype. */
int lp_engine_tensor_info(const lp_engine* e, int id, size_t* offset, int* c, int* c_stored, int* h, int* w);
/* Rows of pred per image (sum over levels of h*w) for the bound shape. */
int lp_engine_num_anchors(const lp_engine* e);

/* Run the graph on the bound shape.  x: device [B,3,H,W] of x_dtype (lp_dtype); pred: device fp32
 * [B, num_anchors, LP_PRED_COLS].  Model.forward's first return value (yolo.py:40); the second one (the
 * three neck maps) stays in the arena (lp_engine_tensor_info). */
int lp_engine_forward(lp_engine* e, const void* x, int x_dtype, float* pred, void* stream);

/* Detections-only forward for callers that go straight on to NMS (Inferer, Evaler.predict, the benchmark): the head does not
 * write the [B,N,290] prediction tensor -- its class-predictor kernels apply non_max_suppression's candidate selection
 * (nms.py:76-96: obj * cls, eight (max, first arg-max), the confidence mask, the score) to the sigmoids while they are still
 * on chip and append the passing anchors to the candidate lists in `workspace` (lp_nms_workspace_bytes(B, N) bytes, 256-byte
 * aligned).  lp_nms_candidates(workspace, ...) on the same stream finishes the job; det / count / keep are bit-identical to
 * lp_nms on the prediction tensor lp_engine_forward would have written.  Replaces Model.forward + the first half of
 * non_max_suppression (yolo.py:32-40, effidehead.py:283-301, nms.py:68-96). */
int lp_engine_forward_det(lp_engine* e, const void* x, int x_dtype, double conf_thres, void* workspace, size_t workspace_bytes,
                          void* stream);

/* enable != 0: lp_engine_forward / lp_engine_forward_det capture their launches into a hipGraph on first use with a given
 * (x, pred or workspace + threshold, dtype, launch geometry) and replay it afterwards; any other pointers re-capture.  For
 * launch-bound shapes (batch 1). */
int lp_engine_set_graph(lp_engine* e, int enable);

/* enable != 0 (default): the forwards issue every kernel on the caller's stream in op order; 0: independent branches of the
 * graph (lp_engine_set_lane) run on side streams forked from / joined into the caller's stream.  With several forwards in flight
 * on several streams the fork / join events cost more than the lanes hide (six batches in flight: 16.2 k images/s on one lane
 * each against 14.9 k; profiles/r03_inflight_lanes.txt); with ONE forward in flight the lanes were worth +2.6 % until the head
 * kernels got shorter -- final state of round 3: one lane +1.2 % at batch 32, +4.5 % at batch 1, lanes +1.8 % on yolov6m 1280x1280
 * (profiles/r03_round_ab.txt).  Results do not depend on it. */
int lp_engine_set_single_lane(lp_engine* e, int enable);

/* MFMA family of the 3x3 layers (call before lp_engine_finalize; default: enabled, LP_NO_MFMA16=1 in the environment
 * disables; stride-2 layers: LP_VARIANT_PIPE16_S2* below).  enable != 0: every 16-bit 3x3 stride-1 MODE_ACT layer whose 16-channel K-chunks are a multiple of four (input channels
 * a multiple of 64) and whose weights are packed in 128-row cout tiles (more than 64 stored output channels) runs on v_mfma_f32_16x16x32 (LP_VARIANT_PIPE16_*:
 * tiles of any number of 16-pixel blocks fill the 256 CUs evenly where 256-pixel tiles leave 22 % of the MFMA slots empty).  That
 * family adds the products in another fp32 order than all other variants: results agree to rounding, not bit for bit.  The choice is
 * a function of the layer alone -- never of the batch size, the input size or a timing -- so image k of a batch equals image k
 * alone, and inside the family the autotuner's variants are bit-identical.  enable == 0: every layer on the 32x32x16 family. */
int lp_engine_set_mfma16(lp_engine* e, int enable);

/* Introspection for benchmarks: ops of the frozen graph and per-op device time (hipEvent pairs on
 * `stream`, one untimed warm run first).  op_ms has lp_engine_num_ops() entries (milliseconds). */
int lp_engine_num_ops(const lp_engine* e);
/* kind: 0 input, 1 conv, 2 deconv, 3 pool, 4 head_cls, 5 head_box.  flops = 2*MAC of the op on the bound
 * shape; bytes = algorithmic activation bytes read + written (each tensor once) + weight bytes. */
int lp_engine_op_info(const lp_engine* e, int op, int* kind, int* ksize, int* cin, int* cout, double* flops,
                      double* bytes);
int lp_engine_profile(lp_engine* e, const void* x, int x_dtype, float* pred, void* stream, float* op_ms, int reps);
/* Same, with `inner` back-to-back launches of every op between its two events (op_ms = event time / inner): the cost of
 * the event pair itself is amortised, so op_ms approaches the kernel duration a rocprofv3 kernel trace reports. */
int lp_engine_profile_ops(lp_engine* e, const void* x, int x_dtype, float* pred, void* stream, float* op_ms, int reps, int inner);

/* Pick, per conv op and for the bound shape, the fastest of the kernel variants (workgroup tile / LDS ring depth)
 * that share the op's weight packing, by timing each in place (hipEvent pairs on `stream`).  The choice is
 * remembered per (B,H,W) and re-applied by lp_engine_bind.  Variants differ only in tiling: every output element
 * is the same fp32 sum in the same K order, so results do not depend on the choice.  lp_engine_op_variant reports
 * the current choice of an op: cfg = a workgroup tile of the implicit-GEMM kernel (0..5 = A..F) with
 * nbuf = LDS ring depth 1 or 2, or LP_VARIANT_STREAM64 / LP_VARIANT_STREAM128 (streaming 1x1 kernel with 64 / 128 output
 * channels per wave, nbuf 2), or LP_VARIANT_ROWS (row-writer form of a head_cls op, nbuf 1), or LP_VARIANT_PIPE_* (pipelined
 * 3x3 stride-1 kernel, nbuf 3).  LP_VARIANT_PIPE_P belongs to the stem only (op 1, over the space-to-depth image the input op
 * writes): with it the stem gathers its pixels from the caller's NCHW frame itself and the input op is skipped, whenever the
 * frame passed to lp_engine_forward has the engine's 16-bit dtype (other dtypes: input op + the stem's other variant).
 * lp_engine_set_op_variant forces one (tests, experiments): LP_ERR_UNSUPPORTED if it does not fit the op. */
enum { LP_VARIANT_STREAM64 = 16, LP_VARIANT_STREAM128 = 17, LP_VARIANT_ROWS = 18,
       /* pipelined 3x3 stride-1 kernel (persistent, 3-slot LDS ring, nbuf 3): 128 couts x 256 px, 64 x 512, 128 x 128, 32 x 512 */
       LP_VARIANT_PIPE_D = 32, LP_VARIANT_PIPE_B = 33, LP_VARIANT_PIPE_F = 34, LP_VARIANT_PIPE_C = 35,
       LP_VARIANT_PIPE_P = 36 /* 32 x 512, the stem reading the NCHW frame (see above) */,
       LP_VARIANT_PIPE16_D = 39, LP_VARIANT_PIPE16_F = 41 /* PIPE_D / PIPE_F on v_mfma_f32_16x16x32 (128-row packing, layers with a multiple of four
                                       * number of 16-channel K-chunks; nbuf 3).  ANOTHER fp32 summation order: equal to the other variants to
                                       * rounding, not bit for bit -- see lp_engine_set_mfma16 */,
       LP_VARIANT_PIPE16_V0 = 42, LP_VARIANT_PIPE16_V1 = 43 /* the same sums as PIPE16_* with tiles of any number of 16-pixel blocks (128 couts x
                                       * <= 448 px as 2 x 4 waves, 128 x <= 224 as 4 x 2): tiles that fill the persistent grid evenly */,
       LP_VARIANT_FUSED_STEM2 = 37 /* op 2 only (3x3 stride 2 behind the stem, <= 64 channels): input op + stem + this layer as ONE kernel
                                      whenever the frame has the engine's 16-bit dtype; the stem's output never reaches memory */,
       LP_VARIANT_FUSED_PW_S2 = 38 /* a 3x3 stride-2 layer (<= 64 channels) whose input comes from a 1x1 layer (64 -> <= 64 channels) that nobody
                                      else reads (BiFusion's downsample(cv2(x)), common.py:504-527): the two as ONE kernel; the 1x1 op is skipped */ };
/* LP_VARIANT_FUSED_BIFUSION (45): the cv3 of a 64-channel BiFusion level (common.py:504-527: 1x1 ReLU over [upsample(x0), cv1(x1), d]) together with
 * the transposed convolution and cv1 as ONE kernel (lp_bifusion_fused.inc): their outputs never reach memory; the two carried ops are skipped.
 * Bit-identical to the three launches.  lp_engine_op_carrier: index of the op whose kernel currently carries `op` (fused stem, fused 1x1 + 3x3
 * stride 2, fused BiFusion; frame_direct != 0: the caller's frame has the engine's dtype), or -1: the op launches its own kernel. */
enum { LP_VARIANT_FUSED_BIFUSION = 45 };
/* LP_VARIANT_BOX_SPARSE (46, the default) / LP_VARIANT_BOX_DENSE (47): a head_box op in the detections-only forward (lp_engine_forward_det) computes
 * reg_preds / cor_preds + dist2bbox / dist2cor (effidehead.py:262-301, general.py:29-66) for the anchors the level's class predictors let pass only,
 * or for every anchor.  The candidate rows lp_nms_candidates reads are the same bits either way; the sparse form needs one execution lane and the
 * op order cls(level), box(level) per level (else the dense form runs whatever is set).  nbuf is ignored. */
enum { LP_VARIANT_BOX_SPARSE = 46, LP_VARIANT_BOX_DENSE = 47 };
/* LP_VARIANT_PIPE16_S2A (48) / _S2B (49): a 3x3 STRIDE-2 layer with 128-row weight packing (more than 64 stored output channels, one
 * destination, 16-bit) on v_mfma_f32_16x16x32: persistent workgroups, three halo + two weight LDS slots (the halo requested two chunks ahead, counted vmcnt), tiles of any number of 16-pixel blocks (128 couts x
 * <= 256 px as 2 x 4 waves, 128 x <= 224 as 4 x 2), nbuf 3 (lp_conv3x3_s2p16.inc; reference: efficientrep.py:57-117, common.py:258-259).
 * Another fp32 summation order than conv_mfma_kernel<KS=3,S=2>'s (equal to rounding, not bit for bit).  Measured 0-9 % faster than that
 * kernel per layer and equal over the whole step (profiles/r04_s2p16_convbench.txt, r04_experiments.txt 15): NOT a default and never picked by the autotuner -- lp_engine_set_op_variant selects it for
 * one op; an engine created under LP_S2P16=1 runs every eligible layer on it (by rule, never by timing). */
enum { LP_VARIANT_PIPE16_S2A = 48, LP_VARIANT_PIPE16_S2B = 49 };
int lp_engine_op_carrier(const lp_engine* e, int op, int frame_direct);
int lp_engine_autotune(lp_engine* e, const void* x, int x_dtype, float* pred, void* stream, int reps);
int lp_engine_op_variant(const lp_engine* e, int op, int* cfg, int* nbuf);
int lp_engine_set_op_variant(lp_engine* e, int op, int cfg, int nbuf);
/* Copy the tuned choices (all shapes) of an engine built from the same graph and dtype, e.g. to the other engines of a
 * several-batches-in-flight pipeline, instead of tuning each. */
int lp_engine_copy_tuning(lp_engine* dst, const lp_engine* src);

/* ---------------------------------------------------------------------------------------------------
 * Post-processing.  Replaces non_max_suppression (yolov6/utils/nms.py:31-130) including its call of
 * torchvision.ops.nms (:121).  All images of the batch are processed by one set of launches.
 *   pred      device fp32 [B,N,290]; columns 13.. are multiplied by column 4 IN PLACE (nms.py:76)
 *   conf/iou  thresholds as the python floats the reference receives (conf is compared in fp32, iou in
 *             double, as torch / torchvision do)
 *   det       device fp32 [B,max_det,28] (16-byte aligned), rows in descending-score order; rows >= count[b] are zero
 *   count     device int32 [B]
 *   keep      device int32 [B,max_det] anchor index of every kept row, or NULL
 *   workspace device scratch of at least lp_nms_workspace_bytes(B,N) bytes
 * ------------------------------------------------------------------------------------------------- */
size_t lp_nms_workspace_bytes(int B, int N);
int lp_nms(float* pred, int B, int N, double conf_thres, double iou_thres, int max_det, float* det, int32_t* count,
           int32_t* keep, void* workspace, size_t workspace_bytes, void* stream);

/* Second half of lp_nms for candidate lists already in `workspace` (written by lp_engine_forward_det): stable descending sort,
 * the > 30000 cut (nms.py:115-116), greedy IoU suppression (torchvision.ops.nms), max_det (nms.py:121-125). */
int lp_nms_candidates(int B, int N, double iou_thres, int max_det, float* det, int32_t* count, int32_t* keep, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Device int32 [B] inside `workspace`: how many anchors of each image passed the confidence mask (nms.py:90-96) in the last
 * lp_nms / lp_engine_forward_det that used it (valid once that call's stream work is done).  Callers use it as a cheap
 * estimate of the candidate density when choosing between the two forms of the path; nothing in the results depends on it. */
const int32_t* lp_nms_candidate_counts(const void* workspace, int B, int N);

/* ---------------------------------------------------------------------------------------------------
 * Callers either side of the path.
 *
 * lp_preprocess_letterbox: Inferer.precess_image (yolov6/core/inferer.py:191-201) with letterbox
 * (yolov6/data/data_augment.py:30-61) for one frame.  img: device uint8 [h0,w0,3] BGR (cv2.imread layout);
 * out: device [3,H,W] of out_dtype, RGB, /255.  The frame is resized (bilinear, OpenCV INTER_LINEAR fixed-point
 * scheme) to rh x rw, placed at (top,left) and surrounded by 114.  rh == h0 && rw == w0 means no resize.
 *
 * lp_rescale_round: Inferer.rescale (inferer.py:203-228) + .round() (:100) on columns 0..11 of n detection rows
 * (row stride 28 floats), in place: v = round_half_even(clamp((v - pad) / ratio, 0, img_w | img_h)).
 * ------------------------------------------------------------------------------------------------- */
int lp_preprocess_letterbox(const unsigned char* img, int h0, int w0, void* out, int out_dtype, int H, int W, int rh, int rw,
                            int top, int left, void* stream);
int lp_rescale_round(float* det, int n, double ratio, double padx, double pady, int img_w, int img_h, void* stream);

/* lp_eval_counts: the matching loops of Evaler.eval (yolov6/core/evaler.py:153-243, box_iou general.py:93-115) for a
 * batch of images, one workgroup per image.
 *   det [B,max_det,28] fp32 + det_count [B]: detections as lp_nms returns them (xyxy, 8 corner coords, 8 confs, 8 ids)
 *   tgt [B,max_t,20] fp32 + tgt_count [B]:   labels (8 ids, xyxy, 8 corner coords), evaler.py:121-128 layout minus column 0
 *   counts: device int64 [LP_EVAL_NCOUNTS], ACCUMULATED into (zero it before the first batch):
 *           [LP_EVAL_TRUE] labels, [LP_EVAL_PRED] labels matched with IoU >= 0.7, then four groups of ten 0.05-wide IoU
 *           bins from 0.5: matched labels, corners right, classes right, both right; [LP_EVAL_UNBINNED] matched labels
 *           whose IoU fits no bin (IoU >= 1.0f) -- the reference re-uses a stale bin index for those, they are skipped here.
 * The ratios (evaler.py:245-283) are host arithmetic on these integers. */
enum { LP_EVAL_TRUE = 0, LP_EVAL_PRED = 1, LP_EVAL_PRED_BINS = 2, LP_EVAL_COR = 12, LP_EVAL_CLS = 22, LP_EVAL_RIGHT = 32,
       LP_EVAL_UNBINNED = 42, LP_EVAL_NCOUNTS = 43 };
int lp_eval_counts(const float* det, const int32_t* det_count, int max_det, const float* tgt, const int32_t* tgt_count,
                   int max_t, int B, long long* counts, void* stream);

/* Verification hook of the detections-only head (lp_engine_forward_det): it takes the largest sigmoid of a head to be the sigmoid of
 * the head's largest logit, which holds iff the kernels' sigmoid (hardware exp2 and reciprocal) is monotone non-decreasing.  Checks
 * every pair of neighbouring fp32 values (2^32 - 1 pairs, NaNs skipped) on the device and leaves the number of violations in
 * *dev_violations (device memory, 8 bytes).  Expected: 0 (tests/test_hip_kernels.py). */
int lp_check_sigmoid_monotone(unsigned long long* dev_violations, void* stream);

/* Verification hook of the greedy NMS step: the kernels decide torchvision's `inter / union > iou_threshold` with two products and
 * compares where the outcome is certain and with the fp32 division only inside a 2^-19-wide band around the threshold (lp_nms.hip,
 * iou_gt).  Evaluates both forms on n box pairs (device fp32 [n][8]: box i xyxy, box j xyxy); dev_out[k] bit 0 = the product form,
 * bit 1 = the plain division.  Expected: both bits equal for every pair, and equal to the fp32 restatement (tests/test_hip_kernels.py). */
int lp_check_iou_predicate(const float* dev_pairs, long long n, double iou_thres, unsigned char* dev_out, void* stream);

/* Test hook of the LDS-ring kernels: fills all 160 KiB of LDS of every CU with 0xFFFF halves (NaN in fp16 / bf16).  LDS keeps its
 * contents between kernels, so a fragment read that runs ahead of its LDS-DMA would otherwise find the (identical) bytes of the
 * previous launch and go unnoticed; after this call it poisons the output (tests/test_hip_kernels.py). */
int lp_debug_poison_lds(void* stream);

/* Host-side planning of the frame-reading stem kernels (no device needed; used by the CPU tests): the `choice`-th best output tile
 * TH x TW for an Ho x Wo output map of stem_planar_kernel (fused == 0: TW % 4 == 0, TH * TW <= 512, planar halo within its 20 KiB LDS
 * slot) or of stem2_fused_kernel (fused != 0: TW even, TH * TW <= 128, frame window <= 21 KiB, stem tile pitch *hpitch >= 2 TW + 1 with
 * (2 TH + 1) * pitch <= 640 positions).  LP_ERR_UNSUPPORTED: no such tile. */
int lp_plan_stem_tile(int fused, int Ho, int Wo, int choice, int* TH, int* TW, int* hpitch);
/* The same for the block-tiled 3x3 kernels: the `choice`-th best output tile of LP_VARIANT_PIPE16_V0 / _V1 (stride 1) or LP_VARIANT_PIPE16_S2A / _S2B
 * (stride 2) for B images of an Ho x Wo OUTPUT map and nct cout tiles -- TH * TW pixels within the variant's pixel blocks, the halo
 * ((TH - 1) s + 3) x ((TW - 1) s + 3) within its LDS slot, fewest rounds of the persistent grid first; *hpitch = the halo row pitch.
 * LP_ERR_UNSUPPORTED: another variant. */
int lp_plan_block_tile(int variant, int Ho, int Wo, int B, int nct, int choice, int* TH, int* TW, int* hpitch);

#ifdef __cplusplus
}
#endif
#endif /* LP_HIP_H */
