// Internal declarations shared by the translation units of libyololp_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include <string>

#include "../../include/lp_hip.h"

// order of the first two chunks' DMA requests in the prologue of the pipelined 3x3 kernels: 0 = halo pieces of both chunks,
// then the weights; 1 = weights first (same-box A/B: profiles/r04_experiments.txt item 8)
#ifndef LP_PRO_ORDER
#define LP_PRO_ORDER 0
#endif

namespace lp {

typedef _Float16 f16;
typedef __bf16 bf16;

// ---- error plumbing -------------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
#define LP_HIP_CHECK(expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return ::lp::fail(LP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)

// Raise a kernel's dynamic-LDS limit (above the 64 KiB default) once PER DEVICE: the attribute belongs to the (function,
// device) pair, and a process may drive several GPUs.  `done` is the caller's per-kernel-instantiation bit mask.
template <typename K> inline int set_max_lds_once(K kernel, int bytes, std::atomic<unsigned long long>& done, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev >= 0 && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return LP_OK;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(LP_ERR_HIP, std::string(what) + ": LDS attribute: " + hipGetErrorString(e));
    if (dev >= 0) done.fetch_or(1ull << dev, std::memory_order_release);
    return LP_OK;
}

inline size_t dtype_size(int dt) { return dt == LP_F32 ? 4 : 2; }
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- conv launch description ----------------------------------------------------------------------
enum ConvMode { MODE_ACT = 0, MODE_PRED = 1, MODE_DECODE = 2 };
enum ConvCfg { CFG_A = 0 /*128 couts x 128 px*/, CFG_B = 1 /*64 x 256*/, CFG_C = 2 /*32 x 256*/, CFG_D = 3 /*128 x 256, 8 waves*/, CFG_E = 4 /*64 x 128, 4 waves of 32x64*/, CFG_F = 5 /*128 x 128, 8 waves of 32x64*/, CFG_COUNT = 6 };

// Pipelined 3x3 stride-1 kernel (lp_conv3x3_pipe.inc): workgroup configurations
enum PipeCfgId { PIPE_D = 0 /*128 couts x 256 px*/, PIPE_B = 1 /*64 x 512*/, PIPE_F = 2 /*128 x 128*/, PIPE_C = 3 /*32 x 512*/, PIPE_P = 4 /*the stem reading the NCHW frame: lp_stem_planar.inc*/, PIPE_COUNT = 5,
                 PIPE_FUSED2 = 5 /*stem + ERBlock_2[0] in one kernel: lp_stem2_fused.inc (not a variant of one layer)*/,
                 PIPE_FUSED_PW = 6 /*1x1 + 3x3 stride 2 in one kernel: lp_pw_s2_fused.inc*/,
                 PIPE_FUSED_BF = 13 /*BiFusion's cv3(cat[upsample(x0), cv1(x1), d]) in one kernel: lp_bifusion_fused.inc*/,
                 PIPE16_D = 7, PIPE16_F = 9 /*D / F on v_mfma_f32_16x16x32 (lp_conv3x3_pipe16.inc): another fp32 summation order,
                                                            chosen per layer by rule, never by timing*/,
                 PIPE16_V0 = 10, PIPE16_V1 = 11 /*the same sums with tiles of any number of 16-pixel blocks (lp_conv3x3_pipe16v.inc): 128 couts x <= 448 px,
                                                  128 x <= 224*/, PIPE_END = 12,
                 PIPE16_S2A = 16, PIPE16_S2B = 17 /*3x3 STRIDE 2 on v_mfma_f32_16x16x32, three halo + two weight slots, tiles of any number of 16-pixel blocks
                                                    (lp_conv3x3_s2p16.inc): 128 couts x <= 256 px (2 x 4 waves), 128 x <= 224 (4 x 2 waves)*/, PIPE_S2_END = 18 };
inline bool pipe_is_16s2(int pcfg) { return pcfg == PIPE16_S2A || pcfg == PIPE16_S2B; }
inline bool pipe_is_16(int pcfg) { return pcfg == PIPE16_D || pcfg == PIPE16_F || pcfg == PIPE16_V0 || pcfg == PIPE16_V1 || pipe_is_16s2(pcfg); }
inline bool pipe_is_16v(int pcfg) { return pcfg == PIPE16_V0 || pcfg == PIPE16_V1; }

struct ConvSrc {
    const void* ptr;
    int cs;  // stored channels (multiple of 8) = pixel stride in elements
};

struct ConvArgs {
    ConvSrc src[LP_MAX_SRC];
    int chunk_begin[LP_MAX_SRC + 1];  // prefix sums of K-chunks per source
    int nsrc;
    const void* w;          // packed [phase][cout_tile][chunk][tap][CB][KC]
    const float* bias;      // [phase?][nct*CB]  (same for every phase)
    unsigned long long* stamps;  // debug builds (-DLP_STAMPS): per-wave phase cycle counters, else null
    const void* zero;       // 128 zero bytes: DMA source of out-of-image / out-of-range granules
    void* trash;            // 16 writable bytes nobody reads: target of stores that must not happen (streaming 1x1 kernel)
    void* out;
    void* out2;             // MODE_ACT, two sibling layers as one launch: output channels >= out_split go to this tensor (channel c - out_split,
    int out_split;          // pixel stride out2_pix_stride, image stride out2_img_stride); null: a single destination
    int out2_pix_stride;
    long long out2_img_stride;
    const void* res;        // residual (same dtype/geometry as out) or null
    int res_cs;
    float alpha;
    int B, H, W;            // input spatial dims
    int Ho, Wo;             // conv output dims (before out_scale)
    int TH, TW, tiles_x, tiles_y;
    unsigned tw_magic, hp_magic;  // multiply-shift reciprocals of TW and hpitch (exact for the index ranges used)
    int hpitch;             // LDS row pitch of the staged halo, in pixels (>= (TW-1)*stride + ksize)
    int nct;                // cout tiles
    int out_c;              // channels to store (multiple of the 16-B granule for MODE_ACT)
    long long out_img_stride;  // elements
    int out_pix_stride;        // elements
    int out_scale;          // 1, or 2 for the transposed conv (phase p writes (2y + p/2, 2x + p%2))
    int nphase;
    long long w_phase_stride;  // elements
    int act;
    // detections-only forward (lp_engine_forward_det): the head writes NMS candidates instead of the prediction tensor.
    //   MODE_DECODE with det_mode: out = candidate rows [B][N][28], columns 0..11 (xyxy + corners) of EVERY anchor;
    //   head_cls_rows_kernel<.., true>: out = candidate rows (columns 12..27 of the anchors that pass the confidence mask),
    //   det_keys [B][det_np] / det_cnt [B] = the candidate lists lp_nms_candidates sorts.
    int det_mode;
    unsigned long long* det_keys;
    int* det_cnt;
    int det_np, det_n, det_anchor0;   // keys per image (power of two), anchors per image, first anchor of this level
    float det_conf;
    //   head_box_det_kernel, sparse form (det_keys set on a MODE_DECODE launch): boxes of the level's candidates only; det_prev [B] = where
    //   the level's entries begin in the key lists (null: 0), det_snap [B] = where they end (written for the next level)
    const int* det_prev;
    int* det_snap;
    // stem2_fused_kernel (lp_stem2_fused.inc): the stem's packed weights, bias, activation and stored channels (src[0] = the frame)
    const void* fz_w1;
    const float* fz_b1;
    int fz_act1, fz_c1;
    // bifusion_fused_kernel (lp_bifusion_fused.inc): src[0..2] = x0 (coarse map), x1, d; fz_w1 / fz_b1 = cv1's operands; the transposed conv's:
    const void* bf_wd;
    const float* bf_bd;
    long long bf_wd_phase_stride;   // elements
    // MODE_DECODE
    int reg_bins;
    const float* proj;
    float stride_px;
};

struct ConvShape {  // compile-time geometry of one kernel configuration, mirrored on the host
    int CB, PB, KC, HPMAX, NT, WP, WGC, WGP;
};
ConvShape conv_shape(int dtype, int cfg, int ksize, int stride);
// Picks the output tile (TH x TW <= PB output pixels, halo <= HPMAX) that needs the fewest blocks.
void conv_pick_tile(const ConvShape& s, int ksize, int stride, int Ho, int Wo, int choice, int* TH, int* TW);
// Halo row pitch (>= halo width) that minimises ds_read_b128 bank conflicts of the pixel-operand reads, found by
// simulating the LDS banking of every fragment read of the tile (exact model: MI355X_MICROARCH.md, LDS table).
int conv_pick_pitch(const ConvShape& s, int dtype, int ksize, int stride, int TH, int TW);
int conv_pick_pitch16(const ConvShape& s, int TH, int TW);   // the same for conv3x3_pipe16_kernel's operand map
// Output tile of conv3x3_pipe16v_kernel (any number of 16-pixel blocks): fewest (rounds of the persistent grid) x (blocks per wave)
void conv_pick_tile16v(const ConvShape& s, int Ho, int Wo, int B, int nct, int choice, int* TH, int* TW, int stride = 1);
int device_cus();
int conv_launch(int dtype, int cfg, int mode, int ksize, int stride, int nbuf, const ConvArgs& a, hipStream_t st);
bool stem_planar_tile(int Ho, int Wo, int choice, int* TH, int* TW);   // lp_stem_planar.inc's output tile (choice = k-th best); false: none
bool stem2_fused_tile(int Ho, int Wo, int choice, int* TH, int* TW, int* hpitch);   // lp_stem2_fused.inc's output tile and stem-tile pitch

// Streaming 1x1 kernel (lp_conv1x1_stream.inc): wc = cout tiles per wave (2 or 4), cb_pack = cout-tile rows of the op's
// weight packing.  conv_stream_lds() < 0: the layer does not fit.
int conv_stream_lds(int dtype, int wc, int nchunks, int cb_pack);
int conv_stream_launch(int dtype, int wc, const ConvArgs& a, int cb_pack, hipStream_t st);

// Pipelined 3x3 stride-1 kernel: geometry of configuration `pcfg` (CB = the weight-packing cout tile it reads), and the launch.
ConvShape conv_pipe_shape(int pcfg);
bool conv_pipe_fits(int dtype, int pcfg, int cb_pack, int ksize, int stride, int mode, int nct, int nphase, int nchunks);
int conv_pipe_launch(int dtype, int pcfg, const ConvArgs& a, hipStream_t st);

// Row-writer form of the class predictors (lp_head_rows.inc).
bool head_rows_fits(int dtype, int nchunks, int cb_pack, int out_c);
bool head_det_fits(int dtype, int nchunks, int cb_pack, int out_c);   // its detections-only form (head_det_kernel)
bool head_box_det_fits(const ConvArgs& a, int cb_pack, int ksize, int stride);
int head_box_det_launch(int dtype, const ConvArgs& a, int cb_pack, hipStream_t st);   // box / corner predictors of the detections-only forward, streaming form
int head_rows_launch(int dtype, const ConvArgs& a, int cb_pack, hipStream_t st);   // a.det_mode: the candidate-writing form

// ---- post-processing pieces shared with the engine's detections-only forward (lp_nms.hip) ----
struct NmsWs {  // carve-up of the caller's NMS workspace
    int32_t* cnt;              // [B]
    unsigned long long* keys;  // [B][NP]
    float* rows;               // [B][N][28]
    float4* sbox;              // [B][N]
    int32_t* kept;             // [B][N]
    int NP;
    size_t bytes;
};
NmsWs nms_carve(void* base, int B, int N);
// score_kernel over `rows_per_img` prediction rows per image (row stride 290 floats, images `rows_per_img` rows apart) whose
// anchors are anchor0.. of N: appends the candidates; write_box false leaves columns 0..11 of the candidate rows alone.
int nms_score_launch(float* pred, int B, int rows_per_img, int anchor0, int N, float conf_f, const NmsWs& w, bool write_box, hipStream_t st);
int zero_counts_launch(int* p, int n, hipStream_t st);   // p[0..n) = 0 by a kernel of our own (no hipMemsetAsync: see lp_nms.hip)

// ---- auxiliary kernels ----------------------------------------------------------------------------
int input_launch(const void* x, int x_dtype, void* dst, int dtype, int B, int H, int W, hipStream_t st);
int input_s2d_launch(const void* x, int x_dtype, void* dst, int dtype, int B, int H, int W, hipStream_t st);
// LDS bytes the pool-chain kernel needs for an h x w plane (its narrowest channel run), and the limit it runs under.
size_t pool_min_lds_bytes(int dtype, int h, int w);
constexpr size_t POOL_MAX_LDS = 128 * 1024;
int pool_launch(const void* src, void* d1, void* d2, void* d3, int dtype, int B, int h, int w, int cs, hipStream_t st);

}  // namespace lp
