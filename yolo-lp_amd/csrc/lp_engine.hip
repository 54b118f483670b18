// Engine: the frozen op graph of one model (built by the python host from folded weights), weight packing
// for the MFMA conv kernels, arena placement and the launch loop.  Host code only; kernels live in
// lp_conv.hip / lp_aux.hip / lp_nms.hip.
#include "lp_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

namespace lp {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

enum OpKind { OP_INPUT = 0, OP_CONV = 1, OP_DECONV = 2, OP_POOL = 3, OP_HEAD_CLS = 4, OP_HEAD_BOX = 5 };

struct Tensor {
    int c, cs, sl;     // logical channels, stored channels, log2 downscale
    bool unused = false;  // declared by the host but no longer read or written (not placed in the arena)
    size_t offset;     // bytes inside the arena (valid after bind)
    int h, w;
};

struct Op {
    int kind = OP_CONV;
    int nsrc = 0;
    int src[LP_MAX_SRC] = {-1, -1, -1, -1};
    int dst = -1, dst2 = -1, dst3 = -1;
    int ksize = 1, stride = 1, act = 0;
    int res = -1;
    float alpha = 0.f;
    int cout = 0;        // logical output channels
    int cin = 0;         // logical input channels (sum over sources)
    int level = 0, reg_bins = 1;
    bool s2d = false;    // input op writes the space-to-depth form (see lp_engine_finalize)
    int alg_kk = 0;      // algorithmic cin*k*k when it differs from the executed one (stem rewrite: 27, not 12*9)
    int lane = 0;        // execution lane (HIP stream) hint given by the host: independent branches overlap
    std::vector<int> deps;   // ops on OTHER lanes whose outputs this op reads (filled by finalize)
    bool signal = false;     // some op on another lane waits for this one
    std::vector<float> weight, bias, proj;  // host fp32, reference layouts
    // filled by finalize
    int cfg = CFG_A, mode = MODE_ACT, nct = 1, nchunks = 0, nphase = 1, nbuf = 1, tile = 0;
    int stream_wc = 0, stream_rd = 2;   // stream_wc != 0: the streaming 1x1 kernel (2 or 4 cout tiles per wave) runs this op
    int rows = 0;                       // OP_HEAD_CLS: the row-writer kernel (lp_head_rows.inc) runs this op
    size_t det_scratch = (size_t)-1;    // OP_HEAD_CLS whose detections-only form needs a prediction scratch: its arena offset (after bind)
    int box_sparse = 1;                 // OP_HEAD_BOX, detections-only forward: boxes of the level's candidates only (LP_VARIANT_BOX_SPARSE / _DENSE)
    int pipe = 0;                       // != 0: the pipelined 3x3 stride-1 kernel (lp_conv3x3_pipe.inc), configuration pipe - 1
    int fused_pw = 0;                   // 3x3 stride-2 layer behind a 1x1 layer (BiFusion's downsample(cv2(x))): != 0: the two run as ONE kernel
                                        // (lp_pw_s2_fused.inc; fused_pw - 1 = its tile choice); the 1x1 op before it is then skipped
    int fused_bf = 0;                   // BiFusion's cv3 (1x1 over [upsample(x0), cv1(x1), d]): != 0: transposed conv, cv1 and cv3 run as ONE kernel
                                        // (lp_bifusion_fused.inc); the two ops it carries (bf_up, bf_cv1) are then skipped
    int bf_up = -1, bf_cv1 = -1;        // ... their indices (lp_engine_finalize: bifusion_fused_possible), and on those ops: bf_carrier = the cv3 op
    int bf_carrier = -1;
    int fused = 0;                      // ERBlock_2[0] (op 2) only: != 0: when the caller's frame has the engine's dtype, input op, stem and this
                                        // layer run as ONE kernel (lp_stem2_fused.inc; fused - 1 = its tile choice)
    int planar = 0;                     // stem over the space-to-depth image only: != 0: when the caller's frame has the engine's dtype, the
                                        // PIPE_P kernel reads it directly and the input op is skipped (planar - 1 = its tile choice)
    int chunk_begin[LP_MAX_SRC + 1] = {0, 0, 0, 0, 0};
    size_t w_off = 0, b_off = 0, proj_off = 0;  // byte offsets in the packed blob
    long long w_phase_stride = 0;              // elements
};

struct Launch {
    ConvArgs a;
    long long pred_off = 0;
    int cfg = 0, mode = 0, ks = 1, st = 1, nbuf = 1;
    int stream_wc = 0, stream_rd = 2, cb_pack = 0, rows = 0, pipe = 0;
    bool is_conv = false;
};

}  // namespace lp

using namespace lp;
struct lp_engine;
static bool bifusion_fused_possible(const lp_engine* e, size_t i, int* up, int* cv1);
static bool op_fam16(const lp_engine* e, const lp::Op& op);
static int fam16_default_pipe(const lp_engine* e, const lp::Op& op);

#ifndef LP_MAX_LANES
#define LP_MAX_LANES 5
#endif
struct lp_engine {
    int dtype = LP_F16;
    bool finalized = false;
    int n_levels = 3;
    std::vector<Tensor> tensors;
    std::vector<Op> ops;
    std::vector<unsigned char> blob;  // packed weights (host)
    char* dev_w = nullptr;            // caller-owned device copy
    char* arena = nullptr;
    size_t arena_bytes = 0;
    int B = 0, H = 0, W = 0, n_anchors = 0;
    std::vector<int> level_off;       // first pred row of each level
    std::vector<hipEvent_t> events;
    int cur_lane = 0;                 // lane assigned to ops added from now on (lp_engine_set_lane)
    int n_lanes = 1;
    hipStream_t lane_stream[LP_MAX_LANES] = {};   // [0] = the caller's stream
    std::vector<hipEvent_t> op_event; // per op, created lazily for ops with signal
    hipEvent_t fork_ev = nullptr, join_ev[LP_MAX_LANES] = {};
    bool single_lane = true;          // lp_engine_set_single_lane (default): every op on the caller's stream, in op order
    bool use_graph = false;           // lp_engine_set_graph: replay the captured forward instead of re-issuing ~80 launches
    bool s2p16 = getenv("LP_S2P16") != nullptr;        // opt-in (measured equal / slower: lp_conv3x3_s2p16.inc): with mfma16, eligible 3x3 STRIDE-2 layers run on conv3x3_s2p16_kernel (op_fam16)
    bool mfma16 = getenv("LP_NO_MFMA16") == nullptr;   // lp_engine_set_mfma16: eligible 3x3 layers run on the 16x16x32 family (op_fam16)
    // Detections-only forward: the box predictors of a level may run for the level's CANDIDATES only (head_box_det_kernel, sparse form) when,
    // in op order, exactly the class predictors of the same level sit between a box op and the box op before it -- then the entries
    // appended to the key lists between the two are this level's.  box_ord[i] = ordinal of box op i among the box ops (-1: not one).
    bool box_sparse_ok = false;
    std::vector<int> box_ord;
    struct CachedGraph {               // one captured forward; valid for exactly these pointers / dtype / tuning state
        hipGraphExec_t exec = nullptr;
        const void* x = nullptr;
        float* pred = nullptr;             // plain forward: the prediction tensor; detections-only forward: null
        void* ws = nullptr;                // detections-only forward: the candidate workspace (and its threshold)
        float conf = 0.f;
        int x_dtype = -1;
        unsigned long long epoch = 0, last_use = 0;
        hipStream_t stream = nullptr;  // stream of its last launch: synchronised before the executable graph is destroyed
    };
    std::vector<CachedGraph> graphs;   // a few, so that alternating buffers do not re-capture (and destroy) every call
    unsigned long long graph_clock = 0;
    hipEvent_t done_ev = nullptr;      // recorded on the caller's stream behind every forward: lp_engine_destroy waits for it (an
    bool done_valid = false;           // event of our own, not the caller's stream handle, which may be gone by then)
    hipStream_t cap_stream = nullptr;  // capture happens here (the caller's stream may be the legacy null stream, which cannot capture)
    unsigned long long epoch = 1;      // changes whenever launches are re-prepared
    std::vector<Launch> launches;     // per op, prepared at bind / after tuning
    Launch stem_planar;               // ops[1] as the PIPE_P kernel (valid when ops[1].planar)
    Launch stem2_fused;               // ops[0..2] as the fused kernel (valid when ops[2].fused)
    std::map<int, Launch> pw_fused;   // op index of the 3x3 layer -> the fused 1x1 + 3x3 launch (valid when that op's fused_pw)
    std::map<int, Launch> bf_fused;   // op index of BiFusion's cv3 -> the fused launch (valid when that op's fused_bf)
    std::map<std::vector<int>, std::vector<std::vector<int>>> tuned;  // (B,H,W) -> per-op {cfg, nbuf, tile, stream_wc, stream_rd, rows, pipe, planar, fused, fused_pw}
};

struct lp_engine;
static int prepare_op(lp_engine* e, size_t idx);
static unsigned long long* g_stamps = nullptr;
extern "C" void lpdbg_set_stamps(void* p) { g_stamps = (unsigned long long*)p; }   // debug hook (LP_STAMPS builds)
static bool valid_tensor(const lp_engine* e, int id) { return id >= 0 && id < (int)e->tensors.size(); }

extern "C" const char* lp_version(void) { return "yololp-hip 0.1 (gfx950)"; }
extern "C" const char* lp_last_error(void) { return g_err.c_str(); }

extern "C" int lp_engine_create(lp_engine** out, int act_dtype) {
    if (!out) return fail(LP_ERR_ARG, "lp_engine_create: null out");
    if (act_dtype != LP_F16 && act_dtype != LP_BF16 && act_dtype != LP_F32) return fail(LP_ERR_ARG, "lp_engine_create: dtype");
    lp_engine* e = new lp_engine();
    e->dtype = act_dtype;
    *out = e;
    return LP_OK;
}

extern "C" void lp_engine_destroy(lp_engine* e) {
    if (!e) return;
    // Nothing of the engine may die under in-flight work: the side lanes may still run kernels of the last forward that
    // read launch arguments' device buffers and wait on / record the events destroyed below (an executable graph destroyed
    // under its last launch aborted the process in round 1).  The last forward on the caller's stream is waited for through
    // an event the engine owns (the stream handle itself may have been destroyed by the caller since).
    for (int l = 1; l < LP_MAX_LANES; ++l)
        if (e->lane_stream[l]) (void)hipStreamSynchronize(e->lane_stream[l]);
    if (e->done_valid) (void)hipEventSynchronize(e->done_ev);
    if (e->done_ev) (void)hipEventDestroy(e->done_ev);
    for (hipEvent_t ev : e->events) (void)hipEventDestroy(ev);
    for (auto& g : e->graphs) {
        if (g.stream) (void)hipStreamSynchronize(g.stream);
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    }
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    for (hipEvent_t ev : e->op_event) if (ev) (void)hipEventDestroy(ev);
    if (e->fork_ev) (void)hipEventDestroy(e->fork_ev);
    for (int l = 1; l < LP_MAX_LANES; ++l) { if (e->join_ev[l]) (void)hipEventDestroy(e->join_ev[l]); if (e->lane_stream[l]) (void)hipStreamDestroy(e->lane_stream[l]); }
    delete e;
}

extern "C" int lp_engine_tensor(lp_engine* e, int channels, int stride_log2) {
    if (!e || e->finalized) return fail(LP_ERR_STATE, "lp_engine_tensor: engine is null or frozen");
    if (channels < 1 || stride_log2 < 0 || stride_log2 > 6) return fail(LP_ERR_ARG, "lp_engine_tensor: bad shape");
    Tensor t;
    t.c = channels;
    t.cs = round_up(channels, 8);
    t.sl = stride_log2;
    t.offset = 0;
    t.h = t.w = 0;
    e->tensors.push_back(t);
    return (int)e->tensors.size() - 1;
}

extern "C" int lp_engine_set_lane(lp_engine* e, int lane) {
    if (!e || e->finalized) return fail(LP_ERR_STATE, "lp_engine_set_lane: engine is null or frozen");
    if (lane < 0 || lane >= LP_MAX_LANES) return fail(LP_ERR_ARG, "lp_engine_set_lane: lane must be 0..4");
    e->cur_lane = lane;
    return LP_OK;
}

extern "C" int lp_engine_add_input(lp_engine* e, int dst) {
    if (!e || e->finalized) return fail(LP_ERR_STATE, "lp_engine_add_input: engine is null or frozen");
    if (!e->ops.empty()) return fail(LP_ERR_STATE, "lp_engine_add_input: must be the first op");
    if (!valid_tensor(e, dst) || e->tensors[dst].c != 3 || e->tensors[dst].sl != 0)
        return fail(LP_ERR_ARG, "lp_engine_add_input: dst must be a 3-channel full-resolution tensor");
    Op op;
    op.kind = OP_INPUT;
    op.dst = dst;
    op.cout = 3;
    op.cin = 3;
    op.lane = e->cur_lane;
    e->ops.push_back(op);
    return LP_OK;
}

extern "C" int lp_engine_add_conv(lp_engine* e, const lp_conv_desc* d) {
    if (!e || e->finalized) return fail(LP_ERR_STATE, "lp_engine_add_conv: engine is null or frozen");
    if (!d || !d->weight || !d->bias) return fail(LP_ERR_ARG, "lp_engine_add_conv: null descriptor / weights");
    if (d->n_src < 1 || d->n_src > LP_MAX_SRC) return fail(LP_ERR_ARG, "lp_engine_add_conv: n_src");
    if (!((d->ksize == 1 && d->stride == 1) || (d->ksize == 3 && (d->stride == 1 || d->stride == 2))))
        return fail(LP_ERR_UNSUPPORTED, "lp_engine_add_conv: only 1x1/s1, 3x3/s1 and 3x3/s2");
    if (d->act < LP_ACT_NONE || d->act > LP_ACT_SILU) return fail(LP_ERR_ARG, "lp_engine_add_conv: act");
    if (!valid_tensor(e, d->dst)) return fail(LP_ERR_ARG, "lp_engine_add_conv: dst");
    Op op;
    op.kind = OP_CONV;
    op.nsrc = d->n_src;
    int cin = 0;
    if (!valid_tensor(e, d->src[0])) return fail(LP_ERR_ARG, "lp_engine_add_conv: src");
    const int sl = e->tensors[d->src[0]].sl;
    for (int i = 0; i < d->n_src; ++i) {
        if (!valid_tensor(e, d->src[i])) return fail(LP_ERR_ARG, "lp_engine_add_conv: src");
        if (e->tensors[d->src[i]].sl != sl) return fail(LP_ERR_ARG, "lp_engine_add_conv: sources differ in resolution");
        if (d->src[i] == d->dst) return fail(LP_ERR_ARG, "lp_engine_add_conv: in-place conv");
        op.src[i] = d->src[i];
        cin += e->tensors[d->src[i]].c;
    }
    const Tensor& td = e->tensors[d->dst];
    if (td.sl != sl + (d->stride == 2 ? 1 : 0)) return fail(LP_ERR_ARG, "lp_engine_add_conv: dst resolution");
    if (d->res >= 0) {
        if (!valid_tensor(e, d->res) || e->tensors[d->res].c != td.c || e->tensors[d->res].sl != td.sl || d->res == d->dst)
            return fail(LP_ERR_ARG, "lp_engine_add_conv: residual shape");
    }
    int cout2 = 0;
    if (d->dst2 > 0) {       // two sibling layers as one launch
        if (!valid_tensor(e, d->dst2) || d->dst2 == d->dst || e->tensors[d->dst2].sl != td.sl) return fail(LP_ERR_ARG, "lp_engine_add_conv: dst2");
        for (int i = 0; i < d->n_src; ++i) if (d->src[i] == d->dst2) return fail(LP_ERR_ARG, "lp_engine_add_conv: in-place conv");
        if (d->res >= 0) return fail(LP_ERR_UNSUPPORTED, "lp_engine_add_conv: no residual with two destinations");
        if (td.c % 8 != 0) return fail(LP_ERR_UNSUPPORTED, "lp_engine_add_conv: the first destination of two needs a multiple of 8 channels");
        cout2 = e->tensors[d->dst2].c;
    }
    op.dst = d->dst;
    op.dst2 = d->dst2 > 0 ? d->dst2 : -1;
    op.ksize = d->ksize;
    op.stride = d->stride;
    op.act = d->act;
    op.res = d->res >= 0 ? d->res : -1;
    op.alpha = d->res_alpha;
    op.cout = td.c + cout2;       // (td.c is its stored width when a second destination follows: the rows are contiguous)
    op.cin = cin;
    op.weight.assign(d->weight, d->weight + (size_t)op.cout * cin * d->ksize * d->ksize);
    op.bias.assign(d->bias, d->bias + op.cout);
    op.lane = e->cur_lane;
    e->ops.push_back(op);
    return LP_OK;
}

extern "C" int lp_engine_add_deconv2x2(lp_engine* e, int src, int dst, const float* weight, const float* bias) {
    if (!e || e->finalized) return fail(LP_ERR_STATE, "lp_engine_add_deconv2x2: engine is null or frozen");
    if (!weight || !bias || !valid_tensor(e, src) || !valid_tensor(e, dst) || src == dst)
        return fail(LP_ERR_ARG, "lp_engine_add_deconv2x2: bad argument");
    if (e->tensors[dst].sl != e->tensors[src].sl - 1) return fail(LP_ERR_ARG, "lp_engine_add_deconv2x2: dst must be 2x the source");
    Op op;
    op.kind = OP_DECONV;
    op.nsrc = 1;
    op.src[0] = src;
    op.dst = dst;
    op.cin = e->tensors[src].c;
    op.cout = e->tensors[dst].c;
    op.weight.assign(weight, weight + (size_t)op.cin * op.cout * 4);
    op.bias.assign(bias, bias + op.cout);
    op.lane = e->cur_lane;
    e->ops.push_back(op);
    return LP_OK;
}

extern "C" int lp_engine_add_pool5_chain(lp_engine* e, int src, int dst1, int dst2, int dst3) {
    if (!e || e->finalized) return fail(LP_ERR_STATE, "lp_engine_add_pool5_chain: engine is null or frozen");
    const int ids[4] = {src, dst1, dst2, dst3};
    for (int i = 0; i < 4; ++i) {
        if (!valid_tensor(e, ids[i])) return fail(LP_ERR_ARG, "lp_engine_add_pool5_chain: tensor id");
        if (e->tensors[ids[i]].c != e->tensors[src].c || e->tensors[ids[i]].sl != e->tensors[src].sl)
            return fail(LP_ERR_ARG, "lp_engine_add_pool5_chain: shapes differ");
        for (int k = 0; k < i; ++k)
            if (ids[k] == ids[i]) return fail(LP_ERR_ARG, "lp_engine_add_pool5_chain: tensors must be distinct");
    }
    Op op;
    op.kind = OP_POOL;
    op.nsrc = 1;
    op.src[0] = src;
    op.dst = dst1;
    op.dst2 = dst2;
    op.dst3 = dst3;
    op.cin = op.cout = e->tensors[src].c;
    op.lane = e->cur_lane;
    e->ops.push_back(op);
    return LP_OK;
}

static int add_head(lp_engine* e, int kind, int src, int level, int cout, int reg_bins, const float* weight, const float* bias,
                    const float* proj) {
    if (!e || e->finalized) return fail(LP_ERR_STATE, "lp_engine_add_head: engine is null or frozen");
    if (!weight || !bias || !valid_tensor(e, src) || level < 0 || level > 3 || cout < 1)
        return fail(LP_ERR_ARG, "lp_engine_add_head: bad argument");
    if (e->tensors[src].sl != 3 + level) return fail(LP_ERR_ARG, "lp_engine_add_head: level / resolution mismatch (strides 8,16,32,64)");
    Op op;
    op.kind = kind;
    op.nsrc = 1;
    op.src[0] = src;
    op.level = level;
    op.reg_bins = reg_bins;
    op.cin = e->tensors[src].c;
    op.cout = cout;
    op.weight.assign(weight, weight + (size_t)cout * op.cin);
    op.bias.assign(bias, bias + cout);
    if (proj) op.proj.assign(proj, proj + reg_bins);
    op.lane = e->cur_lane;
    e->ops.push_back(op);
    return LP_OK;
}

extern "C" int lp_engine_add_head_cls(lp_engine* e, int src, int level, int n_cls, const float* weight, const float* bias) {
    if (n_cls != LP_PRED_COLS - 13) return fail(LP_ERR_ARG, "lp_engine_add_head_cls: n_cls must be 277 (31+24+6*37)");
    return add_head(e, OP_HEAD_CLS, src, level, n_cls, 1, weight, bias, nullptr);
}

extern "C" int lp_engine_add_head_box(lp_engine* e, int src, int level, int reg_bins, const float* weight, const float* bias,
                                      const float* proj) {
    if (reg_bins < 1 || 4 * reg_bins + 8 > 128) return fail(LP_ERR_ARG, "lp_engine_add_head_box: reg_bins");
    if (reg_bins > 1 && !proj) return fail(LP_ERR_ARG, "lp_engine_add_head_box: DFL needs proj");
    return add_head(e, OP_HEAD_BOX, src, level, 4 * reg_bins + 8, reg_bins, weight, bias, proj);
}

// ---- weight packing -------------------------------------------------------------------------------------
static void put_elem(unsigned char* dst, size_t idx, float v, int dtype) {
    if (dtype == LP_F32) {
        memcpy(dst + idx * 4, &v, 4);
    } else if (dtype == LP_F16) {
        const f16 h = (f16)v;      // round-to-nearest-even, like torch's .half()
        memcpy(dst + idx * 2, &h, 2);
    } else {
        const bf16 h = (bf16)v;
        memcpy(dst + idx * 2, &h, 2);
    }
}

// Weight-packing class (cout tile of 128, 64 or 32 rows) of a layer: the least padded output channels, where a row of the
// 32-row class counts 1.6x -- its kernels (32x64 wave tiles, one LDS stage) run at ~60 % of the rate of the others, so a
// 96-channel layer is better off padded to 128 rows of the 64-row class.  Ties go to the larger tile.
static int pick_cfg(int dtype, int ksize, int stride, int cout_store) {
    int best = CFG_A;
    long best_cost = -1;
    const int cfgs[3] = {CFG_A, CFG_B, CFG_C};
    for (int k = 0; k < 3; ++k) {
        const ConvShape s = conv_shape(dtype, cfgs[k], ksize, stride);
        const long cost = (long)ceil_div(cout_store, s.CB) * s.CB * (cfgs[k] == CFG_C ? 16 : 10);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cfgs[k]; }
    }
    return best;
}

extern "C" int lp_engine_finalize(lp_engine* e, int n_levels) {
    if (!e || e->finalized) return fail(LP_ERR_STATE, "lp_engine_finalize: engine is null or already frozen");
    if (e->ops.empty() || e->ops[0].kind != OP_INPUT) return fail(LP_ERR_STATE, "lp_engine_finalize: first op must be the input");
    if (n_levels < 1 || n_levels > 4) return fail(LP_ERR_ARG, "lp_engine_finalize: n_levels");
    e->n_levels = n_levels;
    const int dt = e->dtype;
    const size_t esz = dtype_size(dt);
    std::vector<unsigned char>& blob = e->blob;
    blob.clear();
    blob.resize(256, 0);   // bytes 0..127: zero page, DMA source for padding granules (ConvArgs::zero); 128..143: scratch granule (ConvArgs::trash)
    auto align = [&]() { blob.resize((blob.size() + 255) / 256 * 256, 0); };
    // Stem rewrite: a 3x3 stride-2 conv that is the only reader of the network input becomes a 3x3 stride-1 conv over
    // the space-to-depth form of the image (input_s2d_kernel): same sums, a quarter of the halo, no K=27 special case.
    if (e->ops.size() > 1 && getenv("LP_NO_STEM_S2D") == nullptr) {
        Op& c = e->ops[1];
        const int in_id = e->ops[0].dst;
        bool only_reader = c.kind == OP_CONV && c.ksize == 3 && c.stride == 2 && c.nsrc == 1 && c.src[0] == in_id && c.res < 0;
        for (size_t i = 2; i < e->ops.size() && only_reader; ++i) {
            const Op& o = e->ops[i];
            for (int k = 0; k < o.nsrc; ++k) only_reader &= o.src[k] != in_id;
            only_reader &= o.res != in_id;
        }
        if (only_reader) {
            Tensor t;
            t.c = 12; t.cs = 16; t.sl = 1; t.offset = 0; t.h = t.w = 0;
            e->tensors.push_back(t);
            const int s2d_id = (int)e->tensors.size() - 1;
            e->tensors[in_id].unused = true;
            e->ops[0].s2d = true;
            e->ops[0].dst = s2d_id;
            std::vector<float> w2((size_t)c.cout * 12 * 9, 0.f);
            for (int co = 0; co < c.cout; ++co)
                for (int ch = 0; ch < 3; ++ch)
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const int dy = ky == 0 ? 0 : 1, py = ky == 1 ? 0 : 1;   // image row 2*oy-1+ky = 2*(oy-1+dy) + py
                            const int dx = kx == 0 ? 0 : 1, px = kx == 1 ? 0 : 1;
                            w2[(((size_t)co * 12 + ch * 4 + py * 2 + px) * 3 + dy) * 3 + dx] = c.weight[(((size_t)co * 3 + ch) * 3 + ky) * 3 + kx];
                        }
            c.weight.swap(w2);
            c.src[0] = s2d_id;
            c.stride = 1;
            c.cin = 12;
            c.alg_kk = 27;
        }
    }
    // Cross-lane dependencies: an op waits for the latest earlier op on another lane that wrote one of the tensors it
    // reads (RAW) or that read / wrote a tensor it writes (WAR / WAW).  Same-lane order is stream order.
    if (getenv("LP_SINGLE_LANE")) for (Op& op : e->ops) op.lane = 0;
    e->n_lanes = 1;
    for (size_t i = 0; i < e->ops.size(); ++i) {
        Op& op = e->ops[i];
        if (op.lane + 1 > e->n_lanes) e->n_lanes = op.lane + 1;
        auto reads = [](const Op& o, int t) { for (int k = 0; k < o.nsrc; ++k) if (o.src[k] == t) return true; return o.res == t; };
        auto writes = [](const Op& o, int t) { return o.dst == t || o.dst2 == t || o.dst3 == t; };
        std::vector<int> mine_r, mine_w;
        for (int k = 0; k < op.nsrc; ++k) mine_r.push_back(op.src[k]);
        if (op.res >= 0) mine_r.push_back(op.res);
        for (int t : {op.dst, op.dst2, op.dst3}) if (t >= 0) mine_w.push_back(t);
        for (int lane = 0; lane < LP_MAX_LANES; ++lane) {
            if (lane == op.lane) continue;
            for (size_t j = i; j-- > 0;) {
                const Op& o = e->ops[j];
                if (o.lane != lane) continue;
                bool hit = false;
                for (int t : mine_r) hit |= writes(o, t);
                for (int t : mine_w) hit |= writes(o, t) || reads(o, t);
                if (hit) { op.deps.push_back((int)j); e->ops[j].signal = true; break; }   // the latest one covers the earlier ones
            }
        }
    }
    for (Op& op : e->ops) {
        if (op.kind == OP_INPUT || op.kind == OP_POOL) continue;
        const int ks = op.kind == OP_CONV ? op.ksize : 1;
        const int st = op.kind == OP_CONV ? op.stride : 1;
        int cout_store;
        if (op.kind == OP_HEAD_CLS) { op.mode = MODE_PRED; op.cfg = getenv("LP_PRED_CFG") ? atoi(getenv("LP_PRED_CFG")) : CFG_C; cout_store = op.cout; }
        else if (op.kind == OP_HEAD_BOX) {   // the whole row of box / corner logits must sit in one cout tile: 12 (no DFL) fits 32
            op.mode = MODE_DECODE; op.cfg = op.cout <= 32 ? CFG_C : CFG_A; cout_store = op.cout;
        }
        else {
            op.mode = MODE_ACT;
            cout_store = e->tensors[op.dst].cs + (op.kind == OP_CONV && op.dst2 >= 0 ? e->tensors[op.dst2].cs : 0);
            op.cfg = pick_cfg(dt, ks, st, cout_store);
            // default variant of the 128-cout class; lp_engine_autotune() picks per layer among the variants that
            // share this packing.  LP_TUNE_CFG128 / LP_TUNE_NBUF force one for experiments.
            if (op.cfg == CFG_A) op.cfg = CFG_D;
            const char* tc = getenv("LP_TUNE_CFG128");
            const char* tb = getenv("LP_TUNE_NBUF");
            if (tc && op.cfg == CFG_D) op.cfg = atoi(tc);
            if (tb && op.cfg != CFG_C) op.nbuf = atoi(tb) == 2 ? 2 : 1;
        }
        const ConvShape s = conv_shape(dt, op.cfg, ks, st);
        op.nct = ceil_div(cout_store, s.CB);
        op.nphase = op.kind == OP_DECONV ? 4 : 1;
        // K-chunks: every source starts a new chunk, so a chunk never straddles two tensors
        int cin_off[LP_MAX_SRC + 1] = {0};
        op.chunk_begin[0] = 0;
        for (int i = 0; i < op.nsrc; ++i) {
            const Tensor& t = e->tensors[op.src[i]];
            op.chunk_begin[i + 1] = op.chunk_begin[i] + ceil_div(t.cs, s.KC);
            cin_off[i + 1] = cin_off[i] + t.c;
        }
        for (int i = op.nsrc; i < LP_MAX_SRC; ++i) op.chunk_begin[i + 1] = op.chunk_begin[op.nsrc];
        op.nchunks = op.chunk_begin[op.nsrc];
        if (op_fam16(e, op)) op.pipe = fam16_default_pipe(e, op);     // the layer's family is fixed here; the autotuner picks inside it
        const int taps = ks * ks;
        const size_t per_phase = (size_t)op.nct * op.nchunks * taps * s.CB * s.KC;
        op.w_phase_stride = (long long)per_phase;
        align();
        op.w_off = blob.size();
        blob.resize(blob.size() + per_phase * op.nphase * esz, 0);
        unsigned char* wp = blob.data() + op.w_off;
        for (int ph = 0; ph < op.nphase; ++ph)
            for (int ct = 0; ct < op.nct; ++ct)
                for (int i = 0; i < op.nsrc; ++i) {
                    const Tensor& t = e->tensors[op.src[i]];
                    for (int q = op.chunk_begin[i]; q < op.chunk_begin[i + 1]; ++q) {
                        const int c0 = (q - op.chunk_begin[i]) * s.KC;
                        for (int tp = 0; tp < taps; ++tp)
                            for (int cl = 0; cl < s.CB; ++cl) {
                                const int co = ct * s.CB + cl;
                                if (co >= op.cout) continue;
                                const size_t row = ((((size_t)ph * op.nct + ct) * op.nchunks + q) * taps + tp) * s.CB + cl;
                                const int srow = tp * s.CB + cl;   // row inside the staged slab: decides the granule swizzle
                                for (int kc = 0; kc < s.KC; ++kc) {
                                    const int c = c0 + kc;
                                    if (c >= t.c) break;
                                    const int ci = cin_off[i] + c;
                                    float v;
                                    if (op.kind == OP_DECONV)  // ConvTranspose2d weight [Cin][Cout][2][2], phase = dy*2+dx
                                        v = op.weight[((size_t)ci * op.cout + co) * 4 + ph];
                                    else                        // Conv2d weight [Cout][Cin][k][k]
                                        v = op.weight[((size_t)co * op.cin + ci) * taps + tp];
                                    const int epg = 16 / (int)esz, gpr = s.KC / epg;
                                    const int pos = gpr == 2 ? ((kc / epg) ^ ((srow >> 3) & 1)) : ((kc / epg) ^ ((srow >> 1) & 7));
                                    put_elem(wp, row * s.KC + (size_t)pos * epg + kc % epg, v, dt);
                                }
                            }
                    }
                }
        align();
        op.b_off = blob.size();
        blob.resize(blob.size() + (size_t)op.nct * s.CB * 4, 0);
        memcpy(blob.data() + op.b_off, op.bias.data(), (size_t)op.cout * 4);
        if (!op.proj.empty()) {
            align();
            op.proj_off = blob.size();
            blob.resize(blob.size() + op.proj.size() * 4, 0);
            memcpy(blob.data() + op.proj_off, op.proj.data(), op.proj.size() * 4);
        }
        std::vector<float>().swap(op.weight);  // the fp32 originals are no longer needed
    }
    align();
    for (size_t i = 0; i < e->ops.size(); ++i) {       // BiFusion levels whose three 1x1-type ops can run as one kernel (the autotuner decides)
        int up = -1, c1 = -1;
        if (!bifusion_fused_possible(e, i, &up, &c1)) continue;
        e->ops[i].bf_up = up;
        e->ops[i].bf_cv1 = c1;
        e->ops[(size_t)up].bf_carrier = e->ops[(size_t)c1].bf_carrier = (int)i;
    }
    {   // whether the op order allows the sparse box kernel (see lp_engine::box_sparse_ok)
        e->box_ord.assign(e->ops.size(), -1);
        bool ok = true;
        int nbox = 0, cls_seen = 0, cls_level = -1;
        for (size_t i = 0; i < e->ops.size(); ++i) {
            const Op& o = e->ops[i];
            if (o.kind == OP_HEAD_CLS) { ++cls_seen; cls_level = o.level; }
            if (o.kind == OP_HEAD_BOX) {
                ok = ok && cls_seen == 1 && cls_level == o.level;
                e->box_ord[i] = nbox++;
                cls_seen = 0;
            }
        }
        e->box_sparse_ok = ok && nbox >= 1 && nbox <= 4 && cls_seen == 0;
    }
    e->finalized = true;
    return LP_OK;
}

extern "C" size_t lp_engine_weight_bytes(const lp_engine* e) { return e && e->finalized ? e->blob.size() : 0; }

extern "C" int lp_engine_upload(lp_engine* e, void* dev_weights, void* stream) {
    if (!e || !e->finalized) return fail(LP_ERR_STATE, "lp_engine_upload: finalize first");
    if (!dev_weights || ((uintptr_t)dev_weights & 255)) return fail(LP_ERR_ARG, "lp_engine_upload: need a 256-byte aligned device pointer");
    LP_HIP_CHECK(hipMemcpyAsync(dev_weights, e->blob.data(), e->blob.size(), hipMemcpyHostToDevice, (hipStream_t)stream));
    LP_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    e->dev_w = (char*)dev_weights;
    return LP_OK;
}

static bool rows_fits(const lp_engine* e, const Op& op);
static bool det_fits(const lp_engine* e, const Op& op);

// ---- arena ----------------------------------------------------------------------------------------------
// Tensors back to back; behind them, for the detections-only forward, a prediction scratch [B][h*w][290] fp32 for every
// pyramid level whose class predictors cannot run in the candidate-writing row kernel (too many input channels for its
// resident weights): that level still goes through prediction rows, scored by score_kernel right after.
static size_t place(const lp_engine* e, int B, int H, int W, std::vector<Tensor>* out, std::vector<size_t>* scratch = nullptr) {
    size_t off = 0;
    const size_t esz = dtype_size(e->dtype);
    for (size_t i = 0; i < e->tensors.size(); ++i) {
        Tensor t = e->tensors[i];
        t.h = H >> t.sl;
        t.w = W >> t.sl;
        t.offset = off;
        if (!t.unused) off += ((size_t)B * t.h * t.w * t.cs * esz + 255) / 256 * 256;
        if (out) (*out)[i] = t;
    }
    if (scratch) scratch->assign(e->ops.size(), (size_t)-1);
    for (size_t i = 0; i < e->ops.size(); ++i) {
        const Op& op = e->ops[i];
        if (op.kind != OP_HEAD_CLS || det_fits(e, op)) continue;
        if (scratch) (*scratch)[i] = off;
        off += ((size_t)B * (H >> (3 + op.level)) * (W >> (3 + op.level)) * LP_PRED_COLS * 4 + 255) / 256 * 256;
    }
    return off + 256;
}

// The SPPF pool chain keeps a whole h x w plane of a channel run in LDS twice (lp_aux.hip): stride-32 maps above 4096 px in
// f16 / bf16 (inputs above ~2048x2048) and above 2048 px in f32 (~1440x1440) do not fit.  Checked when a shape is bound,
// not when the first forward fails.
static bool pool_fits(const lp_engine* e, int H, int W) {
    for (const Op& op : e->ops) {
        if (op.kind != OP_POOL) continue;
        const Tensor& t = e->tensors[op.src[0]];
        const int h = H >> t.sl, w = W >> t.sl;
        if (pool_min_lds_bytes(e->dtype, h, w) > POOL_MAX_LDS || h * w >= 65536) return false;
    }
    return true;
}

// H and W must be multiples of the coarsest stride of the graph (32; 64 with a P6 level), at least 32
static int size_granule(const lp_engine* e) {
    int sl = 5;
    for (const Tensor& t : e->tensors) sl = t.sl > sl ? t.sl : sl;
    return 1 << sl;
}

extern "C" size_t lp_engine_arena_bytes(const lp_engine* e, int B, int H, int W) {
    if (!e || B < 1) return 0;
    const int g = size_granule(e);
    if (H < g || W < g || H % g || W % g || !pool_fits(e, H, W)) return 0;
    return place(e, B, H, W, nullptr);
}

extern "C" int lp_engine_bind(lp_engine* e, void* dev_arena, size_t bytes, int B, int H, int W) {
    if (!e || !e->finalized) return fail(LP_ERR_STATE, "lp_engine_bind: finalize first");
    const int gran = size_granule(e);
    if (B < 1 || H < gran || W < gran || H % gran || W % gran)
        return fail(LP_ERR_ARG, "lp_engine_bind: H and W must be positive multiples of " + std::to_string(gran) + " (the coarsest stride of the graph)");
    if (!dev_arena || ((uintptr_t)dev_arena & 255)) return fail(LP_ERR_ARG, "lp_engine_bind: need a 256-byte aligned device pointer");
    if (!pool_fits(e, H, W))
        return fail(LP_ERR_UNSUPPORTED, "lp_engine_bind: the SPPF pool chain keeps a whole stride-32 map in LDS: input too large (limit ~2048x2048 px in "
                                        "f16 / bf16, ~1440x1440 in f32)");
    const size_t need = place(e, B, H, W, nullptr);
    if (bytes < need) return fail(LP_ERR_ARG, "lp_engine_bind: arena too small");
    if ((long long)B * H * W >= (1LL << 31)) return fail(LP_ERR_UNSUPPORTED, "lp_engine_bind: batch too large for 32-bit pixel indices");
    std::vector<size_t> scratch;
    place(e, B, H, W, &e->tensors, &scratch);
    for (size_t i = 0; i < e->ops.size(); ++i) e->ops[i].det_scratch = scratch[i];
    e->arena = (char*)dev_arena;
    e->arena_bytes = bytes;
    e->B = B;
    e->H = H;
    e->W = W;
    e->level_off.assign(e->n_levels + 1, 0);
    for (int l = 0; l < e->n_levels; ++l) e->level_off[l + 1] = e->level_off[l] + (H >> (3 + l)) * (W >> (3 + l));
    e->n_anchors = e->level_off[e->n_levels];
    if (!e->dev_w) return fail(LP_ERR_STATE, "lp_engine_bind: upload the weights first");
    auto it = e->tuned.find({B, H, W});
    if (it != e->tuned.end())
        for (size_t i = 0; i < e->ops.size(); ++i) { e->ops[i].cfg = it->second[i][0]; e->ops[i].nbuf = it->second[i][1]; e->ops[i].tile = it->second[i][2]; e->ops[i].stream_wc = it->second[i][3]; e->ops[i].stream_rd = it->second[i][4]; e->ops[i].rows = it->second[i][5]; e->ops[i].pipe = it->second[i][6]; e->ops[i].planar = it->second[i][7]; e->ops[i].fused = it->second[i][8]; e->ops[i].fused_pw = it->second[i][9]; e->ops[i].fused_bf = it->second[i].size() > 10 ? it->second[i][10] : 0; }
    e->launches.assign(e->ops.size(), Launch());
    for (size_t i = 0; i < e->ops.size(); ++i) {
        int rc = prepare_op(e, i);
        if (rc) return rc;
    }
    return LP_OK;
}

extern "C" int lp_engine_tensor_info(const lp_engine* e, int id, size_t* offset, int* c, int* c_stored, int* h, int* w) {
    if (!e || !valid_tensor(e, id)) return fail(LP_ERR_ARG, "lp_engine_tensor_info: tensor id");
    const Tensor& t = e->tensors[id];
    if (offset) *offset = t.offset;
    if (c) *c = t.c;
    if (c_stored) *c_stored = t.cs;
    if (h) *h = t.h;
    if (w) *w = t.w;
    return LP_OK;
}

extern "C" int lp_engine_num_anchors(const lp_engine* e) { return e ? e->n_anchors : 0; }
extern "C" int lp_engine_num_ops(const lp_engine* e) { return e ? (int)e->ops.size() : 0; }

extern "C" int lp_engine_op_info(const lp_engine* e, int i, int* kind, int* ksize, int* cin, int* cout, double* flops, double* bytes) {
    if (!e || i < 0 || i >= (int)e->ops.size()) return fail(LP_ERR_ARG, "lp_engine_op_info: op index");
    if (!e->arena) return fail(LP_ERR_STATE, "lp_engine_op_info: bind first");
    const Op& op = e->ops[i];
    const double esz = (double)dtype_size(e->dtype);
    double fl = 0, by = 0;
    auto tbytes = [&](int id) { const Tensor& t = e->tensors[id]; return (double)e->B * t.h * t.w * t.c * esz; };
    const Tensor& s0 = e->tensors[op.kind == OP_INPUT ? op.dst : op.src[0]];
    const double in_px = (double)e->B * s0.h * s0.w;
    switch (op.kind) {
        case OP_INPUT: by = (double)e->B * e->H * e->W * 3 * esz + tbytes(op.dst); break;
        case OP_CONV: {
            const Tensor& d = e->tensors[op.dst];
            fl = 2.0 * e->B * d.h * d.w * op.cout * (op.alg_kk ? op.alg_kk : op.cin * op.ksize * op.ksize);
            for (int k = 0; k < op.nsrc; ++k) by += tbytes(op.src[k]);
            by += tbytes(op.dst) + (op.dst2 >= 0 ? tbytes(op.dst2) : 0) + (op.res >= 0 ? tbytes(op.res) : 0) + (double)op.cout * op.cin * op.ksize * op.ksize * esz;
            break;
        }
        case OP_DECONV: fl = 2.0 * in_px * op.cin * op.cout * 4; by = tbytes(op.src[0]) + tbytes(op.dst) + 4.0 * op.cin * op.cout * esz; break;
        case OP_POOL: by = 4 * tbytes(op.src[0]); break;
        case OP_HEAD_CLS:
        case OP_HEAD_BOX:
            fl = 2.0 * in_px * op.cin * op.cout;
            by = tbytes(op.src[0]) + in_px * (op.kind == OP_HEAD_CLS ? op.cout : 13) * 4 + (double)op.cin * op.cout * esz;
            break;
    }
    if (kind) *kind = op.kind;
    if (ksize) *ksize = op.ksize;
    if (cin) *cin = op.cin;
    if (cout) *cout = op.cout;
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    return LP_OK;
}

// ---- execution ------------------------------------------------------------------------------------------
// Whether the streaming 1x1 kernel (wc cout tiles per wave) can run the op: dense 1x1 stride-1 conv whose packing has whole
// cout tiles of the wave, sources made of whole 32-byte K-steps, resident weights + staging within the LDS.
static bool stream_fits(const lp_engine* e, const Op& op, int wc) {
    if (op.kind != OP_CONV || op.ksize != 1 || op.stride != 1 || op.mode != MODE_ACT) return false;
    if (op.dst2 >= 0 && e->tensors[op.dst].cs % (32 * wc) != 0) return false;      // two destinations: the second starts at a cout tile of the wave
    const int kc = 128 / (int)dtype_size(e->dtype);
    for (int i = 0; i < op.nsrc; ++i)
        if (e->tensors[op.src[i]].cs % (kc / 4) != 0) return false;      // whole 32-byte K-steps (a last K-chunk may be partial)
    return conv_stream_lds(e->dtype, wc, op.nchunks, conv_shape(e->dtype, op.cfg, 1, 1).CB) >= 0;
}

// Whether the row-writer kernel can run a class-predictor op.
static bool rows_fits(const lp_engine* e, const Op& op) {
    if (op.kind != OP_HEAD_CLS) return false;
    const int kc = 128 / (int)dtype_size(e->dtype);
    for (int i = 0; i < op.nsrc; ++i) {                     // whole K-chunks, or one partial chunk of whole 16-channel K-steps (32-channel towers)
        const int cs = e->tensors[op.src[i]].cs;
        if (cs % (kc / 4) != 0) return false;                   // whole 16-channel K-steps (a last K-chunk may be partial: 96, 192 channels)
    }
    return head_rows_fits(e->dtype, op.nchunks, conv_shape(e->dtype, op.cfg, 1, 1).CB, op.cout);
}

// Whether the detections-only class-predictor kernel (head_det_kernel) can run the op.
static bool det_fits(const lp_engine* e, const Op& op) {
    if (op.kind != OP_HEAD_CLS || getenv("LP_NO_HEAD_DET")) return false;
    const int kc = 128 / (int)dtype_size(e->dtype);
    for (int i = 0; i < op.nsrc; ++i) {
        const int cs = e->tensors[op.src[i]].cs;
        if (cs % (kc / 4) != 0) return false;                   // whole 16-channel K-steps (a last K-chunk may be partial: 96, 192 channels)
    }
    return head_det_fits(e->dtype, op.nchunks, conv_shape(e->dtype, op.cfg, 1, 1).CB, op.cout);
}

// MFMA family of a layer.  The 16x16x32 kernels (lp_conv3x3_pipe16*.inc) sum in another fp32 order than
// every other variant, so which family computes a layer must not depend on the batch size or on timing: it is this predicate of the
// layer alone (3x3 stride 1, 16-bit, K-chunks a multiple of four, 128-row weight packing: more than 64 stored output channels).  Inside a family the variants (tile
// shapes, wave grids) are bit-identical and the autotuner picks by time.
static bool op_fam16(const lp_engine* e, const Op& op) {
    if (!e->mfma16 || e->dtype == LP_F32 || op.kind != OP_CONV || op.ksize != 3 || op.mode != MODE_ACT) return false;
    if (op.stride == 2)       // stride 2 (lp_conv3x3_s2p16.inc): 128-row packing, one destination; only in engines created under LP_S2P16=1
        return e->s2p16 && op.dst2 < 0 && conv_pipe_fits(e->dtype, PIPE16_S2A, conv_shape(e->dtype, op.cfg, 3, 2).CB, 3, 2, op.mode, op.nct, op.nphase, op.nchunks);
    if (op.stride != 1) return false;
    return conv_pipe_fits(e->dtype, PIPE16_D, conv_shape(e->dtype, op.cfg, 1, 1).CB, 3, 1, op.mode, op.nct, op.nphase, op.nchunks);
}
static int fam16_default_pipe(const lp_engine*, const Op& op) { return (op.stride == 2 ? PIPE16_S2A : PIPE16_D) + 1; }

// Launch geometry of one conv-type op for the bound shape and the op's current kernel variant.
static int prepare_op(lp_engine* e, size_t idx) {
    ++e->epoch;
    const Op& op = e->ops[idx];
    Launch& L = e->launches[idx];
    L.is_conv = !(op.kind == OP_INPUT || op.kind == OP_POOL);
    if (!L.is_conv) return LP_OK;
    const int dt = e->dtype;
    auto tptr = [&](int id) { return (void*)(e->arena + e->tensors[id].offset); };
    const int ks = op.kind == OP_CONV ? op.ksize : 1, stv = op.kind == OP_CONV ? op.stride : 1;
    const int cb_pack = conv_shape(dt, op.cfg, ks, stv).CB;
    const ConvShape s = op.pipe ? conv_pipe_shape(op.pipe - 1) : conv_shape(dt, op.cfg, ks, stv);
    ConvArgs& a = L.a;
    memset(&a, 0, sizeof(a));
    a.nsrc = op.nsrc;
    for (int i = 0; i < op.nsrc; ++i) { a.src[i].ptr = tptr(op.src[i]); a.src[i].cs = e->tensors[op.src[i]].cs; }
    for (int i = 0; i <= LP_MAX_SRC; ++i) a.chunk_begin[i] = op.chunk_begin[i];
    a.zero = e->dev_w;
    a.trash = e->dev_w + 128;
    a.w = e->dev_w + op.w_off;
    a.bias = (const float*)(e->dev_w + op.b_off);
    const Tensor& s0 = e->tensors[op.src[0]];
    a.B = e->B;
    a.H = s0.h;
    a.W = s0.w;
    a.Ho = stv == 2 ? s0.h / 2 : s0.h;
    a.Wo = stv == 2 ? s0.w / 2 : s0.w;
    if (op.pipe && pipe_is_16s2(op.pipe - 1)) {
        conv_pick_tile16v(s, a.Ho, a.Wo, e->B, op.nct, op.tile, &a.TH, &a.TW, 2);
        a.hpitch = 2 * a.TW + 1;  // halo rows stored evens-first, unswizzled, no padding
    } else if (op.pipe && pipe_is_16v(op.pipe - 1)) {
        conv_pick_tile16v(s, a.Ho, a.Wo, e->B, op.nct, op.tile, &a.TH, &a.TW);
        a.hpitch = a.TW + 2;      // unswizzled rows: conflict-free for the 16x16 operand map at any pitch
    } else {
        conv_pick_tile(s, ks, stv, a.Ho, a.Wo, op.tile, &a.TH, &a.TW);
        a.hpitch = (op.pipe && pipe_is_16(op.pipe - 1)) ? conv_pick_pitch16(s, a.TH, a.TW) : conv_pick_pitch(s, dt, ks, stv, a.TH, a.TW);
    }
    a.tw_magic = (unsigned)(((1u << 22) + a.TW - 1) / a.TW);          // n / TW == (n * magic) >> 22 for n * TW < 2^22
    a.hp_magic = (unsigned)(((1u << 22) + a.hpitch - 1) / a.hpitch);
    a.tiles_x = ceil_div(a.Wo, a.TW);
    a.tiles_y = ceil_div(a.Ho, a.TH);
    a.nct = op.nct;
    a.nphase = op.nphase;
    a.w_phase_stride = op.w_phase_stride;
    a.out_scale = op.kind == OP_DECONV ? 2 : 1;
    a.act = op.act;
    a.alpha = op.alpha;
    if (op.mode == MODE_ACT) {
        const Tensor& d = e->tensors[op.dst];
        a.out = tptr(op.dst);
        a.out_c = d.cs;
        a.out_pix_stride = d.cs;
        a.out_img_stride = (long long)d.h * d.w * d.cs;
        if (op.kind == OP_CONV && op.dst2 >= 0) {
            const Tensor& d2 = e->tensors[op.dst2];
            a.out2 = tptr(op.dst2);
            a.out_split = d.cs;
            a.out_c = d.cs + d2.cs;
            a.out2_pix_stride = d2.cs;
            a.out2_img_stride = (long long)d2.h * d2.w * d2.cs;
        }
        if (op.res >= 0) { a.res = tptr(op.res); a.res_cs = e->tensors[op.res].cs; }
    } else {
        a.out = nullptr;   // patched per call: pred + pred_off
        L.pred_off = (long long)e->level_off[op.level] * LP_PRED_COLS + (op.mode == MODE_PRED ? 13 : 0);
        a.out_pix_stride = LP_PRED_COLS;
        a.out_img_stride = (long long)e->n_anchors * LP_PRED_COLS;
        a.out_c = op.cout;
        if (op.mode == MODE_DECODE) {
            a.reg_bins = op.reg_bins;
            a.proj = op.proj.empty() ? nullptr : (const float*)(e->dev_w + op.proj_off);
            a.stride_px = (float)(8 << op.level);
        }
    }
    a.stamps = g_stamps;
    L.cfg = op.cfg;
    L.mode = op.mode;
    L.ks = ks;
    L.st = stv;
    L.nbuf = op.nbuf;
    L.stream_wc = op.stream_wc;
    L.stream_rd = op.stream_rd;
    L.rows = op.rows;
    L.pipe = op.pipe;
    L.cb_pack = cb_pack;
    if (op.planar) {                    // the same layer as the PIPE_P kernel: its own tile geometry, src[0] patched per call
        Launch& P = e->stem_planar;
        P = L;
        ConvArgs& pa = P.a;
        if (pa.W % 4 != 0 || !stem_planar_tile(pa.Ho, pa.Wo, op.planar - 1, &pa.TH, &pa.TW))
            return fail(LP_ERR_UNSUPPORTED, "planar stem: no tile for this frame size");
        pa.tw_magic = (unsigned)(((1u << 22) + pa.TW - 1) / pa.TW);
        pa.tiles_x = ceil_div(pa.Wo, pa.TW);
        pa.tiles_y = ceil_div(pa.Ho, pa.TH);
        pa.src[0].ptr = nullptr;
        P.pipe = PIPE_P + 1;
    }
    if (op.fused) {                     // input op + stem + this layer as one kernel: this layer's arguments + the stem's operands
        const Op& stem = e->ops[1];
        Launch& F = e->stem2_fused;
        F = L;
        ConvArgs& fa = F.a;
        if (!stem2_fused_tile(fa.Ho, fa.Wo, op.fused - 1, &fa.TH, &fa.TW, &fa.hpitch))
            return fail(LP_ERR_UNSUPPORTED, "fused stem: no tile for this frame size");
        fa.tw_magic = (unsigned)(((1u << 22) + fa.TW - 1) / fa.TW);
        fa.hp_magic = (unsigned)(((1u << 22) + fa.hpitch - 1) / fa.hpitch);
        fa.tiles_x = ceil_div(fa.Wo, fa.TW);
        fa.tiles_y = ceil_div(fa.Ho, fa.TH);
        fa.src[0].ptr = nullptr;
        fa.fz_w1 = e->dev_w + stem.w_off;
        fa.fz_b1 = (const float*)(e->dev_w + stem.b_off);
        fa.fz_act1 = stem.act;
        fa.fz_c1 = e->tensors[stem.dst].cs;
        F.pipe = PIPE_FUSED2 + 1;
    }
    if (op.fused_bf) {                  // BiFusion: transposed conv + cv1 + this cv3 as one kernel (this op's arguments + the other two ops' operands)
        const Op& up = e->ops[op.bf_up];
        const Op& c1 = e->ops[op.bf_cv1];
        Launch& F = e->bf_fused[(int)idx];
        F = L;
        ConvArgs& fa = F.a;
        fa.src[0].ptr = tptr(up.src[0]); fa.src[0].cs = e->tensors[up.src[0]].cs;
        fa.src[1].ptr = tptr(c1.src[0]); fa.src[1].cs = e->tensors[c1.src[0]].cs;
        fa.fz_w1 = e->dev_w + c1.w_off;
        fa.fz_b1 = (const float*)(e->dev_w + c1.b_off);
        fa.fz_act1 = c1.act;
        fa.bf_wd = e->dev_w + up.w_off;
        fa.bf_bd = (const float*)(e->dev_w + up.b_off);
        fa.bf_wd_phase_stride = up.w_phase_stride;
        F.pipe = PIPE_FUSED_BF + 1;
    }
    if (op.fused_pw) {                  // the 1x1 layer before this one + this layer as one kernel
        const Op& pw = e->ops[idx - 1];
        Launch& F = e->pw_fused[(int)idx];
        F = L;
        ConvArgs& fa = F.a;
        int hp = 0;
        if (!stem2_fused_tile(fa.Ho, fa.Wo, op.fused_pw - 1, &fa.TH, &fa.TW, &hp))
            return fail(LP_ERR_UNSUPPORTED, "fused 1x1 + 3x3 s2: no tile for this map size");
        fa.hpitch = hp;
        fa.tw_magic = (unsigned)(((1u << 22) + fa.TW - 1) / fa.TW);
        fa.hp_magic = (unsigned)(((1u << 22) + fa.hpitch - 1) / fa.hpitch);
        fa.tiles_x = ceil_div(fa.Wo, fa.TW);
        fa.tiles_y = ceil_div(fa.Ho, fa.TH);
        fa.src[0].ptr = tptr(pw.src[0]);
        fa.src[0].cs = e->tensors[pw.src[0]].cs;
        fa.fz_w1 = e->dev_w + pw.w_off;
        fa.fz_b1 = (const float*)(e->dev_w + pw.b_off);
        fa.fz_act1 = pw.act;
        fa.fz_c1 = e->tensors[pw.dst].cs;
        F.pipe = PIPE_FUSED_PW + 1;
    }
    return LP_OK;
}

// The stem may read the caller's frame itself (PIPE_P) when the frame has the engine's 16-bit dtype.
static bool stem_planar_possible(const lp_engine* e) {
    return e->dtype != LP_F32 && e->ops.size() > 1 && e->ops[0].kind == OP_INPUT && e->ops[0].s2d && e->ops[1].kind == OP_CONV &&
           e->ops[1].ksize == 3 && e->ops[1].stride == 1 && e->ops[1].mode == MODE_ACT && e->ops[1].nct == 1 && e->ops[1].nchunks == 1 &&
           (conv_shape(e->dtype, e->ops[1].cfg, 3, 1).CB == 32 || conv_shape(e->dtype, e->ops[1].cfg, 3, 1).CB == 64) && e->ops[1].res < 0 &&
           e->ops[1].dst2 < 0;
}
// (the frame must have the engine's dtype and be 16-byte aligned -- the kernels copy it in 16-byte pieces; anything else takes the input op)
static bool frame_direct(const lp_engine* e, const void* x, int x_dtype) { return x_dtype == e->dtype && ((uintptr_t)x & 15) == 0; }
static bool stem_planar_now(const lp_engine* e, const void* x, int x_dtype) { return e->ops.size() > 1 && e->ops[1].planar && frame_direct(e, x, x_dtype); }
// ... and run together with the layer behind it (lp_stem2_fused.inc) when that is a 3x3 stride-2 layer of at most 64 output channels
// and the stem's output (at most 32 channels) has no other reader.
static bool stem2_fused_possible(const lp_engine* e) {
    if (!stem_planar_possible(e) || e->ops.size() < 3) return false;
    const Op& s = e->ops[1];
    const Op& c = e->ops[2];
    if (c.kind != OP_CONV || c.ksize != 3 || c.stride != 2 || c.nsrc != 1 || c.src[0] != s.dst || c.res >= 0 || c.dst2 >= 0 || c.mode != MODE_ACT ||
        c.nct != 1 || c.nphase != 1 || c.nchunks < 1 || c.nchunks > 2 || s.act != LP_ACT_RELU || c.act != LP_ACT_RELU) return false;
    const int cb = conv_shape(e->dtype, c.cfg, 3, 2).CB, cs2 = e->tensors[c.dst].cs;
    if ((cb != 32 && cb != 64) || cb != 32 * ((cs2 + 31) / 32) || e->tensors[s.dst].cs > 16 * c.nchunks) return false;
    for (size_t i = 3; i < e->ops.size(); ++i) {
        const Op& o = e->ops[i];
        for (int k = 0; k < o.nsrc; ++k) if (o.src[k] == s.dst) return false;
        if (o.res == s.dst) return false;
    }
    return true;
}
// A 1x1 layer (64 stored input channels, <= 64 output channels) whose only reader is the 3x3 stride-2 layer (<= 64 output
// channels) right behind it on the same lane: BiFusion's downsample(cv2(x)).  `i` = index of the 3x3 layer.
static bool pw_fused_possible(const lp_engine* e, size_t i) {
    if (e->dtype == LP_F32 || i < 2 || i >= e->ops.size()) return false;
    const Op& p = e->ops[i - 1];
    const Op& c = e->ops[i];
    if (c.kind != OP_CONV || c.ksize != 3 || c.stride != 2 || c.nsrc != 1 || c.res >= 0 || c.dst2 >= 0 || c.mode != MODE_ACT || c.nct != 1 || c.nphase != 1) return false;
    if (p.kind != OP_CONV || p.ksize != 1 || p.stride != 1 || p.nsrc != 1 || p.res >= 0 || p.dst2 >= 0 || p.mode != MODE_ACT || p.nct != 1 || p.nphase != 1 ||
        p.nchunks != 1 || p.dst != c.src[0] || p.lane != c.lane || p.signal) return false;
    const int cs0 = e->tensors[p.src[0]].cs, cs1 = e->tensors[p.dst].cs, cs2 = e->tensors[c.dst].cs;
    if (cs0 != 64 || (cs1 != 32 && cs1 != 64) || (cs2 != 32 && cs2 != 64)) return false;
    if (conv_shape(e->dtype, p.cfg, 1, 1).CB != cs1 || conv_shape(e->dtype, c.cfg, 3, 2).CB != cs2 || c.nchunks != cs1 / 16) return false;
    for (size_t k = 0; k < e->ops.size(); ++k) {
        if (k == i) continue;
        const Op& o = e->ops[k];
        for (int q = 0; q < o.nsrc; ++q) if (o.src[q] == p.dst) return false;
        if (o.res == p.dst) return false;
    }
    return true;
}
// BiFusion (common.py:504-527) at 64 channels: cv3 = 1x1 ReLU over [u, a, d] where u is the output of a transposed conv (no activation) of a
// 64-channel coarse map and a the output of a 1x1 ReLU layer of a 128-channel map, u and a read by nobody else, every packing with 64-row cout
// tiles.  Fills the indices of the two carried ops.  (With execution lanes on, a carried op still waits for its producers and records its event
// on its lane -- it only launches nothing -- so the fused op's dependencies on the two carried ops order it behind THEIR inputs.)
static bool bifusion_fused_possible(const lp_engine* e, size_t i, int* up, int* cv1) {
    if (e->dtype == LP_F32 || i >= e->ops.size()) return false;
    const Op& c = e->ops[i];
    if (c.kind != OP_CONV || c.ksize != 1 || c.stride != 1 || c.nsrc != 3 || c.res >= 0 || c.dst2 >= 0 || c.mode != MODE_ACT || c.nct != 1 || c.nphase != 1 ||
        c.nchunks != 3 || c.act != LP_ACT_RELU || conv_shape(e->dtype, c.cfg, 1, 1).CB != 64 || e->tensors[c.dst].cs != 64) return false;
    int ju = -1, ja = -1;
    for (size_t k = 0; k < i; ++k) {
        if (e->ops[k].kind == OP_DECONV && e->ops[k].dst == c.src[0]) ju = (int)k;
        if (e->ops[k].kind == OP_CONV && e->ops[k].dst == c.src[1]) ja = (int)k;
    }
    if (ju < 0 || ja < 0) return false;
    const Op& u = e->ops[ju];
    const Op& p = e->ops[ja];
    if (u.nsrc != 1 || u.nct != 1 || u.nphase != 4 || u.nchunks != 1 || u.act != LP_ACT_NONE ||
        conv_shape(e->dtype, u.cfg, 1, 1).CB != 64 || e->tensors[u.src[0]].cs != 64 || e->tensors[u.dst].cs != 64) return false;
    if (p.ksize != 1 || p.stride != 1 || p.nsrc != 1 || p.res >= 0 || p.dst2 >= 0 || p.mode != MODE_ACT || p.nct != 1 || p.nphase != 1 || p.nchunks != 2 ||
        p.act != LP_ACT_RELU || conv_shape(e->dtype, p.cfg, 1, 1).CB != 64 || e->tensors[p.src[0]].cs != 128 ||
        e->tensors[p.dst].cs != 64 || e->tensors[c.src[2]].cs != 64) return false;
    const Tensor& tf = e->tensors[c.dst];
    const Tensor& tc = e->tensors[u.src[0]];
    if (tf.h != 2 * tc.h || tf.w != 2 * tc.w) return false;
    for (size_t k = 0; k < e->ops.size(); ++k) {
        if (k == i) continue;
        const Op& o = e->ops[k];
        for (int q = 0; q < o.nsrc; ++q) if (o.src[q] == u.dst || o.src[q] == p.dst) return false;
        if (o.res == u.dst || o.res == p.dst) return false;
    }
    if (up) *up = ju;
    if (cv1) *cv1 = ja;
    return true;
}
static bool stem2_fused_now(const lp_engine* e, const void* x, int x_dtype) { return e->ops.size() > 2 && e->ops[2].fused && frame_direct(e, x, x_dtype); }

// Detections-only forward: where the head ops write instead of the prediction tensor.
struct DetCtx {
    NmsWs w;
    float conf;
};

static int run_head_det(lp_engine* e, size_t idx, const DetCtx& dc, hipStream_t st) {
    const Op& op = e->ops[idx];
    const Launch& L = e->launches[idx];
    const int dt = e->dtype;
    ConvArgs a = L.a;
    const int anchor0 = e->level_off[op.level], N = e->n_anchors;
    if (L.mode == MODE_DECODE) {            // columns 0..11 of every anchor's candidate row
        a.det_mode = 1;
        a.out = dc.w.rows + (long long)anchor0 * LP_DET_COLS;
        a.out_pix_stride = LP_DET_COLS;
        a.out_img_stride = (long long)N * LP_DET_COLS;
        static const bool no_box_stream = getenv("LP_NO_BOX_STREAM") != nullptr;     // (A/B switch: the generic decode kernel)
        static const bool no_box_sparse = getenv("LP_NO_BOX_SPARSE") != nullptr;     // (A/B switch: boxes of every anchor)
        if (!no_box_stream && head_box_det_fits(a, L.cb_pack, L.ks, L.st)) {
            // one lane: the class kernel of this level has appended its candidates in front of this launch, nobody else in between
            if (!no_box_sparse && op.box_sparse && e->box_sparse_ok && (e->n_lanes <= 1 || e->single_lane) && e->box_ord[idx] >= 0 && N >= 8) {
                const int k = e->box_ord[idx];
                int* const snaps = dc.w.kept;           // [4][B] ints of the NMS's `kept` array, which lp_nms_candidates only writes later
                a.det_keys = dc.w.keys;
                a.det_cnt = dc.w.cnt;
                a.det_np = dc.w.NP;
                a.det_n = N;
                a.det_anchor0 = anchor0;
                a.det_prev = k > 0 ? snaps + (size_t)(k - 1) * e->B : nullptr;
                a.det_snap = snaps + (size_t)k * e->B;
            }
            return head_box_det_launch(dt, a, L.cb_pack, st);
        }
        return conv_launch(dt, L.cfg, L.mode, L.ks, L.st, L.nbuf, a, st);
    }
    if (op.det_scratch == (size_t)-1) {     // class predictors + candidate selection in one kernel
        a.det_mode = 1;
        a.out = dc.w.rows;
        a.det_keys = dc.w.keys;
        a.det_cnt = dc.w.cnt;
        a.det_np = dc.w.NP;
        a.det_n = N;
        a.det_anchor0 = anchor0;
        a.det_conf = dc.conf;
        return head_rows_launch(dt, a, L.cb_pack, st);
    }
    // this level's class predictors do not fit the row kernel: prediction rows in the scratch, scored right behind
    float* const scratch = (float*)(e->arena + op.det_scratch);
    const int rpi = a.Ho * a.Wo;
    a.out = scratch + 13;
    a.out_img_stride = (long long)rpi * LP_PRED_COLS;
    int rc = conv_launch(dt, L.cfg, L.mode, L.ks, L.st, L.nbuf, a, st);
    if (rc) return rc;
    return nms_score_launch(scratch, e->B, rpi, anchor0, N, dc.conf, dc.w, false, st);
}

static int run_op(lp_engine* e, size_t idx, const void* x, int x_dtype, float* pred, hipStream_t st, const DetCtx* det = nullptr) {
    const Op& op = e->ops[idx];
    const int dt = e->dtype;
    auto tptr = [&](int id) { return (void*)(e->arena + e->tensors[id].offset); };
    if (op.kind == OP_INPUT) {
        if (op.s2d && (stem_planar_now(e, x, x_dtype) || stem2_fused_now(e, x, x_dtype))) return LP_OK;   // the stem reads x itself
        return op.s2d ? input_s2d_launch(x, x_dtype, tptr(op.dst), dt, e->B, e->H, e->W, st)
                      : input_launch(x, x_dtype, tptr(op.dst), dt, e->B, e->H, e->W, st);
    }
    if (op.kind == OP_POOL) {
        const Tensor& t = e->tensors[op.src[0]];
        return pool_launch(tptr(op.src[0]), tptr(op.dst), tptr(op.dst2), tptr(op.dst3), dt, e->B, t.h, t.w, t.cs, st);
    }
    const Launch& L = e->launches[idx];
    if (idx + 1 < e->ops.size() && e->ops[idx + 1].fused_pw) return LP_OK;   // this 1x1 layer runs inside the fused kernel of the next op
    if (op.bf_carrier >= 0 && e->ops[op.bf_carrier].fused_bf) return LP_OK;  // ... inside BiFusion's fused kernel (the cv3 op)
    if (op.fused_bf) return conv_pipe_launch(dt, PIPE_FUSED_BF, e->bf_fused[(int)idx].a, st);
    if (op.fused_pw) return conv_pipe_launch(dt, PIPE_FUSED_PW, e->pw_fused[(int)idx].a, st);
    if (idx == 1 && stem2_fused_now(e, x, x_dtype)) return LP_OK;            // runs inside the fused kernel of op 2
    if (idx == 2 && stem2_fused_now(e, x, x_dtype)) {
        ConvArgs a = e->stem2_fused.a;
        a.src[0].ptr = x;
        return conv_pipe_launch(dt, PIPE_FUSED2, a, st);
    }
    if (idx == 1 && op.planar && stem_planar_now(e, x, x_dtype)) {
        ConvArgs a = e->stem_planar.a;
        a.src[0].ptr = x;
        return conv_pipe_launch(dt, PIPE_P, a, st);
    }
    if (L.mode == MODE_ACT && L.pipe) return conv_pipe_launch(dt, L.pipe - 1, L.a, st);
    if (L.mode == MODE_ACT && L.stream_wc) return conv_stream_launch(dt, L.stream_wc, L.a, L.cb_pack, st);
    if (L.mode == MODE_ACT) return conv_launch(dt, L.cfg, L.mode, L.ks, L.st, L.nbuf, L.a, st);
    if (det) return run_head_det(e, idx, *det, st);
    if (!pred) return fail(LP_ERR_ARG, "forward: pred is null");
    ConvArgs a = L.a;
    a.out = pred + L.pred_off;
    if (L.mode == MODE_PRED && L.rows) return head_rows_launch(dt, a, L.cb_pack, st);
    return conv_launch(dt, L.cfg, L.mode, L.ks, L.st, L.nbuf, a, st);
}

static int check_ready(const lp_engine* e, const void* x, int x_dtype) {
    if (!e || !e->finalized) return fail(LP_ERR_STATE, "forward: engine not finalized");
    if (!e->dev_w) return fail(LP_ERR_STATE, "forward: weights not uploaded");
    if (!e->arena) return fail(LP_ERR_STATE, "forward: arena not bound");
    if (!x) return fail(LP_ERR_ARG, "forward: x is null");
    if (x_dtype != LP_F16 && x_dtype != LP_BF16 && x_dtype != LP_F32) return fail(LP_ERR_ARG, "forward: x dtype");
    return LP_OK;
}

static int ensure_lanes(lp_engine* e) {
    for (int l = 1; l < e->n_lanes; ++l)
        if (!e->lane_stream[l]) LP_HIP_CHECK(hipStreamCreateWithFlags(&e->lane_stream[l], hipStreamNonBlocking));
    if (!e->fork_ev) LP_HIP_CHECK(hipEventCreateWithFlags(&e->fork_ev, hipEventDisableTiming));
    for (int l = 1; l < e->n_lanes; ++l)
        if (!e->join_ev[l]) LP_HIP_CHECK(hipEventCreateWithFlags(&e->join_ev[l], hipEventDisableTiming));
    if (e->op_event.size() != e->ops.size()) e->op_event.assign(e->ops.size(), nullptr);
    for (size_t i = 0; i < e->ops.size(); ++i)
        if (e->ops[i].signal && !e->op_event[i]) LP_HIP_CHECK(hipEventCreateWithFlags(&e->op_event[i], hipEventDisableTiming));
    return LP_OK;
}

static int issue_forward(lp_engine* e, const void* x, int x_dtype, float* pred, hipStream_t main_st, const DetCtx* det = nullptr) {
    int rc;
    if (e->n_lanes <= 1 || e->single_lane) {      // (op order is a topological order of the graph: the host adds producers first)
        for (size_t i = 0; i < e->ops.size(); ++i) {
            rc = run_op(e, i, x, x_dtype, pred, main_st, det);
            if (rc) return rc;
        }
        return LP_OK;
    }
    // Independent branches of the graph (BiFusion inputs, the per-level head towers) run on side streams so that
    // small latency-bound layers overlap; the caller's stream forks them at the start and joins them at the end, so
    // from the outside the whole forward is still ordered on `stream`.
    rc = ensure_lanes(e);
    if (rc) return rc;
    e->lane_stream[0] = main_st;
    LP_HIP_CHECK(hipEventRecord(e->fork_ev, main_st));
    for (int l = 1; l < e->n_lanes; ++l) LP_HIP_CHECK(hipStreamWaitEvent(e->lane_stream[l], e->fork_ev, 0));
    for (size_t i = 0; i < e->ops.size(); ++i) {
        const Op& op = e->ops[i];
        hipStream_t st = e->lane_stream[op.lane];
        for (int d : op.deps) LP_HIP_CHECK(hipStreamWaitEvent(st, e->op_event[d], 0));
        rc = run_op(e, i, x, x_dtype, pred, st, det);
        if (rc) return rc;
        if (op.signal) LP_HIP_CHECK(hipEventRecord(e->op_event[i], st));
    }
    for (int l = 1; l < e->n_lanes; ++l) {
        LP_HIP_CHECK(hipEventRecord(e->join_ev[l], e->lane_stream[l]));
        LP_HIP_CHECK(hipStreamWaitEvent(main_st, e->join_ev[l], 0));
    }
    return LP_OK;
}

extern "C" int lp_engine_set_single_lane(lp_engine* e, int enable) {
    if (!e) return fail(LP_ERR_ARG, "lp_engine_set_single_lane: null engine");
    if (e->single_lane != (enable != 0)) ++e->epoch;      // captured graphs hold the lane structure
    e->single_lane = enable != 0;
    return LP_OK;
}

extern "C" int lp_engine_set_mfma16(lp_engine* e, int enable) {
    if (!e) return fail(LP_ERR_ARG, "lp_engine_set_mfma16: null engine");
    if (e->finalized) return fail(LP_ERR_STATE, "lp_engine_set_mfma16: the families are fixed by lp_engine_finalize");
    e->mfma16 = enable != 0;
    return LP_OK;
}

extern "C" int lp_engine_set_graph(lp_engine* e, int enable) {
    if (!e) return fail(LP_ERR_ARG, "lp_engine_set_graph: null engine");
    e->use_graph = enable != 0;
    return LP_OK;
}

// Marks the end of a forward on the caller's stream with the engine's own event (see lp_engine_destroy).
static int mark_done(lp_engine* e, hipStream_t st) {
    // (no system-scope fence: the event orders device work for lp_engine_destroy, nobody reads memory behind it on the host)
    if (!e->done_ev) LP_HIP_CHECK(hipEventCreateWithFlags(&e->done_ev, hipEventDisableTiming | hipEventDisableSystemFence));
    LP_HIP_CHECK(hipEventRecord(e->done_ev, st));
    e->done_valid = true;
    return LP_OK;
}

// One forward, plain (det == nullptr: writes pred) or detections-only (det: the head writes NMS candidates into det's
// workspace, whose counters are zeroed first), re-issued launch by launch or -- lp_engine_set_graph -- replayed as one hipGraph:
// the ~80 launches (and the lane fork / join events) of a forward are captured once per (input pointer, output pointer or
// workspace + threshold, dtype, launch geometry) and replayed with a single hipGraphLaunch -- what the per-image loop of
// Inferer needs, where the forward is launch-bound.
static int run_forward(lp_engine* e, const void* x, int x_dtype, float* pred, const DetCtx* det, void* ws, hipStream_t main_st) {
    int rc;
    auto issue = [&](hipStream_t st) -> int {
        if (det) { if (int zr = zero_counts_launch(det->w.cnt, e->B, st)) return zr; }   // BEFORE the lanes fork: every head op comes behind it
        return issue_forward(e, x, x_dtype, pred, st, det);
    };
    if (!e->use_graph) {
        rc = issue(main_st);
        return rc ? rc : mark_done(e, main_st);
    }
    constexpr size_t kMaxGraphs = 8;
    const float conf = det ? det->conf : 0.f;
    lp_engine::CachedGraph* hit = nullptr;
    for (auto& g : e->graphs)
        if (g.exec && g.x == x && g.pred == pred && g.ws == ws && g.conf == conf && g.x_dtype == x_dtype && g.epoch == e->epoch) hit = &g;
    if (!hit) {
        // retire stale captures (re-tuned / re-bound engine) and, when the cache is full, the least recently used one; an
        // executable graph must outlive its last launch, so that launch's stream is synchronised first
        for (size_t i = 0; i < e->graphs.size();) {
            auto& g = e->graphs[i];
            bool drop = g.epoch != e->epoch;
            if (!drop && e->graphs.size() >= kMaxGraphs) {
                drop = true;
                for (const auto& o : e->graphs) drop = drop && o.last_use >= g.last_use;
            }
            if (!drop) { ++i; continue; }
            if (g.stream) (void)hipStreamSynchronize(g.stream);
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            e->graphs.erase(e->graphs.begin() + (long)i);
        }
        rc = ensure_lanes(e);
        if (rc) return rc;
        hipGraph_t g = nullptr;
        if (!e->cap_stream) LP_HIP_CHECK(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
        LP_HIP_CHECK(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal));
        rc = issue(e->cap_stream);
        hipError_t ce = hipStreamEndCapture(e->cap_stream, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (ce != hipSuccess) return fail(LP_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
        lp_engine::CachedGraph cg;
        ce = hipGraphInstantiate(&cg.exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ce != hipSuccess) return fail(LP_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ce));
        cg.x = x;
        cg.pred = pred;
        cg.ws = ws;
        cg.conf = conf;
        cg.x_dtype = x_dtype;
        cg.epoch = e->epoch;
        e->graphs.push_back(cg);
        hit = &e->graphs.back();
    }
    LP_HIP_CHECK(hipGraphLaunch(hit->exec, main_st));
    hit->stream = main_st;
    hit->last_use = ++e->graph_clock;
    return mark_done(e, main_st);
}

extern "C" int lp_engine_forward(lp_engine* e, const void* x, int x_dtype, float* pred, void* stream) {
    int rc = check_ready(e, x, x_dtype);
    if (rc) return rc;
    if (!pred) return fail(LP_ERR_ARG, "forward: pred is null");
    return run_forward(e, x, x_dtype, pred, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int lp_engine_forward_det(lp_engine* e, const void* x, int x_dtype, double conf_thres, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    int rc = check_ready(e, x, x_dtype);
    if (rc) return rc;
    if (!workspace || ((uintptr_t)workspace & 255)) return fail(LP_ERR_ARG, "lp_engine_forward_det: need a 256-byte aligned workspace");
    if (!(conf_thres >= 0.0 && conf_thres <= 1.0)) return fail(LP_ERR_ARG, "lp_engine_forward_det: conf_thres must be in [0, 1]");
    DetCtx dc;
    dc.w = nms_carve(workspace, e->B, e->n_anchors);
    dc.conf = (float)conf_thres;
    if (workspace_bytes < dc.w.bytes) return fail(LP_ERR_ARG, "lp_engine_forward_det: workspace too small (lp_nms_workspace_bytes)");
    return run_forward(e, x, x_dtype, nullptr, &dc, workspace, (hipStream_t)stream);
}

static int forward_single_lane(lp_engine* e, const void* x, int x_dtype, float* pred, hipStream_t st) {
    for (size_t i = 0; i < e->ops.size(); ++i) {
        int rc = run_op(e, i, x, x_dtype, pred, st);
        if (rc) return rc;
    }
    return LP_OK;
}

extern "C" int lp_engine_profile_ops(lp_engine* e, const void* x, int x_dtype, float* pred, void* stream, float* op_ms, int reps,
                                     int inner) {
    int rc = check_ready(e, x, x_dtype);
    if (rc) return rc;
    if (!op_ms || reps < 1 || inner < 1) return fail(LP_ERR_ARG, "profile: op_ms / reps / inner");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = e->ops.size();
    while (e->events.size() < n + 1) {
        hipEvent_t ev;
        LP_HIP_CHECK(hipEventCreate(&ev));
        e->events.push_back(ev);
    }
    rc = forward_single_lane(e, x, x_dtype, pred, (hipStream_t)stream);  // untimed warm run
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) op_ms[i] = 0.f;
    for (int r = 0; r < reps; ++r) {
        LP_HIP_CHECK(hipEventRecord(e->events[0], st));
        for (size_t i = 0; i < n; ++i) {
            // `inner` back-to-back launches of the op between two events: the event pair's own cost (a few microseconds of
            // queue bubble) is spread over them, so the figure approaches the kernel's duration as a kernel trace reports it
            // (the ops rewrite their own outputs with identical values)
            for (int k = 0; k < inner; ++k) {
                rc = run_op(e, i, x, x_dtype, pred, st);
                if (rc) return rc;
            }
            LP_HIP_CHECK(hipEventRecord(e->events[i + 1], st));
        }
        LP_HIP_CHECK(hipEventSynchronize(e->events[n]));
        for (size_t i = 0; i < n; ++i) {
            float ms = 0.f;
            LP_HIP_CHECK(hipEventElapsedTime(&ms, e->events[i], e->events[i + 1]));
            op_ms[i] += ms / reps / inner;
        }
    }
    return LP_OK;
}

extern "C" int lp_engine_profile(lp_engine* e, const void* x, int x_dtype, float* pred, void* stream, float* op_ms, int reps) {
    return lp_engine_profile_ops(e, x, x_dtype, pred, stream, op_ms, reps, 1);
}

// Per-layer choice of the conv kernel variant for the bound shape: every variant that shares the op's weight
// packing (same cout tile) is timed in place with hipEvent pairs and the fastest is kept.  The op graph is run
// once first so that every op sees valid inputs; ops rewrite their own outputs with identical values.
extern "C" int lp_engine_autotune(lp_engine* e, const void* x, int x_dtype, float* pred, void* stream, int reps) {
    int rc = check_ready(e, x, x_dtype);
    if (rc) return rc;
    if (reps < 1) reps = 3;
    hipStream_t st = (hipStream_t)stream;
    for (Op& op : e->ops) { op.planar = 0; op.fused = 0; op.fused_pw = 0; op.fused_bf = 0; }   // every op runs on its own while it is tuned
    rc = forward_single_lane(e, x, x_dtype, pred, st);
    if (rc) return rc;
    struct EventPair {      // destroyed on every return path
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } evp;
    LP_HIP_CHECK(hipEventCreate(&evp.a));
    LP_HIP_CHECK(hipEventCreate(&evp.b));
    const hipEvent_t e0 = evp.a, e1 = evp.b;
    for (size_t i = 0; i < e->ops.size(); ++i) {
        Op& op = e->ops[i];
        if (e->launches[i].is_conv && op.kind == OP_HEAD_CLS && rows_fits(e, op) && !getenv("LP_NO_HEAD_ROWS")) {
            // class predictors: tiled generic kernel or the row writer
            float best = -1.f;
            int best_rows = 0;
            for (int rows = 0; rows <= 1; ++rows) {
                op.rows = rows;
                if (prepare_op(e, i) != LP_OK || run_op(e, i, x, x_dtype, pred, st) != LP_OK) continue;
                float ms_min = -1.f;
                for (int round = 0; round < 3; ++round) {
                    LP_HIP_CHECK(hipEventRecord(e0, st));
                    for (int r = 0; r < reps; ++r) run_op(e, i, x, x_dtype, pred, st);
                    LP_HIP_CHECK(hipEventRecord(e1, st));
                    LP_HIP_CHECK(hipEventSynchronize(e1));
                    float ms = 0.f;
                    LP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms_min < 0.f || ms < ms_min) ms_min = ms;
                }
                if (best < 0.f || ms_min < best) { best = ms_min; best_rows = rows; }
            }
            op.rows = best_rows;
            rc = prepare_op(e, i);
            if (rc) return rc;
            continue;
        }
        if (!e->launches[i].is_conv || op.mode != MODE_ACT) continue;
        const int cb = conv_shape(e->dtype, op.cfg, 1, 1).CB;
        int best_cfg = op.cfg, best_nb = op.nbuf, best_tile = op.tile, best_wc = 0, best_rd = 2, best_pipe = 0;
        float best_ms = -1.f;
        int trc = LP_OK;
        auto time_current = [&]() -> float {   // best of three rounds of `reps` launches of the op as prepared; < 0: failed
            if (run_op(e, i, x, x_dtype, pred, st) != LP_OK) return -1.f;   // warm
            float ms_min = -1.f;
            for (int round = 0; round < 3; ++round) {
                if (hipEventRecord(e0, st) != hipSuccess) { trc = LP_ERR_HIP; return -1.f; }
                for (int r = 0; r < reps; ++r) run_op(e, i, x, x_dtype, pred, st);
                float ms = 0.f;
                if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                    hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { trc = LP_ERR_HIP; return -1.f; }
                if (ms_min < 0.f || ms < ms_min) ms_min = ms;
            }
            return ms_min;
        };
        op.stream_wc = 0;
        op.pipe = 0;
        const bool fam16 = op_fam16(e, op);       // only the 16x16x32 variants are candidates then (another fp32 summation order)
        for (int cfg = 0; cfg < CFG_COUNT && !fam16; ++cfg) {
            if (conv_shape(e->dtype, cfg, 1, 1).CB != cb) continue;
            for (int nb = 1; nb <= (cfg == CFG_C ? 1 : 2); ++nb) {
                int last_th = -1, last_tw = -1;
                for (int tile = 0; tile < 3; ++tile) {
                    op.cfg = cfg;
                    op.nbuf = nb;
                    op.tile = tile;
                    if (prepare_op(e, i) != LP_OK) continue;
                    if (e->launches[i].a.TH == last_th && e->launches[i].a.TW == last_tw) break;   // no further candidates
                    last_th = e->launches[i].a.TH;
                    last_tw = e->launches[i].a.TW;
                    const float ms = time_current();
                    if (ms >= 0.f && (best_ms < 0.f || ms < best_ms)) { best_ms = ms; best_cfg = cfg; best_nb = nb; best_tile = tile; }
                }
            }
        }
        // 1x1 stride-1 layers: the streaming kernel reads the same packing
        if (!getenv("LP_NO_STREAM") && !fam16) {
            op.cfg = best_cfg; op.nbuf = best_nb; op.tile = best_tile;
            for (int wc = 2; wc <= 4; wc += 2) {
                if (!stream_fits(e, op, wc)) continue;
                for (int rd = 2; rd <= 2; ++rd) {
                    op.stream_wc = wc;
                    op.stream_rd = rd;
                    if (prepare_op(e, i) != LP_OK) continue;
                    const float ms = time_current();
                    if (ms >= 0.f && (best_ms < 0.f || ms < best_ms)) { best_ms = ms; best_wc = wc; best_rd = rd; }
                }
            }
        }
        // 3x3 stride-1 layers: the pipelined kernel reads the same packing
        int pipe_tile = 0;
        if ((!getenv("LP_NO_PIPE") || fam16) && op.kind == OP_CONV) {
            op.cfg = best_cfg; op.nbuf = best_nb; op.stream_wc = 0;
            for (int pc = fam16 ? PIPE16_D : 0; pc < (fam16 ? PIPE_S2_END : PIPE_COUNT); ++pc) {
                if (fam16 && !pipe_is_16(pc)) continue;
                if (!conv_pipe_fits(e->dtype, pc, cb, op.ksize, op.stride, op.mode, op.nct, op.nphase, op.nchunks)) continue;
                int last_th = -1, last_tw = -1;
                for (int tile = 0; tile < 3; ++tile) {
                    op.pipe = pc + 1;
                    op.tile = tile;
                    if (prepare_op(e, i) != LP_OK) continue;
                    if (e->launches[i].a.TH == last_th && e->launches[i].a.TW == last_tw) break;
                    last_th = e->launches[i].a.TH;
                    last_tw = e->launches[i].a.TW;
                    const float ms = time_current();
                    if (ms >= 0.f && (best_ms < 0.f || ms < best_ms)) { best_ms = ms; best_pipe = pc + 1; pipe_tile = tile; }
                }
            }
            op.pipe = 0;
            if (fam16 && !best_pipe) best_pipe = fam16_default_pipe(e, op);
        }
        if (trc) return fail(trc, "autotune: event timing failed");
        op.cfg = best_cfg;
        op.nbuf = best_nb;
        op.tile = best_pipe ? pipe_tile : best_tile;
        op.stream_wc = best_pipe ? 0 : best_wc;
        op.stream_rd = best_rd;
        op.pipe = best_pipe;
        op.planar = 0;
        op.fused = 0;
        op.fused_pw = 0;
        rc = prepare_op(e, i);
        if (rc) return rc;
        // the stem may read the caller's frame itself and make the input op unnecessary: worth it if it beats the two together
        if (i == 1 && stem_planar_possible(e) && frame_direct(e, x, x_dtype) && !getenv("LP_NO_PLANAR")) {
            float t_in = -1.f;
            for (int round = 0; round < 3 && trc == LP_OK; ++round) {
                float ms = 0.f;
                if (hipEventRecord(e0, st) != hipSuccess) { trc = LP_ERR_HIP; break; }
                for (int r = 0; r < reps; ++r) run_op(e, 0, x, x_dtype, pred, st);
                if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { trc = LP_ERR_HIP; break; }
                if (t_in < 0.f || ms < t_in) t_in = ms;
            }
            int best_planar = 0, last_th = -1, last_tw = -1;
            float best_pl = -1.f;
            for (int tile = 0; tile < 3 && trc == LP_OK && t_in >= 0.f; ++tile) {
                op.planar = tile + 1;
                if (prepare_op(e, i) != LP_OK) continue;
                if (e->stem_planar.a.TH == last_th && e->stem_planar.a.TW == last_tw) break;
                last_th = e->stem_planar.a.TH;
                last_tw = e->stem_planar.a.TW;
                const float ms = time_current();
                if (ms >= 0.f && (best_pl < 0.f || ms < best_pl)) { best_pl = ms; best_planar = tile + 1; }
            }
            if (trc) return fail(trc, "autotune: event timing failed");
            op.planar = (best_planar && best_pl < best_ms + t_in) ? best_planar : 0;
            rc = prepare_op(e, i);
            if (rc) return rc;
        }
        // ... or run as one kernel with the layer behind it: against the three ops as tuned so far
        if (i == 2 && stem2_fused_possible(e) && frame_direct(e, x, x_dtype) && !getenv("LP_NO_PLANAR") && !getenv("LP_NO_FUSED_STEM")) {
            auto time_ops = [&](size_t first, size_t last) -> float {     // best of three rounds of `reps` x (ops first..last)
                float ms_min = -1.f;
                for (int round = 0; round < 3 && trc == LP_OK; ++round) {
                    float ms = 0.f;
                    if (hipEventRecord(e0, st) != hipSuccess) { trc = LP_ERR_HIP; break; }
                    for (int r = 0; r < reps; ++r)
                        for (size_t k = first; k <= last; ++k) run_op(e, k, x, x_dtype, pred, st);
                    if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { trc = LP_ERR_HIP; break; }
                    if (ms_min < 0.f || ms < ms_min) ms_min = ms;
                }
                return ms_min;
            };
            op.fused = 0;
            const float t_sep = time_ops(0, 2);
            int best_f = 0, last_th = -1, last_tw = -1;
            float best_fms = -1.f;
            for (int tile = 0; tile < 3 && trc == LP_OK && t_sep >= 0.f; ++tile) {
                op.fused = tile + 1;
                if (prepare_op(e, i) != LP_OK) continue;
                if (e->stem2_fused.a.TH == last_th && e->stem2_fused.a.TW == last_tw) break;
                last_th = e->stem2_fused.a.TH;
                last_tw = e->stem2_fused.a.TW;
                if (run_op(e, 2, x, x_dtype, pred, st) != LP_OK) continue;
                const float ms = time_ops(2, 2);
                if (ms >= 0.f && (best_fms < 0.f || ms < best_fms)) { best_fms = ms; best_f = tile + 1; }
            }
            if (trc) return fail(trc, "autotune: event timing failed");
            op.fused = (best_f && best_fms < t_sep) ? best_f : 0;
            rc = prepare_op(e, i);
            if (rc) return rc;
        }
        // a 1x1 layer + this 3x3 stride-2 layer as one kernel: against the two ops as tuned
        if (pw_fused_possible(e, i) && !getenv("LP_NO_FUSED_PW")) {
            auto time_pair = [&](size_t first) -> float {
                float ms_min = -1.f;
                for (int round = 0; round < 3 && trc == LP_OK; ++round) {
                    float ms = 0.f;
                    if (hipEventRecord(e0, st) != hipSuccess) { trc = LP_ERR_HIP; break; }
                    for (int r = 0; r < reps; ++r)
                        for (size_t k = first; k <= i; ++k) run_op(e, k, x, x_dtype, pred, st);
                    if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { trc = LP_ERR_HIP; break; }
                    if (ms_min < 0.f || ms < ms_min) ms_min = ms;
                }
                return ms_min;
            };
            op.fused_pw = 0;
            const float t_sep = time_pair(i - 1);
            int best_f = 0, last_th = -1, last_tw = -1;
            float best_fms = -1.f;
            for (int tile = 0; tile < 3 && trc == LP_OK && t_sep >= 0.f; ++tile) {
                op.fused_pw = tile + 1;
                if (prepare_op(e, i) != LP_OK) continue;
                const ConvArgs& fa = e->pw_fused[(int)i].a;
                if (fa.TH == last_th && fa.TW == last_tw) break;
                last_th = fa.TH;
                last_tw = fa.TW;
                if (run_op(e, i, x, x_dtype, pred, st) != LP_OK) continue;
                const float ms = time_pair(i);
                if (ms >= 0.f && (best_fms < 0.f || ms < best_fms)) { best_fms = ms; best_f = tile + 1; }
            }
            if (trc) return fail(trc, "autotune: event timing failed");
            op.fused_pw = (best_f && best_fms < t_sep) ? best_f : 0;
            rc = prepare_op(e, i);
            if (rc) return rc;
        }
        // BiFusion's transposed conv + cv1 + this cv3 as one kernel: against the three ops as tuned
        if (op.bf_up >= 0 && !getenv("LP_NO_FUSED_BF")) {
            auto time_set = [&](bool fused_form) -> float {
                float ms_min = -1.f;
                for (int round = 0; round < 3 && trc == LP_OK; ++round) {
                    float ms = 0.f;
                    if (hipEventRecord(e0, st) != hipSuccess) { trc = LP_ERR_HIP; break; }
                    for (int r = 0; r < reps; ++r) {
                        if (!fused_form) { run_op(e, (size_t)op.bf_up, x, x_dtype, pred, st); run_op(e, (size_t)op.bf_cv1, x, x_dtype, pred, st); }
                        run_op(e, i, x, x_dtype, pred, st);
                    }
                    if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { trc = LP_ERR_HIP; break; }
                    if (ms_min < 0.f || ms < ms_min) ms_min = ms;
                }
                return ms_min;
            };
            op.fused_bf = 0;
            const float t_sep = time_set(false);
            op.fused_bf = 1;
            float t_fused = -1.f;
            if (prepare_op(e, i) == LP_OK && run_op(e, i, x, x_dtype, pred, st) == LP_OK) t_fused = time_set(true);
            if (trc) return fail(trc, "autotune: event timing failed");
            op.fused_bf = (t_sep >= 0.f && t_fused >= 0.f && t_fused < t_sep) ? 1 : 0;
            rc = prepare_op(e, i);
            if (rc) return rc;
        }
    }
    std::vector<std::vector<int>> choice;
    for (const Op& op : e->ops) choice.push_back({op.cfg, op.nbuf, op.tile, op.stream_wc, op.stream_rd, op.rows, op.pipe, op.planar, op.fused, op.fused_pw, op.fused_bf});
    e->tuned[{e->B, e->H, e->W}] = choice;
    return LP_OK;
}

extern "C" int lp_engine_copy_tuning(lp_engine* dst, const lp_engine* src) {
    if (!dst || !src || !dst->finalized || !src->finalized) return fail(LP_ERR_STATE, "lp_engine_copy_tuning: finalize both engines first");
    if (dst->dtype != src->dtype || dst->ops.size() != src->ops.size()) return fail(LP_ERR_ARG, "lp_engine_copy_tuning: engines differ");
    for (size_t i = 0; i < src->ops.size(); ++i) {
        const Op &a = src->ops[i], &b = dst->ops[i];
        if (a.kind != b.kind || a.ksize != b.ksize || a.stride != b.stride || a.cout != b.cout || a.nchunks != b.nchunks || a.nct != b.nct)
            return fail(LP_ERR_ARG, "lp_engine_copy_tuning: engines were built from different graphs");
    }
    dst->tuned = src->tuned;      // applied by lp_engine_bind for the shapes it contains
    if (dst->arena) {
        auto it = dst->tuned.find({dst->B, dst->H, dst->W});
        if (it != dst->tuned.end())
            for (size_t i = 0; i < dst->ops.size(); ++i) {
                Op& op = dst->ops[i];
                op.cfg = it->second[i][0]; op.nbuf = it->second[i][1]; op.tile = it->second[i][2];
                op.stream_wc = it->second[i][3]; op.stream_rd = it->second[i][4]; op.rows = it->second[i][5]; op.pipe = it->second[i][6]; op.planar = it->second[i][7]; op.fused = it->second[i][8]; op.fused_pw = it->second[i][9]; op.fused_bf = it->second[i].size() > 10 ? it->second[i][10] : 0;
                int rc = prepare_op(dst, i);
                if (rc) return rc;
            }
    }
    return LP_OK;
}

extern "C" int lp_engine_set_op_variant(lp_engine* e, int op_idx, int cfg, int nbuf) {
    if (!e || !e->finalized || op_idx < 0 || op_idx >= (int)e->ops.size()) return fail(LP_ERR_ARG, "lp_engine_set_op_variant: op index");
    Op& op = e->ops[op_idx];
    if (op.kind == OP_HEAD_CLS) {        // LP_VARIANT_ROWS: row writer; the op's packing tile (2 = C) selects the tiled kernel again
        if (cfg == LP_VARIANT_ROWS && !rows_fits(e, op)) return fail(LP_ERR_UNSUPPORTED, "lp_engine_set_op_variant: the row-writer kernel does not fit this op");
        if (cfg != LP_VARIANT_ROWS && cfg != op.cfg) return fail(LP_ERR_UNSUPPORTED, "lp_engine_set_op_variant: head_cls takes LP_VARIANT_ROWS or its packing tile");
        op.rows = cfg == LP_VARIANT_ROWS;
        e->tuned.erase({e->B, e->H, e->W});
        if (e->arena && op_idx < (int)e->launches.size()) return prepare_op(e, (size_t)op_idx);
        return LP_OK;
    }
    if (op.kind == OP_HEAD_BOX && (cfg == LP_VARIANT_BOX_SPARSE || cfg == LP_VARIANT_BOX_DENSE)) {
        // detections-only forward: boxes of the level's candidates only / of every anchor (same rows for the candidates; no tuning state involved)
        op.box_sparse = cfg == LP_VARIANT_BOX_SPARSE;
        ++e->epoch;      // (captured graphs hold the launch)
        return LP_OK;
    }
    if (op.kind == OP_INPUT || op.kind == OP_POOL || op.mode != MODE_ACT) return fail(LP_ERR_UNSUPPORTED, "lp_engine_set_op_variant: op has no variants");
    const int ks = op.kind == OP_CONV ? op.ksize : 1, stv = op.kind == OP_CONV ? op.stride : 1;
    const int cb = conv_shape(e->dtype, op.cfg, ks, stv).CB;
    if (cfg == LP_VARIANT_FUSED_BIFUSION) {       // BiFusion: transposed conv + cv1 + this cv3 as one kernel
        if (op.bf_up < 0 || nbuf != 3) return fail(LP_ERR_UNSUPPORTED, "lp_engine_set_op_variant: not the cv3 of a 64-channel BiFusion level");
        op.fused_bf = 1;
        e->tuned.erase({e->B, e->H, e->W});
        if (e->arena && op_idx < (int)e->launches.size()) return prepare_op(e, (size_t)op_idx);
        return LP_OK;
    }
    op.fused_bf = 0;
    if (cfg == LP_VARIANT_FUSED_PW_S2) {          // the 1x1 layer before this 3x3 stride-2 layer + this layer as one kernel
        if (!pw_fused_possible(e, (size_t)op_idx) || nbuf != 3)
            return fail(LP_ERR_UNSUPPORTED, "lp_engine_set_op_variant: no 1x1 layer of at most 64 channels feeds this 3x3 stride-2 layer alone");
        op.fused_pw = 1;
        e->tuned.erase({e->B, e->H, e->W});
        if (e->arena && op_idx < (int)e->launches.size()) return prepare_op(e, (size_t)op_idx);
        return LP_OK;
    }
    op.fused_pw = 0;
    if (cfg == LP_VARIANT_FUSED_STEM2) {          // stem + this layer as one kernel: on top of whatever variants run for other frame dtypes
        if (op_idx != 2 || !stem2_fused_possible(e) || nbuf != 3)
            return fail(LP_ERR_UNSUPPORTED, "lp_engine_set_op_variant: only the layer behind the stem (3x3 stride 2, <= 64 channels) has the fused form");
        op.fused = 1;
        e->tuned.erase({e->B, e->H, e->W});
        if (e->arena && op_idx < (int)e->launches.size()) return prepare_op(e, (size_t)op_idx);
        return LP_OK;
    }
    op.fused = 0;
    if (cfg == LP_VARIANT_PIPE_D + PIPE_P) {      // the stem reading the caller's frame: on top of whatever variant runs for other frame dtypes
        if (op_idx != 1 || !stem_planar_possible(e) || nbuf != 3)
            return fail(LP_ERR_UNSUPPORTED, "lp_engine_set_op_variant: only the stem over the space-to-depth image has the planar form");
        op.planar = 1;
        e->tuned.erase({e->B, e->H, e->W});
        if (e->arena && op_idx < (int)e->launches.size()) return prepare_op(e, (size_t)op_idx);
        return LP_OK;
    }
    op.planar = 0;
    if ((cfg >= LP_VARIANT_PIPE_D && cfg < LP_VARIANT_PIPE_D + PIPE_COUNT) || pipe_is_16(cfg - LP_VARIANT_PIPE_D)) {
        const int pc = cfg - LP_VARIANT_PIPE_D;
        if (!conv_pipe_fits(e->dtype, pc, cb, ks, stv, op.mode, op.nct, op.nphase, op.nchunks) || op.kind != OP_CONV || nbuf != 3)
            return fail(LP_ERR_UNSUPPORTED, "lp_engine_set_op_variant: the pipelined 3x3 kernel does not fit this op");
        op.stream_wc = 0;
        op.pipe = pc + 1;
    } else if (cfg == LP_VARIANT_STREAM64 || cfg == LP_VARIANT_STREAM128) {
        const int wc = cfg == LP_VARIANT_STREAM64 ? 2 : 4;
        if (!stream_fits(e, op, wc) || nbuf != 2)
            return fail(LP_ERR_UNSUPPORTED, "lp_engine_set_op_variant: the streaming 1x1 kernel does not fit this op");
        op.stream_wc = wc;
        op.stream_rd = nbuf;
        op.pipe = 0;
    } else {
        if (cfg < 0 || cfg >= CFG_COUNT || conv_shape(e->dtype, cfg, ks, stv).CB != cb || nbuf < 1 || nbuf > (cfg == CFG_C ? 1 : 2))
            return fail(LP_ERR_UNSUPPORTED, "lp_engine_set_op_variant: variant does not share the op's weight packing");
        op.stream_wc = 0;
        op.pipe = 0;
        op.cfg = cfg;
        op.nbuf = nbuf;
    }
    e->tuned.erase({e->B, e->H, e->W});
    if (e->arena && op_idx < (int)e->launches.size()) return prepare_op(e, (size_t)op_idx);
    return LP_OK;
}

extern "C" int lp_engine_op_carrier(const lp_engine* e, int op, int frame_direct_) {
    if (!e || op < 0 || op >= (int)e->ops.size()) return -1;
    const Op& o = e->ops[(size_t)op];
    if (o.bf_carrier >= 0 && e->ops[(size_t)o.bf_carrier].fused_bf) return o.bf_carrier;
    if ((size_t)op + 1 < e->ops.size() && e->ops[(size_t)op + 1].fused_pw) return op + 1;
    if (frame_direct_ && e->ops.size() > 2 && e->ops[0].kind == OP_INPUT && e->ops[0].s2d) {
        if (e->ops[2].fused && op <= 1) return 2;
        if (e->ops[1].planar && op == 0) return 1;
    }
    return -1;
}

extern "C" int lp_engine_op_variant(const lp_engine* e, int op, int* cfg, int* nbuf) {
    if (!e || op < 0 || op >= (int)e->ops.size()) return fail(LP_ERR_ARG, "lp_engine_op_variant: op index");
    const bool stream = e->ops[op].stream_wc != 0;
    if (e->ops[op].fused_bf) {
        if (cfg) *cfg = LP_VARIANT_FUSED_BIFUSION;
        if (nbuf) *nbuf = 3;
        return LP_OK;
    }
    if (e->ops[op].fused_pw) {
        if (cfg) *cfg = LP_VARIANT_FUSED_PW_S2;
        if (nbuf) *nbuf = 3;
        return LP_OK;
    }
    if (e->ops[op].fused) {
        if (cfg) *cfg = LP_VARIANT_FUSED_STEM2;
        if (nbuf) *nbuf = 3;
        return LP_OK;
    }
    if (e->ops[op].planar) {
        if (cfg) *cfg = LP_VARIANT_PIPE_D + PIPE_P;
        if (nbuf) *nbuf = 3;
        return LP_OK;
    }
    if (e->ops[op].pipe) {
        if (cfg) *cfg = LP_VARIANT_PIPE_D + e->ops[op].pipe - 1;
        if (nbuf) *nbuf = 3;
        return LP_OK;
    }
    if (cfg) *cfg = e->ops[op].rows ? LP_VARIANT_ROWS : stream ? (e->ops[op].stream_wc == 2 ? LP_VARIANT_STREAM64 : LP_VARIANT_STREAM128) : e->ops[op].cfg;
    if (nbuf) *nbuf = stream ? e->ops[op].stream_rd : e->ops[op].nbuf;
    return LP_OK;
}
