// Host-side geometry of the implicit-GEMM conv kernels and the per-dtype launch dispatch.
// The kernels themselves are in lp_conv_kernel.inc (one translation unit per activation dtype).
#include "lp_internal.h"

#include <cstdlib>

namespace lp {

int conv_launch_f16(int cfg, int mode, int ksize, int stride, int nbuf, const ConvArgs& a, hipStream_t st);
int conv_launch_bf16(int cfg, int mode, int ksize, int stride, int nbuf, const ConvArgs& a, hipStream_t st);
int conv_launch_f32(int cfg, int mode, int ksize, int stride, int nbuf, const ConvArgs& a, hipStream_t st);

int conv_stream_launch_f16(int wc, const ConvArgs& a, int cb_pack, int lds, hipStream_t st);
int conv_stream_launch_bf16(int wc, const ConvArgs& a, int cb_pack, int lds, hipStream_t st);
int conv_stream_launch_f32(int wc, const ConvArgs& a, int cb_pack, int lds, hipStream_t st);

int conv_pipe_launch_f16(int pcfg, const ConvArgs& a, int ncu, hipStream_t st);
int conv_pipe_launch_bf16(int pcfg, const ConvArgs& a, int ncu, hipStream_t st);

int head_rows_launch_f16(const ConvArgs& a, int cb_pack, hipStream_t st);
int head_rows_launch_bf16(const ConvArgs& a, int cb_pack, hipStream_t st);
int head_rows_launch_f32(const ConvArgs& a, int cb_pack, hipStream_t st);
int head_box_det_launch_f16(const ConvArgs& a, int cb_pack, hipStream_t st);
int head_box_det_launch_bf16(const ConvArgs& a, int cb_pack, hipStream_t st);
int head_box_det_launch_f32(const ConvArgs& a, int cb_pack, hipStream_t st);

ConvShape conv_shape(int dtype, int cfg, int ksize, int stride) {
    const int sz = (int)dtype_size(dtype);
    ConvShape s;
    int wc = 2, wp = 2, wgc = 1, wgp = 4;
    switch (cfg) {
        case CFG_A: wgc = 2; wgp = 2; break;
        case CFG_B: break;
        case CFG_C: wc = 1; break;
        case CFG_D: wgc = 2; wgp = 4; break;
        case CFG_E: wc = 1; wgc = 2; wgp = 2; break;
        case CFG_F: wc = 1; wgc = 4; wgp = 2; break;
    }
    s.WP = wp;
    s.WGC = wgc;
    s.WGP = wgp;
    s.CB = 32 * wc * wgc;
    s.PB = 32 * wp * wgp;
    s.NT = 64 * wgc * wgp;
    s.KC = ksize == 1 ? 128 / sz : 32 / sz;
    s.HPMAX = ksize == 1 ? s.PB : (stride == 1 ? 2 * s.PB + 128 : 5 * s.PB + 128);
    return s;
}

static inline int swz_host(int row, int g, int gpr) { return gpr == 2 ? (g ^ ((row >> 3) & 1)) : (g ^ ((row >> 1) & 7)); }

int conv_pick_pitch(const ConvShape& s, int dtype, int ksize, int stride, int TH, int TW) {
    const int hw = (TW - 1) * stride + ksize, hh = (TH - 1) * stride + ksize;
    if (ksize == 1) return hw;   // 128-B rows of consecutive pixels: the row swizzle alone is conflict-free
    (void)dtype;
    const int rowb = 32, gpr = 2, npix = TH * TW;
    // the four 16-lane groups one ds_read_b128 is serviced in (lanes >= 32 mirror them)
    static const int G0[16] = {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27};
    static const int G1[16] = {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31};
    long best_cost = -1;
    int best = hw;
    for (int pitch = hw; pitch < hw + 16; ++pitch) {
        if (hh * pitch > s.HPMAX) break;
        long cost = 0;
        for (int blk = 0; blk * 32 < s.PB; ++blk)
            for (int t = 0; t < ksize * ksize; ++t)
                for (int half = 0; half < 2; ++half)
                    for (int g = 0; g < 2; ++g) {
                        const int* lanes = g ? G1 : G0;
                        int addr[16], n = 0, slot_cnt[16] = {0};
                        for (int k = 0; k < 16; ++k) {
                            int pl = blk * 32 + lanes[k];
                            if (pl >= npix) pl = 0;
                            const int ty = pl / TW, tx = pl - ty * TW;
                            const int kx = t % ksize, hwe = (hw + 1) >> 1;
                            // stride 2: rows are stored evens-first (lp_conv_kernel.inc), the walk over tx is dense
                            const int col = stride == 2 ? (kx & 1) * hwe + tx + (kx >> 1) : tx * stride + kx;
                            const int hp = (ty * stride + t / ksize) * pitch + col;
                            const int a = hp * rowb + swz_host(hp, half, gpr) * 16;
                            bool dup = false;
                            for (int m = 0; m < n; ++m) dup |= addr[m] == a;   // identical addresses broadcast
                            if (!dup) { addr[n++] = a; ++slot_cnt[(a >> 4) & 15]; }
                        }
                        int mx = 1;
                        for (int k = 0; k < 16; ++k) mx = slot_cnt[k] > mx ? slot_cnt[k] : mx;
                        cost += mx;
                    }
        cost = cost * 64 + (pitch - hw);   // fewest LDS cycles first, then the least padding
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = pitch; }
    }
    return best;
}

// The same for the 16x16x32 operand map of conv3x3_pipe16_kernel (3x3 stride 1): lane l = (column i = l & 15, K group g = l >> 4)
// reads pixel quad_order(i) of a 16-pixel block at tap ta (g < 2) or tb (g >= 2), granule g & 1.
int conv_pick_pitch16(const ConvShape& s, int TH, int TW) {
    const int hw = TW + 2, hh = TH + 2, rowb = 32, npix = TH * TW;
    static const int GRP[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                   {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    auto qo = [](int i) { return ((i >> 2) == 2 ? 12 : (i >> 2) == 3 ? 8 : (i & 12)) + (i & 3); };
    long best_cost = -1;
    int best = hw;
    for (int pitch = hw; pitch < hw + 16; ++pitch) {
        if (hh * pitch > s.HPMAX) break;
        long cost = 0;
        for (int blk = 0; blk * 16 < s.PB; ++blk)
            for (int st = 0; st < 9; ++st) {
                const int ta = st < 4 ? 2 * st : (st == 4 ? 8 : 2 * st - 9), tb = st == 4 ? 0 : ta + 1;
                for (int gq = 0; gq < 4; ++gq) {
                    int addr[16], n = 0, slot_cnt[16] = {0};
                    for (int k = 0; k < 16; ++k) {
                        const int l = GRP[gq][k], i = l & 15, g = l >> 4;
                        int pl = blk * 16 + qo(i);
                        if (pl >= npix) pl = 0;
                        const int ty = pl / TW, tx = pl - ty * TW, t = g < 2 ? ta : tb;
                        const int hp = (ty + t / 3) * pitch + tx + t % 3;
                        // (step 4: groups 2 / 3 read another ring slot -- a multiple of 256 bytes away, the same banks)
                        const int a = hp * rowb + swz_host(hp, g & 1, 2) * 16 + (g >> 1) * (1 << 20);
                        bool dup = false;
                        for (int m = 0; m < n; ++m) dup |= addr[m] == a;
                        if (!dup) { addr[n++] = a; ++slot_cnt[(a >> 4) & 15]; }
                    }
                    int mx = 1;
                    for (int k = 0; k < 16; ++k) mx = slot_cnt[k] > mx ? slot_cnt[k] : mx;
                    cost += mx;
                }
            }
        cost = cost * 64 + (pitch - hw);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = pitch; }
    }
    return best;
}

void conv_pick_tile16v(const ConvShape& s, int Ho, int Wo, int B, int nct, int choice, int* TH, int* TW, int stride) {
    struct Cand { double cost; int th, tw; };
    Cand best[4];
    int n = 0;
    const int ncu = device_cus();
    for (int th = 1; th <= Ho; ++th)
        for (int tw = 1; tw <= Wo; ++tw) {
            const int hh = (th - 1) * stride + 3, hw = (tw - 1) * stride + 3;     // halo of the tile (stride 2: lp_conv3x3_s2p16.inc)
            if (th * tw > s.PB || hh * hw > s.HPMAX) continue;
            const long tiles = (long)B * ceil_div(Ho, th) * ceil_div(Wo, tw) * nct;
            const int nb = ceil_div(th * tw, 16), per_wave = ceil_div(nb, s.WGP);
            // time ~ (tiles per workgroup) x (K-step length: the busiest wave's blocks + a fixed part) + a little for the halo
            const double cost = (double)ceil_div((int)tiles, ncu) * (per_wave + 0.75) + 1e-4 * hh * hw + 1e-7 * tiles;
            int pos = n < 4 ? n : 4;
            for (int k = 0; k < (n < 4 ? n : 4); ++k)
                if (cost < best[k].cost) { pos = k; break; }
            if (pos >= 4) continue;
            for (int k = (n < 4 ? n : 3); k > pos; --k) best[k] = best[k - 1];
            best[pos] = {cost, th, tw};
            if (n < 4) ++n;
        }
    if (n == 0) { *TH = 1; *TW = 1; return; }
    if (choice < 0) choice = 0;
    if (choice >= n) choice = n - 1;
    *TH = best[choice].th;
    *TW = best[choice].tw;
}

// Candidate output tiles (TH x TW <= PB pixels, halo <= HPMAX), best first: fewest workgroups, then the smallest
// halo.  `choice` selects the n-th distinct candidate (clamped) -- the autotuner times the first few.
void conv_pick_tile(const ConvShape& s, int ksize, int stride, int Ho, int Wo, int choice, int* TH, int* TW) {
    struct Cand { long cost; int th, tw; };
    Cand best[4];
    int n = 0;
    for (int th = 1; th <= s.PB && th <= Ho; ++th) {
        int tw = s.PB / th;
        if (tw > Wo) tw = Wo;
        if (tw < 1) continue;
        const int hh = (th - 1) * stride + ksize, hw = (tw - 1) * stride + ksize;
        if (hh * hw > s.HPMAX) continue;
        const long tiles = (long)ceil_div(Ho, th) * ceil_div(Wo, tw);
        Cand c = {tiles * 100000 + (long)hh * hw, th, tw};
        // insertion into the sorted top-4 (distinct tile counts give distinct costs; equal costs keep the first)
        int pos = n < 4 ? n : 4;
        for (int k = 0; k < (n < 4 ? n : 4); ++k)
            if (c.cost < best[k].cost) { pos = k; break; }
        if (pos >= 4) continue;
        for (int k = (n < 4 ? n : 3); k > pos; --k) best[k] = best[k - 1];
        best[pos] = c;
        if (n < 4) ++n;
    }
    if (n == 0) { *TH = 1; *TW = 1; return; }
    if (choice < 0) choice = 0;
    if (choice >= n) choice = n - 1;
    *TH = best[choice].th;
    *TW = best[choice].tw;
}

int conv_launch(int dtype, int cfg, int mode, int ksize, int stride, int nbuf, const ConvArgs& a, hipStream_t st) {
    // host-side shape checks: the kernel indexes LDS and global memory from these without further tests
    const ConvShape s = conv_shape(dtype, cfg, ksize, stride);
    const int hh = (a.TH - 1) * stride + ksize, hw = (a.TW - 1) * stride + ksize;
    if (a.TH < 1 || a.TW < 1 || a.TH * a.TW > s.PB || a.hpitch < hw || hh * a.hpitch > s.HPMAX)
        return fail(LP_ERR_ARG, "conv: tile does not fit the kernel configuration");
    if (a.tiles_x * a.TW < a.Wo || a.tiles_y * a.TH < a.Ho) return fail(LP_ERR_ARG, "conv: tiles do not cover the output");
    if (a.nsrc < 1 || a.nsrc > LP_MAX_SRC || a.nct < 1 || a.nphase < 1) return fail(LP_ERR_ARG, "conv: bad counts");
    if (mode == MODE_DECODE && (a.nct != 1 || 4 * a.reg_bins + 8 > s.CB)) return fail(LP_ERR_ARG, "decode: cout tile");
    switch (dtype) {
        case LP_F16: return conv_launch_f16(cfg, mode, ksize, stride, nbuf, a, st);
        case LP_BF16: return conv_launch_bf16(cfg, mode, ksize, stride, nbuf, a, st);
        case LP_F32: return conv_launch_f32(cfg, mode, ksize, stride, nbuf, a, st);
    }
    return fail(LP_ERR_ARG, "conv: dtype");
}

int conv_stream_lds(int dtype, int wc, int nchunks, int cb_pack) {
    if ((wc != 2 && wc != 4) || cb_pack % (32 * wc) != 0 || nchunks < 1 || nchunks > 8) return -1;   // 8 = STREAM_MAX_NCH
    const int sz = (int)dtype_size(dtype), cbt = 32 * wc, pxt = 32 * (4 / wc);
    const long bytes = (long)nchunks * cbt * 128 + cbt * 4 + 4L * pxt * (cbt * sz + 16);   // weights, bias, staging
    return bytes <= 160 * 1024 ? (int)bytes : -1;
}

int conv_stream_launch(int dtype, int wc, const ConvArgs& a0, int cb_pack, hipStream_t st) {
    // the streaming kernel treats the tensors as [B*H*W][cs] matrices: 1x1, stride 1, dense NHWC output, one phase
    const int nchunks = a0.chunk_begin[a0.nsrc];
    const int lds = conv_stream_lds(dtype, wc, nchunks, cb_pack);
    if (lds < 0) return fail(LP_ERR_ARG, "conv1x1 stream: layer does not fit");
    if (!a0.trash) return fail(LP_ERR_ARG, "conv1x1 stream: no scratch granule");
    if (a0.nsrc < 1 || a0.nsrc > LP_MAX_SRC || a0.nphase != 1 || a0.out_scale != 1 || a0.Ho != a0.H || a0.Wo != a0.W ||
        a0.out_img_stride != (long long)a0.Ho * a0.Wo * a0.out_pix_stride)
        return fail(LP_ERR_ARG, "conv1x1 stream: not a dense 1x1 stride-1 layer");
    if (a0.out2 && (a0.out_split % (32 * wc) != 0 || a0.res || a0.out2_img_stride != (long long)a0.Ho * a0.Wo * a0.out2_pix_stride))
        return fail(LP_ERR_ARG, "conv1x1 stream: the second destination must start at a cout tile of the wave");
    const int kc = 128 / (int)dtype_size(dtype);
    for (int i = 0; i < a0.nsrc; ++i)
        if (a0.src[i].cs % (kc / 4) != 0) return fail(LP_ERR_ARG, "conv1x1 stream: source channels are not whole 32-byte K-steps");
    ConvArgs a = a0;
    a.nct = ceil_div(a.out_c, 32 * wc);
    switch (dtype) {
        case LP_F16: return conv_stream_launch_f16(wc, a, cb_pack, lds, st);
        case LP_BF16: return conv_stream_launch_bf16(wc, a, cb_pack, lds, st);
        case LP_F32: return conv_stream_launch_f32(wc, a, cb_pack, lds, st);
    }
    return fail(LP_ERR_ARG, "conv1x1 stream: dtype");
}

// ---- pipelined 3x3 stride-1 kernel (lp_conv3x3_pipe.inc) ----
ConvShape conv_pipe_shape(int pcfg) {
    ConvShape s;
    s.KC = 16;
    s.NT = 512;
    s.WP = 2;
    switch (pcfg) {
        case PIPE_B: s.CB = 64; s.WGC = 1; s.WGP = 8; s.HPMAX = 704; break;
        case PIPE_F: case PIPE16_F: s.CB = 128; s.WGC = 4; s.WGP = 2; s.HPMAX = 384; break;
        case PIPE_C: s.CB = 32; s.WGC = 1; s.WGP = 8; s.HPMAX = 736; break;
        case PIPE16_V0: s.CB = 128; s.WGC = 2; s.WGP = 4; s.HPMAX = 512; s.PB = 4 * 7 * 16; return s;
        case PIPE16_V1: s.CB = 128; s.WGC = 4; s.WGP = 2; s.HPMAX = 512; s.PB = 2 * 7 * 16; return s;
        case PIPE16_S2A: s.CB = 128; s.WGC = 2; s.WGP = 4; s.HPMAX = 864; s.PB = 4 * 4 * 16; return s;
        case PIPE16_S2B: s.CB = 128; s.WGC = 4; s.WGP = 2; s.HPMAX = 864; s.PB = 2 * 7 * 16; return s;
        default: s.CB = 128; s.WGC = 2; s.WGP = 4; s.HPMAX = 384; break;
    }
    s.PB = 32 * s.WP * s.WGP;
    return s;
}

bool conv_pipe_fits(int dtype, int pcfg, int cb_pack, int ksize, int stride, int mode, int nct, int nphase, int nchunks) {
    if (pcfg == PIPE_P) return false;     // the planar stem is chosen by the engine (it replaces the input op as well), never as a variant of a layer
    if (pipe_is_16s2(pcfg))               // the stride-2 kernel: any number of K-chunks (five K-steps per chunk), LDS tables of 512 couts / 64 chunks
        return dtype != LP_F32 && ksize == 3 && stride == 2 && mode == MODE_ACT && nphase == 1 && cb_pack == 128 && nct * 128 <= 512 && nchunks >= 1 && nchunks <= 64;
    if (pipe_is_16(pcfg) && (nchunks < 4 || nchunks % 4 != 0)) return false;   // K-steps pair the taps over two chunks, the loop body is two pairs
    if (dtype == LP_F32 || pcfg < 0 || (pcfg >= PIPE_COUNT && !pipe_is_16(pcfg)) || ksize != 3 || stride != 1 || mode != MODE_ACT || nphase != 1) return false;
    const ConvShape s = conv_pipe_shape(pcfg);
    if (pipe_is_16v(pcfg) && (nct * s.CB > 512 || nchunks > 32)) return false;   // PIPE16V_MAXC / PIPE16V_MAXCHUNKS (its LDS tables)
    return s.CB == cb_pack && nct * s.CB <= 1024;   // 1024 = PIPE_MAXC (bias table in LDS)
}

static int g_ncu[16] = {0};
int device_cus() {   // CUs of the current device = persistent workgroups of the pipe kernel (one per CU)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (!g_ncu[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        g_ncu[dev] = n / 8 * 8;
        // LP_PIPE_MAXWG: cap of the persistent grid (experiments: with several batches in flight, kernels of different streams
        // that each take part of the CUs overlap their prologues / epilogues with each other's main loops)
        if (const char* cap = getenv("LP_PIPE_MAXWG")) { const int c = atoi(cap) / 8 * 8; if (c >= 8 && c < g_ncu[dev]) g_ncu[dev] = c; }
    }
    return g_ncu[dev];
}

// Output tile of the planar stem (lp_stem_planar.inc): TW a multiple of 4, TH * TW <= 512, the planar halo within a ring slot;
// fewest tiles first, then the widest rows (longest contiguous runs of the frame).  `choice` = the k-th best.  false: none.
bool stem_planar_tile(int Ho, int Wo, int choice, int* TH, int* TW) {
    struct Cand { long long tiles; int th, tw; };
    std::vector<Cand> cands;
    for (int tw = 4; tw <= 512 && tw < Wo + 4; tw += 4) {
        int th = 512 / tw;
        if (th > Ho) th = Ho;
        while (th >= 1 && ((th + 2) * 6 + 2) * (tw / 4 + 2) * 16 > 20 * 1024) --th;
        if (th < 1) continue;
        cands.push_back({(long long)ceil_div(Ho, th) * ceil_div(Wo, tw), th, tw});
    }
    std::sort(cands.begin(), cands.end(), [](const Cand& x, const Cand& y) { return x.tiles != y.tiles ? x.tiles < y.tiles : x.tw > y.tw; });
    if (choice < 0 || choice >= (int)cands.size()) return false;
    *TH = cands[choice].th;
    *TW = cands[choice].tw;
    return true;
}

// Output tile of the fused stem + ERBlock_2[0] kernel (lp_stem2_fused.inc): TW even, TH * TW <= 128, frame window and stem tile
// within their LDS buffers; fewest stem positions computed in all (tiles x positions per tile).  `choice` = the k-th best.
bool stem2_fused_tile(int Ho, int Wo, int choice, int* TH, int* TW, int* hpitch) {
    struct Cand { long long cost; int th, tw; };
    std::vector<Cand> cands;
    for (int tw = 2; tw <= 128 && tw < Wo + 2; tw += 2)
        for (int th = 1; th <= 128 / tw && th <= Ho; ++th) {
            const int hp4 = tw / 2 + 2, rin = 2 * th + 3, hps = 2 * tw + 1;
            const int ntb = (th * tw + 31) / 32;                      // x cout blocks (1 or 2): 2 or 4 stores per wave only up to 8 / 16 tiles
            if ((rin * 6 + 2) * hp4 * 16 > 21 * 1024 || (rin * 6 * hp4 + 63) / 64 > 24 || (2 * th + 1) * hps > 640 || ntb * 2 > 16) continue;
            cands.push_back({(long long)ceil_div(Ho, th) * ceil_div(Wo, tw) * (2 * th + 1) * hps, th, tw});
        }
    std::sort(cands.begin(), cands.end(), [](const Cand& x, const Cand& y) { return x.cost != y.cost ? x.cost < y.cost : x.tw > y.tw; });
    if (choice < 0 || choice >= (int)cands.size()) return false;
    *TH = cands[choice].th;
    *TW = cands[choice].tw;
    *hpitch = 2 * cands[choice].tw + 1;      // no padding slots: only the last MFMA tile of stage A has positions to zero
    return true;
}

int conv_pipe_launch(int dtype, int pcfg, const ConvArgs& a, hipStream_t st) {
    if (pcfg == PIPE_P || pcfg == PIPE_FUSED2 || pcfg == PIPE_FUSED_PW || pcfg == PIPE_FUSED_BF) {   // the fused / planar kernels check their own geometry
        switch (dtype) {
            case LP_F16: return conv_pipe_launch_f16(pcfg, a, device_cus(), st);
            case LP_BF16: return conv_pipe_launch_bf16(pcfg, a, device_cus(), st);
        }
        return fail(LP_ERR_UNSUPPORTED, "planar stem: 16-bit frames only");
    }
    const ConvShape s = conv_pipe_shape(pcfg);
    if (pipe_is_16s2(pcfg)) {             // stride 2: its own geometry (lp_conv3x3_s2p16.inc); the kernel indexes LDS and global memory from these
        const int hh2 = 2 * a.TH + 1, hw2 = 2 * a.TW + 1, nchunks = a.chunk_begin[a.nsrc];
        if (a.TH < 1 || a.TW < 1 || a.TH * a.TW > s.PB || a.hpitch != hw2 || hh2 * a.hpitch > s.HPMAX)
            return fail(LP_ERR_ARG, "conv3x3 s2p16: tile does not fit the kernel configuration");
        if (a.tiles_x * a.TW < a.Wo || a.tiles_y * a.TH < a.Ho) return fail(LP_ERR_ARG, "conv3x3 s2p16: tiles do not cover the output");
        if (a.nsrc < 1 || a.nsrc > LP_MAX_SRC || a.nct < 1 || a.nct * s.CB > 512 || nchunks < 1 || nchunks > 64 || a.nphase != 1 || a.out_scale != 1 ||
            a.H != 2 * a.Ho || a.W != 2 * a.Wo || a.out2)
            return fail(LP_ERR_ARG, "conv3x3 s2p16: not a 3x3 stride-2 layer this kernel runs");
        const int stride = a.out_pix_stride > a.res_cs ? a.out_pix_stride : a.res_cs;
        if (((long long)a.TH * a.Wo + a.TW) * stride + 2048 >= (1ll << 31)) return fail(LP_ERR_ARG, "conv3x3 s2p16: a tile's rows span more than 2^31 elements");
        switch (dtype) {
            case LP_F16: return conv_pipe_launch_f16(pcfg, a, device_cus(), st);
            case LP_BF16: return conv_pipe_launch_bf16(pcfg, a, device_cus(), st);
        }
        return fail(LP_ERR_UNSUPPORTED, "conv3x3 s2p16: 16-bit activations only");
    }
    const int hh = a.TH + 2, hw = a.TW + 2;
    if (pcfg < 0 || (pcfg >= PIPE_COUNT && !pipe_is_16(pcfg))) return fail(LP_ERR_ARG, "conv3x3 pipe: configuration");
    if (pipe_is_16(pcfg) && (a.chunk_begin[a.nsrc] < 4 || a.chunk_begin[a.nsrc] % 4 != 0)) return fail(LP_ERR_ARG, "conv3x3 pipe16: K-chunks not a multiple of four");
    if (a.TH < 1 || a.TW < 1 || a.TH * a.TW > s.PB || a.hpitch < hw || hh * a.hpitch > s.HPMAX)
        return fail(LP_ERR_ARG, "conv3x3 pipe: tile does not fit the kernel configuration");
    if (a.tiles_x * a.TW < a.Wo || a.tiles_y * a.TH < a.Ho) return fail(LP_ERR_ARG, "conv3x3 pipe: tiles do not cover the output");
    if (pipe_is_16v(pcfg) && (a.nct * s.CB > 512 || a.chunk_begin[a.nsrc] > 32)) return fail(LP_ERR_ARG, "conv3x3 pipe16v: layer exceeds the kernel's LDS tables");
    if (pipe_is_16v(pcfg)) {      // its epilogue addresses a tile's rows with 32-bit element offsets from the tile origin
        int stride = a.out_pix_stride > a.res_cs ? a.out_pix_stride : a.res_cs;
        if (a.out2 && a.out2_pix_stride > stride) stride = a.out2_pix_stride;
        if (((long long)a.TH * a.Wo + a.TW) * stride + 2048 >= (1ll << 31)) return fail(LP_ERR_ARG, "conv3x3 pipe16v: a tile's rows span more than 2^31 elements");
    }
    if (a.nsrc < 1 || a.nsrc > LP_MAX_SRC || a.nct < 1 || a.nct * s.CB > 1024 || a.nphase != 1 || a.out_scale != 1 || a.Ho != a.H || a.Wo != a.W)
        return fail(LP_ERR_ARG, "conv3x3 pipe: not a 3x3 stride-1 layer this kernel runs");
    switch (dtype) {
        case LP_F16: return conv_pipe_launch_f16(pcfg, a, device_cus(), st);
        case LP_BF16: return conv_pipe_launch_bf16(pcfg, a, device_cus(), st);
    }
    return fail(LP_ERR_UNSUPPORTED, "conv3x3 pipe: 16-bit activations only");
}

// Row-writer form of the class predictors (lp_head_rows.inc): whether the op fits, and the launch.
bool head_rows_fits(int dtype, int nchunks, int cb_pack, int out_c) {
    const long lds = (long)nchunks * 9 * 32 * 128 + 9 * 32 * 4 + 32 * 8 + 32 * 1168;
    return nchunks >= 1 && nchunks <= 3 && lds <= 160 * 1024 && (cb_pack == 32 || cb_pack == 64 || cb_pack == 128) && out_c == LP_PRED_COLS - 13;
}
// ... and of its detections-only form (head_det_kernel stages six rows at a time: four K-chunks of resident weights fit)
bool head_det_fits(int dtype, int nchunks, int cb_pack, int out_c) {
    const long lds = (long)nchunks * 9 * 32 * 128 + 9 * 32 * 4 + 32 * 8 * 4 + 16 + 6 * 32 * 8 * 4 + 6 * 1168;
    return nchunks >= 1 && nchunks <= (dtype == LP_F32 ? 3 : 4) && lds <= 160 * 1024 && (cb_pack == 32 || cb_pack == 64 || cb_pack == 128) &&
           out_c == LP_PRED_COLS - 13;
}

// Streaming form of the box / corner predictors in the detections-only forward (lp_head_box.inc): whether the op fits, and the launch.
bool head_box_det_fits(const ConvArgs& a, int cb_pack, int ksize, int stride) {
    const int nchunks = a.chunk_begin[a.nsrc];
    return ksize == 1 && stride == 1 && a.nsrc == 1 && nchunks >= 1 && nchunks <= 4 && a.reg_bins == 1 && a.out_c == 12 && a.nct == 1 &&
           cb_pack >= 32 && a.nphase == 1 && a.out_scale == 1 && a.Ho == a.H && a.Wo == a.W && a.src[0].cs % 8 == 0;
}
int head_box_det_launch(int dtype, const ConvArgs& a, int cb_pack, hipStream_t st) {
    if (!a.det_mode || !a.out || !a.zero || !head_box_det_fits(a, cb_pack, 1, 1)) return fail(LP_ERR_ARG, "head box: op does not fit");
    if (a.out_pix_stride % 4 != 0 || a.out_img_stride % 4 != 0 || ((uintptr_t)a.out & 15)) return fail(LP_ERR_ARG, "head box: candidate rows are not 16-byte aligned");
    switch (dtype) {
        case LP_F16: return head_box_det_launch_f16(a, cb_pack, st);
        case LP_BF16: return head_box_det_launch_bf16(a, cb_pack, st);
        case LP_F32: return head_box_det_launch_f32(a, cb_pack, st);
    }
    return fail(LP_ERR_ARG, "head box: dtype");
}

int head_rows_launch(int dtype, const ConvArgs& a, int cb_pack, hipStream_t st) {
    const int nchunks = a.chunk_begin[a.nsrc];
    if (!(a.det_mode ? head_det_fits(dtype, nchunks, cb_pack, a.out_c) : head_rows_fits(dtype, nchunks, cb_pack, a.out_c)))
        return fail(LP_ERR_ARG, "head rows: op does not fit");
    const int kc = 128 / (int)dtype_size(dtype);
    for (int i = 0; i < a.nsrc; ++i)                       // whole K-chunks, or ONE partial chunk of whole 16-channel K-steps
        if (a.src[i].cs % (kc / 4) != 0)
            return fail(LP_ERR_ARG, "head rows: source channels are not whole 32-byte K-steps");
    if (a.nphase != 1 || a.out_scale != 1 || a.Ho != a.H || a.Wo != a.W || !a.out) return fail(LP_ERR_ARG, "head rows: bad geometry");
    switch (dtype) {
        case LP_F16: return head_rows_launch_f16(a, cb_pack, st);
        case LP_BF16: return head_rows_launch_bf16(a, cb_pack, st);
        case LP_F32: return head_rows_launch_f32(a, cb_pack, st);
    }
    return fail(LP_ERR_ARG, "head rows: dtype");
}

}  // namespace lp

// Host-side planning of the block-tiled 3x3 kernels (conv3x3_pipe16v_kernel, conv3x3_s2p16_kernel), exported for the CPU tests of the host logic
// (no device needed: without one the grid is taken as 256 workgroups).
extern "C" int lp_plan_block_tile(int variant, int Ho, int Wo, int B, int nct, int choice, int* TH, int* TW, int* hpitch) {
    const int pcfg = variant - LP_VARIANT_PIPE_D;
    if (!TH || !TW || Ho < 1 || Wo < 1 || B < 1 || nct < 1) return lp::fail(LP_ERR_ARG, "lp_plan_block_tile: bad arguments");
    if (!lp::pipe_is_16v(pcfg) && !lp::pipe_is_16s2(pcfg)) return lp::fail(LP_ERR_UNSUPPORTED, "lp_plan_block_tile: not a block-tiled variant");
    const lp::ConvShape s = lp::conv_pipe_shape(pcfg);
    const int stride = lp::pipe_is_16s2(pcfg) ? 2 : 1;
    lp::conv_pick_tile16v(s, Ho, Wo, B, nct, choice, TH, TW, stride);
    if (hpitch) *hpitch = (*TW - 1) * stride + 3;
    return LP_OK;
}

// Host-side planning of the two frame-reading stem kernels, exported for the CPU tests of the host logic (no device needed).
extern "C" int lp_plan_stem_tile(int fused, int Ho, int Wo, int choice, int* TH, int* TW, int* hpitch) {
    if (!TH || !TW || Ho < 1 || Wo < 1) return lp::fail(LP_ERR_ARG, "lp_plan_stem_tile: bad arguments");
    int hp = 0;
    const bool ok = fused ? lp::stem2_fused_tile(Ho, Wo, choice, TH, TW, &hp) : lp::stem_planar_tile(Ho, Wo, choice, TH, TW);
    if (hpitch) *hpitch = hp;
    return ok ? LP_OK : LP_ERR_UNSUPPORTED;
}

