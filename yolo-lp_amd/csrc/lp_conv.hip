// Host-side geometry of the implicit-GEMM conv kernels and the per-dtype launch dispatch.
// The kernels themselves are in lp_conv_kernel.inc (one translation unit per activation dtype).
#include "lp_internal.h"

namespace lp {

int conv_launch_f16(int cfg, int mode, int ksize, int stride, int nbuf, const ConvArgs& a, hipStream_t st);
int conv_launch_bf16(int cfg, int mode, int ksize, int stride, int nbuf, const ConvArgs& a, hipStream_t st);
int conv_launch_f32(int cfg, int mode, int ksize, int stride, int nbuf, const ConvArgs& a, hipStream_t st);

ConvShape conv_shape(int dtype, int cfg, int ksize, int stride) {
    const int sz = (int)dtype_size(dtype);
    ConvShape s;
    const int wc = cfg == CFG_C ? 1 : 2, wp = 2, wgc = (cfg == CFG_A || cfg == CFG_D) ? 2 : 1, wgp = cfg == CFG_A ? 2 : 4;
    s.CB = 32 * wc * wgc;
    s.PB = 32 * wp * wgp;
    s.NT = 64 * wgc * wgp;
    s.KC = ksize == 1 ? 128 / sz : 32 / sz;
    s.HPMAX = ksize == 1 ? s.PB : (stride == 1 ? s.PB + s.PB / 2 + 64 : 4 * s.PB + s.PB / 2 + 64);
    return s;
}

void conv_pick_tile(const ConvShape& s, int ksize, int stride, int Ho, int Wo, int* TH, int* TW) {
    long best_cost = -1;
    int bh = 1, bw = 1;
    for (int th = 1; th <= s.PB && th <= Ho + 0; ++th) {
        int tw = s.PB / th;
        if (tw > Wo) tw = Wo;
        if (tw < 1) continue;
        const int hh = (th - 1) * stride + ksize, hw = (tw - 1) * stride + ksize;
        if (hh * hw > s.HPMAX) continue;
        const long tiles = (long)ceil_div(Ho, th) * ceil_div(Wo, tw);
        const long cost = tiles * 100000 + (long)hh * hw;  // fewest blocks first, then the smallest halo
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; bh = th; bw = tw; }
    }
    *TH = bh;
    *TW = bw;
}

int conv_launch(int dtype, int cfg, int mode, int ksize, int stride, int nbuf, const ConvArgs& a, hipStream_t st) {
    // host-side shape checks: the kernel indexes LDS and global memory from these without further tests
    const ConvShape s = conv_shape(dtype, cfg, ksize, stride);
    const int hh = (a.TH - 1) * stride + ksize, hw = (a.TW - 1) * stride + ksize;
    if (a.TH < 1 || a.TW < 1 || a.TH * a.TW > s.PB || hh * hw > s.HPMAX)
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

}  // namespace lp
