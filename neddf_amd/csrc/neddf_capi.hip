// neddf_capi.hip -- C ABI of libneddf_hip.so (include/neddf_hip.h): context,
// weight packing into MFMA fragment order, workspaces, stage entry points and
// the fused render_rays orchestration (all launches on the caller's stream, no
// host round trip inside a call).
#include "../../include/neddf_hip.h"
#include "kernels.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

using namespace neddf;

#include "capi_internal.h"

// one stage launch, bracketed by a hipEvent pair while timing is on
#define STAGE(ctx, stream, which, launch) \
    do { tick((ctx), (stream), (which), true); launch; tick((ctx), (stream), (which), false); } while (0)

static char g_err[256] = "no context";

void neddf_comm_release(neddf_ctx *ctx);       // comm_capi.hip

static int sched_flags() { return 2; }       // bit 1: dynamic tile queue (kernels.h DdfArgs::sched_flags)


// ---------------------------------------------------------------------------
// weight packing
struct Src {
    const float *w;
    int rows, cols;       // logical [in][out]
    bool transposed;      // storage is [out][in] (nn.Linear)
    float at(int k, int n) const { return transposed ? w[(size_t)n * rows + k] : w[(size_t)k * cols + n]; }
};

// round-to-nearest-even fp32 -> bf16 bit pattern (what v_cvt_pk_bf16_f32 does on the device)
static inline uint16_t bf16_bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);      // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// fp32 -> IEEE half bit pattern, round to nearest even (toward_zero = false) or toward zero (true; never overflows to inf)
static inline uint16_t f16_bits(float f, bool toward_zero)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (uint16_t)(sign | (u > 0x7f800000u ? 0x7e00u : (toward_zero ? 0x7bffu : 0x7c00u)));
    if (u >= 0x477ff000u) {                                     // >= 65520: beyond the largest half
        if (toward_zero || u < 0x477ff000u) return (uint16_t)(sign | 0x7bffu);
        return (uint16_t)(sign | (toward_zero ? 0x7bffu : 0x7c00u));
    }
    if (u < 0x33000000u) return (uint16_t)sign;                 // < 2^-25: rounds to zero
    int e = (int)(u >> 23) - 127;
    uint32_t man = (u & 0x7fffffu) | 0x800000u;                 // 24-bit significand
    int shift = e >= -14 ? 13 : 13 + (-14 - e);                 // bits dropped (subnormal halves drop more)
    uint32_t q = man >> shift, rem = man & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    if (!toward_zero && (rem > halfway || (rem == halfway && (q & 1)))) ++q;
    uint32_t h = e >= -14 ? (uint32_t)((e + 15) << 10) + (q - 0x400u) : q;      // q carries into the exponent on overflow
    return (uint16_t)(sign | h);
}

static inline float f16_value(uint16_t h)
{
    const int e = (h >> 10) & 31, m = h & 0x3ff;
    float v = e == 0 ? ldexpf((float)m, -24) : ldexpf((float)(m | 0x400), e - 25);
    return (h & 0x8000) ? -v : v;
}

// fragment-major packing, see kernels.h LayerW.  kmap (engine column -> reference weight row, -1 = zero) is padded to
// a whole number of super-steps: 8 k-values for fp32 fragments (4 floats per lane half), 16 for bf16 (8 per lane half).
static size_t pack_layer(std::vector<float> &blob, const Src &src, std::vector<int> kmap, int nout, int operands, int *ksteps_out)
{
    // operands == 2 (split fp16): two fp16 planes of 2^10 w (h toward zero, m = remainder to nearest), 32 bytes per lane and super-step
    const int step = operands ? 16 : 8, half = step / 2, tiles = nout / 32, planes = operands == 2 ? 2 : 1;     // nout: multiple of 32
    while (kmap.size() % step) kmap.push_back(-1);
    const int ks = (int)kmap.size() / step;
    if (ksteps_out) *ksteps_out = ks;
    size_t off = roundup((int)blob.size(), 64);
    const size_t elems = (size_t)ks * step * nout;
    blob.resize(off + (operands ? elems * planes / 2 : elems), 0.f);
    float *dst = blob.data() + off;
    uint16_t *dst16 = (uint16_t *)dst;
    for (int tile = 0; tile < tiles; ++tile)            // tile-major: tile = wave * NT + t in the kernels
            for (int S = 0; S < ks; ++S)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < half; ++r) {
                        int n = tile * 32 + (lane & 31);
                        int k = kmap[step * S + half * (lane >> 5) + r];
                        float v = (k < 0 || n >= src.cols) ? 0.f : src.at(k, n);
                        size_t frag = (((size_t)tile * ks + S) * 64 + lane);
                        if (operands == 2) {
                            const float sv = v * 1024.0f;
                            uint16_t h = f16_bits(sv, true);
                            dst16[(frag * 2 + 0) * half + r] = h;
                            dst16[(frag * 2 + 1) * half + r] = f16_bits(sv - f16_value(h), false);
                        } else if (operands) dst16[frag * half + r] = bf16_bits(v);
                        else dst[frag * half + r] = v;
                    }
    return off;
}

static size_t put(std::vector<float> &blob, const float *p, size_t n, size_t n_padded = 0)      // zero-padded to n_padded elements
{
    size_t off = roundup((int)blob.size(), 64);
    blob.resize(off + (n_padded > n ? n_padded : n), 0.f);
    memcpy(blob.data() + off, p, n * sizeof(float));
    return off;
}

// engine columns of a hidden state of `width` features (zero-padded to the engine width wp) -> reference weight row base + k
static std::vector<int> hidden_map(int width, int wp, int base)
{
    std::vector<int> m;
    for (int k = 0; k < wp; ++k) m.push_back(k < width ? base + k : -1);
    return m;
}

// engine width of a hidden width: the next multiple of 128 (four waves x 32-column MFMA tiles).  The padding columns carry zero
// weights and biases, so they hold a(0) = 0 under every activation of the reference and feed zero rows downstream: exact.
// A tile row also has to hold both encodings next to a 32-column block (enc_columns): a narrow network with long encodings takes the next
// engine width that does
static int enc_columns(const neddf_field_desc &d) { return 2 * roundup(3 * d.embed_pos_rank, 4) + 2 * roundup(3 * d.embed_dir_rank, 4) + 32; }
static int engine_width(int width, const neddf_field_desc &d)
{
    int w = roundup(width, 128);
    while (w < enc_columns(d)) w += 128;
    return w;
}

// engine columns of the [sin half | cos half] encoding -> reference feature index base+...
static void enc_map(std::vector<int> &m, int rank, int K, int base)
{
    for (int q = 0; q < K; ++q) m.push_back(q < 3 * rank ? base + q : -1);
    for (int q = 0; q < K; ++q) m.push_back(q < 3 * rank ? base + 3 * rank + q : -1);
}



static int build_neddf(neddf_ctx *ctx, Field &f, const float *const *W, const float *const *B, int n_tensors)
{
    const neddf_field_desc &d = f.d;
    const int E = d.embed_pos_rank, Ed = d.embed_dir_rank, n_trunk = d.layer_count - 1, n_col = d.col_layer_count - 1;
    const int KH = roundup(3 * E, 4), KD = roundup(3 * Ed, 4), Cpe = 6 * E, Cdir = 6 * Ed;
    const int operands = d.weight_dtype;      // 0 fp32, 1 bf16, 2 split fp16
    const int Wd = d.layer_width, WP = engine_width(Wd, d);      // hidden width of the reference network / of the tile engine
    if (d.col_layer_width != Wd)
        return fail(ctx, NEDDF_EUNSUPPORTED, "NeDDF: col_layer_width must equal ddf_layer_width (the reference's layer_col_out takes ddf_layer_width inputs, "
                                             "neddf.py:145: its own forward fails otherwise)");
    if (n_tensors != n_trunk + n_col + 3) return fail(ctx, NEDDF_EINVAL, "NeDDF: wrong tensor count");
    if (n_trunk < 1 || n_trunk > kMaxLayers || n_col < 1 || n_col > kMaxLayers) return fail(ctx, NEDDF_EUNSUPPORTED, "NeDDF: layer count out of range");
    for (int i = 0; i < d.n_skips; ++i)
        if (d.skips[i] < 0 || d.skips[i] >= n_trunk - 1)
            return fail(ctx, NEDDF_EUNSUPPORTED, "NeDDF: skip index must address a trunk layer followed by another (the reference itself fails otherwise)");
    std::vector<float> blob;
    DdfArgs &a = f.ddf;
    ColArgs &c = f.col;
    a = DdfArgs{}; c = ColArgs{};
    std::vector<size_t> o_wp(n_trunk), o_b(n_trunk), o_st;
    std::vector<int> pe;
    enc_map(pe, E, KH, 0);
    a.n_layers = n_trunk; a.n_stash = 0;
    for (int l = 0; l < n_trunk; ++l) {
        bool wide = l > 0 && in_skips(d, l - 1);
        int cin = l == 0 ? Cpe : (wide ? Wd + Cpe : Wd);
        Src src{ W[l], cin, Wd, false };
        std::vector<int> km;
        a.layer[l].stash = -1;
        if (l == 0) km = pe;
        else km = hidden_map(Wd, WP, wide ? Cpe : 0);
        o_wp[l] = pack_layer(blob, src, km, WP, operands, &a.layer[l].ksteps);
        if (wide) {
            if (a.n_stash >= kMaxStash) return fail(ctx, NEDDF_EUNSUPPORTED, "NeDDF: more than 4 skip connections");
            o_st.push_back(pack_layer(blob, src, pe, WP, operands, &a.stash[a.n_stash].ksteps));
            a.stash[a.n_stash].col0 = 0;
            a.layer[l].stash = a.n_stash++;
        }
        o_b[l] = put(blob, B[l], Wd, WP);
    }
    const int i_ddf = n_trunk + n_col, i_aux = i_ddf + 1, i_cout = i_ddf + 2;
    size_t o_wddf = put(blob, W[i_ddf], Wd, WP), o_waux = put(blob, W[i_aux], Wd, WP);
    a.b_ddf_out = B[i_ddf][0]; a.b_aux_out = B[i_aux][0];
    // Reverse-mode distance gradient (ddf_rev_kernel): the transposes.  dL/dH_{l-1} = g_l x (hidden rows of W_l)^T is a dense
    // product with B[k][n] = W_l[row0 + n][k]; the gradient of the encoding collects g_0 x W_0^T and g_skip x (encoding rows of
    // W_skip)^T, 64 engine columns wide (same [sin half | cos half] order as the forward encoding in LDS)
    std::vector<size_t> o_wT(n_trunk, 0);
    size_t o_wT_pe0 = 0, o_wT_pes[kMaxStash] = { 0 };
    a.skip_layer = -1;
    {
        const std::vector<int> kall = hidden_map(Wd, WP, 0);
        // narrow transposes: column n of the engine's encoding layout is reference row pe[n] (or padding)
        auto pack_pe_T = [&](const float *Wl) {
            std::vector<float> tmp((size_t)Wd * 64, 0.f);           // [k][n] row-major, n < 64
            for (int n = 0; n < (int)pe.size() && n < 64; ++n)
                if (pe[n] >= 0)
                    for (int k = 0; k < Wd; ++k) tmp[(size_t)k * 64 + n] = Wl[(size_t)pe[n] * Wd + k];
            Src sn{ tmp.data(), Wd, 64, false };
            return pack_layer(blob, sn, kall, 64, operands, nullptr);
        };
        for (int l = 1; l < n_trunk; ++l) {
            const bool wide = in_skips(d, l - 1);
            // logical B[k][n] = W_l[(wide ? Cpe : 0) + n][k], W_l row-major [in][Wd]  ==  Src "transposed" with rows = Wd
            Src st{ W[l] + (size_t)(wide ? Cpe : 0) * Wd, Wd, Wd, true };
            o_wT[l] = pack_layer(blob, st, kall, WP, operands, nullptr);
            if (wide) { a.skip_layer = l; o_wT_pes[a.layer[l].stash] = pack_pe_T(W[l]); }
        }
        o_wT_pe0 = pack_pe_T(W[0]);
    }
    // colour trunk
    std::vector<int> ka;
    enc_map(ka, E, KH, 0);
    enc_map(ka, Ed, KD, Cpe);
    for (int k = 0; k < 3; ++k) ka.push_back(Cpe + Cdir + k);
    std::vector<size_t> c_wp(n_col), c_b(n_col);
    Src s0{ W[n_trunk], Cpe + Cdir + 3 + Wd, Wd, false };
    size_t o_wa = pack_layer(blob, s0, ka, WP, operands, &c.ksteps_a);
    c.n_layers = n_col;
    for (int l = 0; l < n_col; ++l) {
        const std::vector<int> km = hidden_map(Wd, WP, l == 0 ? Cpe + Cdir + 3 : 0);
        Src src{ W[n_trunk + l], l == 0 ? Cpe + Cdir + 3 + Wd : Wd, Wd, false };
        c_wp[l] = pack_layer(blob, src, km, WP, operands, &c.layer[l].ksteps);
        c.layer[l].stash = -1;
        c_b[l] = put(blob, B[n_trunk + l], Wd, WP);
    }
    size_t o_cout = put(blob, W[i_cout], (size_t)Wd * 3, (size_t)WP * 3);
    for (int k = 0; k < 3; ++k) c.b_out[k] = B[i_cout][k];

    if (int rc = ensure(ctx, f.blob, blob.size() * sizeof(float))) return rc;
    HIPCHK(hipMemcpy(f.blob.p, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    const float *base = (const float *)f.blob.p;
    for (int l = 0; l < n_trunk; ++l) { a.layer[l].wp = base + o_wp[l]; a.layer[l].bias = base + o_b[l]; }
    for (int s = 0; s < a.n_stash; ++s) a.stash[s].wp = base + o_st[s];
    a.w_ddf_out = base + o_wddf; a.w_aux_out = base + o_waux;
    for (int l = 1; l < n_trunk; ++l) a.wT[l] = base + o_wT[l];
    a.wT_pe0 = base + o_wT_pe0;
    for (int st = 0; st < a.n_stash; ++st) a.wT_pe_skip[st] = base + o_wT_pes[st];
    a.ks_hidden = WP / (operands ? 16 : 8);
    a.width = c.width = WP;
    c.wp_a = base + o_wa;
    for (int l = 0; l < n_col; ++l) { c.layer[l].wp = base + c_wp[l]; c.layer[l].bias = base + c_b[l]; }
    c.w_out = base + o_cout;
    a.activation = c.activation = d.activation;
    a.density_activation = d.density_activation;
    a.d_near = d.d_near;
    a.operands = c.operands = operands;
    c.final_act = -1;
    for (int k = 0; k < 6; ++k) { c.penalty_weight[k] = d.penalty_weight[k]; c.penalty_has[k] = d.penalty_has[k]; }
    return 0;
}

// NeuS (neddf/network/neus.py): sdf trunk with Jacobian rows on the distance kernel (forward-mode replaces the
// reference's torch.autograd.grad), colour trunk on the colour kernel.
static int build_neus(neddf_ctx *ctx, Field &f, const float *const *W, const float *const *B, int n_tensors)
{
    const neddf_field_desc &d = f.d;
    const int E = d.embed_pos_rank, Ed = d.embed_dir_rank, n_sdf = d.layer_count, n_col = d.col_layer_count;
    const int KH = roundup(3 * E, 4), KD = roundup(3 * Ed, 4), Cpe = 6 * E, Cdir = 6 * Ed;
    const int operands = d.weight_dtype;
    // sdf and colour trunks may have different widths (neus.py:80-99); both run on one engine width, zero-padded
    const int Ws = d.layer_width, Wc = d.col_layer_width, WP = engine_width(Ws > Wc ? Ws : Wc, d);
    if (n_tensors != n_sdf + n_col + 2) return fail(ctx, NEDDF_EINVAL, "NeuS: wrong tensor count");
    if (n_sdf < 1 || n_sdf > kMaxLayers || n_col < 1 || n_col > kMaxLayers) return fail(ctx, NEDDF_EUNSUPPORTED, "NeuS: layer count out of range");
    if (d.activation == NEDDF_ACT_LEAKY) return fail(ctx, NEDDF_EUNSUPPORTED, "NeuS: activation must be ReLU or tanhExp");
    for (int i = 0; i < d.n_skips; ++i)
        if (d.skips[i] < 0 || d.skips[i] >= n_sdf - 1)
            return fail(ctx, NEDDF_EUNSUPPORTED, "NeuS: skip index must address an sdf layer followed by another");
    std::vector<float> blob;
    DdfArgs &a = f.ddf;
    ColArgs &c = f.col;
    a = DdfArgs{}; c = ColArgs{};
    std::vector<size_t> o_wp(n_sdf), o_b(n_sdf), o_st;
    std::vector<int> pe;
    enc_map(pe, E, KH, 0);
    a.n_layers = n_sdf; a.n_stash = 0;
    for (int l = 0; l < n_sdf; ++l) {
        bool wide = l > 0 && in_skips(d, l - 1);
        int cin = l == 0 ? Cpe : (wide ? Ws + Cpe : Ws);
        Src src{ W[l], cin, Ws, true };
        std::vector<int> km;
        a.layer[l].stash = -1;
        if (l == 0) km = pe;
        else km = hidden_map(Ws, WP, 0);                            // cat([hx, embed_pos]): hidden state first
        o_wp[l] = pack_layer(blob, src, km, WP, operands, &a.layer[l].ksteps);
        if (wide) {
            if (a.n_stash >= kMaxStash) return fail(ctx, NEDDF_EUNSUPPORTED, "NeuS: more than 4 skip connections");
            std::vector<int> ps;
            enc_map(ps, E, KH, Ws);
            o_st.push_back(pack_layer(blob, src, ps, WP, operands, &a.stash[a.n_stash].ksteps));
            a.stash[a.n_stash].col0 = 0;
            a.layer[l].stash = a.n_stash++;
        }
        o_b[l] = put(blob, B[l], Ws, WP);
    }
    // Reverse-mode normal (ddf_rev_kernel): the sdf is feature 0 of the last activated layer, so the seed of the reverse pass is
    // e_0 * y'_L -- the "distance head" of the kernel becomes the unit vector e_0 with zero bias, the aux head is zero -- and the
    // transposes are read straight off nn.Linear's [out][in] storage: B[k][n] = W_l[k][n] for the hidden inputs n < 256 (hidden
    // state first in NeuS's concatenation), the encoding inputs through the engine's column map
    std::vector<size_t> o_wT(n_sdf, 0);
    size_t o_wT_pe0 = 0, o_wT_pes[kMaxStash] = { 0 }, o_e0 = 0, o_zero = 0;
    a.skip_layer = -1;
    {
        const std::vector<int> kall = hidden_map(Ws, WP, 0);
        auto pack_pe_T = [&](const float *Wl, int cin, int base) {
            std::vector<int> cols;
            enc_map(cols, E, KH, base);
            std::vector<float> tmp((size_t)Ws * 64, 0.f);           // [k][n] row-major, n < 64
            for (int n = 0; n < (int)cols.size() && n < 64; ++n)
                if (cols[n] >= 0)
                    for (int k = 0; k < Ws; ++k) tmp[(size_t)k * 64 + n] = Wl[(size_t)k * cin + cols[n]];
            Src sn{ tmp.data(), Ws, 64, false };
            return pack_layer(blob, sn, kall, 64, operands, nullptr);
        };
        for (int l = 1; l < n_sdf; ++l) {
            const bool wide = in_skips(d, l - 1);
            // logical B[k][n] = W_l[k][n] over nn.Linear's [out][in] storage, hidden inputs n < Ws only
            Src st{ W[l], Ws, wide ? Ws + Cpe : Ws, false };
            std::vector<float> sq;                                   // pack_layer reads n < src.cols: cut the encoding inputs off
            if (wide) {
                sq.assign((size_t)Ws * Ws, 0.f);
                for (int k = 0; k < Ws; ++k) memcpy(&sq[(size_t)k * Ws], W[l] + (size_t)k * (Ws + Cpe), (size_t)Ws * sizeof(float));
                st = Src{ sq.data(), Ws, Ws, false };
            }
            o_wT[l] = pack_layer(blob, st, kall, WP, operands, nullptr);
            if (wide) { a.skip_layer = l; o_wT_pes[a.layer[l].stash] = pack_pe_T(W[l], Ws + Cpe, Ws); }
        }
        o_wT_pe0 = pack_pe_T(W[0], Cpe, 0);
        std::vector<float> e0(WP, 0.f), zero(WP, 0.f);
        e0[0] = 1.0f;
        o_e0 = put(blob, e0.data(), WP);
        o_zero = put(blob, zero.data(), WP);
    }
    // colour trunk: engine columns [pos 3 | gradient 3 | pad 2 | dir sin KD | dir cos KD]
    std::vector<int> ka;
    for (int k = 0; k < 3; ++k) ka.push_back(k);
    for (int k = 0; k < 3; ++k) ka.push_back(3 + Cdir + k);
    ka.push_back(-1); ka.push_back(-1);
    enc_map(ka, Ed, KD, 3);
    const int in_col = 6 + Cdir + Ws;
    Src s0{ W[n_sdf], in_col, Wc, true };
    size_t o_wa = pack_layer(blob, s0, ka, WP, operands, &c.ksteps_a);
    c.n_layers = n_col;
    std::vector<size_t> c_wp(n_col), c_b(n_col);
    for (int l = 0; l < n_col; ++l) {
        const std::vector<int> km = l == 0 ? hidden_map(Ws, WP, 6 + Cdir) : hidden_map(Wc, WP, 0);
        Src src{ W[n_sdf + l], l == 0 ? in_col : Wc, Wc, true };
        c_wp[l] = pack_layer(blob, src, km, WP, operands, &c.layer[l].ksteps);
        c.layer[l].stash = -1;
        c_b[l] = put(blob, B[n_sdf + l], Wc, WP);
    }
    std::vector<float> wout((size_t)WP * 3, 0.f);         // [3][Wc] -> [WP][3]
    for (int k = 0; k < Wc; ++k)
        for (int o = 0; o < 3; ++o) wout[k * 3 + o] = W[n_sdf + n_col][(size_t)o * Wc + k];
    size_t o_cout = put(blob, wout.data(), wout.size());
    for (int k = 0; k < 3; ++k) c.b_out[k] = B[n_sdf + n_col][k];
    if (int rc = ensure(ctx, f.blob, blob.size() * sizeof(float))) return rc;
    HIPCHK(hipMemcpy(f.blob.p, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    const float *base = (const float *)f.blob.p;
    for (int l = 0; l < n_sdf; ++l) { a.layer[l].wp = base + o_wp[l]; a.layer[l].bias = base + o_b[l]; }
    for (int s = 0; s < a.n_stash; ++s) a.stash[s].wp = base + o_st[s];
    c.wp_a = base + o_wa;
    for (int l = 0; l < n_col; ++l) { c.layer[l].wp = base + c_wp[l]; c.layer[l].bias = base + c_b[l]; }
    c.w_out = base + o_cout;
    for (int l = 1; l < n_sdf; ++l) a.wT[l] = base + o_wT[l];
    a.wT_pe0 = base + o_wT_pe0;
    for (int st = 0; st < a.n_stash; ++st) a.wT_pe_skip[st] = base + o_wT_pes[st];
    a.ks_hidden = WP / (operands ? 16 : 8);
    a.width = c.width = WP;
    a.w_ddf_out = base + o_e0; a.w_aux_out = base + o_zero;
    a.b_ddf_out = 0.f; a.b_aux_out = 0.f;
    a.activation = c.activation = d.activation;
    a.neus = 1;
    a.operands = c.operands = operands;
    a.neus_v10 = W[n_sdf + n_col + 1][0] * 10.0f;
    c.mode = 1;
    c.final_act = d.activation;
    return 0;
}

static int build_nerf(neddf_ctx *ctx, Field &f, const float *const *W, const float *const *B, int n_tensors)
{
    const neddf_field_desc &d = f.d;
    const int E = d.embed_pos_rank, Ed = d.embed_dir_rank, n = d.layer_count;
    const int KH = roundup(3 * E, 4), KD = roundup(3 * Ed, 4), Cpe = 6 * E, Cdir = 6 * Ed;
    const int operands = d.weight_dtype, step = operands ? 16 : 8;
    const int Wn = d.layer_width, WP = engine_width(Wn, d);
    const int Wh = Wn / 2, HC = roundup(Wh > 0 ? Wh : 1, 128);      // colour head's hidden layer (nerf.py:99-103: layer_width // 2) and its engine width
    if (n_tensors != n + 3) return fail(ctx, NEDDF_EINVAL, "NeRF: wrong tensor count");
    if (n < 1 || n > kMaxLayers) return fail(ctx, NEDDF_EUNSUPPORTED, "NeRF: layer count out of range");
    for (int i = 0; i < d.n_skips; ++i)
        if (d.skips[i] < 0 || d.skips[i] >= n - 1)
            return fail(ctx, NEDDF_EUNSUPPORTED, "NeRF: skip index must address a layer followed by another");
    std::vector<float> blob;
    NerfArgs &a = f.nerf;
    a = NerfArgs{};
    std::vector<int> pe;
    enc_map(pe, E, KH, 0);
    std::vector<size_t> o_wp(n), o_b(n), o_st;
    a.n_layers = n; a.n_stash = 0;
    for (int l = 0; l < n; ++l) {
        bool wide = l > 0 && in_skips(d, l - 1);
        int cin = l == 0 ? Cpe : (wide ? Wn + Cpe : Wn);
        Src src{ W[l], cin, Wn, true };
        std::vector<int> km;
        a.layer[l].stash = -1;
        if (l == 0) km = pe;
        else km = hidden_map(Wn, WP, 0);                            // cat([hx, embed_pos]): hidden state first
        o_wp[l] = pack_layer(blob, src, km, WP, operands, &a.layer[l].ksteps);
        if (wide) {
            if (a.n_stash >= kMaxStash - 1) return fail(ctx, NEDDF_EUNSUPPORTED, "NeRF: more than 3 skip connections");
            std::vector<int> ps;
            enc_map(ps, E, KH, Wn);
            o_st.push_back(pack_layer(blob, src, ps, WP, operands, &a.stash[a.n_stash].ksteps));
            a.stash[a.n_stash].col0 = 0;
            a.layer[l].stash = a.n_stash++;
        }
        o_b[l] = put(blob, B[l], Wn, WP);
    }
    size_t o_wd = put(blob, W[n], Wn, WP);
    a.b_density = B[n][0];
    // colour head: Linear(layer_width + dir, layer_width // 2)
    Src sc{ W[n + 1], Wn + Cdir, Wh, true };
    const std::vector<int> km = hidden_map(Wn, WP, 0);
    std::vector<int> kd;
    enc_map(kd, Ed, KD, Wn);
    size_t o_c0 = pack_layer(blob, sc, km, HC, operands, &a.col0.ksteps);
    a.col_stash = a.n_stash;
    size_t o_c0s = pack_layer(blob, sc, kd, HC, operands, &a.stash[a.n_stash].ksteps);
    a.stash[a.n_stash].col0 = roundup(2 * KH, step);          // direction encoding: first super-step boundary after the position encoding
    a.n_stash++;
    size_t o_c0b = put(blob, B[n + 1], Wh, HC);
    std::vector<float> w1((size_t)3 * HC, 0.f);               // [3][Wh] -> [3][HC]
    for (int o = 0; o < 3; ++o) memcpy(&w1[(size_t)o * HC], W[n + 2] + (size_t)o * Wh, (size_t)Wh * sizeof(float));
    size_t o_c1 = put(blob, w1.data(), w1.size());
    for (int k = 0; k < 3; ++k) a.b_col1[k] = B[n + 2][k];

    if (int rc = ensure(ctx, f.blob, blob.size() * sizeof(float))) return rc;
    HIPCHK(hipMemcpy(f.blob.p, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    const float *base = (const float *)f.blob.p;
    for (int l = 0; l < n; ++l) { a.layer[l].wp = base + o_wp[l]; a.layer[l].bias = base + o_b[l]; }
    for (int s = 0; s + 1 < a.n_stash; ++s) a.stash[s].wp = base + o_st[s];
    a.stash[a.col_stash].wp = base + o_c0s;
    a.w_density = base + o_wd;
    a.col0.wp = base + o_c0; a.col0.bias = base + o_c0b; a.col0.stash = a.col_stash;
    a.w_col1 = base + o_c1;
    a.activation = d.activation;
    a.density_activation = d.density_activation;
    a.operands = operands;
    a.width = WP;
    return 0;
}

// ---------------------------------------------------------------------------
// the slot's packed weights were just read by work enqueued on `s`: neddf_set_field waits for exactly this before it repacks
static int mark_use(neddf_ctx *ctx, Field &f, hipStream_t s)
{
    if (!f.last_use) HIPCHK(hipEventCreateWithFlags(&f.last_use, hipEventDisableTiming));
    // one event per slot, shared by every stream that reads it: a stream that takes the event over first waits for the previous
    // holder's mark, so the waits chain and the event always lies behind EVERY earlier reader (two streams reading one slot)
    if (f.last_use_recorded && f.last_use_stream != s) HIPCHK(hipStreamWaitEvent(s, f.last_use, 0));
    HIPCHK(hipEventRecord(f.last_use, s));
    f.last_use_recorded = true;
    f.last_use_stream = s;
    return 0;
}

// marks the slot on every way out of a function that may have launched readers of it (error returns after a chunk included)
struct UseMark {
    neddf_ctx *ctx; Field &f; hipStream_t s;
    ~UseMark() { (void)mark_use(ctx, f, s); }
};

// rays != NULL (neddf_render_rays): the N points are the [B, S] sample grid of these rays; pos / dir / var are then WORKSPACE -- filled by
// the sampling kernel when the route cannot take its points from the rays, untouched when it can (kernels.h RaySrc)
static int field_forward(neddf_ctx *ctx, int slot, const float *pos, const float *dir, const float *var, int64_t N,
                         int out_mode, float *distance, float *density, float *color, float *penalty, float *aux,
                         hipStream_t s, const RaySrc *rays = nullptr)
{
    if (slot < 0 || slot >= NEDDF_NUM_SLOTS || !ctx->field[slot].valid) return fail(ctx, NEDDF_ENOFIELD, "no field in slot");
    if (N <= 0) return 0;
    Field &f = ctx->field[slot];
    UseMark mark{ ctx, f, s };
    const int dt = f.d.weight_dtype;
    const int wid = f.d.kind == NEDDF_FIELD_NERF ? f.nerf.width : f.ddf.width;        // engine width
    // persistent grids: every workgroup slot of the device filled once (all 512 are resident, two per CU: tools/residency_probe.hip)
    const int grid_cap = ctx->cus * nerf_wgs_per_cu(wid);
    const int grid_cap_ddf = ctx->cus * field_wgs_per_cu(dt, wid);
    const int grid_cap_col = ctx->cus * col_wgs_per_cu(dt, wid);
    const int n_parked = f.d.kind == NEDDF_FIELD_NERF ? f.nerf.n_stash : f.ddf.n_stash;      // early partials a kernel may park per workgroup
    if (int rc = ensure(ctx, ctx->scratch, (size_t)(grid_cap > grid_cap_ddf ? grid_cap : grid_cap_ddf) * (n_parked > 1 ? n_parked : 1) * kStashFloatsPerWg * sizeof(float))) return rc;
    auto sample_now = [&]() {       // the sampling tensors after all (a route that reads them)
        STAGE(ctx, s, NEDDF_STAGE_SAMPLING, launch_sampling(rays->rd, rays->ro, rays->view, rays->dists, N / rays->S, rays->S,
                                                              rays->radius, (float *)pos, (float *)dir, (float *)var, s));
    };
    if (f.d.kind == NEDDF_FIELD_NERF) {
        if (rays) sample_now();
        NerfArgs a = f.nerf;
        fill_enc(a.enc, f);
        a.pos = pos; a.dir = dir; a.var = var; a.n_points = N;
        a.scratch = (float *)ctx->scratch.p;
        DevBuf &tmp = ctx->ptaux;   // NeRF needs no hand-off buffers; reuse ptaux for optional sinks
        if (!density || !color) {
            if (int rc = ensure(ctx, tmp, (size_t)N * 4 * sizeof(float))) return rc;
        }
        a.density = density ? density : (float *)tmp.p;
        a.color = color ? color : (float *)tmp.p + N;
        int64_t tiles = (N + nerf_points_per_tile(wid) - 1) / nerf_points_per_tile(wid);
        STAGE(ctx, s, NEDDF_STAGE_NERF, launch_nerf(a, (int)(tiles < grid_cap ? tiles : grid_cap), s));
        HIPCHK(hipGetLastError());
        return 0;
    }
    const bool full = (out_mode == NEDDF_OUT_FULL) && penalty && f.d.kind == NEDDF_FIELD_NEDDF;
    const int fr = full ? 4 : 1;
    // Eval-minimal NeDDF needs the gradient of one scalar (the distance) only: reverse mode halves the matrix work
    // (NEDDF_DDF_REVERSE=0 keeps the forward-mode Jacobian rows, which the penalties of the full mode need anyway)
    static const bool rev_enabled = [] { const char *e = getenv("NEDDF_DDF_REVERSE"); return !e || atoi(e) != 0; }();
    // (all three operand policies gain: fp32 27.5 vs 51.8 ms per 2^21 points, split fp16 12.4 vs 19.9 ms, bf16 6.4 vs 7.4 ms)
    const bool reverse = rev_enabled && !full && (f.d.kind == NEDDF_FIELD_NEDDF || f.d.kind == NEDDF_FIELD_NEUS);
    // Sample points straight from the rays (the cone moments in the field prologue): the reverse-mode distance
    // kernel derives them in its prologue and hands them to the colour kernel in the per-point record; NEDDF_RAYS_IN_FIELD=0 keeps the
    // sampling tensors (the A/B partner: tests/test_gpu_parity.py holds both routes bit-identical)
    // (read per call, a few times per frame: the bit-identity test flips it inside one process)
    const char *rif = rays ? getenv("NEDDF_RAYS_IN_FIELD") : nullptr;
    const bool use_rays = rays && (!rif || atoi(rif) != 0) && reverse && f.d.kind == NEDDF_FIELD_NEDDF;
    if (rays && !use_rays) sample_now();
    // Points per launch of the field kernels.  Every launch boundary drains the persistent grid (workgroups finish up to one tile
    // apart) and refills it: at 2^21 points a 65 536-ray x 128-sample call was four launch pairs, at 2^23 it is one -- fp32 +0.8 %,
    // split fp16 +0.7 %, bf16 +2.8 % (profiles/r04_launch_size.txt).  The hand-off buffers grow with it (1 088 B per point
    // eval-minimal: 9.1 GB at 2^23 of the 288 GB); NEDDF_FIELD_CHUNK_LOG2 overrides.
    static const int chunk_log2 = [] { const char *e = getenv("NEDDF_FIELD_CHUNK_LOG2"); int v = e ? atoi(e) : 23; return v < 16 ? 16 : (v > 25 ? 25 : v); }();
    const int64_t chunk_cap = full ? (1 << 19) : ((int64_t)1 << chunk_log2);
    int64_t chunk = N < chunk_cap ? N : chunk_cap;
    // The hand-off of one launch (features + per-point record: 1 088 B per point eval-minimal at width 256, 9.1 GB at 2^23 points,
    // twice that at engine width 512) is bounded by what the DEVICE has free, not by a constant: several contexts or ranks on one
    // device, or a part with less HBM, get smaller launches (-0.8 .. -2.8 % each halving, profiles/r04_launch_size.txt) instead
    // of NEDDF_EHIP -- at most a quarter of the free memory (counting what this context's own blocks give back when they grow).
    {
        const size_t per_point = ((size_t)fr * wid + kPtAux) * sizeof(float);
        // a launch size settled earlier for this row size stands (the probe is a driver call: not on the hot path of every evaluation)
        if (ctx->handoff_row == per_point && ctx->handoff_chunk > 0 && chunk > ctx->handoff_chunk) chunk = ctx->handoff_chunk;
        if (ctx->features.cap < (size_t)chunk * fr * wid * sizeof(float) || ctx->ptaux.cap < (size_t)chunk * kPtAux * sizeof(float)) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                free_b += ctx->features.cap + ctx->ptaux.cap;
                const int64_t asked = chunk;
                while (chunk > (1 << 16) && (size_t)chunk * per_point + (size_t)chunk * per_point / 8 > free_b / 4) chunk >>= 1;
                if (chunk < asked) { ctx->handoff_row = per_point; ctx->handoff_chunk = chunk; }
            }
        }
    }
    // activations' element type and planes: fp32 1024 B, bf16 512 B, split bf16 (two planes) 1024 B per row
    if (int rc = ensure(ctx, ctx->features, (size_t)chunk * fr * wid * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->ptaux, (size_t)chunk * kPtAux * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->sched, 2 * kSchedInts * sizeof(int))) return rc;
    for (int64_t off = 0; off < N; off += chunk) {
        const int64_t n = (N - off < chunk) ? N - off : chunk;
        DdfArgs a = f.ddf;
        fill_enc(a.enc, f);
        a.pos = pos + off * 3; a.dir = dir + off * 3; a.var = var + off * 3; a.n_points = n;
        if (use_rays) {
            a.pos = a.dir = a.var = nullptr;
            a.rays = *rays;
            a.rays.base = rays->base + off;
        }
        a.aux_grad_scale = f.aux_grad_scale;
        a.scratch = (float *)ctx->scratch.p;
        a.features = (color || full) ? (float *)ctx->features.p : nullptr;      // no colour kernel follows: no hand-off
        a.feat_rows = fr;
        a.ptaux = (float *)ctx->ptaux.p;
        a.distance = distance ? distance + off : nullptr;
        a.density = density ? density + off : nullptr;
        a.aux_grad = aux ? aux + off : nullptr;
        a.sched = (int *)ctx->sched.p;
        a.sched_flags = sched_flags();
        HIPCHK(hipMemsetAsync(a.sched, 0, kSchedInts * sizeof(int), s));
        if (reverse) {          // one scalar's gradient: reverse mode, 64 or 32 points per tile (field_kernels.hip ddf_rev_kernel)
            const int pts = ddf_rev_points(dt, wid), wgs = ddf_rev_wgs_per_cu(dt, wid) * ctx->cus;
            const int64_t tiles = (n + pts - 1) / pts;
            const int grid = (int)(tiles < wgs ? tiles : wgs);
            if (int rc = ensure(ctx, ctx->rev_scratch, (size_t)wgs * ddf_rev_scratch_floats_per_wg(a.n_layers, pts, wid) * sizeof(float))) return rc;
            a.rev_scratch = (float *)ctx->rev_scratch.p;
#ifdef NEDDF_STAMP
            static unsigned long long *d_stamps = nullptr;
            const size_t stamp_bytes = ((size_t)kStampBlocks * 8 * kStampSlots * kStampPairTiles + 4 * kStampWgTail) * sizeof(unsigned long long);
            if (!d_stamps) HIPCHK(hipMalloc((void **)&d_stamps, stamp_bytes));
            HIPCHK(hipMemsetAsync(d_stamps, 0, stamp_bytes, s));
            a.stamps = d_stamps;
#endif
            STAGE(ctx, s, NEDDF_STAGE_DDF, launch_ddf_rev(a, grid, s));
#ifdef NEDDF_STAMP
            if (const char *path = getenv("NEDDF_STAMP_FILE")) {        // the LAST launch's stamps (a diagnostic build: synchronising here is fine)
                HIPCHK(hipStreamSynchronize(s));
                std::vector<unsigned long long> h((size_t)kStampBlocks * 8 * kStampSlots * kStampPairTiles + 4 * kStampWgTail);
                HIPCHK(hipMemcpy(h.data(), d_stamps, stamp_bytes, hipMemcpyDeviceToHost));
                if (FILE *fp = fopen(path, "wb")) { fwrite(h.data(), 1, stamp_bytes, fp); fclose(fp); }
            }
#endif
        } else {
            const int64_t tiles = (n + ddf_points_per_tile(dt, wid) - 1) / ddf_points_per_tile(dt, wid);
            STAGE(ctx, s, NEDDF_STAGE_DDF, launch_ddf(a, (int)(tiles < grid_cap_ddf ? tiles : grid_cap_ddf), s));
        }
        if (color || full) {
            ColArgs c = f.col;
            fill_enc(c.enc, f);
            c.pos = a.pos; c.dir = a.dir; c.var = a.var; c.n_points = n;
            c.rays = use_rays ? 1 : 0;
            c.features = a.features; c.feat_rows = fr; c.ptaux = a.ptaux;
            c.distance_range_max = f.distance_range_max;
            c.penalty = full ? penalty + off : nullptr;
            if (color) c.color = color + off * 3;
            else {      // penalty requested without colour: park colour in the (already consumed) head of ptaux? no -- own sink
                if (int rc = ensure(ctx, ctx->arena, (size_t)chunk * 3 * sizeof(float))) return rc;
                c.color = (float *)ctx->arena.p;
            }
            int ppt = col_points_per_tile(full, dt, wid);
            int64_t ctiles = (n + ppt - 1) / ppt;
            c.sched = (int *)ctx->sched.p + kSchedInts;
            c.sched_flags = sched_flags();
            HIPCHK(hipMemsetAsync(c.sched, 0, kSchedInts * sizeof(int), s));
#ifdef NEDDF_STAMP
            static unsigned long long *d_cstamps = nullptr;
            const size_t cstamp_bytes = ((size_t)kStampBlocks * 8 * kStampSlots * kStampPairTiles + 4 * kStampWgTail) * sizeof(unsigned long long);
            if (!d_cstamps) HIPCHK(hipMalloc((void **)&d_cstamps, cstamp_bytes));
            HIPCHK(hipMemsetAsync(d_cstamps, 0, cstamp_bytes, s));
            c.stamps = d_cstamps;
#endif
            STAGE(ctx, s, NEDDF_STAGE_COL, launch_col(c, (int)(ctiles < grid_cap_col ? ctiles : grid_cap_col), full, s));
#ifdef NEDDF_STAMP
            if (const char *path = getenv("NEDDF_STAMP_FILE_COL")) {    // the LAST colour launch's stamps (tools/stamp_timeline_col.py)
                HIPCHK(hipStreamSynchronize(s));
                std::vector<unsigned long long> h((size_t)kStampBlocks * 8 * kStampSlots * kStampPairTiles + 4 * kStampWgTail);
                HIPCHK(hipMemcpy(h.data(), d_cstamps, cstamp_bytes, hipMemcpyDeviceToHost));
                if (FILE *fp = fopen(path, "wb")) { fwrite(h.data(), 1, cstamp_bytes, fp); fclose(fp); }
            }
#endif
        }
    }
    HIPCHK(hipGetLastError());
    return 0;           // (the slot is marked by `mark` on every way out)
}

// ---------------------------------------------------------------------------
extern "C" {

int neddf_abi_version(void) { return NEDDF_ABI_VERSION; }

const char *neddf_last_error(neddf_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err; }

int neddf_create(int device, neddf_ctx **out)
{
    if (!out) return NEDDF_EINVAL;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || device < 0 || device >= count) {
        snprintf(g_err, sizeof(g_err), "neddf_create: no HIP device %d (%s)", device, hipGetErrorString(e));
        return NEDDF_EHIP;
    }
    neddf_ctx *ctx = new neddf_ctx();
    ctx->device = device;
    DeviceGuard guard_(device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cus = prop.multiProcessorCount;
    if (ctx->cus <= 0) ctx->cus = 256;
    void *p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "neddf_create: hipMalloc failed");
        delete ctx;
        return NEDDF_EHIP;
    }
    (void)hipMemset(p, 0, 256);
    ctx->flags.p = p; ctx->flags.cap = 256;
    *out = ctx;
    return 0;
}

void neddf_destroy(neddf_ctx *ctx)
{
    if (!ctx) return;
    DeviceGuard guard_(ctx->device);
    (void)hipDeviceSynchronize();
    neddf_comm_release(ctx);
    for (auto &f : ctx->field) {
        if (f.blob.p) (void)hipFree(f.blob.p);
        if (f.last_use) (void)hipEventDestroy(f.last_use);
    }
    for (DevBuf *b : { &ctx->features, &ctx->ptaux, &ctx->scratch, &ctx->arena, &ctx->flags, &ctx->rflags, &ctx->rev_scratch, &ctx->sched, &ctx->tpack, &ctx->ttmp, &ctx->tamax })
        if (b->p) (void)hipFree(b->base ? b->base : b->p);
    for (auto &e : ctx->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &e : ctx->pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    delete ctx;
}

int neddf_device_cus(neddf_ctx *ctx) { return ctx ? ctx->cus : 0; }

int neddf_debug_check_guards(neddf_ctx *ctx, int64_t *n_bands, int64_t *n_bad_bytes)
{
    if (!ctx) return NEDDF_EINVAL;
    if (n_bands) *n_bands = 0;
    if (n_bad_bytes) *n_bad_bytes = 0;
    if (!guard_mode()) return 0;            // bands exist only under NEDDF_GUARD=1
    DeviceGuard guard_(ctx->device);
    HIPCHK(hipDeviceSynchronize());
    std::vector<GuardBand> bands = ctx->carve_guards;
    for (DevBuf *b : { &ctx->features, &ctx->ptaux, &ctx->scratch, &ctx->arena, &ctx->flags, &ctx->rflags, &ctx->rev_scratch, &ctx->sched, &ctx->tpack,
                       &ctx->ttmp, &ctx->tamax })
        if (b->base) {
            bands.push_back(GuardBand{ b->base, kGuardBytes });
            bands.push_back(GuardBand{ (char *)b->p + b->cap, kGuardBytes });
        }
    // NEDDF_GUARD_SELFTEST=1: overwrite one byte of the last band first -- the test of the probe itself
    if (const char *e = getenv("NEDDF_GUARD_SELFTEST")) if (atoi(e) && !bands.empty()) HIPCHK(hipMemset((void *)bands.back().p, 0, 1));
    std::vector<unsigned char> h(kGuardBytes);
    int64_t bad = 0;
    for (const GuardBand &g : bands) {
        HIPCHK(hipMemcpy(h.data(), g.p, g.bytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < g.bytes; ++i) bad += h[i] != (unsigned char)kGuardByte;
    }
    if (n_bands) *n_bands = (int64_t)bands.size();
    if (n_bad_bytes) *n_bad_bytes = bad;
    return 0;
}

int neddf_set_field(neddf_ctx *ctx, int slot, const neddf_field_desc *desc, const float *const *W, const float *const *B, int n)
{
    if (!ctx || !desc || !W || !B) return NEDDF_EINVAL;
    if (slot < 0 || slot >= NEDDF_NUM_SLOTS) return fail(ctx, NEDDF_EINVAL, "bad slot");
    DeviceGuard guard_(ctx->device);
    // Hidden widths: any (neddf.py:52-66, nerf.py:34-44, neus.py:80-99 take any); the engine runs them zero-padded to the next
    // multiple of 128, up to 512 (beyond that one 32-row tile no longer fits two workgroups into a CU's LDS).
    // Encoding ranks: the position encoding owns one 64-column block of the tile (2 x roundup(3 E, 4) <= 64, and ten low-pass
    // factors in the argument block) -> E <= 10, the reference's default and the largest frequency (2^9) that still resolves fp32
    // scene coordinates; the direction encoding only has to fit the tile row next to it.
    const int wmax = desc->kind == NEDDF_FIELD_NERF ? desc->layer_width
                                                    : (desc->layer_width > desc->col_layer_width ? desc->layer_width : desc->col_layer_width);
    if (desc->layer_width < 1 || (desc->kind != NEDDF_FIELD_NERF && desc->col_layer_width < 1) || wmax > kMaxWidth)
        return fail(ctx, NEDDF_EUNSUPPORTED, "hidden widths must be in [1, 512]");
    if (desc->kind == NEDDF_FIELD_NERF && desc->layer_width < 2) return fail(ctx, NEDDF_EUNSUPPORTED, "NeRF: layer_width must be at least 2 (the colour head is layer_width // 2 wide)");
    if (desc->embed_pos_rank < 1 || desc->embed_pos_rank > 10 || desc->embed_dir_rank < 1)
        return fail(ctx, NEDDF_EUNSUPPORTED, "embed_pos_rank must be in [1,10] and embed_dir_rank >= 1");
    if (enc_columns(*desc) > kMaxWidth)
        return fail(ctx, NEDDF_EUNSUPPORTED, "embed_dir_rank: the encodings do not fit one tile row of the widest engine (512 columns)");
    if (desc->n_skips < 0 || desc->n_skips > 8) return fail(ctx, NEDDF_EINVAL, "bad n_skips");
    if (desc->activation < 0 || desc->activation > 2 || desc->density_activation < 0 || desc->density_activation > 2)
        return fail(ctx, NEDDF_EINVAL, "bad activation id");
    if (desc->weight_dtype < NEDDF_DTYPE_F32 || desc->weight_dtype > NEDDF_DTYPE_F16_SPLIT) return fail(ctx, NEDDF_EINVAL, "bad weight_dtype");
    Field &f = ctx->field[slot];
    // the packed weights of this slot are about to be overwritten: wait for the LAST launch that reads them (an event recorded
    // on its stream by field_forward), not for the whole device -- other streams and other slots keep running
    if (f.last_use) HIPCHK(hipEventSynchronize(f.last_use));
    f.valid = false;
    f.d = *desc;
    f.aux_grad_scale = 1.1f; f.distance_range_max = 2.0f;
    for (int i = 0; i < 10; ++i) f.lowpass[i] = 1.0f;
    int rc = desc->kind == NEDDF_FIELD_NEDDF ? build_neddf(ctx, f, W, B, n)
           : desc->kind == NEDDF_FIELD_NERF ? build_nerf(ctx, f, W, B, n)
           : desc->kind == NEDDF_FIELD_NEUS ? build_neus(ctx, f, W, B, n)
           : fail(ctx, NEDDF_EINVAL, "bad field kind");
    if (rc) return rc;
    f.valid = true;
    return 0;
}

int neddf_set_iter(neddf_ctx *ctx, int slot, float aux_grad_scale, float distance_range_max, const float *h_lowpass)
{
    if (!ctx) return NEDDF_EINVAL;
    if (slot < 0 || slot >= NEDDF_NUM_SLOTS || !ctx->field[slot].valid) return fail(ctx, NEDDF_ENOFIELD, "no field in slot");
    Field &f = ctx->field[slot];
    f.aux_grad_scale = aux_grad_scale;
    f.distance_range_max = distance_range_max;
    for (int i = 0; i < f.d.embed_pos_rank; ++i) f.lowpass[i] = h_lowpass ? h_lowpass[i] : 1.0f;
    return 0;
}

static CameraArg cam_arg(const neddf_camera *c)
{
    CameraArg a;
    memcpy(a.R, c->R, sizeof(a.R)); memcpy(a.T, c->T, sizeof(a.T)); memcpy(a.calib, c->calib, sizeof(a.calib));
    return a;
}

int neddf_raygen(neddf_ctx *ctx, const void *uv, int uv_type, int64_t n, const neddf_camera *cam, float *rd, float *ro, void *stream)
{
    if (!ctx || !uv || !cam || !rd || !ro) return NEDDF_EINVAL;
    if (uv_type < 0 || uv_type > 3) return fail(ctx, NEDDF_EINVAL, "bad uv_type");
    DeviceGuard guard_(ctx->device);
    STAGE(ctx, (hipStream_t)stream, NEDDF_STAGE_RAYGEN, launch_raygen(uv, uv_type, n, cam_arg(cam), rd, ro, (hipStream_t)stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_sample_coarse(neddf_ctx *ctx, const float *U, int64_t n, int S1, float near_, float far_, float *dists, void *stream)
{
    if (!ctx || !U || !dists || S1 < 2) return NEDDF_EINVAL;
    DeviceGuard guard_(ctx->device);
    STAGE(ctx, (hipStream_t)stream, NEDDF_STAGE_SAMPLE_COARSE, launch_sample_coarse(U, n, S1, near_, far_, dists, (hipStream_t)stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_sampling(neddf_ctx *ctx, const float *rd, const float *ro, const float *dists, int64_t n, int S, double radius,
                   float *pos, float *dir, float *var, void *stream)
{
    if (!ctx || !rd || !ro || !dists || !pos || !dir || !var) return NEDDF_EINVAL;
    if (radius >= 0.0 && S < 2) return fail(ctx, NEDDF_EINVAL, "cone sampling needs at least 2 samples");
    DeviceGuard guard_(ctx->device);
    STAGE(ctx, (hipStream_t)stream, NEDDF_STAGE_SAMPLING, launch_sampling(rd, ro, nullptr, dists, n, S, radius, pos, dir, var, (hipStream_t)stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_sampling_view(neddf_ctx *ctx, const float *rd, const float *ro, const float *view, const float *dists, int64_t n, int S,
                        double radius, float *pos, float *dir, float *var, void *stream)
{
    if (!ctx || !rd || !ro || !view || !dists || !pos || !dir || !var) return NEDDF_EINVAL;
    if (radius >= 0.0 && S < 2) return fail(ctx, NEDDF_EINVAL, "cone sampling needs at least 2 samples");
    DeviceGuard guard_(ctx->device);
    STAGE(ctx, (hipStream_t)stream, NEDDF_STAGE_SAMPLING, launch_sampling(rd, ro, view, dists, n, S, radius, pos, dir, var, (hipStream_t)stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_rays_to_ndc(neddf_ctx *ctx, const float *rd, const float *ro, int64_t n, int width, int height, float fx, float fy,
                      float near_plane, float *nd, float *no, void *stream)
{
    if (!ctx) return NEDDF_EINVAL;
    if (n <= 0) return 0;
    if (!rd || !ro || !nd || !no || width < 1 || height < 1 || !(fx > 0.f) || !(fy > 0.f)) return fail(ctx, NEDDF_EINVAL, "rays_to_ndc: bad argument");
    DeviceGuard guard_(ctx->device);
    STAGE(ctx, (hipStream_t)stream, NEDDF_STAGE_NDC, launch_ndc(rd, ro, n, (float)width, (float)height, fx, fy, near_plane, nd, no, (hipStream_t)stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_field_forward(neddf_ctx *ctx, int slot, const float *pos, const float *dir, const float *var, int64_t N, int out_mode,
                        float *distance, float *density, float *color, float *penalty, float *aux, void *stream)
{
    if (!ctx) return NEDDF_EINVAL;
    if (N <= 0) return 0;
    if (!pos || !dir || !var) return NEDDF_EINVAL;
    DeviceGuard guard_(ctx->device);
    return field_forward(ctx, slot, pos, dir, var, N, out_mode, distance, density, color, penalty, aux, (hipStream_t)stream);
}

int neddf_composite(neddf_ctx *ctx, const float *dists, const float *dens, const float *col, int64_t n, int S, float max_dist,
                    float *w, float *depth, float *color, float *trans, int *nan_flag, void *stream)
{
    if (!ctx || !dists || !dens || !col || !depth || !color || !trans || S < 2) return NEDDF_EINVAL;
    DeviceGuard guard_(ctx->device);
    STAGE(ctx, (hipStream_t)stream, NEDDF_STAGE_COMPOSITE, launch_composite(dists, dens, col, n, S, max_dist, w, depth, color, trans, nan_flag, (hipStream_t)stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_integrate_penalty(neddf_ctx *ctx, const float *dists, const float *pen, int64_t n, int S, float *out, void *stream)
{
    if (!ctx || !dists || !pen || !out) return NEDDF_EINVAL;
    DeviceGuard guard_(ctx->device);
    STAGE(ctx, (hipStream_t)stream, NEDDF_STAGE_PENALTY, launch_integrate_penalty(dists, pen, n, S, out, (hipStream_t)stream));
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_importance_resample(neddf_ctx *ctx, const float *dists, float *weights, const float *U, int64_t n_rays, int n, int nf,
                              int cat, float *out, int64_t *ids, void *stream)
{
    if (!ctx || !dists || !weights || !U || !out || n < 2 || nf < 1) return NEDDF_EINVAL;
    if (n + nf > 8192) return fail(ctx, NEDDF_EUNSUPPORTED, "importance_resample: n + n_fine must be <= 8192");
    DeviceGuard guard_(ctx->device);
    STAGE(ctx, (hipStream_t)stream, NEDDF_STAGE_RESAMPLE, launch_resample(dists, weights, U, n_rays, n, nf, cat, out, ids, (int *)ctx->flags.p + 1, 0, 0, (hipStream_t)stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// carve helper for the render arena
// NEDDF_GUARD=1: a carve ends exactly at its last element and is followed by a poisoned band (capi_internal.h guard_mode)
struct Carver {
    char *p;
    neddf_ctx *ctx = nullptr;
    hipStream_t s = nullptr;
    size_t off = 0;
    float *take(size_t n_floats)
    {
        float *r = (float *)(p + off);
        if (guard_mode() && ctx) {
            const size_t used = (n_floats * sizeof(float) + 15) & ~(size_t)15;
            (void)hipMemsetAsync(p + off + used, kGuardByte, kCarveGuardBytes, s);
            ctx->carve_guards.push_back(GuardBand{ p + off + used, kCarveGuardBytes });
            off += (used + kCarveGuardBytes + 255) & ~(size_t)255;
            return r;
        }
        off += (n_floats * sizeof(float) + 255) & ~(size_t)255;
        return r;
    }
};

static size_t carve_bytes(size_t n_floats)
{
    if (guard_mode()) return (((n_floats * sizeof(float) + 15) & ~(size_t)15) + kCarveGuardBytes + 255) & ~(size_t)255;
    return (n_floats * sizeof(float) + 255) & ~(size_t)255;
}

static int render_pass(neddf_ctx *ctx, int slot, const float *rd, const float *ro, const float *view, const float *dists, int64_t B, int S,
                       const neddf_render_params *rp, float *pos, float *dir, float *var, float *dens, float *col, float *pen,
                       float *w_out, float *depth, float *color, float *trans, float *pen_out, int *nan_flag, hipStream_t s)
{
    RaySrc rays;
    rays.rd = rd; rays.ro = ro; rays.view = view; rays.dists = dists; rays.S = S;
    rays.cone = rp->cone_sampling ? 1 : 0;
    rays.radius = rp->cone_sampling ? rp->ray_radius : -1.0;
    rays.r2 = (float)(rp->ray_radius * rp->ray_radius);          // launch_sampling's own (float)(radius * radius)
    const bool want_pen = pen_out && ctx->field[slot].d.kind == NEDDF_FIELD_NEDDF;
    // a pass whose pixels nobody asked for (the coarse pass of render_image: only its resampling weights are consumed) needs the
    // densities only -- the colour trunk is skipped (the reference evaluates and discards it)
    const bool want_col = depth || color || trans || want_pen;
    int rc = field_forward(ctx, slot, pos, dir, var, B * S, want_pen ? NEDDF_OUT_FULL : NEDDF_OUT_MINIMAL, nullptr, dens, want_col ? col : nullptr,
                           want_pen ? pen : nullptr, nullptr, s, &rays);
    if (rc) return rc;
    STAGE(ctx, s, NEDDF_STAGE_COMPOSITE, launch_composite(dists, dens, want_col ? col : nullptr, B, S, rp->max_dist, w_out, depth, color, trans, nan_flag, s));
    if (want_pen) STAGE(ctx, s, NEDDF_STAGE_PENALTY, launch_integrate_penalty(dists, pen, B, S, pen_out, s));
    return 0;
}

int neddf_render_rays(neddf_ctx *ctx, const void *uv, int uv_type, int64_t B, const neddf_camera *cam, const neddf_render_params *rp,
                      const float *Uc, const float *Uf, const neddf_render_outputs *out, void *stream)
{
    if (!ctx) return NEDDF_EINVAL;
    if (B <= 0) return 0;           // empty batch: nothing to do (pointers of empty tensors may be NULL)
    if (!uv || !cam || !rp || !Uc || !Uf || !out) return NEDDF_EINVAL;
    DeviceGuard guard_(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    const int Sc1 = rp->sample_coarse + 1, Sf1 = rp->sample_fine + 1, S2 = Sc1 + Sf1;
    if (!ctx->field[NEDDF_SLOT_COARSE].valid || !ctx->field[NEDDF_SLOT_FINE].valid) return fail(ctx, NEDDF_ENOFIELD, "render_rays needs coarse and fine fields");
    size_t need = 2 * carve_bytes(B * 3) + carve_bytes(B * Sc1) + carve_bytes(B * S2) + 3 * carve_bytes(B * S2 * 3) +
                  2 * carve_bytes(B * S2) + carve_bytes(B * S2 * 3) + carve_bytes(B * (Sc1 - 1)) + carve_bytes(B * (S2 - 1)) +
                  8 * carve_bytes(B * 3);
    // the arena is also used by field_forward as a colour sink only when colour is not requested; never the case here
    if (int rc = ensure(ctx, ctx->arena, need)) return rc;
    ctx->carve_guards.clear();
    Carver cv{ (char *)ctx->arena.p, ctx, s };
    float *rd = cv.take(B * 3), *ro = cv.take(B * 3);
    float *dc = out->dists_coarse ? out->dists_coarse : cv.take(B * Sc1);
    float *df = out->dists_fine ? out->dists_fine : cv.take(B * S2);
    float *pos = cv.take(B * S2 * 3), *dir = cv.take(B * S2 * 3), *var = cv.take(B * S2 * 3);
    float *dens = cv.take(B * S2), *pen = cv.take(B * S2), *col = cv.take(B * S2 * 3);
    float *wc = out->weight_coarse ? out->weight_coarse : cv.take(B * (Sc1 - 1));
    // coarse pixels only when the caller asked for one of them (render_rays); render_image does not
    const bool coarse_px = out->depth_coarse || out->color_coarse || out->transmittance_coarse || out->fields_penalty_coarse;
    float *depth_c = out->depth_coarse ? out->depth_coarse : (coarse_px ? cv.take(B) : nullptr);
    float *color_c = out->color_coarse ? out->color_coarse : (coarse_px ? cv.take(B * 3) : nullptr);
    float *trans_c = out->transmittance_coarse ? out->transmittance_coarse : (coarse_px ? cv.take(B) : nullptr);
    float *depth = out->depth ? out->depth : cv.take(B);
    float *color = out->color ? out->color : cv.take(B * 3);
    float *trans = out->transmittance ? out->transmittance : cv.take(B);
    int *flags = (int *)ctx->flags.p;
    int *nan_flag = out->nan_flag ? out->nan_flag : flags;

    STAGE(ctx, s, NEDDF_STAGE_RAYGEN, launch_raygen(uv, uv_type, B, cam_arg(cam), rd, ro, s));
    const float *view = nullptr;
    if (rp->ndc_rays) {         // positions follow the NDC ray, the field keeps the world-space viewing direction
        float *nd = cv.take(B * 3), *no = cv.take(B * 3);
        STAGE(ctx, s, NEDDF_STAGE_NDC, launch_ndc(rd, ro, B, (float)rp->ndc_width, (float)rp->ndc_height, cam->calib[0], cam->calib[1], rp->ndc_near, nd, no, s));
        view = rd; rd = nd; ro = no;
    }
    STAGE(ctx, s, NEDDF_STAGE_SAMPLE_COARSE, launch_sample_coarse(Uc, B, Sc1, rp->dist_near, rp->dist_far, dc, s));
    int rc = render_pass(ctx, NEDDF_SLOT_COARSE, rd, ro, view, dc, B, Sc1, rp, pos, dir, var, dens, col, pen, wc, depth_c, color_c,
                         trans_c, out->fields_penalty_coarse, nan_flag, s);
    if (rc) return rc;
    // the reference takes the NaN-fallback decision of sample_pdf per render_rays call, i.e. per `chunk` rays of render_image
    const int64_t group = rp->nan_group > 0 ? rp->nan_group : B;
    const int64_t goff = rp->nan_group > 0 && rp->nan_group_offset > 0 ? rp->nan_group_offset % group : 0;
    if (int rc = ensure(ctx, ctx->rflags, (size_t)((B + goff + group - 1) / group) * sizeof(int))) return rc;
    STAGE(ctx, s, NEDDF_STAGE_RESAMPLE, launch_resample(dc, wc, Uf, B, Sc1, Sf1, 1, df, nullptr, (int *)ctx->rflags.p, group, goff, s));
    rc = render_pass(ctx, NEDDF_SLOT_FINE, rd, ro, view, df, B, S2, rp, pos, dir, var, dens, col, pen, out->weight, depth, color, trans,
                     out->fields_penalty, nan_flag, s);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_render_rays_single(neddf_ctx *ctx, int slot, const void *uv, int uv_type, int64_t B, const neddf_camera *cam,
                             const neddf_render_params *rp, int S1, const float *U, const neddf_render_outputs *out, void *stream)
{
    if (!ctx) return NEDDF_EINVAL;
    if (B <= 0) return 0;
    if (!uv || !cam || !rp || !U || !out || S1 < 2) return NEDDF_EINVAL;
    DeviceGuard guard_(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    if (slot < 0 || slot >= NEDDF_NUM_SLOTS || !ctx->field[slot].valid) return fail(ctx, NEDDF_ENOFIELD, "no field in slot");
    size_t need = 2 * carve_bytes(B * 3) + carve_bytes(B * S1) + 3 * carve_bytes(B * S1 * 3) + 2 * carve_bytes(B * S1) +
                  carve_bytes(B * S1 * 3) + 5 * carve_bytes(B * 3);
    if (int rc = ensure(ctx, ctx->arena, need)) return rc;
    ctx->carve_guards.clear();
    Carver cv{ (char *)ctx->arena.p, ctx, s };
    float *rd = cv.take(B * 3), *ro = cv.take(B * 3);
    float *dc = out->dists_fine ? out->dists_fine : cv.take(B * S1);
    float *pos = cv.take(B * S1 * 3), *dir = cv.take(B * S1 * 3), *var = cv.take(B * S1 * 3);
    float *dens = cv.take(B * S1), *pen = cv.take(B * S1), *col = cv.take(B * S1 * 3);
    float *depth = out->depth ? out->depth : cv.take(B);
    float *color = out->color ? out->color : cv.take(B * 3);
    float *trans = out->transmittance ? out->transmittance : cv.take(B);
    int *nan_flag = out->nan_flag ? out->nan_flag : (int *)ctx->flags.p;
    STAGE(ctx, s, NEDDF_STAGE_RAYGEN, launch_raygen(uv, uv_type, B, cam_arg(cam), rd, ro, s));
    const float *view = nullptr;
    if (rp->ndc_rays) {
        float *nd = cv.take(B * 3), *no = cv.take(B * 3);
        STAGE(ctx, s, NEDDF_STAGE_NDC, launch_ndc(rd, ro, B, (float)rp->ndc_width, (float)rp->ndc_height, cam->calib[0], cam->calib[1], rp->ndc_near, nd, no, s));
        view = rd; rd = nd; ro = no;
    }
    STAGE(ctx, s, NEDDF_STAGE_SAMPLE_COARSE, launch_sample_coarse(U, B, S1, rp->dist_near, rp->dist_far, dc, s));
    int rc = render_pass(ctx, slot, rd, ro, view, dc, B, S1, rp, pos, dir, var, dens, col, pen, out->weight, depth, color, trans,
                         out->fields_penalty, nan_flag, s);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_op_activation(neddf_ctx *ctx, int op, const float *x, const float *J, int64_t N, int C, float *y, float *G, void *stream)
{
    if (!ctx || !x || !y || op < 0 || op > 4 || (J && !G)) return NEDDF_EINVAL;
    if (!J && op > 2) return fail(ctx, NEDDF_EINVAL, "softplus/sigmoid exist only as (value, Jacobian) ops in the reference");
    if (J && op == 4 && C != 1) return fail(ctx, NEDDF_EUNSUPPORTED, "SigmoidGradFunction is defined for one channel (sigmoid.py:42)");
    launch_op_activation(op, x, J, N, C, y, G, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_op_positional_encoding(neddf_ctx *ctx, const float *x, const float *J, const float *scale, int64_t N, int E, float *y,
                                 float *G, void *stream)
{
    if (!ctx || !x || !y || E < 1 || E > 30 || (J && !G)) return NEDDF_EINVAL;
    launch_op_pe(x, J, scale, N, E, y, G, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_op_pe_weights(neddf_ctx *ctx, const float *var, int64_t N, int E, float *w, void *stream)
{
    if (!ctx || !var || !w || E < 1 || E > 30) return NEDDF_EINVAL;
    launch_op_pe_weights(var, N, E, w, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_op_linear_grad(neddf_ctx *ctx, const float *x, const float *J, const float *W, const float *b, int64_t N, int Cin,
                         int Cout, float *y, float *G, void *stream)
{
    if (!ctx || !x || !J || !W || !y || !G) return NEDDF_EINVAL;
    if (Cin < 1 || Cout < 1 || Cin > 65536 || Cout > 65536) return fail(ctx, NEDDF_EINVAL, "op_linear_grad: bad Cin / Cout");
    if (N <= 0) return 0;
    DeviceGuard guard_(ctx->device);
    // any (Cin, Cout) (with_grad/linear.py:87-133 takes any): the product is cut into K blocks of <= 256 input columns (accumulated in
    // the outputs) and N blocks of 256 / 128 output columns (a last partial block runs zero-padded and stores only its valid columns)
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = (N + 31) / 32;
    const int grid = (int)(tiles < ctx->cus ? tiles : ctx->cus);
    for (int n0 = 0; n0 < Cout; n0 += 256) {
        const int nvalid = Cout - n0 < 256 ? Cout - n0 : 256, nblk = nvalid > 128 ? 256 : 128;
        for (int k0 = 0; k0 < Cin; k0 += 256) {
            const int kc = Cin - k0 < 256 ? Cin - k0 : 256;
            std::vector<float> blob;
            std::vector<int> km;
            for (int k = 0; k < roundup(kc, 8); ++k) km.push_back(k < kc ? k0 + k : -1);
            // the block's weights [kc, nblk]: columns n0 .. n0 + nvalid of W (row stride Cout), zero beyond
            std::vector<float> wblk((size_t)Cin * nblk, 0.f);              // (W and b are HOST arrays, include/neddf_hip.h)
            for (int k = k0; k < k0 + kc; ++k)
                for (int n = 0; n < nvalid; ++n) wblk[(size_t)k * nblk + n] = W[(size_t)k * Cout + n0 + n];
            Src src{ wblk.data(), Cin, nblk, false };
            size_t o_w = pack_layer(blob, src, km, nblk, 0, nullptr);
            std::vector<float> bv(nblk, 0.f);
            if (b && k0 == 0) memcpy(bv.data(), b + n0, (size_t)nvalid * sizeof(float));
            size_t o_b = put(blob, bv.data(), nblk);
            if (int rc = ensure(ctx, ctx->features, blob.size() * sizeof(float))) return rc;
            HIPCHK(hipStreamSynchronize(s));                  // the previous block's launch is done with the buffer; blob is a host temporary
            HIPCHK(hipMemcpyAsync(ctx->features.p, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
            const float *base = (const float *)ctx->features.p;
            launch_linear_grad(x + k0, J + k0, N, kc, Cin, nblk, (int)km.size() / 8, base + o_w, base + o_b, y + n0, G + n0, Cout, nvalid, k0 > 0, grid, s);
        }
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_set_timing(neddf_ctx *ctx, int enable)
{
    if (!ctx) return NEDDF_EINVAL;
    ctx->timing = enable != 0;
    return 0;
}

static int drain_events(neddf_ctx *ctx, float *ms, int *launches)
{
    for (auto &e : ctx->events) {
        HIPCHK(hipEventSynchronize(e.b));
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, e.a, e.b));
        ms[e.which] += t;
        launches[e.which] += 1;
        ctx->pool.push_back(e);
    }
    ctx->events.clear();
    return 0;
}

int neddf_get_timings(neddf_ctx *ctx, float *ms, int n)
{
    if (!ctx || !ms || n < 6) return NEDDF_EINVAL;
    DeviceGuard guard_(ctx->device);
    float t[NEDDF_STAGE_COUNT] = { 0 };
    int c[NEDDF_STAGE_COUNT] = { 0 };
    for (int i = 0; i < n; ++i) ms[i] = 0.f;
    if (int rc = drain_events(ctx, t, c)) return rc;
    for (int k = 0; k < 3; ++k) { ms[k] = t[k]; ms[3 + k] = (float)c[k]; }      // distance trunk, colour trunk, NeRF field
    return 0;
}

int neddf_get_stage_timings(neddf_ctx *ctx, float *ms, int *launches, int n_stages)
{
    if (!ctx || !ms || !launches || n_stages < NEDDF_STAGE_COUNT) return NEDDF_EINVAL;
    DeviceGuard guard_(ctx->device);
    for (int i = 0; i < n_stages; ++i) { ms[i] = 0.f; launches[i] = 0; }
    return drain_events(ctx, ms, launches);
}

}  // extern "C"
