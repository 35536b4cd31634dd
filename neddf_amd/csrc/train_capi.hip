// train_capi.hip -- C ABI of the training step (include/neddf_hip.h, "training" section):
// NeDDF.forward with everything its hand-written backward passes need kept in a caller-owned
// workspace, the reverse pass producing parameter gradients in the reference's own
// state-dict layout, and the backward of integrate_volume_render.
//
// The parameters stay where the optimiser updates them (torch tensors on the device, reference
// layout); each call re-packs them into MFMA fragment order on the device (launch_pack), so
// there is no host copy of the weights on the training path.
#include "capi_internal.h"
#include "train_kernels.h"

#include <stdlib.h>

namespace {

struct Plan {
    int E, Ed, Cpe, Cdir, Ca, ldxa, n_trunk, n_col, i_ddf, i_aux, i_cout;
    int WH;                     // hidden width the kernels see: 256, or 512 (NeDDF, per-layer route in 256 x 256 blocks)
    int64_t N, R;
    // workspace offsets (floats)
    size_t o_pes, o_xa, o_pt, o_cr, o_z[kMaxLayers], o_h[kMaxLayers], o_zc[kMaxLayers], o_hc[kMaxLayers], total;
};

constexpr int kLdPe = 64, kLdDir = 64, kLdNarrow = 4;      // row strides of the encoding matrices: 6 x rank columns each (rank <= 10)

// The training kernels (train_kernels.hip) are built for the hidden width of every shipped configuration, 256: tiles, fused layer
// chains and weight-gradient products are 256 columns wide.  Narrower networks reach these entry points zero-padded to 256
// (neddf_amd/network.py _train_tensors: exact).  Round 4: a field up to hidden width 512 (the reference trains whatever it
// constructs, neddf.py:52-66, nerf.py:34-44, neus.py:30-41) arrives padded to 512 and takes the PER-LAYER route with every product cut into 256 x 256 blocks
// (K blocks accumulate in the output, the activation / its backward runs on the last one): the same kernels, correct at any
// multiple of 256, without the fused chains' speed.  Anything wider is refused loudly rather than computed wrongly.

// NEDDF_TRAIN_WIDE_FUSED=0: fields that train padded to 512 columns (hidden widths 257 .. 512) keep round 4's per-layer route with every
// product cut into 256 x 256 blocks instead of the 512-wide fused chains (round 5: mlp_forward_kernel / mlp_backward_kernel / mlp_backward_split_kernel
// over the width on 32-row tiles, point-major [R, 512] matrices, one weight-gradient launch per 256 x 256 block).  Measured (NeDDF
// 8 + 4 layers, 265 k points, forward + backward): 131 ms fused against 162 ms blocked -- after the first version, whose weight
// gradients went through the job-parallel launch, had taken 373 ms (see wide_dw_jobs); split fp16: 71 against 124 ms.
bool wide_fused()
{
    static const bool on = [] { const char *e = getenv("NEDDF_TRAIN_WIDE_FUSED"); return !(e && atoi(e) == 0); }();
    return on;
}

// Weight gradients: the 256-wide routes (fp32 and split fp16) take the job-parallel launch (dw_jobs_kernel / dw_split_jobs_kernel: one launch
// per pass; fp32 39.1 against 39.7 ms per step, split fp16 +1 %), the 512-wide fused routes one launch per 256 x 256 block (the job-parallel
// launch measured 6x slower with their 46 blocks).  The A/B switches of rounds 4-5 left with round 6's pruning (docs/lab_notebook.md R5.18-R5.19).
static constexpr bool dw_jobs_256() { return true; }
static constexpr bool wide_dw_jobs() { return false; }
static constexpr bool split_dw_jobs() { return true; }

// NEDDF_TRAIN_SPLIT_FUSED=0: the split-fp16 policy's backward pass as one GEMM kernel per layer on row-major matrices (rounds 1-4)
// instead of the fused input-gradient chains on point-major ones (round 5: mlp_backward_split_kernel)
bool split_fused()
{
    static const bool on = [] { const char *e = getenv("NEDDF_TRAIN_SPLIT_FUSED"); return !(e && atoi(e) == 0); }();
    return on;
}

int train_supported(neddf_ctx *ctx, const Field &f)
{
    if (f.d.layer_width % kWidth != 0 || f.d.layer_width < kWidth || f.d.layer_width > 2 * kWidth ||
        (f.d.kind != NEDDF_FIELD_NERF && f.d.col_layer_width != f.d.layer_width))
        return fail(ctx, NEDDF_EUNSUPPORTED, "the training kernels take hidden widths 256 and 512 (NeDDF / NeuS: both trunks at the same one): pass other widths zero-padded (neddf_amd does); anything above 512 cannot train (rendering supports 1..512)");
    if (6 * f.d.embed_dir_rank > kLdDir) return fail(ctx, NEDDF_EUNSUPPORTED, "the training kernels take embed_dir_rank <= 10");
    return 0;
}

int make_plan(neddf_ctx *ctx, int slot, int64_t N, int n_tensors, Plan &p)
{
    if (slot < 0 || slot >= NEDDF_NUM_SLOTS || !ctx->field[slot].valid) return fail(ctx, NEDDF_ENOFIELD, "no field in slot");
    const Field &f = ctx->field[slot];
    if (f.d.kind != NEDDF_FIELD_NEDDF) return fail(ctx, NEDDF_EUNSUPPORTED, "unknown field kind");
    if (int rc = train_supported(ctx, f)) return rc;
    p.E = f.d.embed_pos_rank; p.Ed = f.d.embed_dir_rank;
    p.Cpe = 6 * p.E; p.Cdir = 6 * p.Ed; p.Ca = p.Cpe + p.Cdir + 3; p.ldxa = roundup(p.Ca, 8);
    p.n_trunk = f.d.layer_count - 1; p.n_col = f.d.col_layer_count - 1;
    p.i_ddf = p.n_trunk + p.n_col; p.i_aux = p.i_ddf + 1; p.i_cout = p.i_ddf + 2;
    if (n_tensors >= 0 && n_tensors != p.n_trunk + p.n_col + 3) return fail(ctx, NEDDF_EINVAL, "NeDDF: wrong tensor count");
    p.N = N; p.R = 4 * N;
    p.WH = f.d.layer_width;
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
    p.o_pes = take((size_t)p.R * kLdPe);
    p.o_xa = take((size_t)p.R * p.ldxa);
    p.o_pt = take((size_t)N * kTrainPt);
    p.o_cr = take((size_t)p.R * kLdNarrow);
    for (int l = 0; l < p.n_trunk; ++l) { p.o_z[l] = take((size_t)p.R * p.WH); p.o_h[l] = take((size_t)p.R * p.WH); }
    for (int l = 0; l < p.n_col; ++l) { p.o_zc[l] = take((size_t)p.R * p.WH); p.o_hc[l] = take((size_t)p.R * p.WH); }
    p.total = o;
    return 0;
}

void point_args(TrainPointArgs &a, const Field &f, const Plan &p, float *ws)
{
    a = TrainPointArgs{};
    a.N = p.N;
    fill_enc(a.enc, f);
    a.density_activation = f.d.density_activation;
    a.d_near = f.d.d_near;
    a.aux_grad_scale = f.aux_grad_scale;
    a.distance_range_max = f.distance_range_max;
    for (int k = 0; k < 6; ++k) { a.penalty_weight[k] = f.d.penalty_weight[k]; a.penalty_has[k] = f.d.penalty_has[k]; }
    a.ldh = kLdNarrow; a.ldpe = kLdPe; a.ldd = kLdDir; a.ldxa = p.ldxa; a.ldc = kLdNarrow;
    a.XA = ws + p.o_xa; a.PT = ws + p.o_pt; a.CR = ws + p.o_cr;
}

constexpr size_t kPackFloats = (size_t)kWidth * kWidth;        // one packed 256 x 256 segment

// One device scalar per gradient matrix of a backward pass: its producer leaves max |dZ| there, the split-fp16 GEMMs that consume
// it scale their operand by the matching power of two (train_kernels.hip operand_scale).  NULL slots under the fp32 policy.
constexpr int kAmaxSlots = 4 * kMaxLayers + 8;
struct AmaxSlots {
    float *base = nullptr;
    float *dw_tmp = nullptr;        // [256, 256] scratch of the scaled weight-gradient products (launch_dw)
    int next = 0;
    float *take() { return base ? base + next++ : nullptr; }
};
int amax_begin(neddf_ctx *ctx, int split, AmaxSlots &m, hipStream_t s)
{
    m = AmaxSlots{};
    if (!split) return 0;
    if (int rc = ensure(ctx, ctx->tamax, (kAmaxSlots + (size_t)kMaxDwJobs * kPackFloats) * sizeof(float))) return rc;      // (dw_tmp: one [256, 256] scratch per job of a list)
    HIPCHK(hipMemsetAsync(ctx->tamax.p, 0, kAmaxSlots * sizeof(float), s));
    m.base = (float *)ctx->tamax.p;
    m.dw_tmp = m.base + kAmaxSlots;
    return 0;
}

// ---- plain NeRF field (nerf.py:107-165): value rows only, nn.Linear weights [out, in] -------------------------------
struct NerfPlan {
    int E, Ed, Cpe, Cdir, n, i_dens, i_c0, i_c1, in_c0;
    int WH, HC;                 // hidden width the kernels see (256 or 512) and the colour head's hidden width WH / 2 (one 256-column block)
    int64_t N;
    size_t o_pe, o_ed, o_zd, o_zc, o_hc, o_z[kMaxLayers], o_h[kMaxLayers], total;
};

int make_nerf_plan(neddf_ctx *ctx, int slot, int64_t N, int n_tensors, NerfPlan &p)
{
    const Field &f = ctx->field[slot];
    if (int rc = train_supported(ctx, f)) return rc;
    p.E = f.d.embed_pos_rank; p.Ed = f.d.embed_dir_rank;
    p.Cpe = 6 * p.E; p.Cdir = 6 * p.Ed; p.n = f.d.layer_count;
    p.WH = f.d.layer_width; p.HC = p.WH / 2;
    p.i_dens = p.n; p.i_c0 = p.n + 1; p.i_c1 = p.n + 2; p.in_c0 = p.WH + p.Cdir;
    if (n_tensors >= 0 && n_tensors != p.n + 3) return fail(ctx, NEDDF_EINVAL, "NeRF: wrong tensor count");
    p.N = N;
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
    p.o_pe = take((size_t)N * kLdPe);
    p.o_ed = take((size_t)N * kLdDir);
    p.o_zd = take((size_t)N * kLdNarrow);
    p.o_zc = take((size_t)N * kWidth);
    p.o_hc = take((size_t)N * kWidth);
    for (int l = 0; l < p.n; ++l) { p.o_z[l] = take((size_t)N * p.WH); p.o_h[l] = take((size_t)N * p.WH); }
    p.total = o;
    return 0;
}

// Every product of the NeRF route is cut into 256 x 256 blocks like the NeDDF per-layer route (one block each at hidden width 256,
// K blocks accumulating in the output and the activation with the last one at 512); nn.Linear weights [out, in] are read through strides.
int nerf_forward(neddf_ctx *ctx, int slot, const float *const *W, const float *const *B, int n_tensors, const float *pos, const float *dir,
                 const float *var, int64_t N, float *ws, float *density, float *color, hipStream_t s)
{
    NerfPlan p;
    if (int rc = make_nerf_plan(ctx, slot, N, n_tensors, p)) return rc;
    const Field &f = ctx->field[slot];
    const int act = f.d.activation, cus = ctx->cus, WH = p.WH, NBK = WH / kWidth, HC = p.HC;
    const int sp = f.d.weight_dtype == NEDDF_DTYPE_F16_SPLIT;
    if (int rc = ensure(ctx, ctx->tpack, 2 * kPackFloats * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->ttmp, ((size_t)N * kLdNarrow + kWidth) * sizeof(float))) return rc;
    float *wp = (float *)ctx->tpack.p;
    float *CR = (float *)ctx->ttmp.p, *bias_c0 = CR + (size_t)N * kLdNarrow;
    float *PE = ws + p.o_pe, *Ed = ws + p.o_ed;
    EncodeDesc enc;
    fill_enc(enc, f);
    launch_pe_values(pos, dir, var, N, enc, PE, kLdPe, Ed, kLdDir, s);
    const int kpe = (p.Cpe + 3) & ~3, kdir = (p.Cdir + 3) & ~3;
    // Z[N, ldz] (+)= X[N, Kin] x (input columns k_off .. of the [nrows_out, in_total] weight)^T for output rows 0 .. nout_valid (padded to a
    // multiple of 256): Kin <= 256 is one K block of `kload` loaded columns, otherwise Kin = WH
    auto gemm_fw = [&](const float *X, int ldx, int kload, int Kin, const float *Wsrc, int in_total, int k_off, int nout_valid, const float *bias,
                       float *Z, int ldz, int acc0, int act_kind, float *H) {
        const int KB = Kin <= kWidth ? 1 : Kin / kWidth, kc = Kin <= kWidth ? Kin : kWidth, kl = Kin <= kWidth ? kload : kWidth;
        const int NB = (nout_valid + kWidth - 1) / kWidth;
        for (int nb = 0; nb < NB; ++nb)
            for (int kb = 0; kb < KB; ++kb) {
                const bool last = kb == KB - 1;
                const int nv = nout_valid - nb * kWidth < kWidth ? nout_valid - nb * kWidth : kWidth;
                launch_pack(sp, Wsrc, 1, in_total, k_off + kb * kWidth, nb * kWidth, kc, nv, kWidth, wp, s);
                launch_rows_gemm(sp, X + kb * kWidth, N, ldx, kl, wp, gemm_ksteps(kc, sp), (kb == 0 && bias) ? bias + nb * kWidth : nullptr, 1,
                                 Z + nb * kWidth, ldz, (acc0 || kb > 0) ? 1 : 0, last ? act_kind : -1, (last && H) ? H + nb * kWidth : nullptr, cus, s);
            }
    };
    for (int l = 0; l < p.n; ++l) {             // nerf.py:151-155
        float *Z = ws + p.o_z[l], *H = ws + p.o_h[l];
        const bool wide = l > 0 && in_skips(f.d, l - 1);
        const int in_total = l == 0 ? p.Cpe : (wide ? WH + p.Cpe : WH);
        if (l == 0) gemm_fw(PE, kLdPe, kpe, p.Cpe, W[0], in_total, 0, WH, B[0], Z, WH, 0, act, H);
        else if (!wide) gemm_fw(ws + p.o_h[l - 1], WH, WH, WH, W[l], in_total, 0, WH, B[l], Z, WH, 0, act, H);
        else {          // cat([hx, embed_pos]): the hidden state feeds input columns 0 .. WH-1, the encoding WH ..
            gemm_fw(ws + p.o_h[l - 1], WH, WH, WH, W[l], in_total, 0, WH, B[l], Z, WH, 0, -1, nullptr);
            gemm_fw(PE, kLdPe, kpe, p.Cpe, W[l], in_total, WH, WH, nullptr, Z, WH, 1, act, H);
        }
    }
    const float *Hlast = ws + p.o_h[p.n - 1];
    for (int kb = 0; kb < NBK; ++kb) {
        NarrowW dens{};
        dens.nc = 1; dens.wstride = 1; dens.kcount = kWidth;
        dens.w[0] = W[p.i_dens] + kb * kWidth; dens.b[0] = kb == 0 ? B[p.i_dens] : nullptr;
        launch_narrow_forward(Hlast + kb * kWidth, WH, N, dens, 1, ws + p.o_zd, kLdNarrow, s, 0, kb > 0);
    }
    if (density) launch_density_head(f.d.density_activation, ws + p.o_zd, kLdNarrow, N, nullptr, density, 1, s);
    // colour head: Linear(WH + dir, WH / 2) -> ReLU -> Linear(WH / 2, 3); the WH / 2 outputs are one 256-column block (zero weights beyond them)
    HIPCHK(hipMemsetAsync(bias_c0, 0, kWidth * sizeof(float), s));
    HIPCHK(hipMemcpyAsync(bias_c0, B[p.i_c0], (size_t)HC * sizeof(float), hipMemcpyDeviceToDevice, s));
    gemm_fw(Hlast, WH, WH, WH, W[p.i_c0], p.in_c0, 0, HC, bias_c0, ws + p.o_zc, kWidth, 0, -1, nullptr);
    gemm_fw(Ed, kLdDir, kdir, p.Cdir, W[p.i_c0], p.in_c0, WH, HC, nullptr, ws + p.o_zc, kWidth, 1, NEDDF_ACT_RELU, ws + p.o_hc);
    NarrowW c1{};
    c1.nc = 3; c1.wstride = 1; c1.kcount = HC;
    for (int c = 0; c < 3; ++c) { c1.w[c] = W[p.i_c1] + c * HC; c1.b[c] = B[p.i_c1] + c; }
    launch_narrow_forward(ws + p.o_hc, kWidth, N, c1, 1, CR, kLdNarrow, s);
    if (color) launch_copy3(CR, kLdNarrow, color, 3, N, s);
    HIPCHK(hipGetLastError());
    return 0;
}

int nerf_backward(neddf_ctx *ctx, int slot, const float *const *W, int n_tensors, int64_t N, float *ws, const float *g_density,
                  const float *g_color, float *const *gW, float *const *gB, hipStream_t s)
{
    NerfPlan p;
    if (int rc = make_nerf_plan(ctx, slot, N, n_tensors, p)) return rc;
    const Field &f = ctx->field[slot];
    const int act = f.d.activation, cus = ctx->cus, WH = p.WH, NBK = WH / kWidth, HC = p.HC;
    const int sp = f.d.weight_dtype == NEDDF_DTYPE_F16_SPLIT;
    if (int rc = ensure(ctx, ctx->tpack, 2 * kPackFloats * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->ttmp, (size_t)N * (2 * WH + 2 * kLdNarrow) * sizeof(float))) return rc;
    float *wp = (float *)ctx->tpack.p;
    float *dA = (float *)ctx->ttmp.p, *dB = dA + (size_t)N * WH, *GC = dB + (size_t)N * WH, *GD = GC + (size_t)N * kLdNarrow;
    const float *PE = ws + p.o_pe, *Ed = ws + p.o_ed, *Hlast = ws + p.o_h[p.n - 1];
    HIPCHK(hipMemsetAsync(GC, 0, (size_t)N * 2 * kLdNarrow * sizeof(float), s));      // GC and GD
    if (g_color) launch_copy3(g_color, 3, GC, kLdNarrow, N, s);
    // colour head, second layer + the ReLU in front of it (dA = dZ of the first colour layer as one [N, 256] block; padded columns stay zero)
    NarrowW c1{};
    c1.nc = 3; c1.wstride = 1; c1.kcount = HC;
    for (int c = 0; c < 3; ++c) c1.w[c] = W[p.i_c1] + c * HC;
    AmaxSlots am;
    if (int rc = amax_begin(ctx, sp, am, s)) return rc;
    float *mA = am.take();          // max |dA| of the gradient matrix currently in dA
    launch_narrow_backward_act(GC, kLdNarrow, N, c1, nullptr, 0, NEDDF_ACT_RELU, 1, ws + p.o_zc, dA, kWidth, s, mA);
    {
        float *wc[3] = { gW[p.i_c1], gW[p.i_c1] + HC, gW[p.i_c1] + 2 * HC }, *bc[3] = { gB[p.i_c1], gB[p.i_c1] + 1, gB[p.i_c1] + 2 };
        launch_narrow_dw(ws + p.o_hc, kWidth, GC, kLdNarrow, N, 3, wc, 1, bc, 1, HC, s);
    }
    // first colour layer (weights [WH / 2, WH + dir]): weight gradients per 256-row block of the hidden input, then dHlast = dA x W_c0[:, 0:WH]
    for (int kb = 0; kb < NBK; ++kb)
        launch_dw(sp, Hlast + kb * kWidth, WH, kWidth, dA, kWidth, N, gW[p.i_c0] + kb * kWidth, 1, p.in_c0, HC, kb == 0 ? gB[p.i_c0] : nullptr, 1, cus, s, mA, am.dw_tmp);
    launch_dw(sp, Ed, kLdDir, p.Cdir, dA, kWidth, N, gW[p.i_c0] + WH, 1, p.in_c0, HC, nullptr, 1, cus, s, mA, am.dw_tmp);
    for (int nb = 0; nb < NBK; ++nb) {
        launch_pack(sp, W[p.i_c0], p.in_c0, 1, 0, nb * kWidth, HC, kWidth, kWidth, wp, s);     // rows = the WH / 2 outputs, columns = hidden inputs of block nb
        launch_rows_gemm(sp, dA, N, kWidth, HC, wp, gemm_ksteps(HC, sp), nullptr, 1, dB + nb * kWidth, WH, 0, -1, nullptr, cus, s, mA);
    }
    // density head, then the last trunk activation: dA = dZ of the last trunk layer ([N, WH] from here on)
    if (g_density) launch_density_head(f.d.density_activation, ws + p.o_zd, kLdNarrow, N, g_density, GD, kLdNarrow, s);
    mA = am.take();
    for (int kb = 0; kb < NBK; ++kb) {
        NarrowW dens{};
        dens.nc = 1; dens.wstride = 1; dens.kcount = kWidth;
        dens.w[0] = W[p.i_dens] + kb * kWidth;
        launch_narrow_backward_act(GD, kLdNarrow, N, dens, dB + kb * kWidth, 1, act, 1, ws + p.o_z[p.n - 1] + kb * kWidth, dA + kb * kWidth, WH, s, mA);
        float *wd[1] = { gW[p.i_dens] + kb * kWidth }, *bd[1] = { gB[p.i_dens] };
        launch_narrow_dw(Hlast + kb * kWidth, WH, GD, kLdNarrow, N, 1, wd, 1, kb == 0 ? bd : nullptr, 1, kWidth, s);
    }
    // trunk
    auto dw_blocks = [&](const float *X, int ldx, int Kin, float *gWl, int in_total, int col_off, float *gBl) {       // gW[n, col_off + k] += dA^T X
        const int KB = Kin <= kWidth ? 1 : Kin / kWidth, kc = Kin <= kWidth ? Kin : kWidth;
        for (int nb = 0; nb < NBK; ++nb)
            for (int kb = 0; kb < KB; ++kb)
                launch_dw(sp, X + kb * kWidth, ldx, kc, dA + nb * kWidth, WH, N, gWl + (size_t)nb * kWidth * in_total + col_off + kb * kWidth, 1, in_total, kWidth,
                          (kb == 0 && gBl) ? gBl + nb * kWidth : nullptr, 1, cus, s, mA, am.dw_tmp);
    };
    for (int l = p.n - 1; l >= 0; --l) {
        const bool wide = l > 0 && in_skips(f.d, l - 1);
        const int in_total = l == 0 ? p.Cpe : (wide ? WH + p.Cpe : WH);
        if (l == 0) {
            dw_blocks(PE, kLdPe, p.Cpe, gW[0], in_total, 0, gB[0]);
            break;
        }
        dw_blocks(ws + p.o_h[l - 1], WH, WH, gW[l], in_total, 0, gB[l]);
        if (wide) dw_blocks(PE, kLdPe, p.Cpe, gW[l], in_total, WH, nullptr);
        float *mB = am.take();
        for (int nb = 0; nb < NBK; ++nb)             // dB[:, nb] = actback(Z_{l-1}[:, nb]; sum_kb dA[:, kb] x W[kb rows, nb columns])
            for (int kb = 0; kb < NBK; ++kb) {
                launch_pack(sp, W[l], in_total, 1, kb * kWidth, nb * kWidth, kWidth, kWidth, kWidth, wp, s);
                if (kb == NBK - 1)
                    launch_rows_gemm_actback(sp, dA + kb * kWidth, N, WH, kWidth, wp, gemm_ksteps(kWidth, sp), 1, act, ws + p.o_z[l - 1] + nb * kWidth,
                                             dB + nb * kWidth, WH, cus, s, mA, mB, kb > 0);
                else
                    launch_rows_gemm(sp, dA + kb * kWidth, N, WH, kWidth, wp, gemm_ksteps(kWidth, sp), nullptr, 1, dB + nb * kWidth, WH, kb > 0, -1, nullptr, cus, s, mA);
            }
        float *t = dA; dA = dB; dB = t;
        mA = mB;
    }
    HIPCHK(hipGetLastError());
    return 0;
}


// ---- NeuS (neus.py:101-162) ---------------------------------------------------------------------------------------------
// sdf trunk on (value, Jacobian) row groups (the reference's torch.autograd.grad normal is the Jacobian of feature 0, so its
// double backward is the ordinary backward of those rows), colour trunk on value rows; nn.Linear weights [out, in].
// Tensor order as neddf_set_field: layers_sdf.0..n-1, layers_col.0..m, variance (1 element; its bias slot is unused).
static_assert(kActTanhExp == NEDDF_ACT_TANHEXP, "activation ids");
struct NeusPlan {
    int E, Ed, Cpe, Cdir, Ca, ldxa, n_sdf, n_col, i_cout, i_var, in_c0;
    int WH;                     // hidden width the kernels see, both trunks (256 or 512)
    int64_t N, R;
    size_t o_pe, o_ed, o_xa, o_zo, o_z[kMaxLayers], o_h[kMaxLayers], o_zc[kMaxLayers], o_hc[kMaxLayers], total;
};

int make_neus_plan(neddf_ctx *ctx, int slot, int64_t N, int n_tensors, NeusPlan &p)
{
    const Field &f = ctx->field[slot];
    if (int rc = train_supported(ctx, f)) return rc;
    p.E = f.d.embed_pos_rank; p.Ed = f.d.embed_dir_rank;
    p.Cpe = 6 * p.E; p.Cdir = 6 * p.Ed; p.Ca = 6 + p.Cdir; p.ldxa = roundup(p.Ca, 8);
    p.n_sdf = f.d.layer_count; p.n_col = f.d.col_layer_count;
    p.WH = f.d.layer_width;
    p.i_cout = p.n_sdf + p.n_col; p.i_var = p.i_cout + 1; p.in_c0 = p.Ca + p.WH;
    if (n_tensors >= 0 && n_tensors != p.n_sdf + p.n_col + 2) return fail(ctx, NEDDF_EINVAL, "NeuS: wrong tensor count");
    p.N = N; p.R = 4 * N;
    size_t o = 0;
    auto take = [&](size_t n) { size_t at = o; o += (n + 63) & ~(size_t)63; return at; };
    p.o_pe = take((size_t)p.R * kLdPe);
    p.o_ed = take((size_t)N * kLdDir);
    p.o_xa = take((size_t)N * p.ldxa);
    p.o_zo = take((size_t)N * kLdNarrow);
    for (int l = 0; l < p.n_sdf; ++l) { p.o_z[l] = take((size_t)p.R * p.WH); p.o_h[l] = take((size_t)p.R * p.WH); }
    for (int l = 0; l < p.n_col; ++l) { p.o_zc[l] = take((size_t)N * p.WH); p.o_hc[l] = take((size_t)N * p.WH); }
    p.total = o;
    return 0;
}

void neus_point_args(NeusPointArgs &a, const Field &f, const NeusPlan &p, const float *const *W, float *ws)
{
    a = NeusPointArgs{};
    a.N = p.N; a.act = f.d.activation; a.Cdir = p.Cdir;
    a.variance = W[p.i_var];
    a.Ed = ws + p.o_ed; a.ldd = kLdDir;
    a.Hlast = ws + p.o_h[p.n_sdf - 1]; a.Zlast = ws + p.o_z[p.n_sdf - 1]; a.ldh = p.WH;
    a.XA = ws + p.o_xa; a.ldxa = p.ldxa;
    a.ZC = ws + p.o_zo; a.ldc = kLdNarrow;
}

int neus_forward(neddf_ctx *ctx, int slot, const float *const *W, const float *const *B, int n_tensors, const float *pos, const float *dir,
                 int64_t N, float *ws, float *sdf, float *density, float *color, hipStream_t s)
{
    NeusPlan p;
    if (int rc = make_neus_plan(ctx, slot, N, n_tensors, p)) return rc;
    const Field &f = ctx->field[slot];
    const int act = f.d.activation, cus = ctx->cus, WH = p.WH, NBK = WH / kWidth;
    const int sp = f.d.weight_dtype == NEDDF_DTYPE_F16_SPLIT;
    if (int rc = ensure(ctx, ctx->tpack, 2 * kPackFloats * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->ttmp, ((size_t)p.R * kLdPe + (size_t)N * 4) * sizeof(float))) return rc;
    float *wp = (float *)ctx->tpack.p;
    float *PEs = (float *)ctx->ttmp.p, *var0 = PEs + (size_t)p.R * kLdPe;
    float *PE = ws + p.o_pe, *Ed = ws + p.o_ed;
    // plain PositionalEncoding (neus.py:118-119): no variance weights, no low-pass schedule
    HIPCHK(hipMemsetAsync(var0, 0, (size_t)N * 3 * sizeof(float), s));
    EncodeDesc enc;
    fill_enc(enc, f);
    for (int i = 0; i < 10; ++i) enc.lowpass[i] = 1.0f;
    launch_pe_rows(pos, dir, var0, N, enc, PEs, PE, kLdPe, Ed, kLdDir, s);
    const int kpe = (p.Cpe + 3) & ~3;
    int n_wide = 0;
    for (int l = 1; l < p.n_sdf; ++l) n_wide += in_skips(f.d, l - 1) ? 1 : 0;
    static const bool unfused = [] { const char *e = getenv("NEDDF_TRAIN_UNFUSED"); return e && atoi(e) != 0; }();
    // Z[rows, WH] (+)= X[rows, Kin] x (input columns k_off .. of the [WH, in_total] weight)^T, 256 x 256 blocks: Kin <= 256 is one K
    // block of `kload` loaded columns, otherwise Kin = WH; K blocks accumulate in Z, the activation runs with the last one
    auto gemm_fw = [&](const float *X, int64_t rows, int ldx, int kload, int Kin, const float *Wsrc, int in_total, int k_off, const float *bias,
                       int period, float *Z, int acc0, int act_kind, float *H) {
        const int KB = Kin <= kWidth ? 1 : Kin / kWidth, kc = Kin <= kWidth ? Kin : kWidth, kl = Kin <= kWidth ? kload : kWidth;
        for (int nb = 0; nb < NBK; ++nb)
            for (int kb = 0; kb < KB; ++kb) {
                const bool last = kb == KB - 1;
                launch_pack(sp, Wsrc, 1, in_total, k_off + kb * kWidth, nb * kWidth, kc, kWidth, kWidth, wp, s);
                launch_rows_gemm(sp, X + kb * kWidth, rows, ldx, kl, wp, gemm_ksteps(kc, sp), (kb == 0 && bias) ? bias + nb * kWidth : nullptr, period,
                                 Z + nb * kWidth, WH, (acc0 || kb > 0) ? 1 : 0, last ? act_kind : -1, (last && H) ? H + nb * kWidth : nullptr, cus, s);
            }
    };
    if (!unfused && n_wide <= 1 && WH == kWidth) {      // the sdf trunk (neus.py:121-125) as one fused layer stack (train_kernels.h MlpForwardArgs)
        if (int rc = ensure(ctx, ctx->tpack, (size_t)(kMaxLayers + 2) * kPackFloats * sizeof(float))) return rc;
        wp = (float *)ctx->tpack.p;
        float *pack_at = wp;
        auto next_pack = [&]() { float *r = pack_at; pack_at += kPackFloats; return r; };
        MlpForwardArgs m{};
        m.R = p.R; m.X0 = PE; m.ld0 = kLdPe; m.kload0 = kpe; m.ksteps0 = gemm_ksteps(p.Cpe, sp);
        m.n_layers = p.n_sdf; m.skip_layer = -1; m.act_kind = act;
        float *w0 = next_pack();
        launch_pack(sp, W[0], 1, p.Cpe, 0, 0, p.Cpe, kWidth, kWidth, w0, s);
        m.wp0 = w0;
        for (int l = 0; l < p.n_sdf; ++l) {
            const bool wide = l > 0 && in_skips(f.d, l - 1);
            const int in_total = l == 0 ? p.Cpe : (wide ? kWidth + p.Cpe : kWidth);
            m.bias[l] = B[l]; m.Z[l] = ws + p.o_z[l]; m.H[l] = ws + p.o_h[l];
            if (l == 0) continue;
            float *wl = next_pack();
            launch_pack(sp, W[l], 1, in_total, 0, 0, kWidth, kWidth, kWidth, wl, s);       // cat([hx, embed_pos]): hidden state first
            m.wp[l] = wl;
            if (wide) {
                float *wsk = next_pack();
                launch_pack(sp, W[l], 1, in_total, kWidth, 0, p.Cpe, kWidth, kWidth, wsk, s);
                m.skip_layer = l; m.wp_skip = wsk;
            }
        }
        launch_mlp_forward(sp, m, cus, s);
    } else
    for (int l = 0; l < p.n_sdf; ++l) {             // neus.py:121-125
        float *Z = ws + p.o_z[l], *H = ws + p.o_h[l];
        const bool wide = l > 0 && in_skips(f.d, l - 1);
        const int in_total = l == 0 ? p.Cpe : (wide ? WH + p.Cpe : WH);
        if (l == 0) gemm_fw(PE, p.R, kLdPe, kpe, p.Cpe, W[0], in_total, 0, B[0], 4, Z, 0, act, H);
        else if (!wide) gemm_fw(ws + p.o_h[l - 1], p.R, WH, WH, WH, W[l], in_total, 0, B[l], 4, Z, 0, act, H);
        else {          // cat([hx, embed_pos]): hidden state first
            gemm_fw(ws + p.o_h[l - 1], p.R, WH, WH, WH, W[l], in_total, 0, B[l], 4, Z, 0, -1, nullptr);
            gemm_fw(PE, p.R, kLdPe, kpe, p.Cpe, W[l], in_total, WH, nullptr, 4, Z, 1, act, H);
        }
    }
    NeusPointArgs a;
    neus_point_args(a, f, p, W, ws);
    a.pos = pos; a.sdf = sdf; a.density = density; a.color = color;
    launch_neus_head_forward(a, s);
    // colour trunk on value rows (neus.py:146-152): the features are the value rows of the last sdf layer (row stride 4 x WH)
    const float *Hlast = ws + p.o_h[p.n_sdf - 1];
    for (int l = 0; l < p.n_col; ++l) {
        float *Z = ws + p.o_zc[l], *H = ws + p.o_hc[l];
        const float *Wl = W[p.n_sdf + l], *Bl = B[p.n_sdf + l];
        if (l == 0) {
            gemm_fw(ws + p.o_xa, N, p.ldxa, p.ldxa, p.Ca, Wl, p.in_c0, 0, Bl, 1, Z, 0, -1, nullptr);
            gemm_fw(Hlast, N, 4 * WH, WH, WH, Wl, p.in_c0, p.Ca, nullptr, 1, Z, 1, act, H);
        } else
            gemm_fw(ws + p.o_hc[l - 1], N, WH, WH, WH, Wl, WH, 0, Bl, 1, Z, 0, act, H);
    }
    for (int kb = 0; kb < NBK; ++kb) {
        NarrowW cout{};
        cout.nc = 3; cout.wstride = 1; cout.kcount = kWidth;
        for (int c = 0; c < 3; ++c) { cout.w[c] = W[p.i_cout] + c * WH + kb * kWidth; cout.b[c] = kb == 0 ? B[p.i_cout] + c : nullptr; }
        launch_narrow_forward(ws + p.o_hc[p.n_col - 1] + kb * kWidth, WH, N, cout, 1, ws + p.o_zo, kLdNarrow, s, 0, kb > 0);
    }
    if (color) launch_neus_color_forward(a, s);
    HIPCHK(hipGetLastError());
    return 0;
}

int neus_backward(neddf_ctx *ctx, int slot, const float *const *W, int n_tensors, int64_t N, float *ws, const float *g_sdf,
                  const float *g_density, const float *g_color, float *const *gW, float *const *gB, hipStream_t s)
{
    NeusPlan p;
    if (int rc = make_neus_plan(ctx, slot, N, n_tensors, p)) return rc;
    const Field &f = ctx->field[slot];
    const int act = f.d.activation, act4 = neus_backward_act_kind(act), cus = ctx->cus, WH = p.WH, NBK = WH / kWidth;
    const int sp = f.d.weight_dtype == NEDDF_DTYPE_F16_SPLIT;
    if (int rc = ensure(ctx, ctx->tpack, 2 * kPackFloats * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->ttmp, ((size_t)p.R * 2 * WH + (size_t)N * 2 * kLdNarrow) * sizeof(float))) return rc;
    float *wp = (float *)ctx->tpack.p;
    float *dA = (float *)ctx->ttmp.p, *dB = dA + (size_t)p.R * WH, *GC = dB + (size_t)p.R * WH, *DG = GC + (size_t)N * kLdNarrow;
    const float *PE = ws + p.o_pe, *Hlast = ws + p.o_h[p.n_sdf - 1];
    NeusPointArgs a;
    neus_point_args(a, f, p, W, ws);
    a.g_sdf = g_sdf; a.g_density = g_density; a.g_color = g_color;
    a.GC = GC; a.g_variance = gW[p.i_var];
    launch_neus_color_backward(a, s);               // GC = g_color a'(ZC); variance gradient
    AmaxSlots am;
    if (int rc = amax_begin(ctx, sp, am, s)) return rc;
    float *mA = am.take();          // max |dA| of the gradient matrix currently in dA
    // gW[n, col_off + k] += dA[rows, WH]^T X[rows, Kin] in 256 x 256 blocks (+ the bias gradient with the first K block)
    auto dw_blocks = [&](const float *X, int64_t rows, int ldx, int Kin, float *gWl, int in_total, int col_off, float *gBl, int period) {
        const int KB = Kin <= kWidth ? 1 : Kin / kWidth, kc = Kin <= kWidth ? Kin : kWidth;
        for (int nb = 0; nb < NBK; ++nb)
            for (int kb = 0; kb < KB; ++kb)
                launch_dw(sp, X + kb * kWidth, ldx, kc, dA + nb * kWidth, WH, rows, gWl + (size_t)nb * kWidth * in_total + col_off + kb * kWidth, 1, in_total, kWidth,
                          (kb == 0 && gBl) ? gBl + nb * kWidth : nullptr, period, cus, s, mA, am.dw_tmp);
    };
    // dB[rows, WH] = actback(Zprev; dA x (input columns k_off .. k_off + WH of the [WH, in_total] weight)) (act_kind < 0: no activation)
    auto gemm_bw = [&](int64_t rows, const float *Wsrc, int in_total, int k_off, int period, int act_kind, const float *Zprev, float *mB) {
        for (int nb = 0; nb < NBK; ++nb)
            for (int kb = 0; kb < NBK; ++kb) {
                launch_pack(sp, Wsrc, in_total, 1, kb * kWidth, k_off + nb * kWidth, kWidth, kWidth, kWidth, wp, s);      // rows = outputs, columns = inputs
                if (kb == NBK - 1 && act_kind >= 0)
                    launch_rows_gemm_actback(sp, dA + kb * kWidth, rows, WH, kWidth, wp, gemm_ksteps(kWidth, sp), period, act_kind, Zprev + nb * kWidth,
                                             dB + nb * kWidth, WH, cus, s, mA, mB, kb > 0);
                else
                    launch_rows_gemm(sp, dA + kb * kWidth, rows, WH, kWidth, wp, gemm_ksteps(kWidth, sp), nullptr, 1, dB + nb * kWidth, WH, kb > 0, -1, nullptr, cus, s, mA);
            }
    };
    // colour trunk, value rows ([N, WH] matrices; dA holds dZ of the layer in flight)
    for (int kb = 0; kb < NBK; ++kb) {
        NarrowW cout{};
        cout.nc = 3; cout.wstride = 1; cout.kcount = kWidth;
        for (int c = 0; c < 3; ++c) cout.w[c] = W[p.i_cout] + c * WH + kb * kWidth;
        launch_narrow_backward_act(GC, kLdNarrow, N, cout, nullptr, 0, act, 1, ws + p.o_zc[p.n_col - 1] + kb * kWidth, dA + kb * kWidth, WH, s, mA);
        float *wc[3] = { gW[p.i_cout] + kb * kWidth, gW[p.i_cout] + WH + kb * kWidth, gW[p.i_cout] + 2 * WH + kb * kWidth };
        float *bc[3] = { gB[p.i_cout], gB[p.i_cout] + 1, gB[p.i_cout] + 2 };
        launch_narrow_dw(ws + p.o_hc[p.n_col - 1] + kb * kWidth, WH, GC, kLdNarrow, N, 3, wc, 1, kb == 0 ? bc : nullptr, 1, kWidth, s);
    }
    for (int l = p.n_col - 1; l >= 1; --l) {
        dw_blocks(ws + p.o_hc[l - 1], N, WH, WH, gW[p.n_sdf + l], WH, 0, gB[p.n_sdf + l], 1);
        float *mB = am.take();
        gemm_bw(N, W[p.n_sdf + l], WH, 0, 1, act, ws + p.o_zc[l - 1], mB);
        float *t = dA; dA = dB; dB = t;
        mA = mB;
    }
    {   // first colour layer: weights [WH, pos 3 | embed_dir | gradient 3 | features WH]
        const float *W0 = W[p.n_sdf];
        float *gW0 = gW[p.n_sdf];
        dw_blocks(ws + p.o_xa, N, p.ldxa, p.Ca, gW0, p.in_c0, 0, gB[p.n_sdf], 1);
        dw_blocks(Hlast, N, 4 * WH, WH, gW0, p.in_c0, p.Ca, nullptr, 1);
        // of the small inputs only the normal depends on parameters: DG[n, k] = dA[n, :] . W0[:, 3 + Cdir + k]
        for (int kb = 0; kb < NBK; ++kb) {
            NarrowW wn{};
            wn.nc = 3; wn.wstride = p.in_c0; wn.kcount = kWidth;
            for (int c = 0; c < 3; ++c) wn.w[c] = W0 + (size_t)kb * kWidth * p.in_c0 + 3 + p.Cdir + c;
            launch_narrow_forward(dA + kb * kWidth, WH, N, wn, 1, DG, kLdNarrow, s, 0, kb > 0);
        }
        gemm_bw(N, W0, p.in_c0, p.Ca, 1, -1, nullptr, nullptr);      // dF
    }
    // heads: dZ of the last sdf layer on (value, Jacobian) rows
    mA = am.take();
    a.dF = dB; a.DG = DG; a.lddg = kLdNarrow; a.dZ = dA; a.amax_out = mA;
    launch_neus_head_backward(a, s);
    for (int l = p.n_sdf - 1; l >= 0; --l) {
        const bool wide = l > 0 && in_skips(f.d, l - 1);
        const int in_total = l == 0 ? p.Cpe : (wide ? WH + p.Cpe : WH);
        if (l == 0) {
            dw_blocks(PE, p.R, kLdPe, p.Cpe, gW[0], in_total, 0, gB[0], 4);
            break;
        }
        dw_blocks(ws + p.o_h[l - 1], p.R, WH, WH, gW[l], in_total, 0, gB[l], 4);
        if (wide) dw_blocks(PE, p.R, kLdPe, p.Cpe, gW[l], in_total, WH, nullptr, 4);
        float *mB = am.take();
        gemm_bw(p.R, W[l], in_total, 0, 4, act4, ws + p.o_z[l - 1], mB);
        float *t = dA; dA = dB; dB = t;
        mA = mB;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

int64_t neddf_train_workspace_floats(neddf_ctx *ctx, int slot, int64_t n_points)
{
    if (!ctx || n_points < 0) return -1;
    if (slot >= 0 && slot < NEDDF_NUM_SLOTS && ctx->field[slot].valid && ctx->field[slot].d.kind == NEDDF_FIELD_NERF) {
        NerfPlan np;
        if (make_nerf_plan(ctx, slot, n_points, -1, np)) return -1;
        return (int64_t)np.total;
    }
    if (slot >= 0 && slot < NEDDF_NUM_SLOTS && ctx->field[slot].valid && ctx->field[slot].d.kind == NEDDF_FIELD_NEUS) {
        NeusPlan np;
        if (make_neus_plan(ctx, slot, n_points, -1, np)) return -1;
        return (int64_t)np.total;
    }
    Plan p;
    if (make_plan(ctx, slot, n_points, -1, p)) return -1;
    return (int64_t)p.total;
}

int neddf_train_field_forward(neddf_ctx *ctx, int slot, const float *const *W, const float *const *B, int n_tensors,
                              const float *pos, const float *dir, const float *var, int64_t N, float *ws, float *distance,
                              float *density, float *color, float *penalty, float *aux_grad, void *stream)
{
    if (!ctx) return NEDDF_EINVAL;
    if (N <= 0) return 0;
    if (!W || !B || !pos || !dir || !var || !ws) return fail(ctx, NEDDF_EINVAL, "null argument");
    DeviceGuard guard_(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    if (slot >= 0 && slot < NEDDF_NUM_SLOTS && ctx->field[slot].valid && ctx->field[slot].d.kind == NEDDF_FIELD_NERF)
        return nerf_forward(ctx, slot, W, B, n_tensors, pos, dir, var, N, ws, density, color, s);
    if (slot >= 0 && slot < NEDDF_NUM_SLOTS && ctx->field[slot].valid && ctx->field[slot].d.kind == NEDDF_FIELD_NEUS)
        return neus_forward(ctx, slot, W, B, n_tensors, pos, dir, N, ws, distance, density, color, s);
    Plan p;
    if (int rc = make_plan(ctx, slot, N, n_tensors, p)) return rc;
    const Field &f = ctx->field[slot];
    const int act = f.d.activation;
    const int sp = f.d.weight_dtype == NEDDF_DTYPE_F16_SPLIT;      // GEMM operands as two fp16 terms (tile_engine.h)
    // packed weights of a whole layer stack: [layer][256 x 256] + the narrow first-layer / skip segments
    const size_t packf = (size_t)p.WH * p.WH;       // one packed hidden x hidden matrix (kPackFloats at width 256)
    if (int rc = ensure(ctx, ctx->tpack, (size_t)(kMaxLayers + 2) * packf * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->ttmp, ((size_t)p.R * (kLdPe + kLdNarrow) + (size_t)N * kLdDir) * sizeof(float))) return rc;
    float *wp = (float *)ctx->tpack.p;
    float *PEu = (float *)ctx->ttmp.p, *Ed = PEu + (size_t)p.R * kLdPe, *ZH = Ed + (size_t)N * kLdDir;
    float *PEs = ws + p.o_pes;
    TrainPointArgs a;
    point_args(a, f, p, ws);
    launch_pe_rows(pos, dir, var, N, a.enc, PEs, PEu, kLdPe, Ed, kLdDir, s);
    // NEDDF_TRAIN_UNFUSED=1: one GEMM kernel per layer (the round-1 forward), kept for A/B measurements
    static const bool unfused = [] { const char *e = getenv("NEDDF_TRAIN_UNFUSED"); return e && atoi(e) != 0; }();
    const int kpe = (p.Cpe + 3) & ~3;       // loaded width of the encoding matrix (pad columns are zero)
    float *pack_at = wp;
    auto next_pack = [&]() { float *r = pack_at; pack_at += packf; return r; };
    // weight fragments of a fused stack: fp32 in one launch per stack (PackBatch), split fp16 matrix by matrix
    PackBatch pb;
    auto pack = [&](const float *src, int64_t sk, int64_t sn, int k_off, int n_off, int kcount, int ncount, int nout, float *dst) {
        if (sp) launch_pack(1, src, sk, sn, k_off, n_off, kcount, ncount, nout, dst, s);
        else pb.add(src, sk, sn, k_off, n_off, kcount, ncount, nout, dst, s);
    };
    // distance trunk (neddf.py:206-218); the fused kernel holds one skip partial, architectures with more take the per-layer route
    int n_wide = 0;
    for (int l = 1; l < p.n_trunk; ++l) n_wide += in_skips(f.d, l - 1) ? 1 : 0;
    // fp32 policy, both stacks fused: the hidden states are kept point-major (train_kernels.h MlpForwardArgs.point_major), the layout of
    // the fused backward (neddf_train_field_backward takes that route under the same condition)
    const int WH = p.WH, NBK = WH / kWidth;      // hidden width the kernels see, in 256-column blocks
    // the fused chains are 256 wide, and -- round 5, fp32 policy -- 512 wide for the fields that train padded to 512 (wide_fused)
    const bool fused = !unfused && n_wide <= 1 && (WH == kWidth || (WH == 2 * kWidth && wide_fused() && (!sp || split_fused())));
    // (round 5: the split-fp16 policy takes the same fused, point-major route; NEDDF_TRAIN_SPLIT_FUSED=0 keeps its per-layer backward
    // and the row-major matrices that reads -- the A/B partner)
    const int pm = (fused && (!sp || split_fused())) ? 1 : 0;
    // Z[R, WH] (+)= X[R, Kin] x (rows k_off .. k_off + Kin of the [in, WH] weight) (+ bias on value rows); H = a(Z) when act_kind >= 0.
    // Kin <= 256: one K block of `kload` loaded columns; otherwise Kin = WH in 256-row blocks.  Every block is one rows_gemm launch:
    // K blocks accumulate in Z, the activation runs with the last one
    auto gemm_fw = [&](const float *X, int ldx, int kload, int Kin, const float *Wsrc, int k_off, const float *bias, float *Z, int acc0,
                       int act_kind, float *H) {
        const int KB = Kin <= kWidth ? 1 : Kin / kWidth, kc = Kin <= kWidth ? Kin : kWidth, kl = Kin <= kWidth ? kload : kWidth;
        for (int nb = 0; nb < NBK; ++nb)
            for (int kb = 0; kb < KB; ++kb) {
                const bool last = kb == KB - 1;
                launch_pack(sp, Wsrc, WH, 1, k_off + kb * kWidth, nb * kWidth, kc, kWidth, kWidth, wp, s);
                launch_rows_gemm(sp, X + kb * kWidth, p.R, ldx, kl, wp, gemm_ksteps(kc, sp), (kb == 0 && bias) ? bias + nb * kWidth : nullptr, 4,
                                 Z + nb * kWidth, WH, (acc0 || kb > 0) ? 1 : 0, last ? act_kind : -1, (last && H) ? H + nb * kWidth : nullptr, ctx->cus, s);
            }
    };
    // Y[R, kLdNarrow] = X[R, WH] . (nc narrow columns) + bias: one narrow_forward launch per 256-column block of X
    auto narrow_fw = [&](const float *X, NarrowW w, float *Y, int x_pm) {
        const float *w0[4] = { w.w[0], w.w[1], w.w[2], w.w[3] }, *b0[4] = { w.b[0], w.b[1], w.b[2], w.b[3] };
        for (int kb = 0; kb < NBK; ++kb) {
            for (int c = 0; c < w.nc; ++c) { w.w[c] = w0[c] + (size_t)kb * kWidth * w.wstride; w.b[c] = kb == 0 ? b0[c] : nullptr; }
            // (column block kb of a point-major matrix starts 4 x 256 kb floats into a point)
            launch_narrow_forward(X + (x_pm ? 4 : 1) * kb * kWidth, WH, p.R, w, 4, Y, kLdNarrow, s, x_pm, kb > 0);
        }
    };
    if (fused) {
        MlpForwardArgs m{};
        m.R = p.R; m.X0 = PEs; m.ld0 = kLdPe; m.kload0 = kpe; m.ksteps0 = gemm_ksteps(p.Cpe, sp);
        m.n_layers = p.n_trunk; m.skip_layer = -1; m.act_kind = act; m.point_major = pm; m.width = WH;
        float *w0 = next_pack();
        pack(W[0], WH, 1, 0, 0, p.Cpe, WH, WH, w0);
        m.wp0 = w0;
        for (int l = 0; l < p.n_trunk; ++l) {
            const bool wide = l > 0 && in_skips(f.d, l - 1);
            m.bias[l] = B[l]; m.Z[l] = ws + p.o_z[l]; m.H[l] = ws + p.o_h[l];
            if (l == 0) continue;
            float *wl = next_pack();
            pack(W[l], WH, 1, wide ? p.Cpe : 0, 0, WH, WH, WH, wl);
            m.wp[l] = wl;
            if (wide) {         // hx = cat([embed_pos_scaled, hx]): the encoding feeds rows 0 .. Cpe-1 of the weight
                float *wsk = next_pack();
                pack(W[l], WH, 1, 0, 0, p.Cpe, WH, WH, wsk);
                m.skip_layer = l; m.wp_skip = wsk;
            }
        }
        pb.flush(s);
        launch_mlp_forward(sp, m, ctx->cus, s);
    } else {
        for (int l = 0; l < p.n_trunk; ++l) {
            float *Z = ws + p.o_z[l], *H = ws + p.o_h[l];
            const bool wide = l > 0 && in_skips(f.d, l - 1);
            if (l == 0) gemm_fw(PEs, kLdPe, kpe, p.Cpe, W[0], 0, B[0], Z, 0, act, H);
            else if (!wide) gemm_fw(ws + p.o_h[l - 1], WH, WH, WH, W[l], 0, B[l], Z, 0, act, H);
            else {              // hx = cat([embed_pos_scaled, hx]): the encoding feeds rows 0 .. Cpe-1 of the weight
                gemm_fw(ws + p.o_h[l - 1], WH, WH, WH, W[l], p.Cpe, B[l], Z, 0, -1, nullptr);
                gemm_fw(PEs, kLdPe, kpe, p.Cpe, W[l], 0, nullptr, Z, 1, act, H);
            }
        }
    }
    const float *Hlast = ws + p.o_h[p.n_trunk - 1];
    NarrowW heads{};
    heads.nc = 2; heads.wstride = 1; heads.kcount = kWidth;
    heads.w[0] = W[p.i_ddf]; heads.w[1] = W[p.i_aux];
    heads.b[0] = B[p.i_ddf]; heads.b[1] = B[p.i_aux];
    narrow_fw(Hlast, heads, ZH, pm);
    a.ZH = ZH; a.PEu = PEu; a.Ed = Ed;
    a.distance = distance; a.density = density; a.aux_grad = aux_grad;
    launch_point_forward(a, s);
    // colour trunk (neddf.py:243-258)
    if (fused) {
        MlpForwardArgs m{};
        m.R = p.R; m.X0 = ws + p.o_xa; m.ld0 = p.ldxa; m.kload0 = p.ldxa; m.ksteps0 = gemm_ksteps(p.Ca, sp);
        m.X1 = Hlast; m.n_layers = p.n_col; m.skip_layer = -1; m.act_kind = act; m.point_major = pm; m.width = WH;
        pack_at = wp;           // same stream: the trunk kernel is done with the buffer when these packs run
        float *w0 = next_pack(), *w1 = next_pack();
        pack(W[p.n_trunk], WH, 1, 0, 0, p.Ca, WH, WH, w0);
        pack(W[p.n_trunk], WH, 1, p.Ca, 0, WH, WH, WH, w1);
        m.wp0 = w0; m.wp1 = w1;
        for (int l = 0; l < p.n_col; ++l) {
            m.bias[l] = B[p.n_trunk + l]; m.Z[l] = ws + p.o_zc[l]; m.H[l] = ws + p.o_hc[l];
            if (l == 0) continue;
            float *wl = next_pack();
            pack(W[p.n_trunk + l], WH, 1, 0, 0, WH, WH, WH, wl);
            m.wp[l] = wl;
        }
        pb.flush(s);
        launch_mlp_forward(sp, m, ctx->cus, s);
    } else {
        for (int l = 0; l < p.n_col; ++l) {
            float *Z = ws + p.o_zc[l], *H = ws + p.o_hc[l];
            const float *Wl = W[p.n_trunk + l], *Bl = B[p.n_trunk + l];
            if (l == 0) {       // [embed_pos | embed_dir | normal | features] (neddf.py:243)
                gemm_fw(ws + p.o_xa, p.ldxa, p.ldxa, p.Ca, Wl, 0, Bl, Z, 0, -1, nullptr);
                gemm_fw(Hlast, WH, WH, WH, Wl, p.Ca, nullptr, Z, 1, act, H);
            } else gemm_fw(ws + p.o_hc[l - 1], WH, WH, WH, Wl, 0, Bl, Z, 0, act, H);
        }
    }
    NarrowW cout{};
    cout.nc = 3; cout.wstride = 3; cout.kcount = kWidth;
    for (int c = 0; c < 3; ++c) { cout.w[c] = W[p.i_cout] + c; cout.b[c] = B[p.i_cout] + c; }
    narrow_fw(ws + p.o_hc[p.n_col - 1], cout, ws + p.o_cr, pm);
    a.color = color; a.penalty = penalty;
    launch_penalty_forward(a, s);
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_train_field_backward(neddf_ctx *ctx, int slot, const float *const *W, const float *const *B, int n_tensors, int64_t N,
                               const float *ws_, const float *g_distance, const float *g_density, const float *g_color,
                               const float *g_penalty, const float *g_aux_grad, float *const *gW, float *const *gB, void *stream)
{
    if (!ctx) return NEDDF_EINVAL;
    if (N <= 0) return 0;
    if (!W || !B || !ws_ || !gW || !gB) return fail(ctx, NEDDF_EINVAL, "null argument");
    DeviceGuard guard_(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    float *ws = const_cast<float *>(ws_);
    if (slot >= 0 && slot < NEDDF_NUM_SLOTS && ctx->field[slot].valid && ctx->field[slot].d.kind == NEDDF_FIELD_NERF)
        return nerf_backward(ctx, slot, W, n_tensors, N, ws, g_density, g_color, gW, gB, s);
    if (slot >= 0 && slot < NEDDF_NUM_SLOTS && ctx->field[slot].valid && ctx->field[slot].d.kind == NEDDF_FIELD_NEUS)
        return neus_backward(ctx, slot, W, n_tensors, N, ws, g_distance, g_density, g_color, gW, gB, s);
    Plan p;
    if (int rc = make_plan(ctx, slot, N, n_tensors, p)) return rc;
    const Field &f = ctx->field[slot];
    const int act = f.d.activation;
    const int sp = f.d.weight_dtype == NEDDF_DTYPE_F16_SPLIT;      // GEMM operands as two fp16 terms (tile_engine.h)
    // fp32 MFMA policy: the input-gradient chain of each layer stack is ONE kernel (train_kernels.h MlpBackwardArgs); every dZ_l keeps
    // its own matrix for the weight-gradient products.  NEDDF_TRAIN_UNFUSED=1 and the split-fp16 policy take the per-layer route below.
    static const bool unfused = [] { const char *e = getenv("NEDDF_TRAIN_UNFUSED"); return e && atoi(e) != 0; }();
    int n_wide = 0;
    for (int l = 1; l < p.n_trunk; ++l) n_wide += in_skips(f.d, l - 1) ? 1 : 0;
    const int WH = p.WH, NBK = WH / kWidth;
    if ((!sp || split_fused()) && !unfused && n_wide <= 1 && (WH == kWidth || (WH == 2 * kWidth && wide_fused()))) {      // (= the forward's condition for point-major hidden states)
        const int nT = p.n_trunk, nC = p.n_col;
        const size_t slot = (size_t)p.R * WH, packf = (size_t)WH * WH;
        const int B4 = 4 * kWidth;      // a 256-column block of a point-major [R, WH] matrix starts B4 x (block index) floats into a point
        if (int rc = ensure(ctx, ctx->tpack, (size_t)(nT + nC + 1) * packf * sizeof(float))) return rc;
        if (int rc = ensure(ctx, ctx->ttmp, ((size_t)(nT + nC + 1) * slot + (size_t)p.R * 2 * kLdNarrow) * sizeof(float))) return rc;
        float *tb = (float *)ctx->ttmp.p, *pack_at = (float *)ctx->tpack.p;
        auto dZc = [&](int l) { return tb + (size_t)l * slot; };
        auto dZt = [&](int l) { return tb + (size_t)(nC + l) * slot; };
        float *dFeat = tb + (size_t)(nC + nT) * slot, *GZH = dFeat + slot, *GCR = GZH + (size_t)p.R * kLdNarrow;
        auto next_pack = [&]() { float *r = pack_at; pack_at += packf; return r; };
        const float *PEs = ws + p.o_pes;
        TrainPointArgs a;
        point_args(a, f, p, ws);
        a.g_distance = g_distance; a.g_density = g_density; a.g_color = g_color; a.g_penalty = g_penalty; a.g_aux = g_aux_grad;
        a.GZH = GZH; a.GCR = GCR;
        launch_point_backward(a, s);
        PackBatch pbk;         // the transposed weight fragments of a chain: one launch per chain (fp32; split fp16: matrix by matrix)
        auto pack_t = [&](const float *src, int64_t sk, int64_t sn, int k_off, int n_off, int kcount, int ncount, int nout, float *dst) {
            if (sp) launch_pack(1, src, sk, sn, k_off, n_off, kcount, ncount, nout, dst, s);
            else pbk.add(src, sk, sn, k_off, n_off, kcount, ncount, nout, dst, s);
        };
        DwJobs dwj{};          // the weight-gradient products of this pass, job-parallel: one launch per kMaxDwJobs of them (launch_dw_jobs)
        dwj.R = p.R;
        // split fp16: every gradient matrix leaves max |dZ| in a device scalar (the chain kernel publishes it), its weight-gradient
        // products -- one launch each, after the chain that wrote the matrix -- scale their G operand by it
        AmaxSlots am;
        if (int rc = amax_begin(ctx, sp, am, s)) return rc;
        float *mZc[kMaxLayers] = {}, *mZt[kMaxLayers] = {};
        for (int l = 0; l < nC; ++l) mZc[l] = am.take();
        for (int l = 0; l < nT; ++l) mZt[l] = am.take();
        auto flush_dw = [&]() {
            if (sp) launch_dw_split_jobs(dwj, am.dw_tmp, ctx->cus, s);
            else launch_dw_jobs(dwj, ctx->cus, s);
        };
        // one 256 x 256 (or K x 256) weight-gradient product: G is a 256-column block of a point-major gradient matrix
        auto add_dw = [&](const float *X, int ldx, int K, int x_pm, const float *G, const float *amax_g, float *dW, int nvalid, float *db) {
            if (sp && (!split_dw_jobs() || WH != kWidth)) { launch_dw(1, X, ldx, K, G, WH, p.R, dW, WH, 1, nvalid, db, 4, ctx->cus, s, amax_g, am.dw_tmp, x_pm, 1); return; }
            // 512 columns (fp32 probe route): one launch per 256 x 256 block -- the job-parallel launch spreads 46 blocks of 2.2 GB
            // matrices over 8 workgroups each and ran at a sixth of its speed (287 ms per step against 45 ms of products)
            if (!sp && ((WH != kWidth && !wide_dw_jobs()) || (WH == kWidth && !dw_jobs_256()))) { launch_dw(0, X, ldx, K, G, WH, p.R, dW, WH, 1, nvalid, db, 4, ctx->cus, s, nullptr, nullptr, x_pm, 1); return; }
            if (dwj.n == kMaxDwJobs) { flush_dw(); dwj.n = 0; }        // (every G in the list has been produced: the jobs follow their chain)
            dwj.add(X, ldx, K, x_pm, G, WH, dW, WH, 1, nvalid, db, 4);
            dwj.job[dwj.n - 1].amax_g = amax_g;
        };
        // dW[row0 + k, n] += X^T G for a point-major hidden X [R, WH] / a narrow row-major X [R, ldx] (K columns), every 256 x 256 block
        auto dw_hidden = [&](const float *X, const float *G, const float *amax_g, float *dWrow0, float *db) {
            for (int nb = 0; nb < NBK; ++nb)
                for (int kb = 0; kb < NBK; ++kb)
                    add_dw(X + (size_t)B4 * kb, WH, kWidth, 1, G + (size_t)B4 * nb, amax_g, dWrow0 + (size_t)kb * kWidth * WH + nb * kWidth, kWidth,
                           (kb == 0 && db) ? db + nb * kWidth : nullptr);
        };
        auto dw_narrow = [&](const float *X, int ldx, int K, const float *G, const float *amax_g, float *dWrow0, float *db) {
            for (int nb = 0; nb < NBK; ++nb)
                add_dw(X, ldx, K, 0, G + (size_t)B4 * nb, amax_g, dWrow0 + nb * kWidth, kWidth, db ? db + nb * kWidth : nullptr);
        };
        // heads' weight gradients: X point-major, one launch per 256-column block of it
        auto heads_dw = [&](const float *X, const float *G, int nc, float *const *w, int wstride, float *const *b) {
            for (int kb = 0; kb < NBK; ++kb) {
                float *wk[4] = { nullptr, nullptr, nullptr, nullptr };
                for (int c = 0; c < nc; ++c) wk[c] = w[c] + (size_t)kb * kWidth * wstride;
                launch_narrow_dw(X + (size_t)B4 * kb, WH, G, kLdNarrow, p.R, nc, wk, wstride, kb == 0 ? b : nullptr, 4, kWidth, s, 1);
            }
        };
        // colour head: LinearGradFunction.backward (linear.py:62-88) on [HC | JC] rows, then the last colour activation
        NarrowW cout{};
        cout.nc = 3; cout.wstride = 3; cout.kcount = kWidth;
        for (int c = 0; c < 3; ++c) cout.w[c] = W[p.i_cout] + c;
        const float *HClast = ws + p.o_hc[nC - 1], *Hlast = ws + p.o_h[nT - 1];
        {
            float *wc[3] = { gW[p.i_cout], gW[p.i_cout] + 1, gW[p.i_cout] + 2 }, *bc[3] = { gB[p.i_cout], gB[p.i_cout] + 1, gB[p.i_cout] + 2 };
            heads_dw(HClast, GCR, 3, wc, 3, bc);
        }
        {   // colour trunk: dZ of every layer in one kernel
            MlpBackwardArgs m{};
            m.R = p.R; m.n_layers = nC; m.act_kind = act; m.width = WH;
            // prologue: dZ of the last colour layer = activation backward of the head's upstream gradient (3 raw colour columns)
            m.top_G = GCR; m.top_ldg = kLdNarrow; m.top_nc = 3; m.top_wstride = 3;
            for (int c = 0; c < 3; ++c) m.top_w[c] = cout.w[c];
            m.top_Z = ws + p.o_zc[nC - 1]; m.top_out = dZc(nC - 1); m.amax_top = mZc[nC - 1];
            for (int l = 1; l < nC; ++l) {
                float *wl = next_pack();
                pack_t(W[nT + l], 1, WH, 0, 0, WH, WH, WH, wl);             // W_l^T
                m.wT[l] = wl; m.Z[l - 1] = ws + p.o_zc[l - 1]; m.dZ[l - 1] = dZc(l - 1); m.amax_dZ[l - 1] = mZc[l - 1];
            }
            pbk.flush(s);
            launch_mlp_backward(sp, m, ctx->cus, s);
        }
        for (int l = nC - 1; l >= 1; --l) dw_hidden(ws + p.o_hc[l - 1], dZc(l), mZc[l], gW[nT + l], gB[nT + l]);
        dw_narrow(ws + p.o_xa, p.ldxa, p.Ca, dZc(0), mZc[0], gW[nT], gB[nT]);
        dw_hidden(Hlast, dZc(0), mZc[0], gW[nT] + (size_t)p.Ca * WH, nullptr);
        {
            float *wh[2] = { gW[p.i_ddf], gW[p.i_aux] }, *bh[2] = { gB[p.i_ddf], gB[p.i_aux] };
            heads_dw(Hlast, GZH, 2, wh, 1, bh);
        }
        {   // distance trunk
            MlpBackwardArgs m{};
            m.R = p.R; m.n_layers = nT; m.act_kind = act; m.width = WH;
            // prologue: dZ of the last trunk layer = activation backward of (gradient of the features from the colour trunk -- only the
            // feature segment of its first layer propagates: the small colour inputs carry no parameters -- + the distance / aux heads)
            float *wf = next_pack();
            pack_t(W[nT], 1, WH, 0, p.Ca, WH, WH, WH, wf);                  // (feature rows of W_c0)^T
            m.top_src = dZc(0); m.top_wT = wf;
            m.top_G = GZH; m.top_ldg = kLdNarrow; m.top_nc = 2; m.top_wstride = 1;
            m.top_w[0] = W[p.i_ddf]; m.top_w[1] = W[p.i_aux];
            m.top_Z = ws + p.o_z[nT - 1]; m.top_out = dZt(nT - 1); m.amax_top = mZt[nT - 1];
            for (int l = 1; l < nT; ++l) {
                const bool wide = in_skips(f.d, l - 1);
                float *wl = next_pack();
                pack_t(W[l], 1, WH, 0, wide ? p.Cpe : 0, WH, WH, WH, wl);   // (hidden rows of W_l)^T
                m.wT[l] = wl; m.Z[l - 1] = ws + p.o_z[l - 1]; m.dZ[l - 1] = dZt(l - 1); m.amax_dZ[l - 1] = mZt[l - 1];
            }
            pbk.flush(s);
            launch_mlp_backward(sp, m, ctx->cus, s);
        }
        for (int l = nT - 1; l >= 0; --l) {
            const bool wide = l > 0 && in_skips(f.d, l - 1);
            if (l == 0 || wide) dw_narrow(PEs, kLdPe, p.Cpe, dZt(l), mZt[l], gW[l], gB[l]);
            if (l > 0) dw_hidden(ws + p.o_h[l - 1], dZt(l), mZt[l], gW[l] + (size_t)(wide ? p.Cpe : 0) * WH, wide ? nullptr : gB[l]);
        }
        if (dwj.overflow) return fail(ctx, NEDDF_EUNSUPPORTED, "more weight-gradient products than DwJobs holds (train_kernels.h kMaxDwJobs)");
        flush_dw();
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (int rc = ensure(ctx, ctx->tpack, 2 * kPackFloats * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->ttmp, (size_t)p.R * (2 * WH + 2 * kLdNarrow) * sizeof(float))) return rc;
    float *wp = (float *)ctx->tpack.p;
    float *dA = (float *)ctx->ttmp.p, *dB = dA + (size_t)p.R * WH, *GZH = dB + (size_t)p.R * WH, *GCR = GZH + (size_t)p.R * kLdNarrow;
    const float *PEs = ws + p.o_pes;
    TrainPointArgs a;
    point_args(a, f, p, ws);
    a.g_distance = g_distance; a.g_density = g_density; a.g_color = g_color; a.g_penalty = g_penalty; a.g_aux = g_aux_grad;
    a.GZH = GZH; a.GCR = GCR;
    launch_point_backward(a, s);
    AmaxSlots am;
    if (int rc = amax_begin(ctx, sp, am, s)) return rc;
    // Every product of this route is cut into 256 x 256 blocks (NBK = 1 at hidden width 256, 2 at 512).  dA always holds dZ of the layer
    // in flight ([R, WH]); every activation backward is fused into the kernel that produces its upstream gradient.
    // dW[row_off + k, n] += X[R, Kin]^T dA (+ db): Kin <= 256 (an encoding segment) or Kin = WH
    auto dw_blocks = [&](const float *X, int ldx, int Kin, float *gWl, int row_off, float *gBl, const float *mA) {
        const int KB = Kin <= kWidth ? 1 : Kin / kWidth, kc = Kin <= kWidth ? Kin : kWidth;
        for (int nb = 0; nb < NBK; ++nb)
            for (int kb = 0; kb < KB; ++kb)
                launch_dw(sp, X + kb * kWidth, ldx, kc, dA + nb * kWidth, WH, p.R, gWl + (size_t)(row_off + kb * kWidth) * WH + nb * kWidth, WH, 1, kWidth,
                          (kb == 0 && gBl) ? gBl + nb * kWidth : nullptr, 4, ctx->cus, s, mA, am.dw_tmp);
    };
    // dB[R, WH] = [activation backward with Zprev of] dA x (rows row_off .. row_off + WH of the [in, WH] weight)^T: K blocks accumulate
    // in dB, the activation backward runs with the last one
    auto gemm_bw = [&](const float *Wsrc, int row_off, int act_kind, const float *Zprev, const float *mIn, float *mOut) {
        for (int nb = 0; nb < NBK; ++nb)
            for (int kb = 0; kb < NBK; ++kb) {
                launch_pack(sp, Wsrc, 1, WH, kb * kWidth, row_off + nb * kWidth, kWidth, kWidth, kWidth, wp, s);        // W^T block
                if (Zprev && kb == NBK - 1)
                    launch_rows_gemm_actback(sp, dA + kb * kWidth, p.R, WH, kWidth, wp, gemm_ksteps(kWidth, sp), 4, act_kind, Zprev + nb * kWidth,
                                             dB + nb * kWidth, WH, ctx->cus, s, mIn, mOut, kb > 0);
                else
                    launch_rows_gemm(sp, dA + kb * kWidth, p.R, WH, kWidth, wp, gemm_ksteps(kWidth, sp), nullptr, 4, dB + nb * kWidth, WH, kb > 0, -1, nullptr,
                                     ctx->cus, s, mIn);
            }
    };
    // dZ[R, WH] = activation backward with Zprev of ((accumulate ? dH : 0) + sum_c G[., c] w_c), and the narrow columns' weight gradients
    auto narrow_bw_act = [&](const float *G, NarrowW w, const float *dH, int accumulate, const float *Zprev, float *dZ, float *mOut) {
        const float *w0[4] = { w.w[0], w.w[1], w.w[2], w.w[3] };
        for (int kb = 0; kb < NBK; ++kb) {
            for (int c = 0; c < w.nc; ++c) w.w[c] = w0[c] + (size_t)kb * kWidth * w.wstride;
            launch_narrow_backward_act(G, kLdNarrow, p.R, w, dH ? dH + kb * kWidth : nullptr, accumulate, act, 4, Zprev + kb * kWidth, dZ + kb * kWidth, WH, s, mOut);
        }
    };
    auto narrow_dw = [&](const float *X, const float *G, int nc, float *const *w, int wstride, float *const *b) {
        for (int kb = 0; kb < NBK; ++kb) {
            float *wk[4] = { nullptr, nullptr, nullptr, nullptr };
            for (int c = 0; c < nc; ++c) wk[c] = w[c] + (size_t)kb * kWidth * wstride;
            launch_narrow_dw(X + kb * kWidth, WH, G, kLdNarrow, p.R, nc, wk, wstride, kb == 0 ? b : nullptr, 4, kWidth, s);
        }
    };
    // colour head: LinearGradFunction.backward (linear.py:62-88) on [HC | JC] rows
    NarrowW cout{};
    cout.nc = 3; cout.wstride = 3; cout.kcount = kWidth;
    for (int c = 0; c < 3; ++c) cout.w[c] = W[p.i_cout] + c;
    const float *HClast = ws + p.o_hc[p.n_col - 1];
    float *mA = am.take();          // max |dA| of the gradient matrix currently in dA
    narrow_bw_act(GCR, cout, nullptr, 0, ws + p.o_zc[p.n_col - 1], dA, mA);
    {
        float *wc[3] = { gW[p.i_cout], gW[p.i_cout] + 1, gW[p.i_cout] + 2 }, *bc[3] = { gB[p.i_cout], gB[p.i_cout] + 1, gB[p.i_cout] + 2 };
        narrow_dw(HClast, GCR, 3, wc, 3, bc);
    }
    const float *Hlast = ws + p.o_h[p.n_trunk - 1];
    for (int l = p.n_col - 1; l >= 0; --l) {
        const float *Wl = W[p.n_trunk + l];
        float *gWl = gW[p.n_trunk + l], *gBl = gB[p.n_trunk + l];
        if (l > 0) {
            dw_blocks(ws + p.o_hc[l - 1], WH, WH, gWl, 0, gBl, mA);
            float *mB = am.take();
            gemm_bw(Wl, 0, act, ws + p.o_zc[l - 1], mA, mB);
            mA = mB;
        } else {
            dw_blocks(ws + p.o_xa, p.ldxa, p.Ca, gWl, 0, gBl, mA);
            dw_blocks(Hlast, WH, WH, gWl, p.Ca, nullptr, mA);
            // the small colour inputs (encodings, detached normal) carry no parameters: only the feature segment propagates
            gemm_bw(Wl, p.Ca, -1, nullptr, mA, nullptr);
        }
        float *t = dA; dA = dB; dB = t;
    }
    // dA = gradient of the trunk features from the colour trunk; add the distance / aux heads, then the last trunk activation
    NarrowW heads{};
    heads.nc = 2; heads.wstride = 1; heads.kcount = kWidth;
    heads.w[0] = W[p.i_ddf]; heads.w[1] = W[p.i_aux];
    mA = am.take();
    narrow_bw_act(GZH, heads, dA, 1, ws + p.o_z[p.n_trunk - 1], dA, mA);
    {
        float *wh[2] = { gW[p.i_ddf], gW[p.i_aux] }, *bh[2] = { gB[p.i_ddf], gB[p.i_aux] };
        narrow_dw(Hlast, GZH, 2, wh, 1, bh);
    }
    // distance trunk
    for (int l = p.n_trunk - 1; l >= 0; --l) {
        const bool wide = l > 0 && in_skips(f.d, l - 1);
        if (l == 0) {
            dw_blocks(PEs, kLdPe, p.Cpe, gW[0], 0, gB[0], mA);
            break;
        }
        if (wide) {
            dw_blocks(PEs, kLdPe, p.Cpe, gW[l], 0, gB[l], mA);
            dw_blocks(ws + p.o_h[l - 1], WH, WH, gW[l], p.Cpe, nullptr, mA);
        } else dw_blocks(ws + p.o_h[l - 1], WH, WH, gW[l], 0, gB[l], mA);
        float *mB = am.take();
        gemm_bw(W[l], wide ? p.Cpe : 0, act, ws + p.o_z[l - 1], mA, mB);
        float *t = dA; dA = dB; dB = t;
        mA = mB;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int neddf_composite_backward(neddf_ctx *ctx, const float *dists, const float *density, const float *color, int64_t n_rays, int S,
                             float max_dist, const float *g_weight, const float *g_depth, const float *g_color, const float *g_trans,
                             float *g_density, float *g_color_out, void *stream)
{
    if (!ctx) return NEDDF_EINVAL;
    if (n_rays <= 0) return 0;
    if (!dists || !density || !color || !g_density || !g_color_out || S < 2) return fail(ctx, NEDDF_EINVAL, "bad argument");
    DeviceGuard guard_(ctx->device);
    launch_composite_backward(dists, density, color, n_rays, S, max_dist, g_weight, g_depth, g_color, g_trans, g_density, g_color_out,
                              (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"
