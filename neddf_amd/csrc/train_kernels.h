// train_kernels.h -- launchers of the training-step kernels (train_kernels.hip).
#pragma once
#include "kernels.h"

namespace neddf {

// per-point record kept between the forward and the backward pass (float index)
enum { TP_ZD0 = 0,   // raw distance head rows (value, d/dx, d/dy, d/dz)
       TP_ZA0 = 4,   // raw aux head rows
       TP_D = 8, TP_RHO = 9, TP_AUX = 10, TP_U = 11, TP_DGN = 12, TP_DDDT = 13,
       TP_DG0 = 14,  // distance gradient (3)
       TP_ND0 = 17,  // normalised distance gradient (3)
       TP_AGG0 = 20  // aux-gradient gradient (3)
};
constexpr int kTrainPt = 24;

struct TrainPointArgs {
    int64_t N;
    EncodeDesc enc;
    int density_activation;
    float d_near, aux_grad_scale, distance_range_max;
    float penalty_weight[6];
    int penalty_has[6];
    // forward
    const float *ZH; int ldh;             // raw head rows [4N, ldh]: col 0 distance head, col 1 aux head
    const float *PEu; int ldpe;           // unscaled position encoding rows [4N, ldpe]
    const float *Ed; int ldd;             // direction encoding [N, ldd]
    float *XA; int ldxa;                  // colour-trunk small input rows [4N, ldxa] = [embed_pos | embed_dir | normal]
    float *PT;                            // [N, kTrainPt]
    const float *CR; int ldc;             // colour rows [4N, ldc] (cols 0..2)
    float *distance, *density, *aux_grad, *color, *penalty;      // outputs [N] / [N,3]
    // backward
    const float *g_distance, *g_density, *g_aux, *g_color, *g_penalty;
    float *GZH;                           // [4N, ldh]
    float *GCR;                           // [4N, ldc]
};

// logical M[k][n] = src[(k_off + k) * sk + (n_off + n) * sn] for k < kcount, n < ncount (0 elsewhere), packed into the
// fragment-major layout of kernels.h LayerW for nout (128 or 256) output columns and roundup(kcount, 8) / 8 super-steps
// split != 0: split-fp16 fragments (tile_engine.h OpsF16Split), roundup(kcount, 16) / 16 super-steps
void launch_pack(int split, const float *src, int64_t sk, int64_t sn, int k_off, int n_off, int kcount, int ncount, int nout, float *dst,
                 hipStream_t s);
// the same for every matrix of a layer stack in ONE launch (fp32 fragments; the per-matrix launches of a step were 42 x 5 us in a row)
struct PackJob { const float *src; int64_t sk, sn; int k_off, n_off, kcount, ncount, nout, ks; float *dst; };
constexpr int kMaxPackJobs = 16;
struct PackBatch {
    int n = 0;
    PackJob job[kMaxPackJobs];
    void add(const float *src, int64_t sk, int64_t sn, int k_off, int n_off, int kcount, int ncount, int nout, float *dst, hipStream_t s);
    void flush(hipStream_t s);
};
inline int gemm_ksteps(int K, int split) { return split ? (K + 15) / 16 : (K + 7) / 8; }

// Y[R,256] (+)= X[R, 0:kload) x Wpacked (+ bias); kload = loaded width (multiple of 4, <= ldx, zero beyond the logical K);
// act_kind >= 0 with H != NULL additionally writes H = a(Y) on (value, Jacobian) row groups
// amax_in / amax_out / amax_g (device scalars or NULL): range scaling of gradient operands under the split-fp16 policy -- a kernel that
// writes a gradient matrix leaves max |dZ| in amax_out, a split-operand GEMM that consumes it scales by the matching power of two
// (train_kernels.hip operand_scale); ignored by the fp32 MFMA kernels
void launch_rows_gemm(int split, const float *X, int64_t R, int ldx, int kload, const float *wp, int ksteps, const float *bias, int bias_period,
                      float *Y, int ldy, int accumulate, int act_kind, float *H, int cus, hipStream_t s, const float *amax_in = nullptr);
// Forward of a whole stack of (value, Jacobian)-row layers in ONE kernel (round 2): the 64-row tile stays in LDS across the layers
// like in the inference kernels, and every layer leaves its pre-activations Z_l and activations H_l in the workspace from the
// epilogue (what the hand-written backward passes consume) -- 2 KB of HBM traffic per row and layer instead of the 3 KB of one
// GEMM kernel per layer (read H_{l-1}, write Z_l, write H_l), and no per-layer launch / fill / drain.
//   layer 0:  Z_0 = X0[R, 0:kload0) x wp0 (+ X1[R, 256] x wp1) + bias_0        (X1: the colour trunk's 256 trunk features)
//   layer l:  Z_l = H_{l-1} x wp[l] (+ X0 x wp_skip if l == skip_layer) + bias_l     (skip: cat([X0, h]) of neddf.py:217-219)
//   H_l = a(Z_l) on (value, 3 Jacobian) row groups; biases on value rows only (linear.py:43-45)
struct MlpForwardArgs {
    int64_t R;
    const float *X0; int ld0, kload0;     // kload0: loaded width, multiple of 4, zero beyond the logical K
    const float *wp0; int ksteps0;
    const float *X1; const float *wp1;    // optional second input of layer 0 (ld 256, 256 columns) and its packed weights
    int n_layers;
    const float *wp[kMaxLayers];          // [l >= 1] packed 256 x 256
    const float *bias[kMaxLayers];        // [l >= 0] device pointers, 256 floats
    int skip_layer;                       // -1, or the layer that also consumes X0
    const float *wp_skip;                 // its X0 segment (ksteps0 super-steps)
    float *Z[kMaxLayers], *H[kMaxLayers]; // [R, 256] each
    int act_kind;
    // Point-major layout of the [R, 256] matrices (X1, Z, H; fp32 policy, R a multiple of 4): element (row, c) sits at
    //   (row >> 2) * 1024 + 4 c + (row & 3)
    // i.e. the four rows of a point -- (value, d/dx, d/dy, d/dz) -- of one feature are 16 contiguous bytes.  That is exactly what a lane
    // of the 32x32 accumulator layout holds, so the epilogues store (and the backward chain loads) 16 bytes per lane and instruction
    // instead of 4: the row-major side stores of a layer were 128 + 64 scalar memory instructions per wave behind 512 MFMAs, 3.8 of the
    // forward's 13 ms per training step (profiles/r03_train_ablation.txt).  The NeDDF fp32 training route uses it end to end
    // (forward, backward chain, weight gradients, heads); every other route keeps row-major matrices.
    int point_major;
    int width;                            // 0 / 256: the [R, 256] matrices above; 512 (round 5; point-major matrices; fp32 and split fp16): [R, 512] matrices, 512 x 512 packed weights
};
void launch_mlp_forward(int split, const MlpForwardArgs &a, int cus, hipStream_t s);

// Input-gradient chain of a whole stack of (value, Jacobian)-row layers in ONE kernel (round 3), the mirror image of
// launch_mlp_forward: the 64-row tile of gradients stays in LDS from the top layer down,
//   dZ_{l-1} = act_backward(Z_{l-1}; dZ_l x (hidden rows of W_l)^T)          l = n_layers-1 .. 1
// (LinearGradFunction.backward linear.py:62-74 followed by the activation's backward, e.g. tanh_exp.py:57-88), and every dZ_l
// is left in its own [R, 256] matrix for the weight-gradient products that follow.  EVERY [R, 256] matrix of this kernel (dZtop, top_src,
// top_Z, top_out, Z, dZ) is in the POINT-MAJOR layout (MlpForwardArgs.point_major).  Per row and layer the chain reads Z_{l-1}
// and writes dZ_{l-1} (2 KB) where one GEMM kernel per layer also re-read dZ_l (3 KB), and there is no per-layer launch, fill and
// drain.  The activation backward runs on the accumulators themselves: in the 32x32 layout a lane holds the four rows of a point
// for one feature, and Z_{l-1} is requested in that layout before the product.  fp32 MFMA operands (the split-fp16 policy
// keeps the per-layer route: its gradient operands are range-scaled per matrix).
struct MlpBackwardArgs {
    int64_t R;
    const float *dZtop;                   // [R, 256]: gradient of the top layer's pre-activations -- or NULL: formed in the kernel's prologue,
    //   dZ_top = act_backward(Z_top; top_src x top_wT + sum_c top_G[:, c] top_w[c][:])     and left in top_out for its weight gradient:
    const float *top_src, *top_wT;        // optional [R, 256] matrix and packed 256 x 256 weights (the colour trunk's gradient of the features)
    const float *top_G; int top_ldg, top_nc;              // narrow upstream gradient [R, top_ldg], top_nc <= 3 columns (the heads' raw outputs)
    const float *top_w[3]; int top_wstride;               // column c of the heads' weights: top_w[c][feature * top_wstride]
    const float *top_Z;                   // pre-activations of the top layer [R, 256]
    float *top_out;                       // [R, 256]
    int n_layers;
    const float *wT[kMaxLayers];          // [l >= 1] packed (hidden rows of W_l)^T, 256 x 256
    const float *Z[kMaxLayers];           // pre-activations [R, 256]; l = 0 .. n_layers-2 are read
    float *dZ[kMaxLayers];                // outputs [R, 256], l = 0 .. n_layers-2
    int act_kind;                         // backward kind of act_grad2 (0 ReLU, 1 LeakyReLU, 2 tanhExp, 3 tanhExp as NeuS differentiates it)
    // split-fp16 policy (launch_mlp_backward(split = 1, ...), prologue form only): top_wT / wT are split-packed, every product's gradient
    // operand is range-scaled per 64-row tile inside the kernel, and each gradient matrix leaves max |dZ| in a device scalar for the
    // weight-gradient products that follow (launch_dw's amax_g); NULL slots are skipped
    float *amax_dZ[kMaxLayers];           // for dZ[l]
    float *amax_top;                      // for top_out
    int width;                            // 0 / 256, or 512 (round 5; fp32 and split fp16): every [R, 256] above is [R, 512], every packed matrix 512 x 512
};
void launch_mlp_backward(int split, const MlpBackwardArgs &a, int cus, hipStream_t s);

// dW[k * sk + n * sn] += sum_r X[r, k] G[r, n], k < K <= 256, n < nvalid <= 256; db[n] += sum over rows r % bias_period == 0 of G[r, n]
// dZ[R,256] = activation backward (pre-activations Zprev, row period, kind) of X[R, 0:kload) x Wpacked: the input-gradient GEMM of
// a layer fused with the backward of the previous layer's activation
void launch_rows_gemm_actback(int split, const float *X, int64_t R, int ldx, int kload, const float *wp, int ksteps, int period, int act_kind,
                              const float *Zprev, float *dZ, int ldy, int cus, hipStream_t s, const float *amax_in = nullptr,
                              float *amax_out = nullptr, int accumulate = 0);      // accumulate: dZ holds earlier K blocks' partial products
// All weight-gradient products of a backward pass in one launch (fp32 MFMA policy): the workgroups are divided among the products in
// proportion to their cost, each accumulates ITS product over its share of the rows and adds it to dW once.
struct DwJob {
    const float *X; int ldx, K;           // dW[k * sk + n * sn] += sum_r X[r, k] G[r, n], k < K <= 256, n < nvalid <= 256
    const float *G; int ldg;
    float *dW; int64_t sk, sn; int nvalid;
    float *db; int bias_period;           // db[n] += sum over rows r % bias_period == 0 of G[r, n] (or NULL)
    int x_point_major;                    // X is a point-major [R, 256] matrix (MlpForwardArgs.point_major); G always is, R % 4 == 0
    int wg0;                              // first workgroup of the product (set by launch_dw_jobs)
    // split-fp16 policy (launch_dw_split_jobs): max |G| of the gradient matrix (device scalar, or NULL) and a zeroed [256, 256] scratch
    // the range-scaled product lands in before it is added to dW divided by the scale
    const float *amax_g;
    float *tmp;
};
constexpr int kMaxDwJobs = 32;
// the fused NeDDF backward registers one job per trunk layer (distance + colour trunks: <= 2 * kMaxLayers), the wide halves of
// skip layers and the first layers' encoding segments; a job that did not fit would leave a weight gradient silently at zero
static_assert(kMaxDwJobs >= 2 * kMaxLayers + 4, "DwJobs must hold every weight-gradient product of a pass");
struct DwJobs {
    int n;
    bool overflow = false;               // add() past kMaxDwJobs: launch_dw_jobs' caller returns NEDDF_EUNSUPPORTED
    int64_t R;                            // rows of every X / G
    DwJob job[kMaxDwJobs];
    void add(const float *X, int ldx, int K, int x_point_major, const float *G, int ldg, float *dW, int64_t sk, int64_t sn, int nvalid, float *db,
             int bias_period)
    {
        if (n < kMaxDwJobs) job[n++] = DwJob{ X, ldx, K, G, ldg, dW, sk, sn, nvalid, db, bias_period, x_point_major, 0, nullptr, nullptr };
        else overflow = true;
    }
};
void launch_dw_jobs(DwJobs &jobs, int cus, hipStream_t s);
// the same list under the split-fp16 policy (round 5): three fp16 MFMAs per multiply-add on operands of two fp16 terms, G range-scaled by
// its job's amax_g; `tmp` = jobs.n x [256, 256] floats of scratch (zeroed here), one unscale-and-add pass over all jobs at the end
void launch_dw_split_jobs(DwJobs &jobs, float *tmp, int cus, hipStream_t s);
void launch_dw(int split, const float *X, int ldx, int K, const float *G, int ldg, int64_t R, float *dW, int64_t sk, int64_t sn, int nvalid, float *db,
               int bias_period, int cus, hipStream_t s, const float *amax_g = nullptr, float *scaled_tmp = nullptr, int x_point_major = 0,
               int g_point_major = 0);        // (point-major operands: both policies -- the split-fp16 fused route and the fp32 512-wide one --, R % 4 == 0)
// scaled_tmp: [256, 256] floats of scratch, required with amax_g (the scaled product is formed there, then added to dW unscaled)
// heads (1..4 output columns, input width 256): column c of the weight gradient is w[c][k * wstride], b[c] its bias gradient
void launch_narrow_dw(const float *X, int ldx, const float *G, int ldg, int64_t R, int nc, float *const *w, int wstride, float *const *b,
                      int bias_period, int kcount, hipStream_t s, int x_point_major = 0);
void launch_pe_values(const float *pos, const float *dir, const float *var, int64_t N, const EncodeDesc &enc, float *PE, int ld, float *Ed, int ldd,
                      hipStream_t s);
void launch_density_head(int kind, const float *z, int ldz, int64_t N, const float *g, float *out, int ldo, hipStream_t s);
void launch_copy3(const float *in, int ldi, float *out, int ldo, int64_t N, hipStream_t s);
void launch_pe_rows(const float *pos, const float *dir, const float *var, int64_t N, const EncodeDesc &enc, float *PEs, float *PEu, int ld,
                    float *Ed, int ldd, hipStream_t s);
// Y[R, ldy] cols 0..nc-1 = X[R, 256] . w_c (+ b_c on rows r % bias_period == 0); column c of the weights is wcol[c][k * wstride]
struct NarrowW {
    int nc;
    const float *w[4];
    int wstride;
    const float *b[4];        // device pointers to the scalar biases (or NULL)
    int kcount;               // input features present (<= 256; columns beyond are not read)
};
void launch_narrow_forward(const float *X, int ldx, int64_t R, const NarrowW &w, int bias_period, float *Y, int ldy, hipStream_t s,
                           int x_point_major = 0, int accumulate = 0);       // accumulate: Y += (one more 256-column block of a wider input)
// dX[R, ldx] (+)= sum_c G[r, c] * w_c[k]
void launch_narrow_backward(const float *G, int ldg, int64_t R, const NarrowW &w, float *dX, int ldx, int accumulate, hipStream_t s);

// dZ = activation backward (Zprev) of ((accumulate ? dH : 0) + sum_c G[., c] w_c); dH and dZ may alias
void launch_narrow_backward_act(const float *G, int ldg, int64_t R, const NarrowW &w, const float *dH, int accumulate, int act_kind, int period,
                                const float *Zprev, float *dZ, int ld, hipStream_t s, float *amax_out = nullptr);
// NeuS heads (train_kernels.hip "NeuS"): per-point pieces between the sdf trunk and the colour trunk
constexpr int kActTanhExp = 2;                 // = NEDDF_ACT_TANHEXP (include/neddf_hip.h; checked in train_capi.hip)
struct NeusPointArgs {
    int64_t N;
    int act;                              // hidden activation (also applied to the colour outputs)
    int Cdir;                             // 6 * embed_dir_rank
    const float *variance;                // device scalar (the `variance` parameter)
    const float *pos;                     // [N, 3]
    const float *Ed; int ldd;             // direction encoding [N, ldd]
    const float *Hlast, *Zlast;           // last sdf layer, activated / pre-activation, [4N, ldh] (value + Jacobian rows)
    int ldh;                              // hidden width the kernels see: 256 or 512 (also the row stride of dF and dZ)
    float *XA; int ldxa;                  // colour-trunk small input [N, ldxa] = [pos | embed_dir | gradient | 0]
    const float *ZC; int ldc;             // raw colour rows [N, ldc] (cols 0..2)
    float *sdf, *density, *color;         // outputs [N], [N], [N, 3]
    // backward
    const float *g_sdf, *g_density, *g_color;
    float *GC;                            // [N, ldc]: gradient of the raw colour rows
    const float *dF;                      // [N, ldh]: gradient of the features from the colour trunk
    const float *DG; int lddg;            // [N, lddg] (cols 0..2): gradient of the normal from the colour trunk
    float *dZ;                            // [4N, ldh]
    float *g_variance;                    // accumulated
    float *amax_out;                      // max |dZ| (range scaling of split-fp16 operands) or NULL
};
void launch_neus_head_forward(const NeusPointArgs &a, hipStream_t s);
void launch_neus_color_forward(const NeusPointArgs &a, hipStream_t s);
void launch_neus_color_backward(const NeusPointArgs &a, hipStream_t s);
void launch_neus_head_backward(const NeusPointArgs &a, hipStream_t s);
// activation id for the backward of NeuS' (value, Jacobian) rows: tanhExp maps to the reference's double-backward form
int neus_backward_act_kind(int act);
void launch_point_forward(const TrainPointArgs &a, hipStream_t s);
void launch_penalty_forward(const TrainPointArgs &a, hipStream_t s);
void launch_point_backward(const TrainPointArgs &a, hipStream_t s);
void launch_composite_backward(const float *dists, const float *dens, const float *col, int64_t n, int S, float max_dist,
                               const float *g_weight, const float *g_depth, const float *g_color, const float *g_trans, float *g_dens,
                               float *g_col, hipStream_t s);

}  // namespace neddf
