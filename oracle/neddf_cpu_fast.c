/*
 * neddf_cpu_fast.c -- the CPU BASELINE of bench.py: the eval-minimal NeDDF field (what the HIP path computes for rendering)
 * written the way a CPU implementation would be, not the way the reference is.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (like neddf_oracle.c: only tests/, smoke() and bench.py's cpu_baseline leg load it).
 *
 * neddf_oracle.c restates the reference operation by operation (Jacobian rows through BOTH trunks, penalties: 5.15 MFLOP per
 * point, point-at-a-time loops) -- it is the parity checker.  As a *baseline* that flatters the GPU: here the same function
 * values are computed with the algorithm of the HIP kernels -- value rows forward, the distance gradient in reverse mode
 * (neddf.py:206-241 needs the position gradient of ONE scalar), the colour trunk on value rows (its Jacobian is dead code
 * in eval, neddf.py:243-257) = 2.14 MFLOP per point at the shipped architecture -- on blocks of 64 points with a register-blocked
 * single-precision GEMM micro-kernel (AVX-512 when the host has it, AVX2 + FMA otherwise) and OpenMP over blocks.
 * Summation order differs from the reference (like the MFMA's does); tests/test_oracle.py holds it to the 1e-4 gates
 * against neddf_oracle.c.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FAST_MAX_LAYERS 16
enum { ORC_RELU = 0, ORC_LEAKY = 1, ORC_TANHEXP = 2 };

/* same layout as orc_neddf_t (neddf_oracle.c): the Python side fills one struct for both */
typedef struct {
    int embed_pos_rank, embed_dir_rank;
    int n_ddf, ddf_width, n_col, col_width;
    int n_skips;
    int skips[8];
    int activation, density_activation;
    float d_near, aux_grad_scale, distance_range_max;
    float penalty_weight[6];
    int penalty_has[6];
    const float *lowpass;
    const float *ddf_w[FAST_MAX_LAYERS], *ddf_b[FAST_MAX_LAYERS];
    const float *col_w[FAST_MAX_LAYERS], *col_b[FAST_MAX_LAYERS];
    const float *ddf_out_w, *ddf_out_b, *aux_out_w, *aux_out_b, *col_out_w, *col_out_b;
} fast_neddf_t;

/* ---- C[M][N] (+)= A[M][K] x B[K][N], row-major, N a multiple of 16 (callers pad).  MR x NR register tile, k innermost. */
#define GEMM_BODY(MR, NR)                                                                                             \
    for (int i0 = 0; i0 < M; i0 += MR) {                                                                               \
        const int mr = M - i0 < MR ? M - i0 : MR;                                                                      \
        for (int j0 = 0; j0 < N; j0 += NR) {                                                                           \
            float acc[MR][NR];                                                                                         \
            for (int i = 0; i < MR; ++i)                                                                               \
                for (int j = 0; j < NR; ++j) acc[i][j] = (accumulate && i < mr) ? C[(size_t)(i0 + i) * ldc + j0 + j] : 0.f; \
            if (mr == MR) {                                                                                            \
                for (int k = 0; k < K; ++k) {                                                                          \
                    const float *b = B + (size_t)k * ldb + j0;                                                         \
                    for (int i = 0; i < MR; ++i) {                                                                     \
                        const float a = A[(size_t)(i0 + i) * lda + k];                                                 \
                        _Pragma("omp simd") for (int j = 0; j < NR; ++j) acc[i][j] += a * b[j];                        \
                    }                                                                                                  \
                }                                                                                                      \
            } else {                                                                                                   \
                for (int k = 0; k < K; ++k) {                                                                          \
                    const float *b = B + (size_t)k * ldb + j0;                                                         \
                    for (int i = 0; i < mr; ++i) {                                                                     \
                        const float a = A[(size_t)(i0 + i) * lda + k];                                                 \
                        _Pragma("omp simd") for (int j = 0; j < NR; ++j) acc[i][j] += a * b[j];                        \
                    }                                                                                                  \
                }                                                                                                      \
            }                                                                                                          \
            for (int i = 0; i < mr; ++i)                                                                               \
                for (int j = 0; j < NR; ++j) C[(size_t)(i0 + i) * ldc + j0 + j] = acc[i][j];                           \
        }                                                                                                              \
    }

__attribute__((target("avx512f,avx512vl,fma")))
static void gemm_avx512(const float *A, int lda, const float *B, int ldb, float *C, int ldc, int M, int K, int N, int accumulate)
{
    GEMM_BODY(8, 32)         /* 16 zmm accumulators */
}

static void gemm_avx2(const float *A, int lda, const float *B, int ldb, float *C, int ldc, int M, int K, int N, int accumulate)
{
    GEMM_BODY(6, 16)         /* 12 ymm accumulators */
}

static int g_have512 = -1;
static void gemm(const float *A, int lda, const float *B, int ldb, float *C, int ldc, int M, int K, int N, int accumulate)
{
    if (g_have512 < 0) { __builtin_cpu_init(); g_have512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl"); }
    if (g_have512 && N % 32 == 0) gemm_avx512(A, lda, B, ldb, C, ldc, M, K, N, accumulate);
    else gemm_avx2(A, lda, B, ldb, C, ldc, M, K, N, accumulate);
}
int fast_uses_avx512(void) { gemm(0, 0, 0, 0, 0, 0, 0, 0, 32, 0); return g_have512; }

/* Branch-free single-precision exp / tanh(exp) so that the activation loops vectorise (libm's scalar expf / tanhf would be
 * half of the run time): exp by range reduction to 2^n * 2^f with a degree-6 polynomial for 2^f (|rel err| < 2e-7),
 * tanh(u) = 1 - 2 / (e^{2u} + 1) for u >= 0.3 and its odd Taylor polynomial below (the device functions of device_math.h). */
static inline float v_exp(float x)
{
    x = x < -87.0f ? -87.0f : (x > 88.0f ? 88.0f : x);
    const float t = x * 1.4426950408889634f;
    const float n = (t + 12582912.0f) - 12582912.0f;          /* round to nearest (|t| < 2^22): vectorises, roundevenf does not */
    const float f = (x - n * 0.693145751953125f) - n * 1.428606765330187e-06f;      /* x - n ln2, two-term Cody-Waite */
    float p = 1.0f / 720.0f;
    p = p * f + 1.0f / 120.0f; p = p * f + 1.0f / 24.0f; p = p * f + 1.0f / 6.0f; p = p * f + 0.5f; p = p * f + 1.0f; p = p * f + 1.0f;
    union { int32_t i; float f; } s;
    s.i = ((int32_t)n + 127) << 23;
    return p * s.f;
}
static inline float v_tanh_nonneg(float u)
{
    const float big = 1.0f - 2.0f / (v_exp(u + u) + 1.0f);
    const float q = u * u;
    const float poly = ((((62.0f / 2835.0f * q - 17.0f / 315.0f) * q + 2.0f / 15.0f) * q - 1.0f / 3.0f) * q + 1.0f) * u;
    return u < 0.3f ? poly : big;
}
static inline float act_val(int kind, float x)
{
    if (kind == ORC_RELU) return x > 0.f ? x : 0.f;
    if (kind == ORC_LEAKY) return x > 0.f ? x : 0.01f * x;
    const float xc = x > 20.0f ? 20.0f : x;
    const float y = x * v_tanh_nonneg(v_exp(xc));
    return x > 20.0f ? x : y;
}
static inline void act_grad(int kind, float x, float *y, float *dy)
{
    if (kind == ORC_RELU) { float m = x >= 0.f ? 1.f : 0.f; *y = x * m; *dy = m; }
    else if (kind == ORC_LEAKY) { float s = x < 0.f ? 0.01f : 1.f; *y = x * s; *dy = s; }
    else {
        const float xc = x > 20.0f ? 20.0f : x;
        const float ex = v_exp(xc), tx = v_tanh_nonneg(ex);
        const float yy = x * tx, dd = tx - xc * ex * (tx * tx - 1.0f);
        *y = x > 20.0f ? x : yy; *dy = x > 20.0f ? 1.0f : dd;
    }
}

static int in_skips(const fast_neddf_t *n, int id)
{
    for (int i = 0; i < n->n_skips; ++i) if (n->skips[i] == id) return 1;
    return 0;
}

static int rup(int x, int m) { return (x + m - 1) / m * m; }

/* zero-padded copy / transpose of a [rows][cols] row-major block into [prow][pcol] */
static float *pad_copy(const float *src, int rows, int cols, int ld, int prow, int pcol)
{
    float *d = (float *)calloc((size_t)prow * pcol, sizeof(float));
    for (int r = 0; r < rows; ++r) memcpy(d + (size_t)r * pcol, src + (size_t)r * ld, sizeof(float) * cols);
    return d;
}
static float *pad_transpose(const float *src, int rows, int cols, int ld, int prow, int pcol)      /* out[c][r] = src[r][c] */
{
    float *d = (float *)calloc((size_t)prow * pcol, sizeof(float));
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) d[(size_t)c * pcol + r] = src[(size_t)r * ld + c];
    return d;
}

#define FB 64     /* points per block */

/* Eval-minimal NeDDF.forward (neddf.py:162-257 without the training penalties): distance, density, color[N,3], aux_grad. */
void fast_neddf_forward(const fast_neddf_t *net, const float *pos, const float *dir, const float *var, int N,
                        float *distance, float *density, float *color, float *aux_grad_out)
{
    const int E = net->embed_pos_rank, Ed = net->embed_dir_rank, Cpe = 6 * E, Cdir = 6 * Ed, W = net->ddf_width, Wc = net->col_width;
    const int L = net->n_ddf, Lc = net->n_col;
    const int WP = rup(W, 32), WcP = rup(Wc, 32), CpeP = rup(Cpe, 32), Csm = Cpe + Cdir + 3, CsmP = rup(Csm, 8);
    /* padded weights, once per call: forward blocks [K][WP], transposes [WP][.] for the reverse pass */
    float *wf[FAST_MAX_LAYERS], *wpe[FAST_MAX_LAYERS], *wT[FAST_MAX_LAYERS], *wTpe[FAST_MAX_LAYERS], *bf[FAST_MAX_LAYERS];
    float *cw_s = NULL, *cw[FAST_MAX_LAYERS], *cb[FAST_MAX_LAYERS];
    for (int l = 0; l < L; ++l) {
        const int wide = l > 0 && in_skips(net, l - 1);
        const float *Wl = net->ddf_w[l];
        wpe[l] = wTpe[l] = wT[l] = NULL;
        if (l == 0) { wf[l] = pad_copy(Wl, Cpe, W, W, Cpe, WP); wTpe[l] = pad_transpose(Wl, Cpe, W, W, WP, CpeP); }
        else {
            const int off = wide ? Cpe : 0;        /* cat([embed_pos_scaled, h]): encoding rows first (neddf.py:217-219) */
            wf[l] = pad_copy(Wl + (size_t)off * W, W, W, W, WP, WP);
            wT[l] = pad_transpose(Wl + (size_t)off * W, W, W, W, WP, WP);
            if (wide) { wpe[l] = pad_copy(Wl, Cpe, W, W, Cpe, WP); wTpe[l] = pad_transpose(Wl, Cpe, W, W, WP, CpeP); }
        }
        bf[l] = pad_copy(net->ddf_b[l], 1, W, W, 1, WP);
    }
    float *w_ddf = pad_copy(net->ddf_out_w, 1, W, W, 1, WP), *w_aux = pad_copy(net->aux_out_w, 1, W, W, 1, WP);
    for (int l = 0; l < Lc; ++l) {
        const float *Wl = net->col_w[l];
        if (l == 0) { cw_s = pad_copy(Wl, Csm, Wc, Wc, CsmP, WcP); cw[l] = pad_copy(Wl + (size_t)Csm * Wc, W, Wc, Wc, WP, WcP); }
        else cw[l] = pad_copy(Wl, Wc, Wc, Wc, WcP, WcP);
        cb[l] = pad_copy(net->col_b[l], 1, Wc, Wc, 1, WcP);
    }
    const int nblk = (N + FB - 1) / FB;
#pragma omp parallel
    {
        float *pes = (float *)calloc((size_t)FB * Cpe, sizeof(float));        /* embed_pos_scaled */
        float *js = (float *)calloc((size_t)FB * Cpe, sizeof(float));         /* d(embed_pos_scaled)/dx: one non-zero per column */
        float *sm = (float *)calloc((size_t)FB * CsmP, sizeof(float));        /* [embed_pos | embed_dir | normal] */
        float *h = (float *)calloc((size_t)FB * WP, sizeof(float)), *z = (float *)calloc((size_t)FB * (WP > WcP ? WP : WcP), sizeof(float));
        float *yp = (float *)calloc((size_t)L * FB * WP, sizeof(float));      /* a'(z_l) of every trunk layer */
        float *g = (float *)calloc((size_t)FB * WP, sizeof(float)), *g2 = (float *)calloc((size_t)FB * WP, sizeof(float));
        float *gpe = (float *)calloc((size_t)FB * CpeP, sizeof(float));
        float *hc = (float *)calloc((size_t)FB * WcP, sizeof(float));
#pragma omp for schedule(dynamic, 1)
        for (int blk = 0; blk < nblk; ++blk) {
            const int n0 = blk * FB, np = N - n0 < FB ? N - n0 : FB;
            for (int p = 0; p < np; ++p) {
                const float *x = pos + 3 * (size_t)(n0 + p), *v = var + 3 * (size_t)(n0 + p), *dr = dir + 3 * (size_t)(n0 + p);
                for (int e = 0; e < E; ++e) {
                    const float f = ldexpf(1.0f, e), gs = 1.0f / (0.5f * f), lp = net->lowpass[e];
                    for (int d = 0; d < 3; ++d) {
                        const int c = e * 3 + d;
                        const float w = expf(-0.5f * (f * f) * v[d]), ph = f * x[d], sn = sinf(ph), cs = cosf(ph);
                        const float s1 = gs * lp * w, s2 = lp * w;
                        pes[p * Cpe + c] = s1 * sn; pes[p * Cpe + 3 * E + c] = s1 * cs;
                        js[p * Cpe + c] = f * s1 * cs; js[p * Cpe + 3 * E + c] = -(f * s1) * sn;
                        sm[p * CsmP + c] = s2 * sn; sm[p * CsmP + 3 * E + c] = s2 * cs;
                    }
                }
                for (int e = 0; e < Ed; ++e)
                    for (int d = 0; d < 3; ++d) {
                        const float ph = ldexpf(1.0f, e) * dr[d];
                        sm[p * CsmP + Cpe + e * 3 + d] = sinf(ph); sm[p * CsmP + Cpe + 3 * Ed + e * 3 + d] = cosf(ph);
                    }
            }
            /* forward, value rows; keep a'(z_l) */
            for (int l = 0; l < L; ++l) {
                if (l == 0) gemm(pes, Cpe, wf[0], WP, z, WP, np, Cpe, WP, 0);
                else {
                    gemm(h, WP, wf[l], WP, z, WP, np, WP, WP, 0);
                    if (wpe[l]) gemm(pes, Cpe, wpe[l], WP, z, WP, np, Cpe, WP, 1);
                }
                float *ypl = yp + (size_t)l * FB * WP;
                const float *bl = bf[l];
                const int kind = net->activation;
                for (int p = 0; p < np; ++p) {
                    const float *zr = z + p * WP;
                    float *hr = h + p * WP, *yr = ypl + p * WP;
#pragma omp simd
                    for (int j = 0; j < WP; ++j) {
                        float y, dy;
                        act_grad(kind, zr[j] + bl[j], &y, &dy);
                        hr[j] = y; yr[j] = dy;
                    }
                }
            }
            /* heads (value) and the seed of the reverse pass: d z_D / d z_L = w_ddf * a'(z_L) */
            float zD[FB], zA[FB];
            for (int p = 0; p < np; ++p) {
                float s0 = 0.f, s1 = 0.f;
                const float *ypl = yp + (size_t)(L - 1) * FB * WP + p * WP;
                for (int j = 0; j < WP; ++j) { s0 += h[p * WP + j] * w_ddf[j]; s1 += h[p * WP + j] * w_aux[j]; g[p * WP + j] = w_ddf[j] * ypl[j]; }
                zD[p] = s0 + net->ddf_out_b[0]; zA[p] = s1 + net->aux_out_b[0];
            }
            /* reverse pass: g_{l-1} = (g_l W_l^T) * a'(z_{l-1}); the encoding collects g_0 W_0^T and the skip layers' encoding rows */
            memset(gpe, 0, sizeof(float) * FB * CpeP);
            float *gc = g, *gn = g2;
            for (int l = L - 1; l >= 1; --l) {
                if (wTpe[l]) gemm(gc, WP, wTpe[l], CpeP, gpe, CpeP, np, WP, CpeP, 1);
                gemm(gc, WP, wT[l], WP, gn, WP, np, WP, WP, 0);
                const float *ypl = yp + (size_t)(l - 1) * FB * WP;
                for (int i = 0; i < np * WP; ++i) gn[i] *= ypl[i];
                float *t = gc; gc = gn; gn = t;
            }
            gemm(gc, WP, wTpe[0], CpeP, gpe, CpeP, np, WP, CpeP, 1);
            for (int p = 0; p < np; ++p) {
                const int n = n0 + p;
                float gz[3] = { 0.f, 0.f, 0.f };
                for (int e = 0; e < E; ++e)
                    for (int d = 0; d < 3; ++d) {
                        const int c = e * 3 + d;
                        gz[d] += gpe[p * CpeP + c] * js[p * Cpe + c] + gpe[p * CpeP + 3 * E + c] * js[p * Cpe + 3 * E + c];
                    }
                /* softplus.py:38-49, sigmoid.py:38-43, neddf.py:234-241 */
                const float zd = zD[p], big = zd > 20.0f;
                const float sp = big ? zd : logf(1.0f + expf(zd)), dsp = big ? 1.0f : 1.0f / (1.0f + expf(-zd));
                const float D = sp + net->d_near;
                const float dg[3] = { dsp * gz[0], dsp * gz[1], dsp * gz[2] };
                const float sg = (1.0f + tanhf(zA[p] * 0.5f)) * 0.5f, aux = net->aux_grad_scale * sg;
                const float q2 = dg[0] * dg[0] + dg[1] * dg[1] + dg[2] * dg[2], dgn = sqrtf(q2), dDdt = sqrtf(q2 + aux * aux);
                const float rho = act_val(net->density_activation, (1.0f / D) * (1 - dDdt));
                const float ninv = 1.0f / (dgn + 1e-7f);
                for (int i = 0; i < 3; ++i) sm[p * CsmP + Cpe + Cdir + i] = ninv * dg[i];
                if (distance) distance[n] = D;
                if (density) density[n] = rho;
                if (aux_grad_out) aux_grad_out[n] = aux;
            }
            if (color) {      /* colour trunk on value rows (neddf.py:243-257) */
                float *ci = h;
                for (int l = 0; l < Lc; ++l) {
                    if (l == 0) { gemm(sm, CsmP, cw_s, WcP, z, WcP, np, CsmP, WcP, 0); gemm(h, WP, cw[0], WcP, z, WcP, np, WP, WcP, 1); }
                    else gemm(ci, WcP, cw[l], WcP, z, WcP, np, WcP, WcP, 0);
                    const float *bl = cb[l];
                    const int kind = net->activation;
                    for (int p = 0; p < np; ++p) {
                        const float *zr = z + p * WcP;
                        float *hr = hc + p * WcP;
#pragma omp simd
                        for (int j = 0; j < WcP; ++j) hr[j] = act_val(kind, zr[j] + bl[j]);
                    }
                    ci = hc;
                }
                for (int p = 0; p < np; ++p)
                    for (int k = 0; k < 3; ++k) {
                        float s = net->col_out_b[k];
                        for (int j = 0; j < Wc; ++j) s += ci[p * WcP + j] * net->col_out_w[j * 3 + k];
                        color[3 * (size_t)(n0 + p) + k] = s;
                    }
            }
        }
        free(pes); free(js); free(sm); free(h); free(z); free(yp); free(g); free(g2); free(gpe); free(hc);
    }
    for (int l = 0; l < L; ++l) { free(wf[l]); free(wpe[l]); free(wT[l]); free(wTpe[l]); free(bf[l]); }
    free(w_ddf); free(w_aux); free(cw_s);
    for (int l = 0; l < Lc; ++l) { free(cw[l]); free(cb[l]); }
}
