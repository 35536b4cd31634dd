// device_math.h -- activation / encoding arithmetic shared by the fused field
// kernels (field_kernels.hip) and the stand-alone op kernels (op_kernels.hip),
// so that the unit-level parity tests exercise exactly the code the tile
// engine's epilogues run.
#pragma once
#include <hip/hip_runtime.h>

namespace neddf {

// tanh(exp(x)) building blocks on the raw transcendental units: v_exp_f32 is
// exp2 (<= 1 ulp), v_rcp_f32 <= 1 ulp.  u = e^x >= 0; tanh(u) = 1 - 2/(e^{2u}+1)
// loses relative accuracy for small u (cancellation), so u < 0.3 uses the odd
// Taylor polynomial to u^9 (truncation < 6e-8 relative at 0.3).  Measured against
// fp64 over x in [-30, 20]: |err(y)| <= 1.8e-7, |err(y')| <= 1.2e-6 -- inside the
// error of evaluating the reference formula itself in fp32 (2.3e-7 / 2.6e-6).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

__device__ __forceinline__ float tanh_nonneg(float u)
{
    // u * (2 log2 e) and fma(-2, r, 1) are bitwise what (u + u) * log2 e and 1 - 2 r give (doubling is exact), one instruction less each
    float e2u = __builtin_amdgcn_exp2f(u * 2.8853900817779268f);
    float big = fmaf(-2.0f, __builtin_amdgcn_rcpf(e2u + 1.0f), 1.0f);
    float p = u * u;
    float poly = fmaf(p, fmaf(p, fmaf(p, fmaf(p, 62.0f / 2835.0f, -17.0f / 315.0f), 2.0f / 15.0f), -1.0f / 3.0f), 1.0f) * u;
    return u < 0.3f ? poly : big;
}

// Reduced-cost tanhExp of the fused kernels (every operand policy since round 3, tile_engine.h kFastAct): no small-argument
// polynomial, no x > 20 branch (tanh saturates to exactly 1 there; the clamp keeps x e^x finite).  11 VALU instructions
// instead of 23 -- the matrix pipe and the VALU do not overlap, so every instruction is wall time.  Where e^x < 0.3 the
// difference 1 - 2 r loses relative accuracy (absolute error of tanh ~1e-7, of y ~1e-7 |x|): rounding noise next to O(1)
// activations, see the measurements cited at kFastAct.
__device__ __forceinline__ void tanhexp_grad_fast(float x, float &y, float &dy)
{
#if defined(NEDDF_PROBE_NOACT) && NEDDF_PROBE_NOACT
    y = x * 0.5f; dy = x * 0.25f; return;       // timing probe (variant builds only, results invalid): the epilogue without the activation's arithmetic
#endif
    // v_med3_f32 clamps in ONE instruction (fminf costs a NaN-quieting v_max in front of its v_min)
    float ex = fast_exp(__builtin_amdgcn_fmed3f(x, -3.0e38f, 40.0f));
    float e2 = __builtin_amdgcn_exp2f(ex * 2.8853900817779268f);
    float tx = fmaf(-2.0f, __builtin_amdgcn_rcpf(e2 + 1.0f), 1.0f);
    y = x * tx;
    dy = fmaf(-(x * ex), fmaf(tx, tx, -1.0f), tx);
}

// the same, returning tanh(e^x) and the derivative so that the caller can form y = x t and the Jacobian products as packed
// multiplies: (x, z1) * (t, dy) and (z2, z3) * (dy, dy)
__device__ __forceinline__ void tanhexp_parts_fast(float x, float &tx, float &dy)
{
    float ex = fast_exp(__builtin_amdgcn_fmed3f(x, -3.0e38f, 40.0f));
    float e2 = __builtin_amdgcn_exp2f(ex * 2.8853900817779268f);
    tx = fmaf(-2.0f, __builtin_amdgcn_rcpf(e2 + 1.0f), 1.0f);
    dy = fmaf(-(x * ex), fmaf(tx, tx, -1.0f), tx);
}

// The middle form (tile_engine.h kActMode = 2): the closed form where it is accurate (e^x >= 0.2: relative error of tanh <= 5e-7), and
// below that the odd polynomial u (1 + c1 u^2 + c2 u^4) with coefficients fitted to tanh(u)/u on [0, 0.2] (max relative error
// 1.7e-7 evaluated in fp32) -- 17 instructions: the fast form's 11 + two multiplies, two fmas, one compare, one select.  Relative
// accuracy of tanh (and so of small activations) within ~4x of the reference's own libm evaluation for EVERY x, where the fast form
// keeps only the absolute error bounded.
__device__ __forceinline__ void tanhexp_grad_mid(float x, float &y, float &dy)
{
    const float ex = fast_exp(__builtin_amdgcn_fmed3f(x, -3.0e38f, 40.0f));
    const float e2 = __builtin_amdgcn_exp2f(ex * 2.8853900817779268f);
    const float big = fmaf(-2.0f, __builtin_amdgcn_rcpf(e2 + 1.0f), 1.0f);
    const float p = ex * ex;
    const float small = fmaf(p * ex, fmaf(p, 0.13038349f, -0.33329707f), ex);
    const float tx = ex < 0.2f ? small : big;
    y = x * tx;
    dy = fmaf(-(x * ex), fmaf(tx, tx, -1.0f), tx);
}

__device__ __forceinline__ float tanhexp_val_mid(float x)
{
    const float ex = fast_exp(__builtin_amdgcn_fmed3f(x, -3.0e38f, 40.0f));
    const float e2 = __builtin_amdgcn_exp2f(ex * 2.8853900817779268f);
    const float big = fmaf(-2.0f, __builtin_amdgcn_rcpf(e2 + 1.0f), 1.0f);
    const float p = ex * ex;
    const float small = fmaf(p * ex, fmaf(p, 0.13038349f, -0.33329707f), ex);
    return x * (ex < 0.2f ? small : big);
}

// MODE (tile_engine.h Ops::kActMode) selects how the FUSED kernels evaluate tanhExp: 0 = the reference's branches (series below
// e^x = 0.3 and closed form above, both evaluated: 23 instructions), 1 = closed form for every x (11), 2 = the middle form above (17).
// ReLU / LeakyReLU ignore it.
template <int KIND, int MODE = 0>
__device__ __forceinline__ void act_grad(float x, float &y, float &dy)
{
    if (MODE == 1 && KIND == 2) { tanhexp_grad_fast(x, y, dy); return; }
    if (MODE == 2 && KIND == 2) { tanhexp_grad_mid(x, y, dy); return; }
    if (KIND == 0) {            // relu.py:36-38, mask = x >= 0
        float m = (x >= 0.f) ? 1.f : 0.f;
        y = x * m; dy = m;
    } else if (KIND == 1) {     // leaky_relu.py:36-39
        float s = (x < 0.f) ? 0.01f : 1.f;
        y = x * s; dy = s;
    } else {                    // tanh_exp.py:38-46
        // The reference returns (x, 1) for x > 20.  Clamping the exponent's argument at 80 gives exactly that without the
        // comparison and two selects: for e^x >= ~10 the tanh below is exactly 1, so y = x * 1 and y' = 1 - x e^x * (1 - 1) = 1
        // (x e^x stays finite up to the clamp); below 20 nothing changes.  v_med3_f32 is one instruction.
        const float xc = __builtin_amdgcn_fmed3f(x, -3.0e38f, 80.0f);
        float ex = fast_exp(xc);
        float tx = tanh_nonneg(ex);
        y = x * tx;
        dy = fmaf(-(xc * ex), fmaf(tx, tx, -1.0f), tx);     // tx - x*ex*(tx^2 - 1); xc keeps the product finite for any x
    }
}

template <int KIND, int MODE = 0>
__device__ __forceinline__ float act_val(float x)
{
    if (MODE == 2 && KIND == 2) return tanhexp_val_mid(x);
    if (MODE == 1 && KIND == 2) {
        float ex = fast_exp(__builtin_amdgcn_fmed3f(x, -3.0e38f, 40.0f));
        float e2 = __builtin_amdgcn_exp2f(ex * 2.8853900817779268f);
        return x * fmaf(-2.0f, __builtin_amdgcn_rcpf(e2 + 1.0f), 1.0f);
    }
    if (KIND == 0) return x > 0.f ? x : 0.f;            // F.relu
    if (KIND == 1) return x > 0.f ? x : 0.01f * x;       // F.leaky_relu
    return x * tanh_nonneg(fast_exp(__builtin_amdgcn_fmed3f(x, -3.0e38f, 80.0f)));      // nn_module/tanh_exp.py:28-31; x > 20 -> x exactly, see act_grad
}

// ---- the activations over N elements at a time, stage by stage (round 6).
// Written per element, tanhExp is ONE dependent chain of 13 instructions (clamp -> mul -> exp -> mul -> exp -> add -> rcp -> fma -> ...)
// and hipcc's scheduler (which minimises register pressure) emits the elements one after another: every instruction waits for its
// predecessor's result -- ~6.6 cycles per dependent instruction instead of the ~4 a wave can issue, plus an s_nop behind every
// transcendental: the forward epilogue of the bf16 reverse-mode kernel ran at 88 cycles per element for 60 of instruction issue
// (profiles/r06_stamp_timeline_bf16_*.txt).  Here the SAME operations run in the SAME order per element (bit-identical results), but
// interleaved over N independent elements, with scheduling barriers between the stages so that the interleaving survives.
#ifndef NEDDF_ACT_ILP
#define NEDDF_ACT_ILP 4
#endif
#define NEDDF_STAGE() __builtin_amdgcn_sched_barrier(0)
template <int KIND, int MODE, int N>
__device__ __forceinline__ void act_grad_n(const float (&x)[N], float (&y)[N], float (&dy)[N])
{
    if constexpr (KIND == 2 && (MODE == 1 || MODE == 2) && N > 1) {
        float ex[N], e2[N], tx[N], q[N], s[N];
#pragma unroll
        for (int i = 0; i < N; ++i) ex[i] = __builtin_amdgcn_fmed3f(x[i], -3.0e38f, 40.0f) * 1.4426950408889634f;
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) ex[i] = __builtin_amdgcn_exp2f(ex[i]);
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) { e2[i] = ex[i] * 2.8853900817779268f; q[i] = x[i] * ex[i]; }
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) e2[i] = __builtin_amdgcn_exp2f(e2[i]);
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) e2[i] = e2[i] + 1.0f;
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) e2[i] = __builtin_amdgcn_rcpf(e2[i]);
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) tx[i] = fmaf(-2.0f, e2[i], 1.0f);
        if constexpr (MODE == 2) {          // the middle form: the fitted odd polynomial below e^x = 0.2 (tanhexp_grad_mid)
            float p[N], sm[N];
            NEDDF_STAGE();
#pragma unroll
            for (int i = 0; i < N; ++i) p[i] = ex[i] * ex[i];
            NEDDF_STAGE();
#pragma unroll
            for (int i = 0; i < N; ++i) { sm[i] = fmaf(p[i], 0.13038349f, -0.33329707f); p[i] = p[i] * ex[i]; }
            NEDDF_STAGE();
#pragma unroll
            for (int i = 0; i < N; ++i) sm[i] = fmaf(p[i], sm[i], ex[i]);
            NEDDF_STAGE();
#pragma unroll
            for (int i = 0; i < N; ++i) tx[i] = ex[i] < 0.2f ? sm[i] : tx[i];
        }
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) { s[i] = fmaf(tx[i], tx[i], -1.0f); y[i] = x[i] * tx[i]; }
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) dy[i] = fmaf(-q[i], s[i], tx[i]);
        NEDDF_STAGE();
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) act_grad<KIND, MODE>(x[i], y[i], dy[i]);
    }
}

template <int KIND, int MODE, int N>
__device__ __forceinline__ void act_val_n(const float (&x)[N], float (&y)[N])
{
    if constexpr (KIND == 2 && (MODE == 1 || MODE == 2) && N > 1) {
        float ex[N], e2[N], tx[N];
#pragma unroll
        for (int i = 0; i < N; ++i) ex[i] = __builtin_amdgcn_fmed3f(x[i], -3.0e38f, 40.0f) * 1.4426950408889634f;
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) ex[i] = __builtin_amdgcn_exp2f(ex[i]);
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) e2[i] = ex[i] * 2.8853900817779268f;
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) e2[i] = __builtin_amdgcn_exp2f(e2[i]);
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) e2[i] = e2[i] + 1.0f;
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) e2[i] = __builtin_amdgcn_rcpf(e2[i]);
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) tx[i] = fmaf(-2.0f, e2[i], 1.0f);
        if constexpr (MODE == 2) {
            float p[N], sm[N];
            NEDDF_STAGE();
#pragma unroll
            for (int i = 0; i < N; ++i) p[i] = ex[i] * ex[i];
            NEDDF_STAGE();
#pragma unroll
            for (int i = 0; i < N; ++i) { sm[i] = fmaf(p[i], 0.13038349f, -0.33329707f); p[i] = p[i] * ex[i]; }
            NEDDF_STAGE();
#pragma unroll
            for (int i = 0; i < N; ++i) sm[i] = fmaf(p[i], sm[i], ex[i]);
            NEDDF_STAGE();
#pragma unroll
            for (int i = 0; i < N; ++i) tx[i] = ex[i] < 0.2f ? sm[i] : tx[i];
        }
        NEDDF_STAGE();
#pragma unroll
        for (int i = 0; i < N; ++i) y[i] = x[i] * tx[i];
        NEDDF_STAGE();
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) y[i] = act_val<KIND, MODE>(x[i]);
    }
}

__device__ __forceinline__ float act_val_rt(int kind, float x)
{
    if (kind == 0) return act_val<0>(x);
    if (kind == 1) return act_val<1>(x);
    return act_val<2>(x);
}


// SoftplusGradFunction softplus.py:38-49: log(1.0 + exp(x)) (not log1p), threshold 20
__device__ __forceinline__ void softplus_grad(float z, float &y, float &dy)
{
    bool big = z > 20.0f;
    y = big ? z : logf(1.0f + expf(z));
    dy = big ? 1.0f : 1.0f / (1.0f + expf(-z));
}

// the same two head activations on the raw transcendental units (kFast policies: bf16 operands -- their error, ~1e-6 relative, is three
// orders of magnitude below the policy's rounding); the library expf / logf / tanhf above are ~40 instructions each
__device__ __forceinline__ void softplus_grad_fast(float z, float &y, float &dy)
{
    const bool big = z > 20.0f;
    const float e = fast_exp(__builtin_amdgcn_fmed3f(z, -80.0f, 20.0f));
    y = big ? z : __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
    dy = big ? 1.0f : e * __builtin_amdgcn_rcpf(1.0f + e);          // 1 / (1 + e^-z)
}
__device__ __forceinline__ void sigmoid_grad_fast(float a, float &y, float &dy)
{
    const float e = fast_exp(__builtin_amdgcn_fmed3f(-a, -80.0f, 80.0f));
    const float t = __builtin_amdgcn_rcpf(1.0f + e);                // (1 + tanh(a / 2)) / 2
    y = t;
    dy = t * (1 - t);
}

// SigmoidGradFunction sigmoid.py:38-43 (s = 1)
__device__ __forceinline__ void sigmoid_grad(float a, float &y, float &dy)
{
    float t = (1.0f + tanhf(1.0f * a * 0.5f)) * 0.5f;
    y = t;
    dy = 1.0f * t * (1 - t);
}

// One (frequency e, axis d) pair of the integrated positional encoding with J_in = I3:
// weight w = exp(-0.5 4^e var) (sampling.py:55-71), values s*sin / s*cos and the
// non-zero Jacobian entries +-2^e s cos/sin (with_grad/positional_encoding.py:55-87).
// GRADSCALE selects embed_pos_scaled: s = (1/(0.5*2^e)) * lowpass * w (neddf.py:193-204).
// hardware sine / cosine (v_sin_f32 / v_cos_f32 take revolutions); |error| ~ 4e-5 at the largest arguments used here
__device__ __forceinline__ void fast_sincos(float x, float &sn, float &cs)
{
    float r = x * 0.15915494309189535f;
    sn = __builtin_amdgcn_sinf(r);
    cs = __builtin_amdgcn_cosf(r);
}

// sin and cos of the exact (fp32-policy) encodings.  The library sincosf carries a Payne-Hanek path for arguments the
// encodings never reach and costs ~180 instructions, expf ~50; with 30 + 12 (frequency, axis) pairs per sample point the
// colour kernel spent 5.5 % of its time there (profiles/r03_col_ablation.txt).  Here: Cody-Waite reduction by pi/2 with a
// three-term constant (exact products through fma; |x| up to ~1e4 rad, the encodings stay below 2^9 |pos|), then the
// degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4] -- ~25 instructions, |error| <= 2.5e-7 (measured against fp64
// over |x| <= 4096, tests/test_gpu_parity.py::test_layer_ops sweeps it through neddf_op_positional_encoding).
__device__ __forceinline__ void sincos_cw(float x, float &sn, float &cs)
{
    const float k = rintf(x * 0.636619772367581343f);                 // x / (pi/2)
    float r = fmaf(-k, 1.57079637050628662109375f, x);                // pi/2 = c1 + c2 + c3
    r = fmaf(-k, -4.37113882867379116e-8f, r);
    r = fmaf(-k, -1.71512449944288e-15f, r);
    const float r2 = r * r;
    const float sp = fmaf(r * r2, fmaf(r2, fmaf(r2, fmaf(r2, 2.7183114939898219064e-6f, -1.9840874255356029e-4f), 8.3333169820368300e-3f), -1.6666666055418920e-1f), r);
    const float cp = fmaf(r2 * r2, fmaf(r2, fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), fmaf(-0.5f, r2, 1.0f));
    const int q = (int)k;
    const float a = (q & 1) ? cp : sp, b = (q & 1) ? sp : cp;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}

// e^x on the exp2 unit with the rounding error of x * log2(e) carried along: relative error <= 2e-7 for the (non-positive)
// arguments of the integrated encoding's weights (sampling.py:71)
__device__ __forceinline__ float exp_acc(float x)
{
    const float t = x * 1.44269502162933349609375f;
    const float r = fmaf(x, 1.44269502162933349609375f, -t) + x * 1.925963033500011e-8f;
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * 0.693147182464599609375f, e);
}

// Ray.get_sampling_cones ray.py:128-194 / get_sampling_points ray.py:88-126 for sample j of a ray whose S distances start at d: the
// moments along the ray (t_mu, t_var) and across it (r_var).  ONE definition for the stand-alone sampling kernel (render_kernels.hip)
// and for the field kernels that take their sample points straight from the rays (field_kernels.hip): only + - * / in the same order,
// every translation unit is built with -ffp-contract=off, so both give the same bits.
template <bool CONE>
__device__ __forceinline__ void sample_moments(const float *d, int j, int S, float r2, float &t_mu, float &t_var, float &r_var)
{
    // (all three distances requested unconditionally: with the far edge as a branch around its own loads, the field kernels' per-tile staging was
    // a chain of dependent memory round trips.  At the last sample d[S - 1] is d[j] and d[S - 2] is d[j - 1]: the same arithmetic, the same bits.)
    const float dn = d[j], dnx = d[j + 1 < S ? j + 1 : j], dpv = d[j > 0 ? j - 1 : 0];
    t_mu = dn; t_var = 0.f; r_var = 0.f;
    if (CONE) {
        float df = (j + 1 < S) ? dnx : (2 * dn - dpv);
        float mu = 0.5f * (dn + df);
        float sg = 0.5f * (df - dn);
        float mu2 = mu * mu, s2 = sg * sg, s4 = s2 * s2;
        float minv = 1.0f / (3 * mu2 + s2 + 1e-7f);
        const float c13 = (float)(1.0 / 3), c415 = (float)(4.0 / 15), c512 = (float)(5.0 / 12);
        t_mu = mu + (2 * mu * s2) * minv;
        t_var = c13 * s2 - c415 * s4 * (12 * mu2 - s2) * (minv * minv);
        r_var = r2 * (0.25f * mu2 + c512 * s2 - c415 * s4 * minv);
    }
}

// ... and one coordinate of the sample: position o + d t_mu, variance t_var d^2 + r_var (1 - d^2) (zero for point samples)
template <bool CONE>
__device__ __forceinline__ void sample_coord(float dd, float oo, float t_mu, float t_var, float r_var, float &pos, float &var)
{
    const float dsq = dd * dd;
    pos = oo + dd * t_mu;
    var = CONE ? t_var * dsq + r_var * (1.0f - dsq) : 0.0f;
}

template <bool GRADSCALE, bool FAST = false>
__device__ __forceinline__ void pe_pair(int e, float x, float v, float lowpass, float &vs, float &vc, float &js, float &jc)
{
    float f = (float)(1 << e);
    float w = FAST ? fast_exp(-0.5f * (f * f) * v) : exp_acc(-0.5f * (f * f) * v);
    float s = GRADSCALE ? ((1.0f / (0.5f * f)) * lowpass) * w : lowpass * w;
    float sn, cs;
    if (FAST) fast_sincos(f * x, sn, cs);
    else sincos_cw(f * x, sn, cs);
    vs = s * sn;
    vc = s * cs;
    float g = f * s;
    js = g * cs;
    jc = -g * sn;
}

}  // namespace neddf
