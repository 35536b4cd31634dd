// train_kernels.hip -- building blocks of the training step.
//
// Activations are row-major matrices in HBM ([4N, C]: value + Jacobian rows of N sample
// points), so that the forward pass leaves behind exactly what the hand-written
// backward passes of the reference need (neddf/nn_module/with_grad/*.py backward
// staticmethods).  The NeDDF forward runs its two layer stacks fused (mlp_forward_kernel,
// round 2: the inference tile engine with side stores of Z_l / H_l); the backward pass
// and the NeRF / NeuS fields are one kernel per layer.  All dense work -- forward layers, dX = dZ W^T and the weight
// gradient dW = X^T dZ -- runs on the same fp32 MFMA tile engine as the fused
// inference kernels; the elementwise pieces reuse device_math.h.  Training batches
// are ~10^5 points, two orders of magnitude below a rendered frame, so the extra
// HBM round trips (8 KB per point and layer) are affordable while the gradients
// are being pinned against the reference's autograd.
#include "kernels.h"
#include "device_math.h"
#include "tile_engine.h"
#include "train_kernels.h"

#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

namespace neddf {

// ----------------------------------------------------------------------------
// second derivative of the hidden activations as the reference's backward passes define it
template <int KIND>
__device__ __forceinline__ void act_grad2(float x, float &dy, float &d2)
{
    if (KIND == 0) { dy = (x >= 0.f) ? 1.f : 0.f; d2 = 0.f; }            // relu.py backward: mask only
    else if (KIND == 1) { dy = (x < 0.f) ? 0.01f : 1.f; d2 = 0.f; }      // leaky_relu.py backward: scale only
    else {                                                                 // tanh_exp.py:43-51
        float ex = fast_exp(x), tx = tanh_nonneg(ex);
        float t2 = fmaf(tx, tx, -1.0f);
        bool big = x > 20.0f;
        dy = big ? 1.0f : fmaf(-(x * ex), t2, tx);
        if (KIND == 2) d2 = big ? 0.0f : ex * (-x + 2 * ex * x * tx - 2) * t2;
        // KIND 3: what torch's double backward makes of the plain tanhExp Function (nn_module/tanh_exp.py:36-60), which NeuS
        // differentiates twice (neus.py:136-145): its backward computes tx - x ex (tx^2 - 1) from the saved x, ex, tx, and only
        // x carries a graph, so the derivative of y' it propagates is -ex (tx^2 - 1), not y''.  Kept, for identical gradients.
        else d2 = big ? 0.0f : -ex * t2;
    }
}
constexpr int kActTanhExpPlain2 = 3;       // backward-only kind id: tanhExp with the second derivative above

// Backward of one activation on a 4-row group of 4 columns, in place on g (the upstream gradient of the group):
// period 4 = {ReLU,LeakyReLU,TanhExp}GradFunction.backward (dLdx = dLdy y' + sum_i dLdG_i J_i y'', dLdJ_i = dLdG_i y');
// period 1 = the plain activations of NeRF (F.relu / F.leaky_relu / tanhExp, nerf.py:71-79; derivative at 0 as torch's)
__device__ __forceinline__ void act_backward_group(int kind, int period, const f32x4v (&z)[4], f32x4v (&g)[4])
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (period == 4) {
            float dy, d2;
            if (kind == 0) act_grad2<0>(z[0][u], dy, d2); else if (kind == 1) act_grad2<1>(z[0][u], dy, d2);
            else if (kind == 2) act_grad2<2>(z[0][u], dy, d2); else act_grad2<3>(z[0][u], dy, d2);
            float s = g[1][u] * z[1][u];
            s += g[2][u] * z[2][u];
            s += g[3][u] * z[3][u];
            g[0][u] = g[0][u] * dy + s * d2;
            g[1][u] *= dy; g[2][u] *= dy; g[3][u] *= dy;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float x = z[r][u];
                float dy, d2;
                if (kind == 0) dy = x > 0.f ? 1.f : 0.f;
                else if (kind == 1) dy = x > 0.f ? 1.f : 0.01f;
                else act_grad2<2>(x, dy, d2);
                g[r][u] *= dy;
            }
        }
    }
}

// ----------------------------------------------------------------------------
// Operand range of the split-fp16 policy in the backward pass.  Two fp16 terms resolve 2^-24 absolutely, and gradient matrices
// of a mean-over-rays loss sit at 1e-5 .. 1e-8: unscaled, a product with max |dZ| = 1e-5 is already 2e-3 off (1e-6: 1.5e-2).
// Every kernel that writes a gradient matrix therefore also leaves max |dZ| in a device scalar (one atomic per wave), and every
// split-operand GEMM that consumes the matrix multiplies it by the power of two that brings that maximum to [2^13, 2^14) while
// staging it, and the accumulators by the inverse on the way out -- both exact.  (The fp32 MFMA path needs none of this.)
__device__ __forceinline__ float operand_scale_of(float amax)
{
    const int e = (int)((__builtin_bit_cast(unsigned int, amax) >> 23) & 0xffu);       // biased exponent of the maximum
    if (e == 0 || e == 255) return 1.0f;            // zero / denormal / non-finite: leave the operand alone
    int shift = 13 + 127 - e;
    shift = shift > 100 ? 100 : (shift < -100 ? -100 : shift);
    return __builtin_bit_cast(float, (unsigned int)(127 + shift) << 23);
}
__device__ __forceinline__ float operand_scale(const float *amax)
{
    if (!amax) return 1.0f;
    const int e = (int)((__builtin_bit_cast(unsigned int, *amax) >> 23) & 0xffu);       // biased exponent of the maximum
    if (e == 0 || e == 255) return 1.0f;            // zero / denormal / non-finite: leave the operand alone
    int shift = 13 + 127 - e;
    shift = shift > 100 ? 100 : (shift < -100 ? -100 : shift);
    return __builtin_bit_cast(float, (unsigned int)(127 + shift) << 23);
}
__device__ __forceinline__ float pow2_inverse(float p) { return __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(unsigned int, p)); }
__device__ __forceinline__ void publish_amax(float *amax_out, float lmax)       // all lanes of the wave call this
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off, 64));
    // Non-negative floats order like ints.  Hundreds of thousands of waves share the scalar: look first (a stale, smaller value
    // only costs an unnecessary atomic) so that all but the first few skip the same-address atomic, which serialises in L2.
    if ((threadIdx.x & 63) == 0 && lmax > *(volatile const float *)amax_out) atomicMax((int *)amax_out, __builtin_bit_cast(int, lmax));
}

// ----------------------------------------------------------------------------
// Y[R, 256] (+)= X[R, 0:kload) x Wpacked (+ bias on rows r % bias_period == 0), optionally followed by the activation on
// (value, Jacobian) row groups: H = a(Y) (LinearGradFunction.forward + the activation's forward in one pass).
// 64-row tiles, two workgroups per CU (one's loads / stores overlap the other's MFMAs).  The next tile's rows are
// requested (global -> VGPR) before the MFMAs of the current tile issue; results go back through the LDS tile so that
// every global access of the epilogue is a full 16-byte-per-lane row segment, and the activation sees the four rows
// of a point in one thread.
// MODE 0: plain; 1: also H = a(Y) (forward); 2: Y = activation backward of the product with the pre-activations Zp = H
// (the dX GEMM of layer l immediately followed by the backward of layer l-1's activation: dZ_{l-1}, no dH round trip)
// Ops = OpsF32 (exact fp32 MFMA) or OpsF16Split (fp32 operands as two fp16 terms, tile_engine.h): the operand tile in LDS has
// the policy's layout, the result tile that the epilogue reads back is always fp32 [64][kActLd] in the same LDS bytes.
template <int MODE, class Ops>
__global__ __launch_bounds__(kThreads, 2) void rows_gemm_kernel(const float *X, int64_t R, int ldx, int kload, const float *wp, int ksteps,
                                                                const float *bias, int bias_period, float *Y, int ldy, int accumulate,
                                                                int act_kind, float *H, const float *amax_in, float *amax_out)
{
    typedef typename Ops::act_t act_t;
    constexpr int MT = 2, NT = 2, ROWS = MT * 32, NPF = ROWS * (kWidth / 4) / kThreads, LD = Ops::kLd;
    constexpr bool SCALED = Ops::kPlanes == 2;       // split-fp16 operands: X is range-scaled (see operand_scale)
    const float xs = SCALED ? operand_scale(amax_in) : 1.0f;
    const float unscale = SCALED ? pow2_inverse(xs) * (1.0f / Ops::kWScale) : 1.0f / Ops::kWScale;
    float lmax = 0.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *act = smem;                       // result tile (fp32)
    act_t *opd = (act_t *)smem;              // operand tile (policy layout)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const act_t *act_lane = act_lane_ptr<Ops>(opd, lane);
    const typename Ops::bfrag *wl = (const typename Ops::bfrag *)wp + (size_t)wave * NT * ksteps * 64 + lane;
    const int64_t ntiles = (R + ROWS - 1) / ROWS;
    const int c4n = kload >> 2, kpack = Ops::kStep * ksteps;
    f32x4v pf[NPF];
    // 16-byte chunk idx = tid + i * kThreads of the tile is (row idx / c4n, column idx % c4n): one division per thread, then
    // steps of kThreads chunks (as 2 x 16 runtime divisions per tile they were 800 of the tile's vector instructions)
    const int step_r = kThreads / c4n, step_c = kThreads - step_r * c4n;
    const int row0 = tid / c4n, col0 = tid - row0 * c4n;
    auto fetch = [&](int64_t tile) {
        const int64_t r0 = tile * ROWS;
        int r = row0, c = col0;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            f32x4v v = { 0.f, 0.f, 0.f, 0.f };
            if (r < ROWS && r0 + r < R) v = *(const f32x4v *)(X + (r0 + r) * ldx + 4 * c);
            pf[i] = v;
            r += step_r; c += step_c;
            if (c >= c4n) { c -= c4n; ++r; }
        }
    };
    int64_t tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * ROWS;
        __syncthreads();                // the previous tile's epilogue is done with the LDS tile
        {
            int r = row0, c = col0;
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                if (r < ROWS) Ops::put4(opd + r * LD + 4 * c, SCALED ? pf[i] * xs : pf[i]);
                r += step_r; c += step_c;
                if (c >= c4n) { c -= c4n; ++r; }
            }
        }
        for (int i = tid; i < ROWS * (kpack - kload); i += kThreads) {      // packed width beyond the loaded width
            int w = kpack - kload, r = i / w, c = i - r * w;
            Ops::zero(opd + r * LD + kload + c);
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[MT][NT];
        if (bias_period == 4) acc_init<MT, NT, true>(acc, bias, wave, lane, Ops::kWScale);
        else acc_init<MT, NT, false>(acc, bias, wave, lane, Ops::kWScale);
        dense<MT, NT, Ops>(acc, act_lane, wl, ksteps);
        __syncthreads();                // every wave finished reading the A operands
        const int j = lane & 31, h = lane >> 5;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float *o = act + (mt * 32 + 4 * h) * kActLd + (wave * NT + t) * 32 + j;
#pragma unroll
                for (int q = 0; q < 16; ++q) o[(8 * (q >> 2) + (q & 3)) * kActLd] = acc[mt][t][q] * unscale;
            }
        __syncthreads();
        // items: (4-row group, 4 columns); rows of a group are consecutive rows of the tile
        for (int it = tid; it < (ROWS / 4) * (kWidth / 4); it += kThreads) {
            const int grp = it >> 6, c4 = it & 63;
            const int64_t row = r0 + 4 * grp;
            if (row >= R) continue;
            f32x4v z[4];
            if (MODE == 2) {
                f32x4v zp[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    z[r] = *(const f32x4v *)(act + (4 * grp + r) * kActLd + 4 * c4);
                    const f32x4v zero = { 0.f, 0.f, 0.f, 0.f };
                    zp[r] = row + r < R ? *(const f32x4v *)(H + (row + r) * ldy + 4 * c4) : zero;
                    // K-blocked products (hidden widths above 256): the earlier blocks' partial sums wait in Y
                    if (accumulate && row + r < R) z[r] += *(const f32x4v *)(Y + (row + r) * ldy + 4 * c4);
                }
                act_backward_group(act_kind, bias_period, zp, z);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row + r < R) {
                        *(f32x4v *)(Y + (row + r) * ldy + 4 * c4) = z[r];
#pragma unroll
                        for (int u = 0; u < 4; ++u) lmax = fmaxf(lmax, fabsf(z[r][u]));
                    }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                z[r] = *(const f32x4v *)(act + (4 * grp + r) * kActLd + 4 * c4);
                if (row + r < R) {
                    float *yp = Y + (row + r) * ldy + 4 * c4;
                    if (accumulate) z[r] += *(const f32x4v *)yp;
                    *(f32x4v *)yp = z[r];
                }
            }
            constexpr bool ACT = MODE == 1;
            if (ACT && bias_period != 4) {          // plain rows (NeRF): H = a(Y) elementwise
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (row + r >= R) break;
                    f32x4v y;
#pragma unroll
                    for (int u = 0; u < 4; ++u) y[u] = act_val_rt(act_kind, z[r][u]);
                    *(f32x4v *)(H + (row + r) * ldy + 4 * c4) = y;
                }
            } else if (ACT) {   // period-4 groups: row 0 value, rows 1..3 Jacobian
                f32x4v y, dy;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float yy, dd;
                    if (act_kind == 0) act_grad<0>(z[0][u], yy, dd); else if (act_kind == 1) act_grad<1>(z[0][u], yy, dd); else act_grad<2>(z[0][u], yy, dd);
                    y[u] = yy; dy[u] = dd;
                }
                *(f32x4v *)(H + row * ldy + 4 * c4) = y;
#pragma unroll
                for (int r = 1; r < 4; ++r) *(f32x4v *)(H + (row + r) * ldy + 4 * c4) = dy * z[r];
            }
        }
    }
    if (MODE == 2 && amax_out) publish_amax(amax_out, lmax);
}

template <class Ops>
static void launch_rows_gemm_ops(int mode, const float *X, int64_t R, int ldx, int kload, const float *wp, int ksteps, const float *bias,
                                 int period, float *Y, int ldy, int accumulate, int act_kind, float *H, int cus, hipStream_t s,
                                 const float *amax_in, float *amax_out)
{
    constexpr size_t kOpd = (size_t)64 * Ops::kLd * sizeof(typename Ops::act_t), kRes = (size_t)64 * kActLd * sizeof(float);
    constexpr size_t lds = kOpd > kRes ? kOpd : kRes;
    static bool once = ((void)hipFuncSetAttribute((const void *)rows_gemm_kernel<0, Ops>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                        (void)hipFuncSetAttribute((const void *)rows_gemm_kernel<1, Ops>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                        (void)hipFuncSetAttribute((const void *)rows_gemm_kernel<2, Ops>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    int64_t tiles = (R + 63) / 64;
    int grid = (int)(tiles < 2 * cus ? tiles : 2 * cus);
    if (mode == 1)
        hipLaunchKernelGGL((rows_gemm_kernel<1, Ops>), dim3(grid), dim3(kThreads), lds, s, X, R, ldx, kload, wp, ksteps, bias, period, Y, ldy, accumulate, act_kind, H, amax_in, amax_out);
    else if (mode == 2)
        hipLaunchKernelGGL((rows_gemm_kernel<2, Ops>), dim3(grid), dim3(kThreads), lds, s, X, R, ldx, kload, wp, ksteps, bias, period, Y, ldy, accumulate, act_kind, H, amax_in, amax_out);
    else
        hipLaunchKernelGGL((rows_gemm_kernel<0, Ops>), dim3(grid), dim3(kThreads), lds, s, X, R, ldx, kload, wp, ksteps, bias, period, Y, ldy, accumulate, -1, nullptr, amax_in, amax_out);
}

static void launch_rows_gemm_mode(int mode, int split, const float *X, int64_t R, int ldx, int kload, const float *wp, int ksteps,
                                  const float *bias, int period, float *Y, int ldy, int accumulate, int act_kind, float *H, int cus,
                                  hipStream_t s, const float *amax_in, float *amax_out)
{
    if (R <= 0) return;
    if (split) launch_rows_gemm_ops<OpsF16Split>(mode, X, R, ldx, kload, wp, ksteps, bias, period, Y, ldy, accumulate, act_kind, H, cus, s, amax_in, amax_out);
    else launch_rows_gemm_ops<OpsF32>(mode, X, R, ldx, kload, wp, ksteps, bias, period, Y, ldy, accumulate, act_kind, H, cus, s, nullptr, amax_out);
}

void launch_rows_gemm(int split, const float *X, int64_t R, int ldx, int kload, const float *wp, int ksteps, const float *bias, int bias_period,
                      float *Y, int ldy, int accumulate, int act_kind, float *H, int cus, hipStream_t s, const float *amax_in)
{
    launch_rows_gemm_mode((act_kind >= 0 && H) ? 1 : 0, split, X, R, ldx, kload, wp, ksteps, bias, bias_period, Y, ldy, accumulate, act_kind, H,
                          cus, s, amax_in, nullptr);
}

void launch_rows_gemm_actback(int split, const float *X, int64_t R, int ldx, int kload, const float *wp, int ksteps, int period, int act_kind,
                              const float *Zprev, float *dZ, int ldy, int cus, hipStream_t s, const float *amax_in, float *amax_out, int accumulate)
{
    launch_rows_gemm_mode(2, split, X, R, ldx, kload, wp, ksteps, nullptr, period, dZ, ldy, accumulate, act_kind, const_cast<float *>(Zprev), cus, s,
                          amax_in, amax_out);
}


// ----------------------------------------------------------------------------
// Fused forward of a layer stack (train_kernels.h MlpForwardArgs): the inference tile engine with side stores.
// 64-row tiles, two workgroups per CU.  The skip layer's X0 product is taken at tile start, while X0 sits in LDS, and held in
// registers until its layer (as in ddf_trunk_kernel).  Z_l / H_l leave from the accumulator registers: a store instruction
// covers 2 rows x 32 consecutive columns = two full 128-byte lines.
// accumulators -> Z_l (global), H_l = a(Z_l) (global + LDS tile for the next layer)
// PM: Z_l / H_l leave in the POINT-MAJOR layout (train_kernels.h): the lane's four rows of a point are one 16-byte store
template <int KIND, bool FULL, bool LAST, int MT, int NT, class Ops, bool PM>
__device__ __forceinline__ void mlp_epilogue(const f32x16 (&acc)[MT][NT], typename Ops::act_t *act, float *Zl, float *Hl, int64_t r0,
                                             int64_t R, int wave, int lane)
{
    constexpr int LD = Ops::kLd, W = Ops::kWid;         // W: columns of the [R, W] matrices (the engine width: 256, or 512 for wide fields)
    constexpr float unscale = 1.0f / Ops::kWScale;
    const int j = lane & 31, h = lane >> 5;
    // one 64-bit base per lane; everything else is a compile-time offset from it
    const int64_t base = (r0 + 4 * h) * W + wave * NT * 32 + j;
    float *zb = Zl + base, *hb = Hl + base;
    const int64_t pbase = ((r0 >> 2) + h) * (4 * W) + (wave * NT * 32 + j) * 4;      // point r0 / 4 + h, this lane's first column
    float *zpm = Zl + pbase, *hpm = Hl + pbase;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            typename Ops::act_t *o = act + (mt * 32 + 4 * h) * LD + (wave * NT + t) * 32 + j;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float z[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) z[r] = acc[mt][t][4 * g + r] * unscale;
                float y, dy;
                act_grad<KIND>(z[0], y, dy);
                const float hv[4] = { y, dy * z[1], dy * z[2], dy * z[3] };
                const int off = (mt * 32 + 8 * g) * W + t * 32;
                if (FULL || r0 + mt * 32 + 8 * g + 4 * h < R) {     // R is a multiple of 4: a point's rows are all inside or all outside
                    if constexpr (PM) {
                        const int poff = (mt * 8 + 2 * g) * (4 * W) + t * 128;
                        __builtin_nontemporal_store(f32x4v{ z[0], z[1], z[2], z[3] }, (f32x4v *)(zpm + poff));
                        __builtin_nontemporal_store(f32x4v{ hv[0], hv[1], hv[2], hv[3] }, (f32x4v *)(hpm + poff));
                    } else
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        zb[off + r * W] = z[r];
                        hb[off + r * W] = hv[r];
                    }
                }
                if (!LAST) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Ops::put(o + (8 * g + r) * LD, hv[r]);
                }
            }
        }
}

// HOLD: the skip partial waits in 64 registers (fp32); otherwise X0 is staged a second time at the skip layer (split fp16: its
// conversion-heavy epilogue has no registers to spare -- 136 spilled with the partial held)
template <class Ops, bool HOLD, bool PM>
__global__ __launch_bounds__(kThreads, 2) void mlp_forward_kernel(const MlpForwardArgs a)
{
    typedef typename Ops::act_t act_t;
    typedef typename Ops::bfrag frag;
    // 256 columns: 64-row tiles, two column tiles per wave; 512 (round 5: wide fields, fp32): 32-row tiles, four column tiles per wave --
    // the same accumulator file and the same 66 KB LDS tile, two workgroups per CU either way (the rendering engine's rule, geo_w)
    constexpr int W = Ops::kWid, MT = W <= 256 ? 2 : 1, NT = W / 128, ROWS = MT * 32, LD = Ops::kLd;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    act_t *act = (act_t *)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const act_t *act_lane = act_lane_ptr<Ops>(act, lane);
    const int64_t ntiles = (a.R + ROWS - 1) / ROWS;
    const int c4n = a.kload0 >> 2, kpack0 = Ops::kStep * a.ksteps0;
    auto frags = [&](const float *wp, int ksteps) { return (const frag *)wp + (size_t)wave * NT * ksteps * 64 + lane; };
    auto stage = [&](const float *X, int ld, int c4cols, int kpack, int64_t r0) {       // X[r0 .. r0+64, 0:4*c4cols) -> LDS, zero padded to kpack
        // four chunks per thread and batch, requested together (row clamped, zero selected afterwards): as one load per iteration hipcc waits for
        // every chunk before it requests the next -- a tile started with up to sixteen memory round trips in a row
        const int total = ROWS * c4cols;
        for (int base = 0; base < total; base += 4 * kThreads) {
            f32x4v v[4];
            int rr[4], cc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = base + tid + i * kThreads, ok = idx < total ? idx : total - 1;
                rr[i] = ok / c4cols; cc[i] = ok - rr[i] * c4cols;
                const int64_t row = r0 + rr[i] < a.R ? r0 + rr[i] : a.R - 1;
                v[i] = *(const f32x4v *)(X + row * ld + 4 * cc[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (base + tid + i * kThreads >= total) continue;
                if (r0 + rr[i] >= a.R) v[i] = (f32x4v){ 0.f, 0.f, 0.f, 0.f };
                Ops::put4(act + rr[i] * LD + 4 * cc[i], v[i]);
            }
        }
        const int w = kpack - 4 * c4cols;
        for (int i = tid; i < ROWS * w; i += kThreads) {
            const int r = i / w, c = i - r * w;
            Ops::zero(act + r * LD + 4 * c4cols + c);
        }
    };
    auto stage_pm = [&](const float *X, int64_t r0) {        // the tile's rows of a point-major [R, W] matrix -> the row-major LDS tile
        constexpr int NV = (ROWS / 4) * W / kThreads;         // chunks per thread (W = 256: 16, one point per chunk)
        static_assert(NV % 4 == 0 && (ROWS / 4) * W == NV * kThreads, "point-major staging in batches of four chunks");
        const int64_t last = (a.R >> 2) - 1;                  // R is a multiple of 4 (a point's rows are all inside or all outside)
#pragma unroll
        for (int b = 0; b < NV; b += 4) {                     // (batches of four, as in stage)
            f32x4v v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + (b + i) * kThreads, p = idx / W, c = idx - p * W;
                const int64_t pt = (r0 >> 2) + p;
                v[i] = *(const f32x4v *)(X + (pt < last ? pt : last) * (4 * W) + 4 * c);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + (b + i) * kThreads, p = idx / W, c = idx - p * W;
                if (r0 + 4 * p >= a.R) v[i] = (f32x4v){ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int q = 0; q < 4; ++q) Ops::put(act + (4 * p + q) * LD + c, v[i][q]);
            }
        }
    };
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * ROWS;
        __syncthreads();                // the previous tile is done with the LDS tile
        stage(a.X0, a.ld0, c4n, kpack0, r0);
        __syncthreads();
        f32x16 acc[MT][NT], held[HOLD ? MT : 1][HOLD ? NT : 1];
        if constexpr (HOLD) {
            if (a.skip_layer >= 0) {
                acc_init<MT, NT, true>(held, nullptr, wave, lane);
                dense<MT, NT, Ops>(held, act_lane, frags(a.wp_skip, a.ksteps0), a.ksteps0);
            }
        }
        acc_init<MT, NT, true>(acc, a.bias[0], wave, lane, Ops::kWScale);
        dense<MT, NT, Ops>(acc, act_lane, frags(a.wp0, a.ksteps0), a.ksteps0);
        if (a.X1) {
            __syncthreads();
            if constexpr (PM) stage_pm(a.X1, r0); else stage(a.X1, W, W / 4, W, r0);
            __syncthreads();
            dense<MT, NT, Ops>(acc, act_lane, frags(a.wp1, W / Ops::kStep), W / Ops::kStep);
        }
        for (int l = 0; l < a.n_layers; ++l) {
            if (l > 0) {
                acc_init<MT, NT, true>(acc, a.bias[l], wave, lane, Ops::kWScale);
                if constexpr (HOLD) {
                    if (l == a.skip_layer) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int t = 0; t < NT; ++t) acc[mt][t] += held[mt][t];
                    }
                }
                dense<MT, NT, Ops>(acc, act_lane, frags(a.wp[l], W / Ops::kStep), W / Ops::kStep);
                if constexpr (!HOLD) {
                    if (l == a.skip_layer) {
                        __syncthreads();            // every wave finished reading the hidden state
                        stage(a.X0, a.ld0, c4n, kpack0, r0);
                        __syncthreads();
                        dense<MT, NT, Ops>(acc, act_lane, frags(a.wp_skip, a.ksteps0), a.ksteps0);
                    }
                }
            }
            __syncthreads();            // every wave finished reading the previous activations
            // one straight-line epilogue per (activation, interior / ragged tile, last layer or not): no per-group branches
            const bool full = r0 + ROWS <= a.R, last = l + 1 == a.n_layers;
            auto run = [&](auto kind, auto is_full, auto is_last) {
                mlp_epilogue<decltype(kind)::value, decltype(is_full)::value, decltype(is_last)::value, MT, NT, Ops, PM>(acc, act, a.Z[l], a.H[l], r0, a.R, wave, lane);
            };
            auto by_shape = [&](auto kind) {
                if (full) { if (last) run(kind, std::true_type{}, std::true_type{}); else run(kind, std::true_type{}, std::false_type{}); }
                else { if (last) run(kind, std::false_type{}, std::true_type{}); else run(kind, std::false_type{}, std::false_type{}); }
            };
            if (a.act_kind == 0) by_shape(std::integral_constant<int, 0>{});
            else if (a.act_kind == 1) by_shape(std::integral_constant<int, 1>{});
            else by_shape(std::integral_constant<int, 2>{});
            if (l + 1 < a.n_layers) __syncthreads();        // the next layer reads what this epilogue wrote
        }
    }
}

template <class Ops, bool HOLD, bool PM>
static void launch_mlp_forward_ops(const MlpForwardArgs &a, int cus, hipStream_t s)
{
    constexpr int ROWS = Ops::kWid <= 256 ? 64 : 32;
    constexpr size_t lds = (size_t)ROWS * Ops::kLd * sizeof(typename Ops::act_t);
    static bool once = ((void)hipFuncSetAttribute((const void *)mlp_forward_kernel<Ops, HOLD, PM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    const int64_t tiles = (a.R + ROWS - 1) / ROWS;
    hipLaunchKernelGGL((mlp_forward_kernel<Ops, HOLD, PM>), dim3((unsigned)(tiles < 2 * cus ? tiles : 2 * cus)), dim3(kThreads), lds, s, a);
}

void launch_mlp_forward(int split, const MlpForwardArgs &a, int cus, hipStream_t s)
{
    if (a.R <= 0) return;
    if (a.width == 512) {       // round 5: wide fields (point-major) on the fused chain too
        if (split) launch_mlp_forward_ops<OpsF16SplitT<512>, false, true>(a, cus, s);
        else launch_mlp_forward_ops<OpsF32T<512>, true, true>(a, cus, s);
        return;
    }
    if (split && a.point_major) launch_mlp_forward_ops<OpsF16Split, false, true>(a, cus, s);      // (round 5: the split policy's fused NeDDF route)
    else if (split) launch_mlp_forward_ops<OpsF16Split, false, false>(a, cus, s);      // (the per-layer backward reads row-major matrices)
    else if (a.point_major) launch_mlp_forward_ops<OpsF32, true, true>(a, cus, s);
    else launch_mlp_forward_ops<OpsF32, true, false>(a, cus, s);
}

// ----------------------------------------------------------------------------
// Fused input-gradient chain of a layer stack (train_kernels.h MlpBackwardArgs): see the header.  64-row tiles, two workgroups
// per CU, fp32 MFMA.
template <int KIND, int W = kWidth, int MT = 2, int NT = 2>
__device__ __forceinline__ void mlp_backward_epilogue(const f32x16 (&acc)[MT][NT], const f32x16 (&zp)[MT][NT], float *act, float *dZl, int64_t r0,
                                                      int64_t R, int wave, int lane)
{
    constexpr int LD = OpsF32T<W>::kLd;
    const int j = lane & 31, h = lane >> 5;
    float *gb = dZl + ((r0 >> 2) + h) * (4 * W) + (wave * NT * 32 + j) * 4;      // point-major: point r0 / 4 + h, this lane's first column
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float *o = act + (mt * 32 + 4 * h) * LD + (wave * NT + t) * 32 + j;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // rows 8 g + 4 h + {0, 1, 2, 3} of the tile = (value, d/dx, d/dy, d/dz) of one point, one feature
                float dy, d2;
                act_grad2<KIND>(zp[mt][t][4 * g], dy, d2);
                const float g0 = acc[mt][t][4 * g], g1 = acc[mt][t][4 * g + 1], g2 = acc[mt][t][4 * g + 2], g3 = acc[mt][t][4 * g + 3];
                float sj = g1 * zp[mt][t][4 * g + 1];
                sj += g2 * zp[mt][t][4 * g + 2];
                sj += g3 * zp[mt][t][4 * g + 3];
                const float ov[4] = { g0 * dy + sj * d2, g1 * dy, g2 * dy, g3 * dy };
                if (r0 + mt * 32 + 8 * g + 4 * h < R)
                    __builtin_nontemporal_store(f32x4v{ ov[0], ov[1], ov[2], ov[3] }, (f32x4v *)(gb + (mt * 8 + 2 * g) * (4 * W) + t * 128));
#pragma unroll
                for (int r = 0; r < 4; ++r) o[(8 * g + r) * LD] = ov[r];
            }
        }
}

// W = 512 (round 5): the [R, 512] matrices of a wide field on 32-row tiles, four column tiles per wave (see mlp_forward_kernel)
template <int KIND, int W = kWidth>      // backward kind of the stack's activation: one straight-line epilogue per kernel, not four behind run-time branches
__global__ __launch_bounds__(kThreads, 2) void mlp_backward_kernel(const MlpBackwardArgs a)
{
    typedef OpsF32T<W> Ops;
    typedef typename Ops::bfrag frag;
    constexpr int MT = W <= 256 ? 2 : 1, NT = W / 128, ROWS = MT * 32, LD = Ops::kLd, KS = W / Ops::kStep;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *act = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *act_lane = act_lane_ptr<Ops>(act, lane);
    const int j = lane & 31, h = lane >> 5;
    const int64_t ntiles = (a.R + ROWS - 1) / ROWS;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * ROWS;
        __syncthreads();                // the previous tile is done with the LDS tile
        auto stage = [&](const float *src) {            // the tile's rows of a point-major matrix -> the row-major LDS tile
            for (int idx = tid; idx < (ROWS / 4) * W; idx += kThreads) {
                const int p = idx / W, c = idx - p * W;
                f32x4v v = { 0.f, 0.f, 0.f, 0.f };
                if (r0 + 4 * p < a.R) v = *(const f32x4v *)(src + ((r0 >> 2) + p) * (4 * W) + 4 * c);
#pragma unroll
                for (int q = 0; q < 4; ++q) act[(4 * p + q) * LD + c] = v[q];
            }
        };
        // Z of this lane's accumulator positions: the four rows of a point for one feature = one 16-byte load in the point-major layout
        auto load_z = [&](const float *Z, f32x16 (&zp)[MT][NT]) {
            const float *zb = Z + ((r0 >> 2) + h) * (4 * W) + (wave * NT * 32 + j) * 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4v v = { 0.f, 0.f, 0.f, 0.f };
                        if (r0 + mt * 32 + 8 * g + 4 * h < a.R) v = __builtin_nontemporal_load((const f32x4v *)(zb + (mt * 8 + 2 * g) * (4 * W) + t * 128));
#pragma unroll
                        for (int r = 0; r < 4; ++r) zp[mt][t][4 * g + r] = v[r];
                    }
        };
        if (a.dZtop) {
            stage(a.dZtop);
            __syncthreads();
        } else {
            // prologue: the top layer's gradient is formed here instead of by two more passes over HBM (a plain GEMM kernel for the
            // colour trunk's feature gradient + a kernel that adds the heads and applies the top activation's backward)
            f32x16 zp[MT][NT], acc[MT][NT];
            load_z(a.top_Z, zp);
            acc_init<MT, NT, false>(acc, nullptr, wave, lane);
            if (a.top_src) {
                stage(a.top_src);
                __syncthreads();
                dense<MT, NT, Ops>(acc, act_lane, (const frag *)a.top_wT + (size_t)wave * NT * KS * 64 + lane, KS);
                __syncthreads();        // every wave finished reading the staged matrix
            }
            float hw[3][NT];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int t = 0; t < NT; ++t) hw[c][t] = c < a.top_nc ? a.top_w[c][(size_t)(wave * NT * 32 + t * 32 + j) * a.top_wstride] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t row = r0 + mt * 32 + 8 * g + 4 * h + r;
                        float gv[3] = { 0.f, 0.f, 0.f };
                        if (row < a.R)
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                if (c < a.top_nc) gv[c] = a.top_G[row * a.top_ldg + c];
#pragma unroll
                        for (int t = 0; t < NT; ++t)
#pragma unroll
                            for (int c = 0; c < 3; ++c) acc[mt][t][4 * g + r] = fmaf(gv[c], hw[c][t], acc[mt][t][4 * g + r]);
                    }
            mlp_backward_epilogue<KIND, W, MT, NT>(acc, zp, act, a.top_out, r0, a.R, wave, lane);
            __syncthreads();
        }
        for (int l = a.n_layers - 1; l >= 1; --l) {
            // Z_{l-1} of this lane's accumulator positions, requested before the product (rows past R: zero; R is a multiple of 4)
            f32x16 zp[MT][NT];
            load_z(a.Z[l - 1], zp);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc[MT][NT];
            acc_init<MT, NT, false>(acc, nullptr, wave, lane);
            dense<MT, NT, Ops>(acc, act_lane, (const frag *)a.wT[l] + (size_t)wave * NT * KS * 64 + lane, KS);
            __syncthreads();            // every wave finished reading dZ_l
            mlp_backward_epilogue<KIND, W, MT, NT>(acc, zp, act, a.dZ[l - 1], r0, a.R, wave, lane);
            if (l > 1) __syncthreads(); // the next layer reads what this epilogue wrote
        }
    }
}

// ----------------------------------------------------------------------------
// The same chain under the split-fp16 policy (round 5; train_kernels.h MlpBackwardArgs with `split`): every product is three fp16 MFMAs on
// operands of two fp16 terms, the activation backward stays fp32 on the accumulators.  Gradient matrices sit at 1e-5 .. 1e-8 (a mean over
// rays), below what two fp16 terms resolve, so every gradient tile that enters a product is RANGE-SCALED by a power of two -- like the
// per-layer route does per matrix (operand_scale), but per 64-row TILE, the only maximum a fused chain can know: the epilogue runs in
// two passes around the barrier it needs anyway -- (1) the new gradient into the accumulator registers and out to dZ_l (unscaled fp32),
// the lane / wave maximum of |dZ_l| into LDS; barrier (every wave is done reading the old tile AND the four wave maxima are visible);
// (2) scaled by the tile's power of two, split into two fp16 terms, into the LDS tile.  The accumulators of the next product carry
// that scale and the weights' 2^10: both come off (exactly) at the start of the next epilogue.  Each dZ_l also leaves its global
// maximum in a device scalar for the weight-gradient products that follow (dw_split_kernel scales its G operand by it).
template <int KIND, int W = kWidth>
__global__ __launch_bounds__(kThreads, 2) void mlp_backward_split_kernel(const MlpBackwardArgs a)
{
    typedef OpsF16SplitT<W> Ops;
    typedef typename Ops::bfrag frag;
    typedef typename Ops::act_t act_t;
    constexpr int MT = W <= 256 ? 2 : 1, NT = W / 128, ROWS = MT * 32, LD = Ops::kLd, KS = W / Ops::kStep;      // (512: 32-row tiles, see mlp_forward_kernel)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    act_t *act = (act_t *)smem;
    float *smax = (float *)(act + ROWS * LD);           // [4]: the waves' maxima of the gradient tile in flight
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const act_t *act_lane = act_lane_ptr<Ops>(act, lane);
    const int j = lane & 31, h = lane >> 5;
    const int64_t ntiles = (a.R + ROWS - 1) / ROWS;
    auto wave_max = [&](float v) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
        return v;
    };
    auto publish = [&](float *amax_out, float wmax) {       // (see publish_amax: look first, most waves skip the same-address atomic)
        if (amax_out && lane == 0 && wmax > *(volatile const float *)amax_out) atomicMax((int *)amax_out, __builtin_bit_cast(int, wmax));
    };
    auto tile_scale = [&]() { return operand_scale_of(fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]))); };
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * ROWS;
        __syncthreads();                // the previous tile is done with the LDS tile (and with smax)
        // Z of this lane's accumulator positions: the four rows of a point for one feature = one 16-byte load in the point-major layout
        auto load_z = [&](const float *Z, f32x16 (&zp)[MT][NT]) {
            const float *zb = Z + ((r0 >> 2) + h) * (4 * W) + (wave * NT * 32 + j) * 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4v v = { 0.f, 0.f, 0.f, 0.f };
                        if (r0 + mt * 32 + 8 * g + 4 * h < a.R) v = __builtin_nontemporal_load((const f32x4v *)(zb + (mt * 8 + 2 * g) * (4 * W) + t * 128));
#pragma unroll
                        for (int r = 0; r < 4; ++r) zp[mt][t][4 * g + r] = v[r];
                    }
        };
        // pass 1 of an epilogue: acc (upstream gradient of H_l in this lane's accumulator positions, already unscaled) -> dZ_l through the
        // activation backward, in place; out to the point-major matrix; returns the lane's maximum of |dZ_l|
        auto backward_values = [&](f32x16 (&acc)[MT][NT], const f32x16 (&zp)[MT][NT], float *dZl) {
            float lmax = 0.f;
            float *gb = dZl + ((r0 >> 2) + h) * (4 * W) + (wave * NT * 32 + j) * 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float dy, d2;
                        act_grad2<KIND>(zp[mt][t][4 * g], dy, d2);
                        const float g0 = acc[mt][t][4 * g], g1 = acc[mt][t][4 * g + 1], g2 = acc[mt][t][4 * g + 2], g3 = acc[mt][t][4 * g + 3];
                        float sj = g1 * zp[mt][t][4 * g + 1];
                        sj += g2 * zp[mt][t][4 * g + 2];
                        sj += g3 * zp[mt][t][4 * g + 3];
                        const float ov[4] = { g0 * dy + sj * d2, g1 * dy, g2 * dy, g3 * dy };
                        const bool in = r0 + mt * 32 + 8 * g + 4 * h < a.R;
                        if (in) __builtin_nontemporal_store(f32x4v{ ov[0], ov[1], ov[2], ov[3] }, (f32x4v *)(gb + (mt * 8 + 2 * g) * (4 * W) + t * 128));
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc[mt][t][4 * g + r] = in ? ov[r] : 0.f;
                            lmax = fmaxf(lmax, fabsf(acc[mt][t][4 * g + r]));
                        }
                    }
            return lmax;
        };
        // pass 2: the gradient tile, scaled and split, into LDS
        auto to_lds = [&](const f32x16 (&acc)[MT][NT], float scale) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    act_t *o = act + (mt * 32 + 4 * h) * LD + (wave * NT + t) * 32 + j;
#pragma unroll
                    for (int q = 0; q < 16; ++q) Ops::put(o + (8 * (q >> 2) + (q & 3)) * LD, acc[mt][t][q] * scale);
                }
        };
        float s_cur = 1.0f;             // the power of two the gradient tile in LDS carries
        {
            // prologue: the top layer's gradient (MlpBackwardArgs: top_src x top_wT + the narrow heads, then the top activation's backward)
            f32x16 zp[MT][NT], acc[MT][NT];
            load_z(a.top_Z, zp);
            acc_init<MT, NT, false>(acc, nullptr, wave, lane);
            float unscale = 1.0f / Ops::kWScale;
            if (a.top_src) {
                // 64 rows of the point-major gradient matrix -> registers, their maximum -> the tile's scale -> the LDS tile
                // (read twice -- the second time from L2 -- rather than held: 64 registers next to Z and the accumulators would spill)
                constexpr int NV = (ROWS / 4) * W / kThreads;
                auto piece = [&](int i) {
                    const int idx = tid + i * kThreads, p = idx / W, c = idx - p * W;
                    f32x4v v = { 0.f, 0.f, 0.f, 0.f };
                    if (r0 + 4 * p < a.R) v = *(const f32x4v *)(a.top_src + ((r0 >> 2) + p) * (4 * W) + 4 * c);
                    return v;
                };
                float lmax = 0.f;
#pragma unroll 4
                for (int i = 0; i < NV; ++i) {
                    const f32x4v v = piece(i);
#pragma unroll
                    for (int q = 0; q < 4; ++q) lmax = fmaxf(lmax, fabsf(v[q]));
                }
                const float wmax = wave_max(lmax);
                if (lane == 0) smax[wave] = wmax;
                __syncthreads();
                const float sc = tile_scale();
#pragma unroll 4
                for (int i = 0; i < NV; ++i) {
                    const int idx = tid + i * kThreads, p = idx / W, c = idx - p * W;
                    const f32x4v v = piece(i);
#pragma unroll
                    for (int q = 0; q < 4; ++q) Ops::put(act + (4 * p + q) * LD + c, v[q] * sc);
                }
                __syncthreads();
                dense<MT, NT, Ops>(acc, act_lane, (const frag *)a.top_wT + (size_t)wave * NT * KS * 64 + lane, KS);
                unscale = pow2_inverse(sc) * (1.0f / Ops::kWScale);
            }
            float hw[3][NT];
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int t = 0; t < NT; ++t) hw[c][t] = c < a.top_nc ? a.top_w[c][(size_t)(wave * NT * 32 + t * 32 + j) * a.top_wstride] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t row = r0 + mt * 32 + 8 * g + 4 * h + r;
                        float gv[3] = { 0.f, 0.f, 0.f };
                        if (row < a.R)
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                if (c < a.top_nc) gv[c] = a.top_G[row * a.top_ldg + c];
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            float v = acc[mt][t][4 * g + r] * unscale;
#pragma unroll
                            for (int c = 0; c < 3; ++c) v = fmaf(gv[c], hw[c][t], v);
                            acc[mt][t][4 * g + r] = v;
                        }
                    }
            const float wmax = wave_max(backward_values(acc, zp, a.top_out));
            publish(a.amax_top, wmax);
            if (a.n_layers >= 2) {
                if (lane == 0) smax[wave] = wmax;
                __syncthreads();        // every wave finished reading the staged matrix; the maxima are visible
                s_cur = tile_scale();
                to_lds(acc, s_cur);
                __syncthreads();
            }
        }
        for (int l = a.n_layers - 1; l >= 1; --l) {
            f32x16 zp[MT][NT];
            load_z(a.Z[l - 1], zp);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc[MT][NT];
            acc_init<MT, NT, false>(acc, nullptr, wave, lane);
            dense<MT, NT, Ops>(acc, act_lane, (const frag *)a.wT[l] + (size_t)wave * NT * KS * 64 + lane, KS);
            const float unscale = pow2_inverse(s_cur) * (1.0f / Ops::kWScale);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[mt][t][q] *= unscale;
            const float wmax = wave_max(backward_values(acc, zp, a.dZ[l - 1]));
            publish(a.amax_dZ[l - 1], wmax);
            if (l > 1) {
                if (lane == 0) smax[wave] = wmax;
                __syncthreads();        // every wave finished reading dZ_l; the maxima are visible
                s_cur = tile_scale();
                to_lds(acc, s_cur);
                __syncthreads();        // the next layer reads what this epilogue wrote
            }
        }
    }
}

void launch_mlp_backward(int split, const MlpBackwardArgs &a, int cus, hipStream_t s)
{
    if (a.R <= 0 || (a.n_layers < 2 && a.dZtop)) return;         // a one-layer stack still has a prologue to run
    if (split && a.dZtop) {             // a caller's error, not a data condition: the split chain exists in its prologue form only
        fprintf(stderr, "neddf: launch_mlp_backward(split = 1) takes the top gradient through top_G / top_src, not dZtop\n");
        abort();
    }
    if (split && a.width == 512) {      // wide fields under the split policy (round 5)
        constexpr size_t lds5 = (size_t)32 * OpsF16SplitT<512>::kLd * sizeof(OpsF16Split::act_t) + 16 * sizeof(float);
        static bool once5 = ((void)hipFuncSetAttribute((const void *)mlp_backward_split_kernel<0, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5),
                             (void)hipFuncSetAttribute((const void *)mlp_backward_split_kernel<1, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5),
                             (void)hipFuncSetAttribute((const void *)mlp_backward_split_kernel<2, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5),
                             (void)hipFuncSetAttribute((const void *)mlp_backward_split_kernel<3, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5), true);
        (void)once5;
        const int64_t tiles5 = (a.R + 31) / 32;
        const dim3 grid5((unsigned)(tiles5 < 2 * cus ? tiles5 : 2 * cus));
        if (a.act_kind == 0) hipLaunchKernelGGL((mlp_backward_split_kernel<0, 512>), grid5, dim3(kThreads), lds5, s, a);
        else if (a.act_kind == 1) hipLaunchKernelGGL((mlp_backward_split_kernel<1, 512>), grid5, dim3(kThreads), lds5, s, a);
        else if (a.act_kind == 2) hipLaunchKernelGGL((mlp_backward_split_kernel<2, 512>), grid5, dim3(kThreads), lds5, s, a);
        else hipLaunchKernelGGL((mlp_backward_split_kernel<3, 512>), grid5, dim3(kThreads), lds5, s, a);
        return;
    }
    if (split) {        // (prologue form only: the NeDDF route, the one caller)
        constexpr size_t lds = (size_t)64 * OpsF16Split::kLd * sizeof(OpsF16Split::act_t) + 16 * sizeof(float);
        static bool once = ((void)hipFuncSetAttribute((const void *)mlp_backward_split_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                            (void)hipFuncSetAttribute((const void *)mlp_backward_split_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                            (void)hipFuncSetAttribute((const void *)mlp_backward_split_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                            (void)hipFuncSetAttribute((const void *)mlp_backward_split_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
        (void)once;
        const int64_t tiles = (a.R + 63) / 64;
        const dim3 grid((unsigned)(tiles < 2 * cus ? tiles : 2 * cus));
        if (a.act_kind == 0) hipLaunchKernelGGL(mlp_backward_split_kernel<0>, grid, dim3(kThreads), lds, s, a);
        else if (a.act_kind == 1) hipLaunchKernelGGL(mlp_backward_split_kernel<1>, grid, dim3(kThreads), lds, s, a);
        else if (a.act_kind == 2) hipLaunchKernelGGL(mlp_backward_split_kernel<2>, grid, dim3(kThreads), lds, s, a);
        else hipLaunchKernelGGL(mlp_backward_split_kernel<3>, grid, dim3(kThreads), lds, s, a);
        return;
    }
    if (a.width == 512) {       // wide fields (round 5): 32-row tiles of [R, 512] matrices
        constexpr size_t lds5 = (size_t)32 * OpsF32T<512>::kLd * sizeof(float);
        static bool once5 = ((void)hipFuncSetAttribute((const void *)mlp_backward_kernel<0, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5),
                             (void)hipFuncSetAttribute((const void *)mlp_backward_kernel<1, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5),
                             (void)hipFuncSetAttribute((const void *)mlp_backward_kernel<2, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5),
                             (void)hipFuncSetAttribute((const void *)mlp_backward_kernel<3, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5), true);
        (void)once5;
        const int64_t tiles5 = (a.R + 31) / 32;
        const dim3 grid5((unsigned)(tiles5 < 2 * cus ? tiles5 : 2 * cus));
        if (a.act_kind == 0) hipLaunchKernelGGL((mlp_backward_kernel<0, 512>), grid5, dim3(kThreads), lds5, s, a);
        else if (a.act_kind == 1) hipLaunchKernelGGL((mlp_backward_kernel<1, 512>), grid5, dim3(kThreads), lds5, s, a);
        else if (a.act_kind == 2) hipLaunchKernelGGL((mlp_backward_kernel<2, 512>), grid5, dim3(kThreads), lds5, s, a);
        else hipLaunchKernelGGL((mlp_backward_kernel<3, 512>), grid5, dim3(kThreads), lds5, s, a);
        return;
    }
    constexpr size_t lds = (size_t)64 * OpsF32::kLd * sizeof(float);
    static bool once = ((void)hipFuncSetAttribute((const void *)mlp_backward_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                        (void)hipFuncSetAttribute((const void *)mlp_backward_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                        (void)hipFuncSetAttribute((const void *)mlp_backward_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                        (void)hipFuncSetAttribute((const void *)mlp_backward_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    const int64_t tiles = (a.R + 63) / 64;
    const dim3 grid((unsigned)(tiles < 2 * cus ? tiles : 2 * cus));
    if (a.act_kind == 0) hipLaunchKernelGGL(mlp_backward_kernel<0>, grid, dim3(kThreads), lds, s, a);
    else if (a.act_kind == 1) hipLaunchKernelGGL(mlp_backward_kernel<1>, grid, dim3(kThreads), lds, s, a);
    else if (a.act_kind == 2) hipLaunchKernelGGL(mlp_backward_kernel<2>, grid, dim3(kThreads), lds, s, a);
    else hipLaunchKernelGGL(mlp_backward_kernel<3>, grid, dim3(kThreads), lds, s, a);
}

// ----------------------------------------------------------------------------
// dW[K, 256] += X[R, 0:K]^T x G[R, 0:256], db[n] += sum over rows r % bias_period == 0 of G[r, n]
// (LinearGradFunction.backward, linear.py:75-82: x^T dLdy + J^T dLdG is one product over the stacked value + Jacobian rows).
// Each workgroup owns a contiguous range of rows and the WHOLE K x 256 output in accumulators (wave w: all K-tiles x output
// columns [64w, 64w+64)), so X and G are read from HBM exactly once; 32-row chunks are staged through LDS with the next
// chunk in flight (global -> VGPR) during the MFMAs.  The contraction index of v_mfma_f32_32x32x2_f32 is the row:
// A[i = k][kk = row parity], B[kk][j = n]; LDS row strides are 32 mod 64 floats so the two row parities hit disjoint banks.

// rows [rb, re) of one weight-gradient product; the LDS layout and the accumulator file are those of KT k-tiles, KTN <= KT of them
// carry data (K <= 32 KTN) and are multiplied -- a compile-time count, so that the job-parallel kernel below can serve products of
// different K from one accumulator allocation without run-time tests between its MFMAs
// XPM / GPM: X / G arrive in the POINT-MAJOR layout (train_kernels.h; 256 columns, ld 256) and are staged as they are: a 32-row chunk
// is 8 points x 256 columns x 4 rows, and a lane's operands of TWO row pairs are one 8-byte LDS read.  The contraction pairs rows
// (4p + e, 4p + 2 + e), e = 0, 1, of a point in either layout (any pairing is the same sum; the two matrices only have to agree).
typedef float f32x2v __attribute__((ext_vector_type(2)));
template <int KT, int KTN = KT, bool XPM = false, bool GPM = false>
__device__ __forceinline__ void dw_tile_rows(const float *X, int ldx, int K, const float *G, int ldg, int64_t rb, int64_t re, float *dW,
                                             int64_t sk, int64_t sn, int nvalid, float *db, int bias_period)
{
    constexpr int RC = 32, KP = 32 * KT, LDX = ((KP + 32) % 64 == 32) ? KP + 32 : KP + 64, LDG = kWidth + 32;
    constexpr int XPF = (RC * (KP / 4) + kThreads - 1) / kThreads, GPF = RC * (kWidth / 4) / kThreads;
    constexpr int PS = 4 * kWidth;              // floats per point in the point-major layout
    static_assert(!XPM || KT == 8, "a point-major X has 256 columns");
    static_assert(kThreads == kWidth, "point-major staging: thread = column");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Xs = smem, *Gs = smem + RC * LDX;    // (a point-major chunk is 8 PS floats <= RC LDX)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int n0 = wave * 64;
    if (rb >= re) return;
    const int k4 = (K + 3) >> 2;                 // float4 columns actually present (ldx >= 4 * k4)
    f32x16 acc[KT][2];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[kt][t][q] = 0.f;
    // this thread's share of db.  Row-major G: columns 4 (tid & 63) .. +3 of rows wave + 4 i; point-major: column tid (component 0)
    f32x4v bsum = { 0.f, 0.f, 0.f, 0.f };
    f32x4v xp[XPF], gp[GPF];
    // Which 16-byte pieces of a 32-row chunk this thread stages never changes: its element offsets from the chunk's first row are
    // computed once (-1: a padding column of X), and a FULL chunk is fetched with one 64-bit base per matrix and those offsets.
    // Per chunk the index arithmetic, three bounds tests and the 64-bit address of every piece had been ~370 vector
    // instructions per wave in front of 256 MFMAs, with nothing to overlap them at one wave per SIMD (PMC: 1.45 per MFMA).
    // (point-major: piece i = point i of the chunk, column tid; the chunk's first element is at c0 * 256 either way)
    int xoff[XPF], goff[GPF];
#pragma unroll
    // (point-major operands: ldx / ldg = the columns of the WHOLE matrix -- 256, or 512 for a wide field, whose 256-column block the
    // caller selects through the base pointer --, so a point is 4 ld floats in memory; in LDS a chunk keeps PS floats per point)
    for (int i = 0; i < XPF; ++i) {
        const int idx = tid + i * kThreads, r = idx / (KP / 4), c = idx - r * (KP / 4);
        xoff[i] = XPM ? i * 4 * ldx + 4 * tid : ((r < RC && c < k4) ? r * ldx + 4 * c : -1);
    }
#pragma unroll
    for (int i = 0; i < GPF; ++i) {
        const int idx = tid + i * kThreads;
        goff[i] = GPM ? i * 4 * ldg + 4 * tid : (idx >> 6) * ldg + 4 * (idx & 63);
    }
    auto xrow = [&](int i) { return XPM ? 4 * i : (tid + i * kThreads) / (KP / 4); };       // first chunk row of piece i
    auto grow = [&](int i) { return GPM ? 4 * i : (tid + i * kThreads) >> 6; };
    auto fetch = [&](int64_t c0) {
        const float *xb = X + c0 * ldx, *gb = G + c0 * ldg;
        if (c0 + RC <= re) {            // a full chunk: no row tests
#pragma unroll
            for (int i = 0; i < XPF; ++i) {
                f32x4v v = { 0.f, 0.f, 0.f, 0.f };
                if (xoff[i] >= 0) v = *(const f32x4v *)(xb + xoff[i]);
                xp[i] = v;
            }
#pragma unroll
            for (int i = 0; i < GPF; ++i) gp[i] = *(const f32x4v *)(gb + goff[i]);
            return;
        }
#pragma unroll
        for (int i = 0; i < XPF; ++i) {
            f32x4v v = { 0.f, 0.f, 0.f, 0.f };
            if (xoff[i] >= 0 && c0 + xrow(i) < re) v = *(const f32x4v *)(xb + xoff[i]);
            xp[i] = v;
        }
#pragma unroll
        for (int i = 0; i < GPF; ++i) {
            f32x4v v = { 0.f, 0.f, 0.f, 0.f };
            if (c0 + grow(i) < re) v = *(const f32x4v *)(gb + goff[i]);
            gp[i] = v;
        }
    };
    fetch(rb);
    for (int64_t c0 = rb; c0 < re; c0 += RC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < XPF; ++i) {
            int idx = tid + i * kThreads, r = idx / (KP / 4), c = idx - r * (KP / 4);
            if (XPM) { *(f32x4v *)(Xs + i * PS + 4 * tid) = xp[i]; }
            else if (r < RC) *(f32x4v *)(Xs + r * LDX + 4 * c) = xp[i];
        }
#pragma unroll
        for (int i = 0; i < GPF; ++i) {
            int idx = tid + i * kThreads;
            // db: column sums over the value rows, taken here from the staged registers (row-major: row idx >> 6 = wave + 4 i; chunks
            // start at multiples of 32 and bias_period is 1 or 4, so the row's phase is its phase in the chunk; rows past `re` are zero).
            // Inside the MFMA loop -- first as a 64-bit vector modulo per row pair, 280 of that loop's 314 vector instructions per
            // 64 MFMAs, then as a masked add -- every one of these instructions broke the MFMA issue chain
            if (GPM) {
                *(f32x4v *)(Gs + i * PS + 4 * tid) = gp[i];
                if (db) bsum[0] += bias_period == 4 ? gp[i][0] : (gp[i][0] + gp[i][1]) + (gp[i][2] + gp[i][3]);
            } else {
                *(f32x4v *)(Gs + (idx >> 6) * LDG + 4 * (idx & 63)) = gp[i];
                if (db && (((idx >> 6) & (bias_period - 1)) == 0)) bsum += gp[i];
            }
        }
        __syncthreads();
        if (c0 + RC < re) fetch(c0 + RC);
        __builtin_amdgcn_sched_barrier(0);
        // Software pipeline over the points of the chunk (2 row pairs each): the operands of point p + 1 are requested before the
        // 4 KT MFMAs of point p issue.  Left to itself hipcc reads each pair of k-tiles right in front of its four MFMAs and waits
        // (`ds_read2_b32; s_waitcnt lgkmcnt(0)` every 4 MFMAs: the LDS latency of 16 loads per chunk row pair, un-overlapped with
        // one wave per SIMD) -- the kernel streamed at 68 % of the matrix peak (PMC: 1.45 vector instructions per MFMA, matrix pipe
        // busy 68 %)
        const float *xr = XPM ? Xs + 4 * j + 2 * h : Xs + 2 * h * LDX + j;
        const float *gr = GPM ? Gs + 4 * (n0 + j) + 2 * h : Gs + 2 * h * LDG + n0 + j;
        float a0[KT][2], a1[KT][2], b0[2][2], b1[2][2];
        auto load_point = [&](float (&a)[KT][2], float (&b)[2][2], int p) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (GPM) { const f32x2v v = *(const f32x2v *)(gr + p * PS + t * 128); b[t][0] = v[0]; b[t][1] = v[1]; }
                else { b[t][0] = gr[(4 * p) * LDG + 32 * t]; b[t][1] = gr[(4 * p + 1) * LDG + 32 * t]; }
            }
#pragma unroll
            for (int kt = 0; kt < KTN; ++kt) {
                if (XPM) { const f32x2v v = *(const f32x2v *)(xr + p * PS + kt * 128); a[kt][0] = v[0]; a[kt][1] = v[1]; }
                else { a[kt][0] = xr[(4 * p) * LDX + 32 * kt]; a[kt][1] = xr[(4 * p + 1) * LDX + 32 * kt]; }
            }
        };
        auto mfma_point = [&](const float (&a)[KT][2], const float (&b)[2][2]) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int kt = 0; kt < KTN; ++kt) {
                    acc[kt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kt][e], b[0][e], acc[kt][0], 0, 0, 0);
                    acc[kt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kt][e], b[1][e], acc[kt][1], 0, 0, 0);
                }
        };
        load_point(a0, b0, 0);
#pragma unroll
        for (int p = 0; p < RC / 4; p += 2) {
            load_point(a1, b1, p + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_point(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (p + 2 < RC / 4) load_point(a0, b0, p + 2);
            __builtin_amdgcn_sched_barrier(0);
            mfma_point(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int kt = 0; kt < KTN; ++kt) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                int k = 32 * kt + 8 * (q >> 2) + 4 * h + (q & 3), n = n0 + 32 * t + j;
                if (k < K && n < nvalid) atomicAdd(&dW[k * sk + n * sn], acc[kt][t][q]);
            }
    }
    if (GPM) {
        if (db && tid < nvalid) atomicAdd(&db[tid], bsum[0]);
    } else if (db && (wave & (bias_period - 1)) == 0) {        // staged rows are wave + 4 i: with bias_period 4 only wave 0 holds value rows
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (4 * lane + q < nvalid) atomicAdd(&db[4 * lane + q], bsum[q]);
    }
}

template <int KT, int KTN = KT, bool XPM = false, bool GPM = false>
__global__ __launch_bounds__(kThreads, 1) void dw_tile_kernel(const float *X, int ldx, int K, const float *G, int ldg, int64_t R,
                                                              int64_t rows_per_wg, float *dW, int64_t sk, int64_t sn, int nvalid, float *db,
                                                              int bias_period)
{
    const int64_t rb = (int64_t)blockIdx.x * rows_per_wg;
    dw_tile_rows<KT, KTN, XPM, GPM>(X, ldx, K, G, ldg, rb, rb + rows_per_wg < R ? rb + rows_per_wg : R, dW, sk, sn, nvalid, db, bias_period);
}

// Job-parallel weight gradients (train_kernels.h DwJobs): every product of a backward pass in ONE launch, each workgroup working on ONE
// of them.  A workgroup's epilogue is 65 536 device-scope atomic adds whatever its share of rows (profiles/r03_dw_ablation.txt: 1.8 ms
// of a step when every product is its own launch over all CUs); with the workgroups divided among the products instead of every
// workgroup visiting every product, a pass issues one such epilogue per workgroup, not one per workgroup and product.
__global__ __launch_bounds__(kThreads, 1) void dw_jobs_kernel(const DwJobs jobs)
{
    int jb = 0;
    while (jb + 1 < jobs.n && (int)blockIdx.x >= jobs.job[jb + 1].wg0) ++jb;
    const DwJob &J = jobs.job[jb];
    const int w = (int)blockIdx.x - J.wg0, nw = (jb + 1 < jobs.n ? jobs.job[jb + 1].wg0 : (int)gridDim.x) - J.wg0;
    const int64_t chunks = (jobs.R + 31) / 32, per = (chunks + nw - 1) / nw;
    const int64_t rb = (int64_t)w * per * 32, re_ = rb + per * 32;
    const int64_t re = re_ < jobs.R ? re_ : jobs.R;
    // G (a dZ of the fused backward chain) is point-major in every job; X is point-major when it is a hidden state (K = 256), row-major
    // when it is one of the narrow first-layer inputs
    if (J.x_point_major) dw_tile_rows<8, 8, true, true>(J.X, J.ldx, J.K, J.G, J.ldg, rb, re, J.dW, J.sk, J.sn, J.nvalid, J.db, J.bias_period);
    else if (J.K <= 64) dw_tile_rows<8, 2, false, true>(J.X, J.ldx, J.K, J.G, J.ldg, rb, re, J.dW, J.sk, J.sn, J.nvalid, J.db, J.bias_period);
    else if (J.K <= 96) dw_tile_rows<8, 3, false, true>(J.X, J.ldx, J.K, J.G, J.ldg, rb, re, J.dW, J.sk, J.sn, J.nvalid, J.db, J.bias_period);
    else dw_tile_rows<8, 8, false, true>(J.X, J.ldx, J.K, J.G, J.ldg, rb, re, J.dW, J.sk, J.sn, J.nvalid, J.db, J.bias_period);
}

void launch_dw_jobs(DwJobs &jobs, int cus, hipStream_t s)
{
    if (jobs.n <= 0 || jobs.R <= 0) return;
    // workgroups in proportion to the products' cost: matrix work ~ K, plus the streaming of G that every product pays
    float cost[kMaxDwJobs], total = 0.f;
    // (the constant = the staging / streaming share of a chunk, ~a quarter of a 256-wide product's time; measured flat between 32 and 192)
    for (int i = 0; i < jobs.n; ++i) { const int K = jobs.job[i].K; cost[i] = 64.0f + (float)(K <= 64 ? 64 : K <= 96 ? 96 : 256); total += cost[i]; }
    int grid = cus > jobs.n ? cus : jobs.n, at = 0;
    for (int i = 0; i < jobs.n; ++i) {
        jobs.job[i].wg0 = at;
        int share = (int)(cost[i] / total * grid + 0.5f);
        if (share < 1) share = 1;
        const int left = jobs.n - 1 - i;
        if (at + share > grid - left) share = grid - left - at;
        at += share;
    }
    grid = at;
    constexpr int KP = 256, LDX = KP + 32, LDG = kWidth + 32;
    const size_t lds = (size_t)32 * (LDX + LDG) * sizeof(float);
    static bool once = ((void)hipFuncSetAttribute((const void *)dw_jobs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(32 * (LDX + LDG) * sizeof(float))), true);
    (void)once;
    hipLaunchKernelGGL(dw_jobs_kernel, dim3(grid), dim3(kThreads), lds, s, jobs);
}

template <int KT, int KTN = KT, bool XPM = false, bool GPM = false>
static void launch_dw_tile(const float *X, int ldx, int K, const float *G, int ldg, int64_t R, float *dW, int64_t sk, int64_t sn, int nvalid,
                           float *db, int bias_period, int cus, hipStream_t s)
{
    constexpr int KP = 32 * KT, LDX = ((KP + 32) % 64 == 32) ? KP + 32 : KP + 64, LDG = kWidth + 32;
    const size_t lds = (size_t)32 * (LDX + LDG) * sizeof(float);
    static bool once = ((void)hipFuncSetAttribute((const void *)dw_tile_kernel<KT, KTN, XPM, GPM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(32 * (LDX + LDG) * sizeof(float))), true);
    (void)once;
    int64_t chunks = (R + 31) / 32;
    int grid = (int)(chunks < cus ? chunks : cus);
    int64_t rows_per_wg = ((chunks + grid - 1) / grid) * 32;
    hipLaunchKernelGGL((dw_tile_kernel<KT, KTN, XPM, GPM>), dim3(grid), dim3(kThreads), lds, s, X, ldx, K, G, ldg, R, rows_per_wg, dW, sk, sn, nvalid, db, bias_period);
}

// Heads with 1..4 output columns: dW_c[k] += sum_r X[r, k] G[r, c], db_c += sum over value rows of G[r, c].
// One thread per input feature k, rows strided over workgroups; X is streamed once, fully coalesced (HBM-bound).
struct NarrowGrad {
    int nc;
    float *w[4];          // column c of the weight gradient: w[c][k * wstride]
    int wstride;
    float *b[4];          // scalar bias gradients (or NULL)
    int kcount;           // input features present (<= 256)
};
template <bool PM>      // PM: X point-major (ld 256); rows_per_wg is a multiple of 16
__global__ __launch_bounds__(kThreads) void narrow_dw_kernel(const float *X, int ldx, const float *G, int ldg, int64_t R, int64_t rows_per_wg,
                                                             NarrowGrad o, int bias_period)
{
    const int k = threadIdx.x;
    const int64_t rb = (int64_t)blockIdx.x * rows_per_wg, re = rb + rows_per_wg < R ? rb + rows_per_wg : R;
    float acc[4] = { 0.f, 0.f, 0.f, 0.f }, bs[4] = { 0.f, 0.f, 0.f, 0.f };
    if (PM && ldg == 4) {
        // Sixteen rows (four points) per pass, the next sixteen in flight; the rows of G (64 floats) as ONE coalesced load per wave,
        // broadcast with readlane (per-thread loads of the same address would be 16 x nc more vector-memory instructions than the X
        // stream itself).  Rows beyond `re` read as zero (R is a multiple of 4: points are whole), so the loop has no branches.
        const f32x4v zero = { 0.f, 0.f, 0.f, 0.f };
        f32x4v xn[4];
        float gn;
        auto fetch = [&](int64_t r0) {
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) xn[pt] = r0 + 4 * pt < re ? *(const f32x4v *)(X + ((r0 >> 2) + pt) * (4 * (int64_t)ldx) + 4 * k) : zero;
            const int64_t gi = r0 * 4 + (k & 63);
            gn = gi < re * 4 ? G[gi] : 0.f;
        };
        fetch(rb);
        for (int64_t r0 = rb; r0 < re; r0 += 16) {
            f32x4v x[4];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) x[pt] = xn[pt];
            const int gl = __builtin_bit_cast(int, gn);
            fetch(r0 + 16);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool vr = (u & (bias_period - 1)) == 0;       // bias_period is 1 or 4; r0 is a multiple of 16
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < o.nc) {
                        const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(gl, u * 4 + c));
                        acc[c] = fmaf(x[u >> 2][u & 3], g, acc[c]);
                        bs[c] += vr ? g : 0.f;
                    }
            }
        }
    } else
    for (int64_t r0 = rb; r0 < re; r0 += 4) {
        float x[4];
        if (PM) {
            const f32x4v v = *(const f32x4v *)(X + (r0 >> 2) * (4 * (int64_t)ldx) + 4 * k);      // R is a multiple of 4: the point is whole
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = v[u];
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = r0 + u < re ? X[(r0 + u) * ldx + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = r0 + u;
            if (r >= re) break;
            const bool vr = ((int)r & (bias_period - 1)) == 0;      // bias_period is 1 or 4
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < o.nc) {
                    float g = G[r * ldg + c];
                    acc[c] = fmaf(x[u], g, acc[c]);
                    if (vr) bs[c] += g;
                }
        }
    }
    for (int c = 0; c < o.nc; ++c) {
        if (k < o.kcount) atomicAdd(&o.w[c][(size_t)k * o.wstride], acc[c]);
        if (k == 0 && o.b[c]) atomicAdd(o.b[c], bs[c]);
    }
}
void launch_narrow_dw(const float *X, int ldx, const float *G, int ldg, int64_t R, int nc, float *const *w, int wstride, float *const *b,
                      int bias_period, int kcount, hipStream_t s, int x_point_major)
{
    if (R <= 0) return;
    NarrowGrad o{};
    o.nc = nc; o.wstride = wstride; o.kcount = kcount;
    for (int c = 0; c < nc; ++c) { o.w[c] = w[c]; o.b[c] = b ? b[c] : nullptr; }
    // every workgroup ends with 256 x nc same-address atomics into the one gradient: few, long workgroups (eight per CU) instead of many short ones
    int grid = (int)((R + 255) / 256);
    if (grid > 2048) grid = 2048;
    int64_t rows_per_wg = ((R + grid - 1) / grid + 15) & ~(int64_t)15;
    if (x_point_major) hipLaunchKernelGGL(narrow_dw_kernel<true>, dim3(grid), dim3(kThreads), 0, s, X, ldx, G, ldg, R, rows_per_wg, o, bias_period);
    else hipLaunchKernelGGL(narrow_dw_kernel<false>, dim3(grid), dim3(kThreads), 0, s, X, ldx, G, ldg, R, rows_per_wg, o, bias_period);
}

// The same product with split-fp16 operands (tile_engine.h OpsF16Split; three fp16 MFMAs per multiply-add).  The contraction
// index of v_mfma_f32_32x32x16_f16 is the ROW, eight consecutive rows per lane, so both matrices are staged TRANSPOSED in LDS
// ([column][row], two fp16 planes each): thread c owns column c, loads it one row at a time (a wave reads 256 contiguous
// bytes of a row), splits in registers and stores eight rows per 16-byte LDS write.  32-row chunks, next chunk in flight
// during the MFMAs, whole K x 256 output in accumulators as in dw_tile_kernel.
// XPM / GPM: the operand is a POINT-MAJOR [R, 256] matrix (train_kernels.h MlpForwardArgs.point_major): column tid of the four rows of
// a point is one 16-byte load (R % 4 == 0, chunks start at multiples of 32)
// (KTN <= KT: the LDS layout and the accumulator file are those of KT k-tiles, KTN of them carry data and are multiplied -- so that the
// job-parallel kernel below serves products of different K from one allocation, like dw_tile_rows)
template <int KT, bool XPM = false, bool GPM = false, int KTN = KT>
__device__ __forceinline__ void dw_split_rows(const float *X, int ldx, int K, const float *G, int ldg, int64_t rb, int64_t re,
                                              float *dW, int64_t sk, int64_t sn, int nvalid, float *db, int bias_period, const float *amax_g)
{
    static_assert(!XPM || KT == 8, "a point-major X has 256 columns");
    constexpr int RC = 32, LDT = RC + 8;            // halves per transposed column: 80 B, 16-byte aligned row octets
    const float gs = operand_scale(amax_g);      // G is a gradient matrix: range-scaled while staged; the launcher undoes it
    constexpr int KP = 32 * KT;
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16 *Xh = (_Float16 *)smem, *Xm = Xh + KP * LDT, *Gh = Xm + KP * LDT, *Gm = Gh + kWidth * LDT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int n0 = wave * 64;
    if (rb >= re) return;
    f32x16 acc[KT][2];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[kt][t][q] = 0.f;
    float bsum = 0.f;
    const bool xcol = tid < KP && tid < K;          // thread tid stages column tid of X (if any) and column tid of G
    float xr[RC], gr[RC];
    auto fetch = [&](int64_t c0) {
        if constexpr (XPM || GPM) {
#pragma unroll
            for (int p = 0; p < RC / 4; ++p) {
                const bool in = c0 + 4 * p < re;
                const f32x4v zero = { 0.f, 0.f, 0.f, 0.f };
                if constexpr (XPM) {
                    const f32x4v v = (in && xcol) ? *(const f32x4v *)(X + ((c0 >> 2) + p) * (4 * (int64_t)ldx) + 4 * tid) : zero;
#pragma unroll
                    for (int q = 0; q < 4; ++q) xr[4 * p + q] = v[q];
                }
                if constexpr (GPM) {
                    const f32x4v v = in ? *(const f32x4v *)(G + ((c0 >> 2) + p) * (4 * (int64_t)ldg) + 4 * tid) : zero;
#pragma unroll
                    for (int q = 0; q < 4; ++q) gr[4 * p + q] = v[q];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RC; ++r) {
            const bool in = c0 + r < re;
            if constexpr (!XPM) xr[r] = (in && xcol) ? X[(c0 + r) * (int64_t)ldx + tid] : 0.f;
            if constexpr (!GPM) gr[r] = in ? G[(c0 + r) * (int64_t)ldg + tid] : 0.f;
        }
    };
    auto stage = [&](const float (&v)[RC], _Float16 *ph, _Float16 *pm, float scale) {        // column tid: 32 rows -> 4 + 4 LDS writes of 8 halves
#pragma unroll
        for (int o = 0; o < RC / 8; ++o) {
            h8 vh, vm;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float x = v[8 * o + i] * scale;
                h2 t = __builtin_amdgcn_cvt_pkrtz(x, x);
                vh[i] = (_Float16)t[0];
                vm[i] = (_Float16)__builtin_amdgcn_fmed3f(x - (float)vh[i], -65504.0f, 65504.0f);
            }
            *(h8 *)(ph + tid * LDT + 8 * o) = vh;
            *(h8 *)(pm + tid * LDT + 8 * o) = vm;
        }
    };
    fetch(rb);
    for (int64_t c0 = rb; c0 < re; c0 += RC) {
        __syncthreads();
        if (tid < 32 * KTN) stage(xr, Xh, Xm, 1.0f);
        stage(gr, Gh, Gm, gs);
        if (db) {
#pragma unroll
            for (int r = 0; r < RC; ++r)
                if ((r & (bias_period - 1)) == 0) bsum += gr[r];      // rows past `re` are zero; chunk starts are multiples of 32
        }
        __syncthreads();
        if (c0 + RC < re) fetch(c0 + RC);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int S = 0; S < RC / 16; ++S) {
            const int ro = 16 * S + 8 * h;
            h8 bh[2], bm[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bh[t] = *(const h8 *)(Gh + (n0 + 32 * t + j) * LDT + ro);
                bm[t] = *(const h8 *)(Gm + (n0 + 32 * t + j) * LDT + ro);
            }
#pragma unroll
            for (int kt = 0; kt < KTN; ++kt) {
                const h8 ah = *(const h8 *)(Xh + (32 * kt + j) * LDT + ro), am = *(const h8 *)(Xm + (32 * kt + j) * LDT + ro);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    acc[kt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh[t], acc[kt][t], 0, 0, 0);
                    acc[kt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm[t], acc[kt][t], 0, 0, 0);
                    acc[kt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[t], acc[kt][t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int kt = 0; kt < KTN; ++kt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                int k = 32 * kt + 8 * (q >> 2) + 4 * h + (q & 3), n = n0 + 32 * t + j;
                if (k < K && n < nvalid) atomicAdd(&dW[k * sk + n * sn], acc[kt][t][q]);
            }
    if (db && tid < nvalid) atomicAdd(&db[tid], bsum);
}

template <int KT, bool XPM = false, bool GPM = false>
__global__ __launch_bounds__(kThreads, 1) void dw_split_kernel(const float *X, int ldx, int K, const float *G, int ldg, int64_t R,
                                                               int64_t rows_per_wg, float *dW, int64_t sk, int64_t sn, int nvalid, float *db,
                                                               int bias_period, const float *amax_g)
{
    const int64_t rb = (int64_t)blockIdx.x * rows_per_wg;
    dw_split_rows<KT, XPM, GPM>(X, ldx, K, G, ldg, rb, rb + rows_per_wg < R ? rb + rows_per_wg : R, dW, sk, sn, nvalid, db, bias_period, amax_g);
}

// Job-parallel split-fp16 weight gradients (round 5; train_kernels.h launch_dw_split_jobs): every product of a backward pass in ONE
// launch, the workgroups divided among the products as in dw_jobs_kernel -- one 65 536-atomic epilogue per workgroup and pass instead
// of one per workgroup and product.  A job with amax_g accumulates its range-scaled product into its own dense [K, 256] scratch.
__global__ __launch_bounds__(kThreads, 1) void dw_split_jobs_kernel(const DwJobs jobs)
{
    int jb = 0;
    while (jb + 1 < jobs.n && (int)blockIdx.x >= jobs.job[jb + 1].wg0) ++jb;
    const DwJob &J = jobs.job[jb];
    const int w = (int)blockIdx.x - J.wg0, nw = (jb + 1 < jobs.n ? jobs.job[jb + 1].wg0 : (int)gridDim.x) - J.wg0;
    const int64_t chunks = (jobs.R + 31) / 32, per = (chunks + nw - 1) / nw;
    const int64_t rb = (int64_t)w * per * 32, re_ = rb + per * 32;
    const int64_t re = re_ < jobs.R ? re_ : jobs.R;
    const bool scaled = J.amax_g && J.tmp;
    float *out = scaled ? J.tmp : J.dW;
    const int64_t osk = scaled ? kWidth : J.sk, osn = scaled ? 1 : J.sn;
    const float *am = scaled ? J.amax_g : nullptr;
    if (J.x_point_major) dw_split_rows<8, true, true, 8>(J.X, J.ldx, J.K, J.G, J.ldg, rb, re, out, osk, osn, J.nvalid, J.db, J.bias_period, am);
    else if (J.K <= 64) dw_split_rows<8, false, true, 2>(J.X, J.ldx, J.K, J.G, J.ldg, rb, re, out, osk, osn, J.nvalid, J.db, J.bias_period, am);
    else if (J.K <= 96) dw_split_rows<8, false, true, 3>(J.X, J.ldx, J.K, J.G, J.ldg, rb, re, out, osk, osn, J.nvalid, J.db, J.bias_period, am);
    else dw_split_rows<8, false, true, 8>(J.X, J.ldx, J.K, J.G, J.ldg, rb, re, out, osk, osn, J.nvalid, J.db, J.bias_period, am);
}

// dW += tmp / scale(amax) for every scaled job of the list: block (x, y) = 256 elements x of job y
__global__ void dw_unscale_add_jobs_kernel(const DwJobs jobs)
{
    const DwJob &J = jobs.job[blockIdx.y];
    if (!(J.amax_g && J.tmp)) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= J.K * kWidth) return;
    const int k = i >> 8, n = i & 255;
    if (n < J.nvalid) J.dW[k * J.sk + n * J.sn] += J.tmp[i] * pow2_inverse(operand_scale(J.amax_g));
}

void launch_dw_split_jobs(DwJobs &jobs, float *tmp, int cus, hipStream_t s)
{
    if (jobs.n <= 0 || jobs.R <= 0) return;
    // workgroups in proportion to the products' cost.  Under this policy the staging of G (256 columns split into two fp16 terms per
    // chunk, whatever K) outweighs the matrix work: measured per launch 0.226 / 0.258 / 0.346 ms for K <= 64 / <= 96 / 256
    // (profiles/r05_train_step_kernel_stats_f16_split_fused.csv) -- with the fp32 route's weights (128 : 160 : 320) the narrow products
    // finished 1.6x late and the launch was 4.4 ms SLOWER than one launch per product
    float cost[kMaxDwJobs], total = 0.f;
    for (int i = 0; i < jobs.n; ++i) { const int K = jobs.job[i].K; cost[i] = K <= 64 ? 226.0f : K <= 96 ? 258.0f : 346.0f; total += cost[i]; }
    int grid = cus > jobs.n ? cus : jobs.n, at = 0;
    for (int i = 0; i < jobs.n; ++i) {
        jobs.job[i].wg0 = at;
        int share = (int)(cost[i] / total * grid + 0.5f);
        if (share < 1) share = 1;
        const int left = jobs.n - 1 - i;
        if (at + share > grid - left) share = grid - left - at;
        at += share;
        if (jobs.job[i].amax_g) jobs.job[i].tmp = tmp + (size_t)i * kWidth * kWidth;
    }
    grid = at;
    (void)hipMemsetAsync(tmp, 0, (size_t)jobs.n * kWidth * kWidth * sizeof(float), s);
    constexpr size_t lds = (size_t)2 * (32 * 8 + kWidth) * 40 * sizeof(_Float16);
    static bool once = ((void)hipFuncSetAttribute((const void *)dw_split_jobs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    hipLaunchKernelGGL(dw_split_jobs_kernel, dim3(grid), dim3(kThreads), lds, s, jobs);
    hipLaunchKernelGGL(dw_unscale_add_jobs_kernel, dim3(kWidth * kWidth / 256, jobs.n), dim3(256), 0, s, jobs);
}

// dW[k * sk + n * sn] += T[k, n] / scale(amax): the accumulators of dw_split_kernel stay in the accumulation registers this way
// (scaling them in the kernel moved all 256 of them into VGPRs and spilled)
__global__ void dw_unscale_add_kernel(const float *T, int K, int nvalid, float *dW, int64_t sk, int64_t sn, const float *amax_g)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * kWidth) return;
    const int k = i >> 8, n = i & 255;
    if (n < nvalid) dW[k * sk + n * sn] += T[i] * pow2_inverse(operand_scale(amax_g));
}

template <int KT, bool XPM = false, bool GPM = false>
static void launch_dw_split(const float *X, int ldx, int K, const float *G, int ldg, int64_t R, float *dW, int64_t sk, int64_t sn, int nvalid,
                            float *db, int bias_period, int cus, hipStream_t s, const float *amax_g)
{
    const size_t lds = (size_t)2 * (32 * KT + kWidth) * 40 * sizeof(_Float16);
    static bool once = ((void)hipFuncSetAttribute((const void *)dw_split_kernel<KT, XPM, GPM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)((size_t)2 * (32 * KT + kWidth) * 40 * sizeof(_Float16))), true);
    (void)once;
    int64_t chunks = (R + 31) / 32;
    int grid = (int)(chunks < cus ? chunks : cus);
    int64_t rows_per_wg = ((chunks + grid - 1) / grid) * 32;
    hipLaunchKernelGGL((dw_split_kernel<KT, XPM, GPM>), dim3(grid), dim3(kThreads), lds, s, X, ldx, K, G, ldg, R, rows_per_wg, dW, sk, sn, nvalid, db,
                       bias_period, amax_g);
}

// dW[k * sk + n * sn] += sum_r X[r, k] G[r, n] for k < K <= 256, n < nvalid <= 256 (G has 256 columns, the rest zero);
// (sk, sn) = (256, 1) for LinearGradLayer weights [in, out], (1, in_total) for nn.Linear weights [out, in]
void launch_dw(int split, const float *X, int ldx, int K, const float *G, int ldg, int64_t R, float *dW, int64_t sk, int64_t sn, int nvalid,
               float *db, int bias_period, int cus, hipStream_t s, const float *amax_g, float *scaled_tmp, int x_point_major, int g_point_major)
{
    if (R <= 0 || K <= 0) return;
    if (bias_period != 1 && bias_period != 2 && bias_period != 4) return;       // the kernels test row phases with a mask
    if (split && g_point_major) {       // round 5: the split policy's fused NeDDF route keeps its hidden states and gradients point-major
        const bool scaled = amax_g && scaled_tmp;
        float *out = scaled ? scaled_tmp : dW;
        const int64_t osk = scaled ? kWidth : sk, osn = scaled ? 1 : sn;
        if (scaled) (void)hipMemsetAsync(scaled_tmp, 0, (size_t)K * kWidth * sizeof(float), s);
        const float *am = scaled ? amax_g : nullptr;
        if (x_point_major) launch_dw_split<8, true, true>(X, ldx, K, G, ldg, R, out, osk, osn, nvalid, db, bias_period, cus, s, am);
        else if (K <= 64) launch_dw_split<2, false, true>(X, ldx, K, G, ldg, R, out, osk, osn, nvalid, db, bias_period, cus, s, am);
        else if (K <= 96) launch_dw_split<3, false, true>(X, ldx, K, G, ldg, R, out, osk, osn, nvalid, db, bias_period, cus, s, am);
        else launch_dw_split<8, false, true>(X, ldx, K, G, ldg, R, out, osk, osn, nvalid, db, bias_period, cus, s, am);
        if (scaled) hipLaunchKernelGGL(dw_unscale_add_kernel, dim3((K * kWidth + 255) / 256), dim3(256), 0, s, scaled_tmp, K, nvalid, dW, sk, sn, amax_g);
        return;
    }
    if (split) {
        // with a range-scaled G the product lands in a dense [K, 256] scratch first and is added to dW divided by the scale
        const bool scaled = amax_g && scaled_tmp;
        float *out = scaled ? scaled_tmp : dW;
        const int64_t osk = scaled ? kWidth : sk, osn = scaled ? 1 : sn;
        if (scaled) (void)hipMemsetAsync(scaled_tmp, 0, (size_t)K * kWidth * sizeof(float), s);
        if (K <= 64) launch_dw_split<2>(X, ldx, K, G, ldg, R, out, osk, osn, nvalid, db, bias_period, cus, s, scaled ? amax_g : nullptr);
        else if (K <= 96) launch_dw_split<3>(X, ldx, K, G, ldg, R, out, osk, osn, nvalid, db, bias_period, cus, s, scaled ? amax_g : nullptr);
        else launch_dw_split<8>(X, ldx, K, G, ldg, R, out, osk, osn, nvalid, db, bias_period, cus, s, scaled ? amax_g : nullptr);
        if (scaled) hipLaunchKernelGGL(dw_unscale_add_kernel, dim3((K * kWidth + 255) / 256), dim3(256), 0, s, scaled_tmp, K, nvalid, dW, sk, sn, amax_g);
        return;
    }
    if (g_point_major) {        // (fp32, one launch per product on point-major operands: the wide fused route, whose 46 products the job-parallel launch serves badly)
        if (x_point_major) launch_dw_tile<8, 8, true, true>(X, ldx, K, G, ldg, R, dW, sk, sn, nvalid, db, bias_period, cus, s);
        else if (K <= 64) launch_dw_tile<8, 2, false, true>(X, ldx, K, G, ldg, R, dW, sk, sn, nvalid, db, bias_period, cus, s);
        else if (K <= 96) launch_dw_tile<8, 3, false, true>(X, ldx, K, G, ldg, R, dW, sk, sn, nvalid, db, bias_period, cus, s);
        else launch_dw_tile<8, 8, false, true>(X, ldx, K, G, ldg, R, dW, sk, sn, nvalid, db, bias_period, cus, s);
        return;
    }
    if (K <= 64) launch_dw_tile<2>(X, ldx, K, G, ldg, R, dW, sk, sn, nvalid, db, bias_period, cus, s);
    else if (K <= 96) launch_dw_tile<3>(X, ldx, K, G, ldg, R, dW, sk, sn, nvalid, db, bias_period, cus, s);
    else launch_dw_tile<8>(X, ldx, K, G, ldg, R, dW, sk, sn, nvalid, db, bias_period, cus, s);
}

// ----------------------------------------------------------------------------
// positional encodings as row matrices in the REFERENCE feature order (sin half c = e*3+d, cos half 3E + c), so that
// weight-gradient rows line up with the reference's weight rows: PEs / PEu [4N, ld] (value + Jacobian rows), Ed [N, ldd]
__global__ void pe_rows_kernel(const float *pos, const float *dir, const float *var, int64_t N, EncodeDesc enc, float *PEs, float *PEu,
                               int ld, float *Ed, int ldd)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K3 = 3 * enc.E, K3d = 3 * enc.Ed;
    if (i >= N * (K3 + K3d)) return;
    int64_t n = i / (K3 + K3d);
    int q = (int)(i - n * (K3 + K3d));
    if (q == 0)         // pad columns up to the leading dimension are read by the vectorised GEMM loads: keep them zero
        for (int r = 0; r < 4; ++r)
            for (int c = 2 * K3; c < ld; ++c) { PEs[(n * 4 + r) * ld + c] = 0.f; PEu[(n * 4 + r) * ld + c] = 0.f; }
    if (q < K3) {
        int e = q / 3, d = q - 3 * e;
        float vs, vc, js, jc, us, uc, ujs, ujc;
        pe_pair<true>(e, pos[n * 3 + d], var[n * 3 + d], enc.lowpass[e], vs, vc, js, jc);
        pe_pair<false>(e, pos[n * 3 + d], var[n * 3 + d], enc.lowpass[e], us, uc, ujs, ujc);
        for (int r = 0; r < 4; ++r) {
            float *ps = PEs + (n * 4 + r) * ld, *pu = PEu + (n * 4 + r) * ld;
            bool on = r == 1 + d;
            ps[q] = r == 0 ? vs : (on ? js : 0.f);  ps[K3 + q] = r == 0 ? vc : (on ? jc : 0.f);
            pu[q] = r == 0 ? us : (on ? ujs : 0.f); pu[K3 + q] = r == 0 ? uc : (on ? ujc : 0.f);
        }
    } else {
        q -= K3;
        int e = q / 3, d = q - 3 * e;
        float sn, cs;
        sincosf((float)(1 << e) * dir[n * 3 + d], &sn, &cs);
        Ed[n * ldd + q] = sn;
        Ed[n * ldd + K3d + q] = cs;
    }
}

void launch_pe_rows(const float *pos, const float *dir, const float *var, int64_t N, const EncodeDesc &enc, float *PEs, float *PEu, int ld,
                    float *Ed, int ldd, hipStream_t s)
{
    int64_t t = N * (3 * enc.E + 3 * enc.Ed);
    if (t > 0) hipLaunchKernelGGL(pe_rows_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, pos, dir, var, N, enc, PEs, PEu, ld, Ed, ldd);
}

// ----------------------------------------------------------------------------
// heads + density (neddf.py:220-241) on the raw head rows ZH[4N, ldh] (col 0 = ddf_out, col 1 = aux_out, biases included
// on the value row); writes the per-point record PT[N, kTrainPt] and the small-input part of the colour trunk's input,
// XA[4N, ldxa] = [embed_pos rows | embed_dir | norm_dir.detach()] (neddf.py:243-253; the features follow as a second segment)
__global__ void point_forward_kernel(TrainPointArgs a)
{
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.N) return;
    const float *zh = a.ZH + n * 4 * a.ldh;
    float zd[4], za[4];
    for (int r = 0; r < 4; ++r) { zd[r] = zh[r * a.ldh]; za[r] = zh[r * a.ldh + 1]; }
    float sp, dsp, sg, dsg;
    softplus_grad(zd[0], sp, dsp);
    sigmoid_grad(za[0], sg, dsg);
    float D = sp + a.d_near;
    float dg[3] = { dsp * zd[1], dsp * zd[2], dsp * zd[3] };
    float aux = a.aux_grad_scale * sg;
    float q2 = dg[0] * dg[0] + dg[1] * dg[1] + dg[2] * dg[2];
    float dgn = sqrtf(q2), dDdt = sqrtf(q2 + aux * aux);
    float Dinv = 1.0f / D;
    float u = Dinv * (1 - dDdt);
    float rho = act_val_rt(a.density_activation, u);
    float ninv = 1.0f / (dgn + 1e-7f);
    float *pt = a.PT + n * kTrainPt;
    pt[TP_ZD0] = zd[0]; pt[TP_ZD0 + 1] = zd[1]; pt[TP_ZD0 + 2] = zd[2]; pt[TP_ZD0 + 3] = zd[3];
    pt[TP_ZA0] = za[0]; pt[TP_ZA0 + 1] = za[1]; pt[TP_ZA0 + 2] = za[2]; pt[TP_ZA0 + 3] = za[3];
    pt[TP_D] = D; pt[TP_RHO] = rho; pt[TP_AUX] = aux; pt[TP_U] = u; pt[TP_DGN] = dgn; pt[TP_DDDT] = dDdt;
    for (int i = 0; i < 3; ++i) {
        pt[TP_DG0 + i] = dg[i];
        pt[TP_ND0 + i] = ninv * dg[i];
        pt[TP_AGG0 + i] = a.aux_grad_scale * (dsg * za[1 + i]);
    }
    if (a.distance) a.distance[n] = D;
    if (a.density) a.density[n] = rho;
    if (a.aux_grad) a.aux_grad[n] = aux;
}

// colour-trunk small-input rows XA[4N, ldxa] = [embed_pos rows | embed_dir | norm_dir.detach() | 0] (neddf.py:243-253), one thread per
// element so that reads and writes run along rows; the normal comes from the per-point record written above
__global__ void xa_fill_kernel(TrainPointArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N * 4 * a.ldxa) return;
    const int64_t row = i / a.ldxa;
    const int c = (int)(i - row * a.ldxa), r = (int)(row & 3);
    const int64_t n = row >> 2;
    const int Cpe = 6 * a.enc.E, Cdir = 6 * a.enc.Ed;
    float v = 0.f;
    if (c < Cpe) v = a.PEu[row * a.ldpe + c];
    else if (r == 0 && c < Cpe + Cdir) v = a.Ed[n * a.ldd + c - Cpe];
    else if (r == 0 && c < Cpe + Cdir + 3) v = a.PT[n * kTrainPt + TP_ND0 + c - Cpe - Cdir];
    a.XA[i] = v;
}

// field penalties (neddf.py:260-300) from the colour rows CR[4N, ldc] (cols 0..2) + the per-point record
__global__ void penalty_forward_kernel(TrainPointArgs a)
{
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.N) return;
    const float *pt = a.PT + n * kTrainPt;
    const float *cr = a.CR + n * 4 * a.ldc;
    float c[4][3];
    for (int r = 0; r < 4; ++r)
        for (int k = 0; k < 3; ++k) c[r][k] = cr[r * a.ldc + k];
    float D = pt[TP_D], aux = pt[TP_AUX], Dinv = 1.0f / D;
    float pen[6];
    float d2 = pt[TP_AGG0] * pt[TP_ND0] + pt[TP_AGG0 + 1] * pt[TP_ND0 + 1] + pt[TP_AGG0 + 2] * pt[TP_ND0 + 2];
    float rest = 3 * aux * Dinv;
    pen[0] = (aux * pt[TP_DGN] * D) * ((d2 - rest) * (d2 - rest));
    float t1 = fmaxf(-1.0f + pt[TP_DDDT], 0.f);
    pen[1] = t1 * t1;
    float a1 = fmaxf(-4.6f - pt[TP_ZD0], 0.f), a2 = fmaxf(-a.distance_range_max + pt[TP_ZD0], 0.f);
    pen[2] = (a1 + a2) * (a1 + a2);
    float b1 = fmaxf(-4.6f - pt[TP_ZA0], 0.f), b2 = fmaxf(-4.6f + pt[TP_ZA0], 0.f);
    pen[3] = (b1 + b2) * (b1 + b2);
    pen[4] = 0.f; pen[5] = 0.f;
    for (int k = 0; k < 3; ++k) {
        float c1 = fmaxf(-0.0f - c[0][k], 0.f), c2 = fmaxf(-1.0f + c[0][k], 0.f);
        pen[4] += (c1 + c2) * (c1 + c2);
        float s = c[1][k] * pt[TP_DG0] + c[2][k] * pt[TP_DG0 + 1] + c[3][k] * pt[TP_DG0 + 2];
        pen[5] += s * s;
    }
    float tot = 0.f;
    for (int k = 0; k < 6; ++k) tot += a.penalty_has[k] ? pen[k] * a.penalty_weight[k] : pen[k];
    if (a.penalty) a.penalty[n] = tot;
    if (a.color) { a.color[n * 3] = c[0][0]; a.color[n * 3 + 1] = c[0][1]; a.color[n * 3 + 2] = c[0][2]; }
}

// reverse of the two kernels above: upstream gradients of (distance, density, color, fields_penalty, aux_grad) ->
// gradients of the raw head rows GZH[4N, ldh] (cols 0, 1) and of the colour rows GCR[4N, ldc] (cols 0..2)
__global__ void point_backward_kernel(TrainPointArgs a)
{
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.N) return;
    const float *pt = a.PT + n * kTrainPt;
    const float *cr = a.CR + n * 4 * a.ldc;
    float c[4][3];
    for (int r = 0; r < 4; ++r)
        for (int k = 0; k < 3; ++k) c[r][k] = cr[r * a.ldc + k];
    const float zd0 = pt[TP_ZD0], za0 = pt[TP_ZA0];
    const float zdJ[3] = { pt[TP_ZD0 + 1], pt[TP_ZD0 + 2], pt[TP_ZD0 + 3] }, zaJ[3] = { pt[TP_ZA0 + 1], pt[TP_ZA0 + 2], pt[TP_ZA0 + 3] };
    const float D = pt[TP_D], aux = pt[TP_AUX], dgn = pt[TP_DGN], dDdt = pt[TP_DDDT], u = pt[TP_U];
    const float dg[3] = { pt[TP_DG0], pt[TP_DG0 + 1], pt[TP_DG0 + 2] }, nd[3] = { pt[TP_ND0], pt[TP_ND0 + 1], pt[TP_ND0 + 2] };
    const float agg[3] = { pt[TP_AGG0], pt[TP_AGG0 + 1], pt[TP_AGG0 + 2] };
    const float Dinv = 1.0f / D, s_ = a.aux_grad_scale;
    float sp, dsp, sg, dsg;
    softplus_grad(zd0, sp, dsp);
    sigmoid_grad(za0, sg, dsg);
    const float gP = a.g_penalty ? a.g_penalty[n] : 0.f;
    float gpen[6];
    for (int k = 0; k < 6; ++k) gpen[k] = a.penalty_has[k] ? gP * a.penalty_weight[k] : gP;
    float g_c[4][3] = {};
    float g_zd0 = 0.f, g_za0 = 0.f, g_aux = a.g_aux ? a.g_aux[n] : 0.f, g_dDdt = 0.f, g_D = a.g_distance ? a.g_distance[n] : 0.f;
    float g_agg[3], g_nd[3], g_dg[3] = { 0.f, 0.f, 0.f };
    // pen0 = scale.detach() * (d2 - rest)^2, d2 = sum agg*nd, rest = 3*aux*Dinv.detach()
    float d2 = agg[0] * nd[0] + agg[1] * nd[1] + agg[2] * nd[2];
    float rest = 3 * aux * Dinv;
    float g_d2 = gpen[0] * (aux * dgn * D) * 2 * (d2 - rest);
    g_aux += -g_d2 * 3 * Dinv;
    for (int i = 0; i < 3; ++i) { g_agg[i] = g_d2 * nd[i]; g_nd[i] = g_d2 * agg[i]; }
    // pen1 = relu(dDdt - 1)^2
    g_dDdt += gpen[1] * 2 * fmaxf(dDdt - 1.0f, 0.f);
    // pen2, pen3: range penalties on the raw head outputs
    {
        float a1 = -4.6f - zd0, a2 = zd0 - a.distance_range_max;
        float t = fmaxf(a1, 0.f) + fmaxf(a2, 0.f);
        g_zd0 += gpen[2] * 2 * t * ((a2 > 0.f ? 1.f : 0.f) - (a1 > 0.f ? 1.f : 0.f));
        float b1 = -4.6f - za0, b2 = za0 - 4.6f;
        float t2 = fmaxf(b1, 0.f) + fmaxf(b2, 0.f);
        g_za0 += gpen[3] * 2 * t2 * ((b2 > 0.f ? 1.f : 0.f) - (b1 > 0.f ? 1.f : 0.f));
    }
    // pen4 (colour range), pen5 (colour constraint, distance_grad detached), colour itself
    for (int k = 0; k < 3; ++k) {
        float c1 = -c[0][k], c2 = c[0][k] - 1.0f;
        float t = fmaxf(c1, 0.f) + fmaxf(c2, 0.f);
        g_c[0][k] = (a.g_color ? a.g_color[n * 3 + k] : 0.f) + gpen[4] * 2 * t * ((c2 > 0.f ? 1.f : 0.f) - (c1 > 0.f ? 1.f : 0.f));
        float sk = c[1][k] * dg[0] + c[2][k] * dg[1] + c[3][k] * dg[2];
        for (int i = 0; i < 3; ++i) g_c[1 + i][k] = gpen[5] * 2 * sk * dg[i];
    }
    // density = act(u), u = Dinv (1 - dDdt)
    float g_rho = a.g_density ? a.g_density[n] : 0.f;
    float dact;
    if (a.density_activation == 0) dact = u > 0.f ? 1.f : 0.f;                       // F.relu
    else if (a.density_activation == 1) dact = u > 0.f ? 1.f : 0.01f;                // F.leaky_relu
    else { float yy; act_grad<2>(u, yy, dact); }                                     // tanhExp backward (nn_module/tanh_exp.py:52-54)
    float g_u = g_rho * dact;
    float g_Dinv = g_u * (1 - dDdt);
    g_dDdt += -g_u * Dinv;
    g_D += -g_Dinv * Dinv * Dinv;
    // dDdt = ||(dg, aux)||, nd = dg / (||dg|| + 1e-7)
    if (dDdt > 0.f) {
        for (int i = 0; i < 3; ++i) g_dg[i] += g_dDdt * dg[i] / dDdt;
        g_aux += g_dDdt * aux / dDdt;
    }
    float ninv = 1.0f / (dgn + 1e-7f);
    float g_ninv = 0.f;
    for (int i = 0; i < 3; ++i) { g_dg[i] += g_nd[i] * ninv; g_ninv += g_nd[i] * dg[i]; }
    float g_dgn = -g_ninv * ninv * ninv;
    if (dgn > 0.f) for (int i = 0; i < 3; ++i) g_dg[i] += g_dgn * dg[i] / dgn;
    // SigmoidGradFunction.backward (sigmoid.py:76-81) with dLdy = s*g_aux, dLdG_i = s*g_agg_i
    float g_sg = s_ * g_aux;
    float gG[3] = { s_ * g_agg[0], s_ * g_agg[1], s_ * g_agg[2] };
    float d2s = dsg * (1.0f - 2 * sg);
    g_za0 += g_sg * dsg + d2s * (zaJ[0] * gG[0] + zaJ[1] * gG[1] + zaJ[2] * gG[2]);
    float g_za[3] = { gG[0] * dsg, gG[1] * dsg, gG[2] * dsg };
    // SoftplusGradFunction.backward (softplus.py:81-87) with dLdy = g_D, dLdG_i = g_dg_i
    float d2p = zd0 > 20.0f ? 0.f : (1 - dsp) * dsp;
    g_zd0 += g_D * dsp + d2p * (zdJ[0] * g_dg[0] + zdJ[1] * g_dg[1] + zdJ[2] * g_dg[2]);
    float g_zd[3] = { g_dg[0] * dsp, g_dg[1] * dsp, g_dg[2] * dsp };
    float *gz = a.GZH + n * 4 * a.ldh;
    gz[0] = g_zd0; gz[1] = g_za0;
    for (int i = 0; i < 3; ++i) { gz[(1 + i) * a.ldh] = g_zd[i]; gz[(1 + i) * a.ldh + 1] = g_za[i]; }
    float *gc = a.GCR + n * 4 * a.ldc;
    for (int r = 0; r < 4; ++r)
        for (int k = 0; k < 3; ++k) gc[r * a.ldc + k] = g_c[r][k];
}

void launch_point_forward(const TrainPointArgs &a, hipStream_t s)
{
    if (a.N <= 0) return;
    hipLaunchKernelGGL(point_forward_kernel, dim3((unsigned)((a.N + 127) / 128)), dim3(128), 0, s, a);
    const int64_t t = a.N * 4 * a.ldxa;
    hipLaunchKernelGGL(xa_fill_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, a);
}
void launch_penalty_forward(const TrainPointArgs &a, hipStream_t s)
{
    if (a.N > 0) hipLaunchKernelGGL(penalty_forward_kernel, dim3((unsigned)((a.N + 127) / 128)), dim3(128), 0, s, a);
}
void launch_point_backward(const TrainPointArgs &a, hipStream_t s)
{
    if (a.N > 0) hipLaunchKernelGGL(point_backward_kernel, dim3((unsigned)((a.N + 127) / 128)), dim3(128), 0, s, a);
}

// ----------------------------------------------------------------------------
// NeRF field (nerf.py:139-165), value rows only: encodings as row matrices in the reference feature order with zero pad
// columns, and the density head's activation forward / backward
__global__ void pe_values_kernel(const float *pos, const float *dir, const float *var, int64_t N, EncodeDesc enc, float *PE, int ld, float *Ed,
                                 int ldd)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K3 = 3 * enc.E, K3d = 3 * enc.Ed;
    if (i >= N * (K3 + K3d)) return;
    int64_t n = i / (K3 + K3d);
    int q = (int)(i - n * (K3 + K3d));
    if (q == 0) {
        for (int c = 2 * K3; c < ld; ++c) PE[n * ld + c] = 0.f;
        for (int c = 2 * K3d; c < ldd; ++c) Ed[n * ldd + c] = 0.f;
    }
    if (q < K3) {
        int e = q / 3, d = q - 3 * e;
        float vs, vc, js, jc;
        pe_pair<false>(e, pos[n * 3 + d], var[n * 3 + d], enc.lowpass[e], vs, vc, js, jc);
        PE[n * ld + q] = vs;
        PE[n * ld + K3 + q] = vc;
    } else {
        q -= K3;
        int e = q / 3, d = q - 3 * e;
        float sn, cs;
        sincosf((float)(1 << e) * dir[n * 3 + d], &sn, &cs);
        Ed[n * ldd + q] = sn;
        Ed[n * ldd + K3d + q] = cs;
    }
}
void launch_pe_values(const float *pos, const float *dir, const float *var, int64_t N, const EncodeDesc &enc, float *PE, int ld, float *Ed, int ldd,
                      hipStream_t s)
{
    int64_t t = N * (3 * enc.E + 3 * enc.Ed);
    if (t > 0) hipLaunchKernelGGL(pe_values_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, pos, dir, var, N, enc, PE, ld, Ed, ldd);
}

// density = density_activation(z) (forward, g == NULL) or g_z = g_density * density_activation'(z) (backward)
__global__ void density_head_kernel(int kind, const float *z, int ldz, int64_t N, const float *g, float *out, int ldo)
{
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float x = z[n * ldz];
    if (!g) { out[n * ldo] = act_val_rt(kind, x); return; }
    float dy, d2;
    if (kind == 0) dy = x > 0.f ? 1.f : 0.f;
    else if (kind == 1) dy = x > 0.f ? 1.f : 0.01f;
    else act_grad2<2>(x, dy, d2);
    out[n * ldo] = g[n] * dy;
}
void launch_density_head(int kind, const float *z, int ldz, int64_t N, const float *g, float *out, int ldo, hipStream_t s)
{
    if (N > 0) hipLaunchKernelGGL(density_head_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, kind, z, ldz, N, g, out, ldo);
}

// out[n, 0:3] = in[n, 0:3] with different leading dimensions (colour head output / upstream gradient staging)
__global__ void copy3_kernel(const float *in, int ldi, float *out, int ldo, int64_t N)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 3) return;
    int64_t n = i / 3;
    int c = (int)(i - 3 * n);
    out[n * ldo + c] = in[n * ldi + c];
}
void launch_copy3(const float *in, int ldi, float *out, int ldo, int64_t N, hipStream_t s)
{
    if (N > 0) hipLaunchKernelGGL(copy3_kernel, dim3((unsigned)((N * 3 + 255) / 256)), dim3(256), 0, s, in, ldi, out, ldo, N);
}

// ----------------------------------------------------------------------------
// NeuS (neus.py:118-156).  The sdf trunk runs on (value, Jacobian) row groups like NeDDF's distance trunk; sdf is feature 0
// of the last activated layer and the normal ("gradients", torch.autograd.grad in the reference) its three Jacobian rows.
__device__ __forceinline__ void neus_density(float v10, float sdf, float &rho, float &drho_ds, float &drho_dv10)
{
    const float ex = expf(-v10 * sdf);                     // neus.py:153-156
    const float den = 1.0f + ex, r = 1.0f / (den * den);
    rho = v10 * ex * r;
    const float q = ex * (1.0f - ex) * r / den;            // e (1 - e) / (1 + e)^3
    drho_ds = -v10 * v10 * q;
    drho_dv10 = ex * r - v10 * sdf * q;
}

// XA[N, ldxa] = [pos | embed_dir | gradient | 0] (the small-input segment of the colour trunk, neus.py:146-149), sdf, density
__global__ void neus_head_forward_kernel(NeusPointArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N * a.ldxa) return;
    const int64_t n = i / a.ldxa;
    const int c = (int)(i - n * a.ldxa);
    const float *h = a.Hlast + n * 4 * a.ldh;
    float v = 0.f;
    if (c < 3) v = a.pos[n * 3 + c];
    else if (c < 3 + a.Cdir) v = a.Ed[n * a.ldd + c - 3];
    else if (c < 6 + a.Cdir) v = h[(1 + c - 3 - a.Cdir) * a.ldh];
    a.XA[i] = v;
    if (c == 0) {
        const float sdf = h[0];
        float rho, ds, dv;
        neus_density(10.0f * a.variance[0], sdf, rho, ds, dv);
        if (a.sdf) a.sdf[n] = sdf;
        if (a.density) a.density[n] = rho;
    }
}
void launch_neus_head_forward(const NeusPointArgs &a, hipStream_t s)
{
    int64_t t = a.N * a.ldxa;
    if (t > 0) hipLaunchKernelGGL(neus_head_forward_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, a);
}

// colour = activation(ZC) on the three outputs (neus.py:150-152: the activation follows every colour layer, the last included)
__global__ void neus_color_forward_kernel(NeusPointArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N * 3) return;
    const int64_t n = i / 3;
    const int c = (int)(i - 3 * n);
    a.color[i] = act_val_rt(a.act, a.ZC[n * a.ldc + c]);
}
void launch_neus_color_forward(const NeusPointArgs &a, hipStream_t s)
{
    if (a.N > 0) hipLaunchKernelGGL(neus_color_forward_kernel, dim3((unsigned)((a.N * 3 + 255) / 256)), dim3(256), 0, s, a);
}

// GC[N, ldc] = g_color * activation'(ZC) (cols 0..2, col 3 = 0); g_variance += 10 sum_n g_density drho/d(10 variance)
__global__ __launch_bounds__(256) void neus_color_backward_kernel(NeusPointArgs a)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float gv = 0.f;
    if (n < a.N) {
        float *gc = a.GC + n * a.ldc;
        for (int c = 0; c < 3; ++c) {
            const float x = a.ZC[n * a.ldc + c];
            float dy, d2;
            if (a.act == 0) dy = x > 0.f ? 1.f : 0.f;
            else if (a.act == 1) dy = x > 0.f ? 1.f : 0.01f;
            else act_grad2<2>(x, dy, d2);
            gc[c] = a.g_color ? a.g_color[n * 3 + c] * dy : 0.f;
        }
        for (int c = 3; c < a.ldc; ++c) gc[c] = 0.f;
        if (a.g_density) {
            float rho, ds, dv;
            neus_density(10.0f * a.variance[0], a.Hlast[n * 4 * a.ldh], rho, ds, dv);
            gv = 10.0f * a.g_density[n] * dv;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) gv += __shfl_xor(gv, off, 64);
    if ((threadIdx.x & 63) == 0 && gv != 0.f) atomicAdd(a.g_variance, gv);
}
void launch_neus_color_backward(const NeusPointArgs &a, hipStream_t s)
{
    if (a.N > 0) hipLaunchKernelGGL(neus_color_backward_kernel, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, s, a);
}

// dZ[4N, ldh] of the last sdf layer: the colour trunk's gradient of the features (dF, value rows) and of the normal (DG, feature 0
// of the Jacobian rows), the upstream gradients of sdf and density (feature 0 of the value row), through the last activation.
__global__ void neus_head_backward_kernel(NeusPointArgs a, int act_kind)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = a.ldh / 4;          // 64 or 128 threads per point: whole waves either way
    if (i >= a.N * per) return;
    const int64_t n = i / per;
    const int c4 = (int)(i - n * per);
    const f32x4v zero = { 0.f, 0.f, 0.f, 0.f };
    f32x4v g[4] = { *(const f32x4v *)(a.dF + n * a.ldh + 4 * c4), zero, zero, zero }, z[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) z[r] = *(const f32x4v *)(a.Zlast + (n * 4 + r) * a.ldh + 4 * c4);
    if (c4 == 0) {
        float gs = a.g_sdf ? a.g_sdf[n] : 0.f;
        if (a.g_density) {
            float rho, ds, dv;
            neus_density(10.0f * a.variance[0], a.Hlast[n * 4 * a.ldh], rho, ds, dv);
            gs = fmaf(a.g_density[n], ds, gs);
        }
        g[0][0] += gs;
        for (int k = 0; k < 3; ++k) g[1 + k][0] = a.DG[n * a.lddg + k];
    }
    act_backward_group(act_kind, 4, z, g);
    float lmax = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        *(f32x4v *)(a.dZ + (n * 4 + r) * a.ldh + 4 * c4) = g[r];
#pragma unroll
        for (int u = 0; u < 4; ++u) lmax = fmaxf(lmax, fabsf(g[r][u]));
    }
    if (a.amax_out) publish_amax(a.amax_out, lmax);
}
int neus_backward_act_kind(int act) { return act == kActTanhExp ? kActTanhExpPlain2 : act; }
void launch_neus_head_backward(const NeusPointArgs &a, hipStream_t s)
{
    int64_t t = a.N * (a.ldh / 4);
    if (t > 0) hipLaunchKernelGGL(neus_head_backward_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, a, neus_backward_act_kind(a.act));
}
// ----------------------------------------------------------------------------
// on-device weight packing (the parameters live in torch tensors and change every optimiser step)
__global__ void pack_kernel(const float *src, int64_t sk, int64_t sn, int k_off, int n_off, int kcount, int ncount, int nout, int ks, float *dst)
{
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)ks * 8 * nout) return;
    int r = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    int64_t rest = idx >> 8;
    int S = (int)(rest % ks);
    int wt = (int)(rest / ks);          // wave * NT + t
    int n = wt * 32 + (lane & 31);
    int k = 8 * S + 4 * (lane >> 5) + r;
    dst[idx] = (k < kcount && n < ncount) ? src[(int64_t)(k_off + k) * sk + (int64_t)(n_off + n) * sn] : 0.f;
}
// the same matrix as split-fp16 fragments (OpsF16Split: 16 k-values per super-step, planes h / m of 2^10 w)
__global__ void pack_split_kernel(const float *src, int64_t sk, int64_t sn, int k_off, int n_off, int kcount, int ncount, int nout, int ks,
                                  unsigned short *dst)
{
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)ks * 16 * nout) return;
    int r = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    int64_t frag = idx >> 3;            // ((wave * NT + t) * ks + S) * 64 + lane
    int64_t rest = idx >> 9;
    int S = (int)(rest % ks);
    int wt = (int)(rest / ks);
    int n = wt * 32 + (lane & 31);
    int k = 16 * S + 8 * (lane >> 5) + r;
    float w = (k < kcount && n < ncount) ? src[(int64_t)(k_off + k) * sk + (int64_t)(n_off + n) * sn] : 0.f;
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    const float sv = w * OpsF16Split::kWScale;
    h2 t = __builtin_amdgcn_cvt_pkrtz(sv, sv);
    const _Float16 h = (_Float16)t[0];
    const _Float16 m = (_Float16)(sv - (float)h);
    dst[(frag * 2 + 0) * 8 + r] = __builtin_bit_cast(unsigned short, h);
    dst[(frag * 2 + 1) * 8 + r] = __builtin_bit_cast(unsigned short, m);
}
void launch_pack(int split, const float *src, int64_t sk, int64_t sn, int k_off, int n_off, int kcount, int ncount, int nout, float *dst,
                 hipStream_t s)
{
    if (split) {
        int ks = (kcount + 15) / 16;
        int64_t t = (int64_t)ks * 16 * nout;
        hipLaunchKernelGGL(pack_split_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, src, sk, sn, k_off, n_off, kcount, ncount, nout, ks,
                           (unsigned short *)dst);
        return;
    }
    int ks = (kcount + 7) / 8;
    int64_t t = (int64_t)ks * 8 * nout;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, src, sk, sn, k_off, n_off, kcount, ncount, nout, ks, dst);
}

struct PackJobs { PackJob job[kMaxPackJobs]; };
__global__ void pack_batch_kernel(const PackJobs jobs)
{
    const PackJob &jb = jobs.job[blockIdx.y];
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)jb.ks * 8 * jb.nout) return;
    int r = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    int64_t rest = idx >> 8;
    int S = (int)(rest % jb.ks);
    int wt = (int)(rest / jb.ks);
    int n = wt * 32 + (lane & 31);
    int k = 8 * S + 4 * (lane >> 5) + r;
    jb.dst[idx] = (k < jb.kcount && n < jb.ncount) ? jb.src[(int64_t)(jb.k_off + k) * jb.sk + (int64_t)(jb.n_off + n) * jb.sn] : 0.f;
}
void PackBatch::add(const float *src, int64_t sk, int64_t sn, int k_off, int n_off, int kcount, int ncount, int nout, float *dst, hipStream_t s)
{
    if (n == kMaxPackJobs) flush(s);
    job[n++] = PackJob{ src, sk, sn, k_off, n_off, kcount, ncount, nout, (kcount + 7) / 8, dst };
}
void PackBatch::flush(hipStream_t s)
{
    if (!n) return;
    PackJobs jobs{};
    int64_t tmax = 0;
    for (int i = 0; i < n; ++i) {
        jobs.job[i] = job[i];
        const int64_t t = (int64_t)job[i].ks * 8 * job[i].nout;
        if (t > tmax) tmax = t;
    }
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)((tmax + 255) / 256), (unsigned)n), dim3(256), 0, s, jobs);
    n = 0;
}

// narrow heads (1..4 output columns from a 256-wide input): one wavefront per four rows.  A lane holds four features of each row, so
// the wave has 16 partial sums per lane (4 rows x 4 columns) to add across 64 lanes: a reduce-scatter butterfly (xor 32, 16, 8, 4: each
// step sends half of the lane's values to the partner and keeps the other half) leaves ONE value per lane after 15 exchanges, two plain
// steps finish it -- 17 cross-lane moves instead of 96.  Value u * 4 + c ends in the lanes whose bits 5..2 spell it.
__device__ __forceinline__ float narrow_reduce16(float (&v)[16], int lane)
{
    float a8[8], a4[4], a2[2];
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float keep = b5 ? v[8 + i] : v[i], send = b5 ? v[i] : v[8 + i];
        a8[i] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b4 ? a8[4 + i] : a8[i], send = b4 ? a8[i] : a8[4 + i];
        a4[i] = keep + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b3 ? a4[2 + i] : a4[i], send = b3 ? a4[i] : a4[2 + i];
        a2[i] = keep + __shfl_xor(send, 8, 64);
    }
    float r = (b2 ? a2[1] : a2[0]) + __shfl_xor(b2 ? a2[0] : a2[1], 4, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 1, 64);
    return r;
}

template <bool PM>      // PM: X point-major (ld 256, R a multiple of 4): the wave's four rows are one point
__global__ __launch_bounds__(256) void narrow_forward_kernel(const float *X, int ldx, int64_t R, NarrowW w, int bias_period, float *Y, int ldy, int accumulate)
{
    const int lane = threadIdx.x & 63;
    float wv[4][4];                 // this lane's four input features of every output column
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[c][q] = (c < w.nc && 4 * lane + q < w.kcount) ? w.w[c][(size_t)(4 * lane + q) * w.wstride] : 0.f;
    // the value this lane ends up with: row `mu` of the group, output column `mc`
    const int idx = ((lane >> 5) & 1) << 3 | ((lane >> 4) & 1) << 2 | ((lane >> 3) & 1) << 1 | ((lane >> 2) & 1), mu = idx >> 2, mc = idx & 3;
    const float mbias = (mc < w.nc && w.b[mc]) ? w.b[mc][0] : 0.f;
    const int64_t nw = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const f32x4v zero = { 0.f, 0.f, 0.f, 0.f };
    auto fetch = [&](int64_t base, f32x4v (&x)[4]) {        // PM: x[q] = the four rows of feature 4 lane + q; else x[u] = four features of row u
        if (PM) {
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = base < R ? *(const f32x4v *)(X + (base >> 2) * (4 * (int64_t)ldx) + 4 * (4 * lane + q)) : zero;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = base + u < R ? *(const f32x4v *)(X + (base + u) * ldx + 4 * lane) : zero;
        }
    };
    f32x4v xn[4];
    fetch(w0 * 4, xn);
    for (int64_t base = w0 * 4; base < R; base += nw * 4) {         // four rows per wave per pass, the next four in flight
        f32x4v x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = xn[q];
        fetch(base + nw * 4, xn);
        float v[16];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float sum = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) sum = fmaf(PM ? x[q][u] : x[u][q], wv[c][q], sum);
                v[u * 4 + c] = sum;
            }
        const float r = narrow_reduce16(v, lane);
        if ((lane & 3) == 0 && mc < w.nc && base + mu < R) {        // accumulate: the input is wider than 256 columns and this call adds one more block
            float *y = Y + (base + mu) * ldy + mc;
            const float val = r + ((((int)(base + mu) & (bias_period - 1)) == 0) ? mbias : 0.f);
            *y = accumulate ? *y + val : val;
        }
    }
}
void launch_narrow_forward(const float *X, int ldx, int64_t R, const NarrowW &w, int bias_period, float *Y, int ldy, hipStream_t s, int x_point_major,
                           int accumulate)
{
    if (R <= 0) return;
    int64_t wgs = (R + 15) / 16;
    if (wgs > 8192) wgs = 8192;
    if (x_point_major) hipLaunchKernelGGL(narrow_forward_kernel<true>, dim3((unsigned)wgs), dim3(256), 0, s, X, ldx, R, w, bias_period, Y, ldy, accumulate);
    else hipLaunchKernelGGL(narrow_forward_kernel<false>, dim3((unsigned)wgs), dim3(256), 0, s, X, ldx, R, w, bias_period, Y, ldy, accumulate);
}
__global__ void narrow_backward_kernel(const float *G, int ldg, int64_t R, NarrowW w, float *dX, int ldx, int accumulate)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * kWidth) return;
    int64_t r = i >> 8;
    int k = (int)(i & 255);
    float s = 0.f;
    if (k < w.kcount)
        for (int c = 0; c < w.nc; ++c) s = fmaf(G[r * ldg + c], w.w[c][(size_t)k * w.wstride], s);
    float *d = dX + r * ldx + k;
    *d = accumulate ? *d + s : s;
}
void launch_narrow_backward(const float *G, int ldg, int64_t R, const NarrowW &w, float *dX, int ldx, int accumulate, hipStream_t s)
{
    int64_t t = R * kWidth;
    if (t > 0) hipLaunchKernelGGL(narrow_backward_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, G, ldg, R, w, dX, ldx, accumulate);
}

// The same followed by the backward of the activation that produced the head's input: one thread per (4-row group, 4 columns);
// dZ = act_backward(Zprev, (accumulate ? dH : 0) + sum_c G[., c] w_c).  dH and dZ may alias.
__global__ void narrow_backward_act_kernel(const float *G, int ldg, int64_t R, NarrowW w, const float *dH, int accumulate, int act_kind,
                                           int period, const float *Zprev, float *dZ, int ld, float *amax_out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t groups = (R + 3) >> 2;
    if (i >= groups * (kWidth / 4)) return;     // 64 threads per row group: a wave leaves as one (publish_amax below needs whole waves)
    const int64_t row = (i >> 6) * 4;
    const int c4 = (int)(i & 63);
    f32x4v g[4], z[4];
    const f32x4v zero = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool in = row + r < R;
        g[r] = (accumulate && in) ? *(const f32x4v *)(dH + (row + r) * ld + 4 * c4) : zero;
        z[r] = in ? *(const f32x4v *)(Zprev + (row + r) * ld + 4 * c4) : zero;
        if (in)
            for (int c = 0; c < w.nc; ++c) {
                const float gv = G[(row + r) * ldg + c];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (4 * c4 + u < w.kcount) g[r][u] = fmaf(gv, w.w[c][(size_t)(4 * c4 + u) * w.wstride], g[r][u]);
            }
    }
    act_backward_group(act_kind, period, z, g);
    float lmax = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (row + r < R) {
            *(f32x4v *)(dZ + (row + r) * ld + 4 * c4) = g[r];
#pragma unroll
            for (int u = 0; u < 4; ++u) lmax = fmaxf(lmax, fabsf(g[r][u]));
        }
    if (amax_out) publish_amax(amax_out, lmax);     // 64 threads per row group: whole waves reach this point together
}
void launch_narrow_backward_act(const float *G, int ldg, int64_t R, const NarrowW &w, const float *dH, int accumulate, int act_kind, int period,
                                const float *Zprev, float *dZ, int ld, hipStream_t s, float *amax_out)
{
    int64_t t = ((R + 3) >> 2) * (kWidth / 4);
    if (t > 0) hipLaunchKernelGGL(narrow_backward_act_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, G, ldg, R, w, dH, accumulate,
                                  act_kind, period, Zprev, dZ, ld, amax_out);
}

// ----------------------------------------------------------------------------
// integrate_volume_render backward (base_neural_render.py:148-171), one wavefront per ray.
// w_j = o_j T_j, T_j = prod_{k<j} a_k, a_k = 1 - o_k + 1e-7, o_k = 1 - exp(-rho_k delta_k).
// With q_j = gw_j + gC . c_j + gd t_j the quantity each weight is dotted with, and gT_end = gT + gd max_dist:
//   dL/do_j = T_j q_j - (sum_{m>j} o_m T_m q_m + gT_end T_end) / a_j ;  dL/drho_j = dL/do_j * delta_j (1 - o_j)
// The suffix sum is a reversed wave scan (fp64 like the forward product).
__global__ __launch_bounds__(256) void composite_backward_kernel(const float *dists, const float *dens, const float *col, int64_t n, int S,
                                                                 float max_dist, const float *g_weight, const float *g_depth,
                                                                 const float *g_color, const float *g_trans, float *g_dens, float *g_col)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= n) return;
    const float *d = dists + b * S, *r = dens + b * S, *c = col + b * S * 3;
    const float gd = g_depth ? g_depth[b] : 0.f, gT = g_trans ? g_trans[b] : 0.f;
    const float gc0 = g_color ? g_color[3 * b] : 0.f, gc1 = g_color ? g_color[3 * b + 1] : 0.f, gc2 = g_color ? g_color[3 * b + 2] : 0.f;
    const int nchunk = (S - 1 + 63) / 64;
    // pass 1: T_end
    double carry = 1.0;
    for (int ch = 0; ch < nchunk; ++ch) {
        int j = ch * 64 + lane;
        bool on = j < S - 1;
        float o = on ? 1.0f - expf(-r[j] * (d[j + 1] - d[j])) : 0.f;
        double aj = on ? (double)(1.0f - o + 1e-7f) : 1.0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { double t = __shfl_up(aj, off, 64); if (lane >= off) aj *= t; }
        carry *= __shfl(aj, 63, 64);
    }
    const double Tend = (double)(float)carry;
    double suffix = (double)(gT + gd * max_dist) * Tend;      // sum_{m>j} o_m T_m q_m + gT_end T_end, for j beyond this chunk
    // pass 2: chunks in reverse; recompute T_j by a forward scan inside each chunk from the chunk's entry value
    for (int ch = nchunk - 1; ch >= 0; --ch) {
        // entry transmittance of this chunk = product of all a_k before it (rounded per element like cumprod)
        double entry = 1.0;
        for (int cc = 0; cc < ch; ++cc) {
            int j = cc * 64 + lane;
            float o = 1.0f - expf(-r[j] * (d[j + 1] - d[j]));
            double aj = (double)(1.0f - o + 1e-7f);
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { double t = __shfl_up(aj, off, 64); if (lane >= off) aj *= t; }
            entry *= __shfl(aj, 63, 64);
        }
        int j = ch * 64 + lane;
        bool on = j < S - 1;
        float dj = on ? d[j] : 0.f, delta = on ? d[j + 1] - dj : 0.f;
        float e = on ? expf(-r[j] * delta) : 1.f;
        float o = 1.0f - e;
        float af = 1.0f - o + 1e-7f;
        double aj = on ? (double)af : 1.0, incl = aj;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { double t = __shfl_up(incl, off, 64); if (lane >= off) incl *= t; }
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0;
        float Tj = (float)(excl * entry);
        float q = 0.f;
        if (on) q = (g_weight ? g_weight[b * (S - 1) + j] : 0.f) + gc0 * c[3 * j] + gc1 * c[3 * j + 1] + gc2 * c[3 * j + 2] + gd * dj;
        double term = on ? (double)(o * Tj) * (double)q : 0.0;
        // exclusive suffix sum within the chunk (lanes above this one) + what lies beyond the chunk
        double inc = term;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { double t = __shfl_down(inc, off, 64); if (lane + off < 64) inc += t; }
        double after = inc - term + suffix;
        if (on) {
            float go = Tj * q - (float)(after / (double)af);
            g_dens[b * S + j] = go * delta * e;
            g_col[(b * S + j) * 3 + 0] = o * Tj * gc0;
            g_col[(b * S + j) * 3 + 1] = o * Tj * gc1;
            g_col[(b * S + j) * 3 + 2] = o * Tj * gc2;
        }
        suffix += __shfl(inc, 0, 64);
    }
    if (lane == 0) {        // the last sample only closes the last interval: no gradient (densities[:, :-1], colors[:, :-1])
        g_dens[b * S + S - 1] = 0.f;
        g_col[(b * S + S - 1) * 3 + 0] = 0.f; g_col[(b * S + S - 1) * 3 + 1] = 0.f; g_col[(b * S + S - 1) * 3 + 2] = 0.f;
    }
}

void launch_composite_backward(const float *dists, const float *dens, const float *col, int64_t n, int S, float max_dist,
                               const float *g_weight, const float *g_depth, const float *g_color, const float *g_trans, float *g_dens,
                               float *g_col, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(composite_backward_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, dists, dens, col, n, S, max_dist,
                                  g_weight, g_depth, g_color, g_trans, g_dens, g_col);
}

}  // namespace neddf
