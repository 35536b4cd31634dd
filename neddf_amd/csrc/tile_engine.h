// tile_engine.h -- the MFMA tile engine shared by the fused field kernels
// (field_kernels.hip) and the layer-by-layer training kernels (train_kernels.hip).
// See the header of field_kernels.hip for the design.
#pragma once
#include "kernels.h"
#include "device_math.h"

namespace neddf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));


// ----------------------------------------------------------------------------
// dense: acc[mt][t] += act[rows, k0 .. k0+8*ksteps) x Wpacked
template <int MT, int NT>
__device__ __forceinline__ void dense_load(f32x4v (&a)[MT], f32x4v (&b)[NT], const float *act_lane, const f32x4v *wl, int ksteps, int S)
{
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = wl[((size_t)t * ksteps + S) * 64];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = *(const f32x4v *)(act_lane + mt * 32 * kActLd + 8 * S);
}

template <int MT, int NT>
__device__ __forceinline__ void dense_mfma(f32x16 (&acc)[MT][NT], const f32x4v (&a)[MT], const f32x4v (&b)[NT])
{
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][r], b[t][r], acc[mt][t], 0, 0, 0);
}

// Software pipeline, ping-pong operand registers: the operands of super-step
// S+1 are requested (global -> VGPR for B, LDS -> VGPR for A) before the 32
// MFMAs (2048 cycles) of super-step S issue.
// Operands that do not depend on the previous layer's activations (first weight
// fragments + bias) are requested BEFORE the activation epilogue / barriers of the
// previous layer, so their L2 latency is off the critical path.
template <int NT>
struct LayerPre {
    f32x4v b[NT];
    float bias[NT];
};

template <int NT>
__device__ __forceinline__ void layer_prefetch(LayerPre<NT> &p, const float *wp, const float *bias, int ksteps, int wave, int lane)
{
    const f32x4v *wl = (const f32x4v *)wp + (size_t)wave * NT * ksteps * 64 + lane;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        p.b[t] = wl[(size_t)t * ksteps * 64];
        p.bias[t] = bias ? bias[(wave * NT + t) * 32 + (lane & 31)] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int MT, int NT, bool ROWS4>
__device__ __forceinline__ void acc_init_pre(f32x16 (&acc)[MT][NT], const LayerPre<NT> &p)
{
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mt][t][q] = (ROWS4 && (q & 3)) ? 0.f : p.bias[t];
}

template <int MT, int NT>
__device__ __forceinline__ void dense_pre(f32x16 (&acc)[MT][NT], const float *act_lane, const f32x4v *wl, int ksteps,
                                          const LayerPre<NT> &p)
{
    f32x4v a0[MT], b0[NT], a1[MT], b1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b0[t] = p.b[t];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a0[mt] = *(const f32x4v *)(act_lane + mt * 32 * kActLd);
    for (int S = 0; S < ksteps; S += 2) {
        const bool more = S + 1 < ksteps;
        dense_load<MT, NT>(a1, b1, act_lane, wl, ksteps, more ? S + 1 : S);
        __builtin_amdgcn_sched_barrier(0);
        dense_mfma<MT, NT>(acc, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            dense_load<MT, NT>(a0, b0, act_lane, wl, ksteps, S + 2 < ksteps ? S + 2 : S);
            __builtin_amdgcn_sched_barrier(0);
            dense_mfma<MT, NT>(acc, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int MT, int NT>
__device__ __forceinline__ void dense(f32x16 (&acc)[MT][NT], const float *act_lane, const f32x4v *wl, int ksteps)
{
    f32x4v a0[MT], b0[NT], a1[MT], b1[NT];
    dense_load<MT, NT>(a0, b0, act_lane, wl, ksteps, 0);
    for (int S = 0; S < ksteps; S += 2) {
        const bool more = S + 1 < ksteps;
        dense_load<MT, NT>(a1, b1, act_lane, wl, ksteps, more ? S + 1 : S);
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of the MFMA block (hipcc sinks it otherwise)
        dense_mfma<MT, NT>(acc, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            dense_load<MT, NT>(a0, b0, act_lane, wl, ksteps, S + 2 < ksteps ? S + 2 : S);
            __builtin_amdgcn_sched_barrier(0);
            dense_mfma<MT, NT>(acc, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int MT, int NT, bool ROWS4>
__device__ __forceinline__ void acc_init(f32x16 (&acc)[MT][NT], const float *bias, int wave, int lane)
{
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float bv = bias ? bias[(wave * NT + t) * 32 + (lane & 31)] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mt][t][q] = (ROWS4 && (q & 3)) ? 0.f : bv;   // bias only on value rows (linear.py:43-45)
    }
}

template <int MT, int NT>
__device__ __forceinline__ void stash_store(const f32x16 (&acc)[MT][NT], float *slot, int wave, int lane)
{
    f32x4v *dst = (f32x4v *)slot + (size_t)wave * (MT * NT * 4) * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4v v = { acc[mt][t][4 * g], acc[mt][t][4 * g + 1], acc[mt][t][4 * g + 2], acc[mt][t][4 * g + 3] };
                dst[((mt * NT + t) * 4 + g) * 64] = v;
            }
}

template <int MT, int NT>
__device__ __forceinline__ void stash_add(f32x16 (&acc)[MT][NT], const float *slot, int wave, int lane)
{
    const f32x4v *src = (const f32x4v *)slot + (size_t)wave * (MT * NT * 4) * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4v v = src[((mt * NT + t) * 4 + g) * 64];
                acc[mt][t][4 * g] += v[0]; acc[mt][t][4 * g + 1] += v[1];
                acc[mt][t][4 * g + 2] += v[2]; acc[mt][t][4 * g + 3] += v[3];
            }
}

// activation epilogue: registers -> LDS activations (columns [0, NT*128))
template <int MT, int NT, bool ROWS4, int KIND>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[MT][NT], float *act, int wave, int lane)
{
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float *o = act + (mt * 32 + 4 * h) * kActLd + (wave * NT + t) * 32 + j;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (ROWS4) {
                    float y, dy;
                    act_grad<KIND>(acc[mt][t][4 * g], y, dy);
                    o[(8 * g + 0) * kActLd] = y;
                    o[(8 * g + 1) * kActLd] = dy * acc[mt][t][4 * g + 1];
                    o[(8 * g + 2) * kActLd] = dy * acc[mt][t][4 * g + 2];
                    o[(8 * g + 3) * kActLd] = dy * acc[mt][t][4 * g + 3];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[(8 * g + r) * kActLd] = act_val<KIND>(acc[mt][t][4 * g + r]);
                }
            }
        }
}

template <int MT, int NT, bool ROWS4>
__device__ __forceinline__ void epilogue_rt(const f32x16 (&acc)[MT][NT], float *act, int kind, int wave, int lane)
{
    if (kind == 0) epilogue<MT, NT, ROWS4, 0>(acc, act, wave, lane);
    else if (kind == 1) epilogue<MT, NT, ROWS4, 1>(acc, act, wave, lane);
    else epilogue<MT, NT, ROWS4, 2>(acc, act, wave, lane);
}

// ----------------------------------------------------------------------------
// input encodings
__device__ __forceinline__ void zero_cols(float *act, int rows, int ncols, int tid)
{
    for (int i = tid; i < rows * ncols; i += kThreads) {
        int r = i / ncols, c = i - r * ncols;
        act[r * kActLd + c] = 0.f;
    }
}

}  // namespace neddf
