// tile_inputs.h -- what a field kernel's tile starts from: the positional encodings written into the LDS tile (value rows, and the
// Jacobian rows of the forward-mode kernels), the sample point of a ray (cone / point moments), the tile queue, and the phase stamps of
// the diagnostic builds (`make stamp`).  Included by field_kernels.hip only.
#pragma once
#include "kernels.h"
#include "device_math.h"
#include "tile_engine.h"

namespace neddf {

// Integrated positional encoding of the sample position into act columns
// [col0, col0+2*KH) as [sin half | cos half] (sampling.py:55-71 weights,
// with_grad/positional_encoding.py:55-87 values + Jacobian for J_in = I3).
// Region must be pre-zeroed.  GRADSCALE selects embed_pos_scaled (neddf.py:200-204).
template <bool ROWS4, bool GRADSCALE, class Ops = OpsF32, int THREADS = kThreads>
__device__ __forceinline__ void encode_pos(typename Ops::act_t *act, int col0, const EncodeDesc &enc, const float *lp, const float *pos,
                                           const float *var, int64_t p0, int64_t N, int P, int tid, bool use_var = true, int stride = 3)
{
    const int K3 = 3 * enc.E, KH = enc.KH;
    if (!ROWS4 && (P & 63) == 0 && THREADS % P == 0) {
        // value-row tiles of 64 / 128 points: thread -> point p = tid mod P for ALL its items, item -> pair q = tid / P, + THREADS / P, ...
        // A wave's 64 lanes are 64 consecutive points of ONE (frequency, axis) pair: q, e, d, 2^e and the low-pass factor are scalars,
        // the two runtime divisions per item (by 3 E and by 3: half of the loop's instructions) are gone, and the point's six inputs
        // are loaded once per tile instead of once per item.  The same pe_pair per (point, pair): bit-identical values.
        const int p = tid % P;
        const int64_t gp = p0 + p < N ? p0 + p : N - 1;
        float x[3], v[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { x[d] = pos[gp * stride + d]; v[d] = use_var ? var[gp * stride + d] : 0.0f; }
        typename Ops::act_t *row = act + p * Ops::kLd + col0;
        for (int q = __builtin_amdgcn_readfirstlane(tid / P); q < K3; q += THREADS / P) {
            const int e = q / 3, d = q - 3 * e;
            const float xd = d == 0 ? x[0] : d == 1 ? x[1] : x[2], vd = d == 0 ? v[0] : d == 1 ? v[1] : v[2];
            float vs, vc, js, jc;
            pe_pair<GRADSCALE, Ops::kFast>(e, xd, vd, lp[e], vs, vc, js, jc);
            Ops::put(row + q, vs);
            Ops::put(row + q + KH, vc);
        }
        return;
    }
    for (int item = tid; item < P * K3; item += THREADS) {
        int p = item / K3, q = item - p * K3;
        int e = q / 3, d = q - 3 * e;
        int64_t gp = p0 + p < N ? p0 + p : N - 1;
        float vs, vc, js, jc;
        pe_pair<GRADSCALE, Ops::kFast>(e, pos[gp * stride + d], use_var ? var[gp * stride + d] : 0.0f, lp[e], vs, vc, js, jc);
        constexpr int LD = Ops::kLd;
        if (ROWS4) {
            typename Ops::act_t *r0 = act + (4 * p) * LD + col0 + q;
            Ops::put(r0, vs);
            Ops::put(r0 + KH, vc);
            Ops::put(r0 + (1 + d) * LD, js);
            Ops::put(r0 + (1 + d) * LD + KH, jc);
        } else {
            typename Ops::act_t *r0 = act + p * LD + col0 + q;
            Ops::put(r0, vs);
            Ops::put(r0 + KH, vc);
        }
    }
}

// PositionalEncoding of the view direction (positional_encoding.py:51-65), value rows only.
template <bool ROWS4, class Ops = OpsF32, int THREADS = kThreads>
__device__ __forceinline__ void encode_dir(typename Ops::act_t *act, int col0, const EncodeDesc &enc, const float *dir, int64_t p0,
                                           int64_t N, int P, int tid, int stride = 3)
{
    const int K3 = 3 * enc.Ed, KD = enc.KD;
    if (!ROWS4 && (P & 63) == 0 && THREADS % P == 0) {         // (as in encode_pos: one point per thread, one (frequency, axis) pair per wave and step)
        const int p = tid % P;
        const int64_t gp = p0 + p < N ? p0 + p : N - 1;
        float x[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) x[d] = dir[gp * stride + d];
        typename Ops::act_t *row = act + p * Ops::kLd + col0;
        for (int q = __builtin_amdgcn_readfirstlane(tid / P); q < K3; q += THREADS / P) {
            const int e = q / 3, d = q - 3 * e;
            const float xd = d == 0 ? x[0] : d == 1 ? x[1] : x[2];
            float sn, cs;
            if (Ops::kFast) fast_sincos((float)(1 << e) * xd, sn, cs);
            else sincos_cw((float)(1 << e) * xd, sn, cs);
            Ops::put(row + q, sn);
            Ops::put(row + q + KD, cs);
        }
        return;
    }
    for (int item = tid; item < P * K3; item += THREADS) {
        int p = item / K3, q = item - p * K3;
        int e = q / 3, d = q - 3 * e;
        int64_t gp = p0 + p < N ? p0 + p : N - 1;
        float sn, cs;
        if (Ops::kFast) fast_sincos((float)(1 << e) * dir[gp * stride + d], sn, cs);
        else sincos_cw((float)(1 << e) * dir[gp * stride + d], sn, cs);
        typename Ops::act_t *r0 = act + (ROWS4 ? 4 * p : p) * Ops::kLd + col0 + q;
        Ops::put(r0, sn);
        Ops::put(r0 + KD, cs);
    }
}

// ----------------------------------------------------------------------------
// Tile scheduling: tiles are pulled from a global queue (one atomicAdd per tile,
// issued a whole tile ahead of its use) instead of a static stride, so workgroups
// that progress unevenly (two share a CU) do not unbalance the launch tail.
// sched_flags bit 1 = dynamic queue (always on in the shipped library).
// ctl[0] = next tile index, written by thread 0.
// Phase time stamps (-DNEDDF_STAMP, `make stamp`): lane 0 of every wave of the first kStampBlocks workgroups records s_memtime at
// the phase boundaries of its kStampTile-th tile; neddf_capi.hip dumps them after the launch, tools/stamp_timeline.py prints them.
#if defined(NEDDF_STAMP) && defined(NEDDF_STAMP_PAIRS)
// pair mode (tools/stamp_pairs.py): four consecutive tiles of workgroups {0..3, 256..259} -- with 512 workgroups on 256 CUs, b and b + 256
// are the candidates for sharing a CU (slot 0 carries HW_ID | XCC_ID << 32 to check) -- to see how the two workgroups' phases line up
#define NEDDF_STAMP_DECL int sidx_ = 1, stile_ = 0; const unsigned long long swall0_ = wall_clock64(), scyc0_ = __builtin_readcyclecounter(); unsigned long long *sbuf_ = (a.stamps && blockIdx.x < 512 && (blockIdx.x & 255) < 4 && lane == 0) ? a.stamps + ((size_t)((blockIdx.x & 255) + 4 * (blockIdx.x >> 8)) * 8 + wave) * (kStampSlots * kStampPairTiles) : nullptr; \
    if (sbuf_) sbuf_[0] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32)
#define NEDDF_STAMP_TILE() do { ++stile_; } while (0)
#define STAMP() do { if (sbuf_ && stile_ >= kStampTile && stile_ < kStampTile + kStampPairTiles && sidx_ < kStampSlots * kStampPairTiles) sbuf_[sidx_++] = __builtin_readcyclecounter(); } while (0)
#define STAMP_WALL(k) do { } while (0)
#elif defined(NEDDF_STAMP)
#define NEDDF_STAMP_DECL int sidx_ = 0, stile_ = 0; const unsigned long long swall0_ = wall_clock64(), scyc0_ = __builtin_readcyclecounter(); unsigned long long *sbuf_ = (a.stamps && blockIdx.x < kStampBlocks && lane == 0) ? a.stamps + ((size_t)blockIdx.x * 8 + wave) * kStampSlots : nullptr
#define NEDDF_STAMP_TILE() do { sidx_ = 0; ++stile_; } while (0)
#define STAMP() do { if (sbuf_ && stile_ == kStampTile && sidx_ < kStampSlots - 2) sbuf_[sidx_++] = __builtin_readcyclecounter(); } while (0)
// the constant 100 MHz clock (s_memrealtime) at the start (k = 0) and the end (k = 1) of the stamped tile, in the last two slots: cycles of
// the tile over its wall time = the shader clock the part actually held while this kernel ran (tools/stamp_timeline*.py print it)
#define STAMP_WALL(k) do { if (sbuf_ && stile_ == kStampTile) sbuf_[kStampSlots - 2 + (k)] = wall_clock64(); } while (0)
#endif
#if defined(NEDDF_STAMP)
// at the end of the kernel: how many tiles this workgroup took from the queue, and where it ran
#define NEDDF_STAMP_EXIT() do { if (a.stamps && threadIdx.x == 0 && blockIdx.x < kStampWgTail) { unsigned long long *t_ = a.stamps + (size_t)kStampBlocks * 8 * kStampSlots * kStampPairTiles + blockIdx.x; \
    t_[0] = (unsigned long long)(stile_ & 0xfffff) | ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15) << 20) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 24); \
    t_[kStampWgTail] = swall0_; t_[2 * kStampWgTail] = wall_clock64(); t_[3 * kStampWgTail] = __builtin_readcyclecounter() - scyc0_; } } while (0)      /* + its first and last moment on the constant 100 MHz clock, and its shader cycles between them */
#else
#define NEDDF_STAMP_EXIT() do { } while (0)
#define NEDDF_STAMP_DECL
#define NEDDF_STAMP_TILE() do { } while (0)
#define STAMP() do { } while (0)
#define STAMP_WALL(k) do { } while (0)
#endif

// position / variance / direction of point `gpt` of the [B, S] sample grid, straight from the rays (kernels.h RaySrc)
__device__ __forceinline__ void ray_point(const RaySrc &r, int64_t gpt, float (&pos)[3], float (&var)[3], float (&dir)[3])
{
    const int64_t b = gpt / r.S;
    const int j = (int)(gpt - b * r.S);
    float t_mu, t_var, r_var;
    if (r.cone) sample_moments<true>(r.dists + b * r.S, j, r.S, r.r2, t_mu, t_var, r_var);
    else sample_moments<false>(r.dists + b * r.S, j, r.S, r.r2, t_mu, t_var, r_var);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float dd = r.rd[3 * b + k];
        if (r.cone) sample_coord<true>(dd, r.ro[3 * b + k], t_mu, t_var, r_var, pos[k], var[k]);
        else sample_coord<false>(dd, r.ro[3 * b + k], t_mu, t_var, r_var, pos[k], var[k]);
        dir[k] = r.view ? r.view[3 * b + k] : dd;
    }
}

__device__ __forceinline__ int64_t sched_begin(int *sched, int flags, int *ctl, int tid)
{
    if (tid == 0) ctl[0] = (flags & 2) ? atomicAdd(&sched[0], 1) : (int)blockIdx.x;
    __syncthreads();
    return ctl[0];
}

__device__ __forceinline__ int sched_next(int *sched, int flags, int64_t tile)
{
    return (flags & 2) ? atomicAdd(&sched[0], 1) : (int)(tile + gridDim.x);
}

}  // namespace neddf
