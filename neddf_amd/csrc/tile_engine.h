// tile_engine.h -- the MFMA tile engine shared by the fused field kernels
// (field_kernels.hip) and the layer-by-layer training kernels (train_kernels.hip).
// See the header of field_kernels.hip for the design.
//
// The engine is written once over an operand policy `Ops`:
//   OpsF32   activations fp32 in LDS, weights fp32, v_mfma_f32_32x32x2_f32 (exact fp32; the parity path)
//   OpsBF16  activations bf16 in LDS, weights bf16, v_mfma_f32_32x32x16_bf16 with fp32 accumulation
//            (BASELINE.json configs[4]: "bf16 MLP weights on MFMA"; 16x the fp32 matrix rate)
//   OpsF16Split  fp32 data on the fp16 matrix instructions: every operand split into two fp16 terms (21-22 bits), a.w
//            accumulated from the three products above 2^-22 in fp32.  3 MFMAs at 1/16 of the fp32 MFMA's cost each: 5.3x
//            the fp32 matrix rate, errors at the level of the fp32 MFMA path's own
// Both read one 16-byte fragment per lane per super-step for A (ds_read_b128) and for B
// (global_load_dwordx4); a super-step covers Ops::kStep values of k, lane half h = lane>>5
// holding k = kStep*S + (kStep/2)*h ... +kStep/2-1 -- the same mapping for A and B, which is all
// the contraction needs.
#pragma once
#include "kernels.h"
#include "device_math.h"

namespace neddf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

// `make exactact` (-DNEDDF_EXACT_ACT -> libneddf_hip_exactact.so) builds the fp32 and split-fp16 policies with the reference's
// branch-exact tanhExp in the fused kernels too, so that the reduced-cost form stays A/B-testable on any network
// (tests/test_gpu_parity.py::test_fast_activation_against_the_branch_exact_build)
// (-DNEDDF_ACT_F32=<mode> / -DNEDDF_ACT_SPLIT=<mode> build any other pairing for A/B timing: tools/act_variants.sh)
#ifdef NEDDF_EXACT_ACT
#define NEDDF_ACT_F32 0
#define NEDDF_ACT_SPLIT 0
#endif
#ifndef NEDDF_ACT_F32
#define NEDDF_ACT_F32 2
#endif
#ifndef NEDDF_ACT_SPLIT
#define NEDDF_ACT_SPLIT 2
#endif

// Every policy is written over the engine width WID (hidden width padded to a multiple of 128: four waves x 32-column MFMA
// tiles); the 256-wide aliases below are what the shipped configurations and the training kernels use.
template <int WID>
struct OpsF32T {
    static constexpr int kWid = WID;
    typedef float act_t;
    typedef f32x4v frag;
    typedef frag afrag;                      // activation (A) and weight (B) fragments have the same type
    typedef frag bfrag;
    static constexpr int kLd = WID + 4;      // LDS row stride in elements (260 floats at width 256): stride = 4 dwords mod 64 -> conflict-free ds_read_b128
    static constexpr int kStep = 8;          // k values per super-step
    static constexpr int kSub = 4;           // MFMA instructions per fragment
    static constexpr bool kFast = false;     // reference-exact elementwise math
    // tanhExp in the fused kernels: the middle form (device_math.h tanhexp_grad_mid, 17 instructions: closed form where it keeps
    // relative accuracy, a fitted odd polynomial below e^x = 0.2).  Rounds 3-4 shipped the closed form alone (mode 1, 11 instructions,
    // +1.4 % rays/s: fp32 MFMA and VALU work do not overlap, so the activation is wall time); on the negative-bias stress network its
    // density error was 2.0x the reference's own fp32 error against fp64 -- inside every gate, but with no margin to speak of.  The
    // middle form is as close to fp64 as the branch-exact build to the printed digits (profiles/r04_act_modes.txt), so the parity
    // path pays the 1.4 % and the tests hold density to 1.5x (was 2.5x).  The stand-alone ops and the training kernels keep the
    // branch-exact form; -DNEDDF_ACT_F32=1 builds the closed form for A/B.
    static constexpr int kActMode = NEDDF_ACT_F32;
    static constexpr int kPlanes = 1, kPlane = 0;
    static constexpr bool kPackedRows = false;
    static constexpr float kWScale = 1.0f;   // packed weights = kWScale * w
    static constexpr bool kStash16 = false;  // y' of the reverse-mode kernel travels in fp32
    static constexpr bool kDeepPrefetch = false;
    static constexpr bool kEncInLds = false; // no LDS to spare next to two 67 KB tiles
    static constexpr bool kTransposed = false, kStashF16 = false;       // (see OpsBF16RT)
    static __device__ __forceinline__ frag load_a(const act_t *p) { return *(const frag *)p; }
    static __device__ __forceinline__ void zero(act_t *p) { *p = 0.f; }
    static __device__ __forceinline__ void put4(act_t *p, const f32x4v &v) { *(f32x4v *)p = v; }     // 4 consecutive columns
    static __device__ __forceinline__ void put(act_t *p, float v) { *p = v; }
    static __device__ __forceinline__ void put2(act_t *pa, act_t *pb, float va, float vb) { *pa = va; *pb = vb; }   // two values, two places
    static __device__ __forceinline__ float get(const act_t *p) { return *p; }
    static __device__ __forceinline__ void load4(const act_t *p, float (&x)[4])
    {
        f32x4v v = *(const f32x4v *)p;
        x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3];
    }
    static __device__ __forceinline__ f32x16 mfma(const frag &a, const frag &b, const f32x16 &c, int r)
    {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[r], c, 0, 0, 0);
    }
};

typedef OpsF32T<256> OpsF32;

template <int WID>
struct OpsBF16T {
    static constexpr int kWid = WID;
    typedef unsigned short act_t;            // bf16 bit pattern
    typedef bf16x8 frag;
    typedef frag afrag;
    typedef frag bfrag;
    static constexpr int kLd = WID + 8;      // 528 B rows at width 256: same bank pattern as the fp32 tile (row stride = 4 dwords mod 64)
    static constexpr int kStep = 16;
    static constexpr int kSub = 1;
    static constexpr bool kFast = true;      // reduced-cost elementwise math (device_math.h), invisible after bf16 rounding
    static constexpr int kActMode = 1;
    static constexpr int kPlanes = 1, kPlane = 0;
    static constexpr float kWScale = 1.0f;
    static constexpr bool kStash16 = true;   // ... as bf16 here
    static constexpr bool kDeepPrefetch = true;      // weight fragments two super-steps ahead (dense_pipeline3)
    static constexpr bool kEncInLds = true;          // the reverse-mode kernel keeps the encoding in a second LDS tile for its skip layers (a bf16 tile is 33 KB)
    static constexpr bool kTransposed = false, kStashF16 = false;
    static __device__ __forceinline__ frag load_a(const act_t *p) { return *(const frag *)p; }
    static __device__ __forceinline__ void zero(act_t *p) { *p = 0; }
    static __device__ __forceinline__ unsigned short cvt(float v)       // round to nearest even (v_cvt_pk_bf16_f32)
    {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 in = { v, v };
        bf16x2 o = __builtin_convertvector(in, bf16x2);
        return __builtin_bit_cast(unsigned int, o) & 0xffffu;
    }
    static __device__ __forceinline__ void put(act_t *p, float v) { *p = cvt(v); }
    static __device__ __forceinline__ void put2(act_t *pa, act_t *pb, float va, float vb)      // one packed conversion for both
    {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const unsigned int w = __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2){ va, vb }, bf16x2));
        *pa = (unsigned short)w; *pb = (unsigned short)(w >> 16);
    }
    // four values of one column to four consecutive rows: two packed conversions, the upper halves stored with d16_hi
    static constexpr bool kPackedRows = true;
    static __device__ __forceinline__ void put_rows4(act_t *p, int ld, float v0, float v1, float v2, float v3)
    {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const unsigned int a = __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2){ v0, v1 }, bf16x2));
        const unsigned int b = __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2){ v2, v3 }, bf16x2));
        p[0] = (unsigned short)a; p[ld] = (unsigned short)(a >> 16);
        p[2 * ld] = (unsigned short)b; p[3 * ld] = (unsigned short)(b >> 16);
    }
    static __device__ __forceinline__ float get(const act_t *p) { return __builtin_bit_cast(float, (unsigned int)*p << 16); }
    static __device__ __forceinline__ void load4(const act_t *p, float (&x)[4])
    {
        u16x4 v = *(const u16x4 *)p;
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = __builtin_bit_cast(float, (unsigned int)v[i] << 16);
    }
    static __device__ __forceinline__ f32x16 mfma(const frag &a, const frag &b, const f32x16 &c, int)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
typedef OpsBF16T<256> OpsBF16;

// The bf16 policy of the REVERSE-MODE distance kernel (round 6): the products are taken TRANSPOSED.  The 32x32x16 MFMA is symmetric
// in its operands' lane layout (lane (i, h) holds 8 consecutive k of row i of A / of column i of B), so the same two loads -- the
// packed weight fragment, the activation fragment from LDS -- can enter in either order; with the weights as A the accumulator has
// FEATURES in its rows and POINTS in its columns: lane (j, h) owns point j and, per register group g, the four CONSECUTIVE features
// 8 g + 4 h .. + 3.  What that buys in the epilogues, which are this policy's bound (vector ALU 51 %, matrix pipe 30 %:
// docs/lab_notebook.md R5.9): four activations leave as ONE ds_write_b64 (two packed conversions) instead of four ds_write_b16 --
// 6 cycles of the LDS store path instead of 16 --, and the bias / the reverse seed become one ds_read_b128 of four consecutive
// features from an LDS-resident vector straight into the accumulator registers instead of 64 v_mov per layer.  y' travels as fp16
// pairs (range [-0.14, 1.07]: 11 bits instead of bf16's 8) so that the reverse pass multiplies an accumulator by its half in ONE
// mixed-precision instruction (v_fma_mix_f32) instead of unpack + multiply.
template <int WID>
struct OpsBF16RT : OpsBF16T<WID> {
    typedef OpsBF16T<WID> Base;
    typedef typename Base::act_t act_t;
    typedef typename Base::frag frag;
    static constexpr bool kTransposed = true, kStashF16 = true;
    static __device__ __forceinline__ f32x16 mfma(const frag &a, const frag &b, const f32x16 &c, int)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0);       // weights (b) as the A operand
    }
    static __device__ __forceinline__ void put4(act_t *p, float v0, float v1, float v2, float v3)      // four consecutive columns, 8-byte aligned
    {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        u32x2 w = { __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2){ v0, v1 }, bf16x2)),
                    __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2){ v2, v3 }, bf16x2)) };
        *(u32x2 *)p = w;
    }
};

// Split-fp16 operands: fp16 carries 11 mantissa bits, so TWO terms per operand (a = h + m, h rounded toward zero -- which
// also saturates instead of overflowing --, m the remainder rounded to nearest) hold 21-22 bits, and a.w needs only the
// three products above 2^-22 (m.h, h.m, h.h).  Weights are scaled by 2^10 on the host so that their second term stays in
// fp16's normal range; the accumulators are scaled back (exactly) on their way into the activation.  3 MFMAs at 1/16 of
// the fp32 MFMA's cost: 5.3x the fp32 matrix rate, errors at the fp32 MFMA path's level.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct hfrag2 {
    f16x8 h, m;
};

template <int WID>
struct OpsF16SplitT {
    static constexpr int kWid = WID;
    typedef unsigned short act_t;
    typedef hfrag2 afrag;
    typedef hfrag2 bfrag;
    static constexpr int kPlanes = 2, kPlane = WID + 8;
    // row = [first terms: WID + 8 | remainders: WID + 8 | 8 of padding]: 2 (WID + 8) halves alone are a stride of 8 dwords mod 64 and
    // the sixteen rows of a ds_read_b128 wave quarter hit each bank twice (34 % of this policy's LDS cycles were conflicts,
    // profiles/r04_pmc_c2_slab.csv); with the padding the stride is 12 dwords mod 64 and the sixteen 16-byte fragments tile the 64 banks
    static constexpr int kLd = 2 * (WID + 8) + 8;
    static constexpr int kStep = 16;
    static constexpr int kSub = 3;
    static constexpr bool kFast = false;
    static constexpr int kActMode = NEDDF_ACT_SPLIT;      // the middle form: this policy is sold on the fp32 gates, and the closed form alone
                                                          // triples its density error where most pre-activations are very negative (profiles/r04_act_modes.txt)
    static constexpr bool kPackedRows = false;
    static constexpr float kWScale = 1024.0f;            // weights are packed as 2^10 w
    static constexpr bool kStash16 = false;
    static constexpr bool kDeepPrefetch = false;         // measured: the colour kernel spills with a third operand set (dense_pipeline3)
    static constexpr bool kEncInLds = false;
    static constexpr bool kTransposed = false, kStashF16 = false;
    static __device__ __forceinline__ float f(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
    static __device__ __forceinline__ void put(act_t *p, float v)
    {
        typedef __fp16 h2 __attribute__((ext_vector_type(2)));
        h2 t = __builtin_amdgcn_cvt_pkrtz(v, v);            // toward zero: |h| <= |v| and +-65504 at most, never inf
        const _Float16 h = (_Float16)t[0];
        const _Float16 m = (_Float16)__builtin_amdgcn_fmed3f(v - (float)h, -65504.0f, 65504.0f);
        p[0] = __builtin_bit_cast(unsigned short, h);
        p[kPlane] = __builtin_bit_cast(unsigned short, m);
    }
    // (a packed variant -- both first terms from one v_cvt_pkrtz, both remainders from a second one, rounded toward zero -- was
    // measured: 1 % faster, 20 dB of PSNR lost to the truncated remainders; not kept)
    static __device__ __forceinline__ void put2(act_t *pa, act_t *pb, float va, float vb) { put(pa, va); put(pb, vb); }
    static __device__ __forceinline__ void zero(act_t *p) { p[0] = 0; p[kPlane] = 0; }
    static __device__ __forceinline__ void put4(act_t *p, const f32x4v &v)                               // 4 consecutive columns
    {
        typedef __fp16 h2 __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 h, m;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h2 t = __builtin_amdgcn_cvt_pkrtz(v[i], v[i]);
            h[i] = (_Float16)t[0];
            m[i] = (_Float16)__builtin_amdgcn_fmed3f(v[i] - (float)h[i], -65504.0f, 65504.0f);
        }
        *(f16x4 *)p = h;
        *(f16x4 *)(p + kPlane) = m;
    }
    static __device__ __forceinline__ float get(const act_t *p) { return f(p[kPlane]) + f(p[0]); }
    static __device__ __forceinline__ void load4(const act_t *p, float (&x)[4])
    {
        u16x4 h = *(const u16x4 *)p, m = *(const u16x4 *)(p + kPlane);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = f(m[i]) + f(h[i]);
    }
    static __device__ __forceinline__ afrag load_a(const act_t *p)
    {
        afrag a;
        a.h = *(const f16x8 *)p; a.m = *(const f16x8 *)(p + kPlane);
        return a;
    }
    static __device__ __forceinline__ f32x16 mfma(const afrag &a, const bfrag &b, const f32x16 &c, int r)
    {
        switch (r) {
        case 0: return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.m, b.h, c, 0, 0, 0);
        case 1: return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.m, c, 0, 0, 0);
        default: return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, c, 0, 0, 0);
        }
    }
};
typedef OpsF16SplitT<256> OpsF16Split;

// the same policy over a narrow side tile (the encoding kept in LDS for the skip layers): 64 columns + 8 of padding per row, i.e. a row
// stride of 36 dwords -- 16 rows x 16-byte fragments tile the 64 banks exactly
constexpr int kEncLd = 72;
template <class Ops>
struct EncView : Ops {
    static constexpr int kLd = kEncLd;
};

// per-lane base of the A fragments of M-tile 0
template <class Ops>
__device__ __forceinline__ const typename Ops::act_t *act_lane_ptr(const typename Ops::act_t *act, int lane)
{
    return act + (lane & 31) * Ops::kLd + (Ops::kStep / 2) * (lane >> 5);
}

// ----------------------------------------------------------------------------
// dense: acc[mt][t] += act[rows, k0 .. k0+kStep*ksteps) x Wpacked
//
// Operand addresses cost no vector instructions inside the product loop (on gfx950 a VALU instruction between two MFMAs of
// a wave costs ~16 cycles of matrix time, profiles/r02_shadow_ubench.txt): the weight pointer of a wave is wave-uniform
// apart from the lane's slot, so the fragments are fetched as buffer loads -- a scalar resource (the wave's base), the
// constant lane offset in a VGPR and the super-step's offset in an SGPR (`buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen`)
// -- instead of one 64-bit v_lshl_add_u64 per global load.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct WeightStream {
    __amdgpu_buffer_rsrc_t rsrc;    // base = the wave's first fragment, raw buffer without range clamp
    unsigned lane_off;              // lane * sizeof(fragment)
};

template <class F>
__device__ __forceinline__ WeightStream weight_stream(const F *wl)
{
    // wl = (wave-uniform fragment pointer) + lane: recover the uniform part and hand it to the scalar unit
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const unsigned long long u = (unsigned long long)(wl - lane);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    void *base = (void *)(((unsigned long long)hi << 32) | lo);
    return { __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)0xffffffffu, 0x00020000), lane * (unsigned)sizeof(F) };
}

template <class F>
__device__ __forceinline__ F stream_load(const WeightStream &w, unsigned frag_index)
{
    static_assert(sizeof(F) % 16 == 0, "fragments are fetched 16 bytes at a time");
    struct Raw { u32x4 v[sizeof(F) / 16]; } raw;
#pragma unroll
    for (unsigned i = 0; i < sizeof(F) / 16; ++i)
        raw.v[i] = __builtin_amdgcn_raw_buffer_load_b128(w.rsrc, w.lane_off + 16 * i, frag_index * (64 * (unsigned)sizeof(F)), 0);
    return __builtin_bit_cast(F, raw);
}

// B fragments of super-step S (scalar index, clamped by the caller) and A fragments at `ap` (the lane's pointer for that
// super-step: the caller advances it, so the LDS reads carry immediate offsets and cost no address arithmetic either)
template <int MT, int NT, class Ops = OpsF32>
__device__ __forceinline__ void dense_load(typename Ops::afrag (&a)[MT], typename Ops::bfrag (&b)[NT], const typename Ops::act_t *ap,
                                           const WeightStream &w, int ksteps, int S)
{
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = stream_load<typename Ops::bfrag>(w, (unsigned)(t * ksteps + S));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = Ops::load_a(ap + mt * 32 * Ops::kLd);
}

template <int MT, int NT, class Ops = OpsF32>
__device__ __forceinline__ void dense_mfma(f32x16 (&acc)[MT][NT], const typename Ops::afrag (&a)[MT], const typename Ops::bfrag (&b)[NT])
{
#pragma unroll
    for (int r = 0; r < Ops::kSub; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[mt][t] = Ops::mfma(a[mt], b[t], acc[mt][t], r);
}

// Software pipeline, ping-pong operand registers: the operands of super-step
// S+1 are requested (global -> VGPR for B, LDS -> VGPR for A) before the MFMAs
// of super-step S issue.
// Operands that do not depend on the previous layer's activations (first weight
// fragments + bias) are requested BEFORE the activation epilogue / barriers of the
// previous layer, so their L2 latency is off the critical path.
template <int NT, class Ops = OpsF32>
struct LayerPre {
    typename Ops::bfrag b[NT];
    float bias[NT];
};

template <int NT, class Ops = OpsF32>
__device__ __forceinline__ void layer_prefetch(LayerPre<NT, Ops> &p, const void *wp, const float *bias, int ksteps, int wave, int lane)
{
    const typename Ops::bfrag *wl = (const typename Ops::bfrag *)wp + (size_t)wave * NT * ksteps * 64 + lane;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        p.b[t] = wl[(size_t)t * ksteps * 64];
        p.bias[t] = bias ? Ops::kWScale * bias[(wave * NT + t) * 32 + (lane & 31)] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int MT, int NT, bool ROWS4, class Ops = OpsF32>
__device__ __forceinline__ void acc_init_pre(f32x16 (&acc)[MT][NT], const LayerPre<NT, Ops> &p)
{
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mt][t][q] = (ROWS4 && (q & 3)) ? 0.f : p.bias[t];
}

// bf16 policy (Ops::kDeepPrefetch): the weight fragments travel TWO super-steps ahead (ring of three B sets), the LDS fragments one.
// A bf16 super-step is 4 MFMAs = 128 matrix cycles, a fraction of one L2 round trip (300-500 cycles under load): with one super-step
// of distance a wave had 2 KB of weights in flight where latency x the CU's 64 B/clk asks for ~4 KB.  Measured in one call
// (tools/r04_prefetch.sh): bf16 distance kernel 24.69 -> 23.96 ms per launch (C2 2.024 -> 2.072 M rays/s, C5 1.056 -> 1.080 M); split fp16
// (384 matrix cycles per super-step) distance kernel 50.06 -> 49.88 ms but its colour kernel 13.40 -> 13.86 ms (the third set spills
// there): not taken for that policy.
// timing probes (variant builds only, results invalid): the product loop without its weight-fragment loads / without its LDS loads
#ifndef NEDDF_PROBE_NOB
#define NEDDF_PROBE_NOB 0
#endif
#ifndef NEDDF_PROBE_NOY
#define NEDDF_PROBE_NOY 0          // the reverse-mode kernel's y' round trip (field_kernels.hip: bit 0 stores, bit 1 loads, bit 2 one slot for all layers)
#endif
#ifndef NEDDF_PROBE_NOA
#define NEDDF_PROBE_NOA 0
#endif
template <int MT, int NT, class Ops>
__device__ __forceinline__ void dense_pipeline3(f32x16 (&acc)[MT][NT], typename Ops::afrag (&a0)[MT], typename Ops::bfrag (&b0)[NT],
                                                const typename Ops::act_t *act_lane, const WeightStream &wl, int ksteps)
{

    typename Ops::afrag a[2][MT];
    typename Ops::bfrag b[3][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[0][mt] = a0[mt];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        b[0][t] = b0[t];
        b[1][t] = stream_load<typename Ops::bfrag>(wl, (unsigned)(t * ksteps + (ksteps > 1 ? 1 : 0)));
    }
    const typename Ops::act_t *ap = act_lane;
    for (int S = 0; S < ksteps; S += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            if (S + u >= ksteps) break;                 // wave-uniform
            const int s2 = S + u + 2 < ksteps ? S + u + 2 : ksteps - 1;     // (clamped: the index could leave the allocation)
#if !NEDDF_PROBE_NOB
#pragma unroll
            for (int t = 0; t < NT; ++t) b[(u + 2) % 3][t] = stream_load<typename Ops::bfrag>(wl, (unsigned)(t * ksteps + s2));
#else
            (void)s2;
#pragma unroll
            for (int t = 0; t < NT; ++t) b[(u + 2) % 3][t] = b[u % 3][t];
#endif
#if !NEDDF_PROBE_NOA
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[(u + 1) & 1][mt] = Ops::load_a(ap + (u + 1) * Ops::kStep + mt * 32 * Ops::kLd);
#else
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[(u + 1) & 1][mt] = a[u & 1][mt];
#endif
            __builtin_amdgcn_sched_barrier(0);
            dense_mfma<MT, NT, Ops>(acc, a[u & 1], b[u % 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
        ap += 6 * Ops::kStep;
    }
}

template <int MT, int NT, class Ops = OpsF32>
__device__ __forceinline__ void dense_pipeline(f32x16 (&acc)[MT][NT], typename Ops::afrag (&a0)[MT], typename Ops::bfrag (&b0)[NT],
                                               const typename Ops::act_t *act_lane, const WeightStream &wl, int ksteps)
{
    if constexpr (Ops::kDeepPrefetch) {
        dense_pipeline3<MT, NT, Ops>(acc, a0, b0, act_lane, wl, ksteps);
        return;
    }
    typename Ops::afrag a1[MT];
    typename Ops::bfrag b1[NT];
    // A fragments past the last super-step are fetched like the others and never used (the reads stay inside the workgroup's
    // LDS: at most two super-steps beyond a row's columns); the weight index is clamped instead, it could leave the allocation
    const typename Ops::act_t *ap = act_lane;
    for (int S = 0; S < ksteps; S += 2) {
        const bool more = S + 1 < ksteps;
        dense_load<MT, NT, Ops>(a1, b1, ap + Ops::kStep, wl, ksteps, more ? S + 1 : S);
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of the MFMA block (hipcc sinks it otherwise)
        dense_mfma<MT, NT, Ops>(acc, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            dense_load<MT, NT, Ops>(a0, b0, ap + 2 * Ops::kStep, wl, ksteps, S + 2 < ksteps ? S + 2 : S);
            __builtin_amdgcn_sched_barrier(0);
            dense_mfma<MT, NT, Ops>(acc, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        ap += 2 * Ops::kStep;
    }
}

template <int MT, int NT, class Ops = OpsF32>
__device__ __forceinline__ void dense_pre(f32x16 (&acc)[MT][NT], const typename Ops::act_t *act_lane, const typename Ops::bfrag *wl,
                                          int ksteps, const LayerPre<NT, Ops> &p)
{
    typename Ops::afrag a0[MT];
    typename Ops::bfrag b0[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b0[t] = p.b[t];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a0[mt] = Ops::load_a(act_lane + mt * 32 * Ops::kLd);
    dense_pipeline<MT, NT, Ops>(acc, a0, b0, act_lane, weight_stream(wl), ksteps);
}

template <int MT, int NT, class Ops = OpsF32>
__device__ __forceinline__ void dense(f32x16 (&acc)[MT][NT], const typename Ops::act_t *act_lane, const typename Ops::bfrag *wl, int ksteps)
{
    typename Ops::afrag a0[MT];
    typename Ops::bfrag b0[NT];
    const WeightStream w = weight_stream(wl);
    dense_load<MT, NT, Ops>(a0, b0, act_lane, w, ksteps, 0);
    dense_pipeline<MT, NT, Ops>(acc, a0, b0, act_lane, w, ksteps);
}

// acc = act x W (no bias).  (Starting from the MFMA's zero constant instead of cleared accumulators -- no 16 v_mov per tile -- measured SLOWER under
// every policy: bf16 -2 %, fp32 -1 %, split fp16 -2.5 %; profiles/r06_bf16_steps.txt, r06_f32_regression_check.txt)
template <int MT, int NT, class Ops = OpsF32>
__device__ __forceinline__ void dense_from_zero(f32x16 (&acc)[MT][NT], const typename Ops::act_t *act_lane, const typename Ops::bfrag *wl, int ksteps)
{
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mt][t][q] = 0.f;
    dense<MT, NT, Ops>(acc, act_lane, wl, ksteps);
}

template <int MT, int NT, bool ROWS4>
__device__ __forceinline__ void acc_init(f32x16 (&acc)[MT][NT], const float *bias, int wave, int lane, float scale = 1.0f)
{
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float bv = bias ? scale * bias[(wave * NT + t) * 32 + (lane & 31)] : 0.f;
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

// the same in bf16: 16 accumulators = two 16-byte chunks per lane
template <int MT, int NT>
__device__ __forceinline__ void stash_store16(const f32x16 (&acc)[MT][NT], float *slot, int wave, int lane)
{
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    u32x4 *dst = (u32x4 *)slot + (size_t)wave * (MT * NT * 2) * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                u32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[i] = __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2){ acc[mt][t][8 * c + 2 * i], acc[mt][t][8 * c + 2 * i + 1] }, bf16x2));
                dst[((mt * NT + t) * 2 + c) * 64] = v;
            }
}

__device__ __forceinline__ void stash_load16(f32x16 &dst, const u32x4 *src)     // src: this lane's first chunk of the accumulator tile
{
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const u32x4 v = src[c * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[8 * c + 2 * i] = __builtin_bit_cast(float, v[i] << 16);
            dst[8 * c + 2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xffff0000u);
        }
    }
}

// the same as fp16 pairs (OpsBF16RT): two packed conversions per four values, 16 bytes per lane and store
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int MT, int NT>
__device__ __forceinline__ void stash_store_f16(const f32x16 (&acc)[MT][NT], float *slot, int wave, int lane)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    u32x4 *dst = (u32x4 *)slot + (size_t)wave * (MT * NT * 2) * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                u32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    v[i] = __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2){ acc[mt][t][8 * c + 2 * i], acc[mt][t][8 * c + 2 * i + 1] }, f16x2));
                dst[((mt * NT + t) * 2 + c) * 64] = v;
            }
}

// x times one fp16 half of w in ONE instruction (v_fma_mix_f32: the half is widened inside the multiplier; + 0 because the instruction
// is a fused multiply-add).  hipcc does not form it from (float)half * x, it converts and multiplies.
template <int HI>
__device__ __forceinline__ float mul_f16_half(unsigned w, float x)
{
    float d;
    if constexpr (HI) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "v"(x));
    else asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "v"(x));
    return d;
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
template <int MT, int NT, bool ROWS4, int KIND, class Ops = OpsF32>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[MT][NT], typename Ops::act_t *act, int wave, int lane)
{
    constexpr int LD = Ops::kLd;
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            typename Ops::act_t *o = act + (mt * 32 + 4 * h) * LD + (wave * NT + t) * 32 + j;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float z[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    z[r] = acc[mt][t][4 * g + r];
                    if constexpr (Ops::kWScale != 1.0f) z[r] *= (1.0f / Ops::kWScale);      // exact: a power of two
                }
                if constexpr (Ops::kPackedRows) {
                    // 16-bit policies are VALU-bound (profiles/r02_geometry_sweep.md): four scalar multiplies (measured faster than
                    // two v_pk_mul_f32 + the move that pairs their operands), two packed conversions, four 2-byte stores
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    if (ROWS4) {
                        f32x2 lo, hi;
                        if constexpr (KIND == 2 && Ops::kActMode == 1) {
                            float tx, dy;
                            tanhexp_parts_fast(z[0], tx, dy);
                            lo = (f32x2){ z[0] * tx, z[1] * dy };
                            hi = (f32x2){ z[2] * dy, z[3] * dy };
                        } else {
                            float y, dy;
                            act_grad<KIND, Ops::kActMode>(z[0], y, dy);
                            lo = (f32x2){ y, dy * z[1] };
                            hi = (f32x2){ z[2] * dy, z[3] * dy };
                        }
                        Ops::put_rows4(o + (8 * g) * LD, LD, lo[0], lo[1], hi[0], hi[1]);
                    } else {
                        Ops::put_rows4(o + (8 * g) * LD, LD, act_val<KIND, Ops::kActMode>(z[0]), act_val<KIND, Ops::kActMode>(z[1]),
                                       act_val<KIND, Ops::kActMode>(z[2]), act_val<KIND, Ops::kActMode>(z[3]));
                    }
                } else if (ROWS4) {
                    float y, dy;
                    act_grad<KIND, Ops::kActMode>(z[0], y, dy);
                    Ops::put2(o + (8 * g + 0) * LD, o + (8 * g + 1) * LD, y, dy * z[1]);
                    Ops::put2(o + (8 * g + 2) * LD, o + (8 * g + 3) * LD, dy * z[2], dy * z[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r += 2)
                        Ops::put2(o + (8 * g + r) * LD, o + (8 * g + r + 1) * LD, act_val<KIND, Ops::kActMode>(z[r]), act_val<KIND, Ops::kActMode>(z[r + 1]));
                }
            }
        }
}

template <int MT, int NT, bool ROWS4, class Ops = OpsF32>
__device__ __forceinline__ void epilogue_rt(const f32x16 (&acc)[MT][NT], typename Ops::act_t *act, int kind, int wave, int lane)
{
    if (kind == 0) epilogue<MT, NT, ROWS4, 0, Ops>(acc, act, wave, lane);
    else if (kind == 1) epilogue<MT, NT, ROWS4, 1, Ops>(acc, act, wave, lane);
    else epilogue<MT, NT, ROWS4, 2, Ops>(acc, act, wave, lane);
}

// ----------------------------------------------------------------------------
// input encodings
template <class Ops = OpsF32, int THREADS = kThreads>
__device__ __forceinline__ void zero_cols(typename Ops::act_t *act, int rows, int ncols, int tid)
{
    typedef typename Ops::act_t act_t;
    constexpr int E = 16 / (int)sizeof(act_t);          // elements per 16-byte store
    if (ncols % E == 0 && (Ops::kLd * (int)sizeof(act_t)) % 16 == 0 && (Ops::kPlane * (int)sizeof(act_t)) % 16 == 0) {
        // every caller's ncols is a whole number of MFMA k-steps: 16 bytes per store instead of one element (and one runtime division
        // per ELEMENT: 28 of each per thread of the bf16 colour tile of rounds 4-5, 9.6 % of its span)
        const int cpr = ncols / E;
        for (int i = tid; i < rows * cpr; i += THREADS) {
            const int r = i / cpr, c = i - r * cpr;
#pragma unroll
            for (int pl = 0; pl < Ops::kPlanes; ++pl) *(f32x4v *)(act + r * Ops::kLd + pl * Ops::kPlane + c * E) = (f32x4v){ 0.f, 0.f, 0.f, 0.f };
        }
        return;
    }
    for (int i = tid; i < rows * ncols; i += THREADS) {
        int r = i / ncols, c = i - r * ncols;
        Ops::zero(act + r * Ops::kLd + c);
    }
}

}  // namespace neddf
