// field_kernels.hip -- the coordinate-MLP hot path (99.8 % of render time in the
// reference) as hand-written gfx950 kernels.
//
// Tile engine.  A workgroup of 4 waves (one per SIMD) owns a tile of MT*32
// activation ROWS that stay in LDS for the whole network; every dense layer is
//     acc[rows, 256] (+)= act[rows, K] x W[K, 256]
// on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak = the number the roofline
// is priced against).  Wave w owns output columns [64w, 64w+64) for all rows:
//   * A operand  = activations, ds_read_b128 from LDS (row stride 260 floats ->
//     conflict-free for the 32x32x2 A fragment: lane = row, 4 consecutive k);
//   * B operand  = weights, read ONCE per workgroup straight from global/L2
//     into registers, pre-packed on the host in fragment order so that each
//     lane issues one global_load_dwordx4 per four MFMA k-steps (1 KiB/wave,
//     fully coalesced); no LDS staging, no inter-wave traffic for weights;
//   * the k order inside a super-step of 8 is permuted (lane-half h reads
//     k = 8S+4h..8S+4h+3) -- legal because A and B use the same permutation.
// For the NeDDF distance trunk the 4 rows of a sample point (value, d/dx,
// d/dy, d/dz -- the forward-mode Jacobian the density is defined from) are
// 4 CONSECUTIVE rows, so in the 32x32 accumulator layout
//     row = 8*(reg>>2) + 4*(lane>>5) + (reg&3),  col = lane&31
// one lane holds all four rows of a point for one feature in acc[4g..4g+3]:
// the activation + JVP epilogue (y = a(x), G = a'(x)*J) is register-local, one
// transcendental evaluation per four accumulators.
// Skip-connection inputs (positional encoding concatenated in front of the
// hidden state) never exist as a concatenated tensor: their partial product is
// computed at tile start, while the encoding sits in LDS, and parked in a
// per-workgroup global scratch in accumulator layout ("stash").
#include "kernels.h"
#include "device_math.h"
#include "tile_engine.h"
#include "tile_inputs.h"

namespace neddf {

// ----------------------------------------------------------------------------
// NeDDF distance trunk
// Geometry (MT, WPS, NW): MT*32 rows per tile, WPS workgroups per CU, NW waves per workgroup.  NW = 4: one wave per SIMD,
// each owning 64 output columns (NT = 2 column tiles) -- every shipped shape.  The kernels also compile with NW = 8 (two waves per
// SIMD, each owning 32 columns for ALL rows of a 128-row tile: the bf16 colour trunk ran on it in rounds 4-5, until round 6 made
// three four-wave workgroups per CU the faster shape); the packed weights are tile-major (tile = wave * NT + t), so either reads
// the same blob.
template <int MT, int WPS, class Ops, int NW = kWaves>
__global__ __launch_bounds__(64 * NW, WPS * NW / 4) void ddf_trunk_kernel(const DdfArgs a)
{
    typedef typename Ops::act_t act_t;
    typedef typename Ops::bfrag frag;            // weight fragments
    constexpr int WID = Ops::kWid, NT = WID / 32 / NW, THREADS = 64 * NW, ROWS = MT * 32, P = MT * 8, LD = Ops::kLd;
    static_assert(NT >= 1 && NT * 32 * NW == WID && MT * NT <= 8, "engine width = NW waves x NT column tiles of 32; a stash slot holds MT x NT <= 8 tiles per wave");
    constexpr int REG_BUDGET = 512 / (WPS * NW / 4);        // registers per lane at this occupancy
    // ping-pong operand registers of the dense pipeline
    constexpr int OPREGS = 2 * (MT * (int)sizeof(typename Ops::afrag) / 4 + NT * (int)sizeof(typename Ops::bfrag) / 4);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    act_t *act = (act_t *)smem;
    float *hd = (float *)(act + ROWS * LD);  // [HSPLIT][2][ROWS] head dot products
    float *lp = hd + 6 * ROWS;              // [16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const act_t *act_lane = act_lane_ptr<Ops>(act, lane);
    float *scratch = a.scratch + (size_t)blockIdx.x * kStashFloatsPerWg * (a.n_stash > 1 ? a.n_stash : 1);
    if (tid == 0) {
#pragma unroll
        for (int e = 0; e < 10; ++e) lp[e] = a.enc.lowpass[e];
    }
    const int kin = Ops::kStep * a.layer[0].ksteps;          // encoding columns incl. zero padding
    const int64_t ntiles = (a.n_points + P - 1) / P;

    int *ctl = (int *)(lp + 12);
    int64_t tile = sched_begin(a.sched, a.sched_flags, ctl, tid);
    while (tile < ntiles) {
        const int64_t p0 = tile * P;
        LayerPre<NT, Ops> pre;
        layer_prefetch<NT, Ops>(pre, a.layer[0].wp, a.layer[0].bias, a.layer[0].ksteps, wave, lane);
        zero_cols<Ops, THREADS>(act, ROWS, kin, tid);
        __syncthreads();
        int next_tile = 0;
        if (tid == 0) next_tile = sched_next(a.sched, a.sched_flags, tile);     // consumed at the end of this tile
        if (a.neus) encode_pos<true, false, Ops, THREADS>(act, 0, a.enc, lp, a.pos, a.var, p0, a.n_points, P, tid, false);   // plain PE (neus.py:118)
        else encode_pos<true, true, Ops, THREADS>(act, 0, a.enc, lp, a.pos, a.var, p0, a.n_points, P, tid);
        __syncthreads();

        f32x16 acc[MT][NT];
        // Early partials of skip layers (neddf.py:217-219).  With one skip connection (every shipped
        // config) and the 64-row tile the partial stays in registers until its layer; otherwise it is
        // parked in the per-workgroup global scratch.
        constexpr bool REG_STASH = NW == 4 ? ((MT == 2 && WPS <= 2) || (MT == 1 && WPS <= 4))      // denser packings of a CU have no registers to spare
                                           : (2 * MT * NT * 16 + OPREGS + 64 <= REG_BUDGET);
        // 128-row tiles at two workgroups per CU (16-bit operands: each fetched weight fragment feeds four M-tiles) have neither
        // the registers for a held partial nor the HBM bandwidth for a parked one: the skip layer re-encodes the positions into
        // the tile's first columns after its 256-wide product and multiplies them then (three more barriers per tile).
        constexpr bool REENCODE = (MT == 4) && (NW == 8 ? !REG_STASH : WPS == 2);
        const bool in_regs = REG_STASH && a.n_stash == 1;
        f32x16 held[REG_STASH ? MT : 1][REG_STASH ? NT : 1];
        for (int s = 0; s < (REENCODE ? 0 : a.n_stash); ++s) {
            const frag *wl = (const frag *)a.stash[s].wp + (size_t)wave * NT * a.stash[s].ksteps * 64 + lane;
            if constexpr (REG_STASH) {
                if (in_regs) {
                    acc_init<MT, NT, true>(held, nullptr, wave, lane);
                    dense<MT, NT, Ops>(held, act_lane + a.stash[s].col0, wl, a.stash[s].ksteps);
                    continue;
                }
            }
            acc_init<MT, NT, true>(acc, nullptr, wave, lane);
            dense<MT, NT, Ops>(acc, act_lane + a.stash[s].col0, wl, a.stash[s].ksteps);
            stash_store<MT, NT>(acc, scratch + (size_t)s * kStashFloatsPerWg, wave, lane);
        }
        for (int l = 0; l < a.n_layers; ++l) {      // neddf.py:214-216
            const LayerW &L = a.layer[l];
            acc_init_pre<MT, NT, true, Ops>(acc, pre);
            if (L.stash >= 0) {
                bool done = false;
                if constexpr (REG_STASH) {
                    if (in_regs) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int t = 0; t < NT; ++t) acc[mt][t] += held[mt][t];
                        done = true;
                    }
                }
                if (!done && !REENCODE) stash_add<MT, NT>(acc, scratch + (size_t)L.stash * kStashFloatsPerWg, wave, lane);
            }
            const frag *wl = (const frag *)L.wp + (size_t)wave * NT * L.ksteps * 64 + lane;
            dense_pre<MT, NT, Ops>(acc, act_lane, wl, L.ksteps, pre);
            if constexpr (REENCODE) {
                if (L.stash >= 0) {
                    const StashW &sw = a.stash[L.stash];
                    __syncthreads();                        // every wave finished reading the hidden state
                    zero_cols<Ops, THREADS>(act, ROWS, kin, tid);
                    __syncthreads();
                    if (a.neus) encode_pos<true, false, Ops, THREADS>(act, 0, a.enc, lp, a.pos, a.var, p0, a.n_points, P, tid, false);
                    else encode_pos<true, true, Ops, THREADS>(act, 0, a.enc, lp, a.pos, a.var, p0, a.n_points, P, tid);
                    __syncthreads();
                    const frag *ws_ = (const frag *)sw.wp + (size_t)wave * NT * sw.ksteps * 64 + lane;
                    dense<MT, NT, Ops>(acc, act_lane + sw.col0, ws_, sw.ksteps);
                }
            }
            if (l + 1 < a.n_layers)                 // next layer's first fragments fly during the epilogue
                layer_prefetch<NT, Ops>(pre, a.layer[l + 1].wp, a.layer[l + 1].bias, a.layer[l + 1].ksteps, wave, lane);
            __syncthreads();                        // every wave finished reading the previous activations
            epilogue_rt<MT, NT, true, Ops>(acc, act, a.activation, wave, lane);
            __syncthreads();
        }
        if (a.neus) {       // NeuS: sdf = feature 0 of the last activated layer, normal = its Jacobian rows (neus.py:132-145)
            if (tid < P && p0 + tid < a.n_points) {
                const int64_t gp = p0 + tid;
                float sdf = Ops::get(act + (4 * tid) * LD);
                float ex = expf(-a.neus_v10 * sdf);
                float den = 1 + ex;
                float rho = a.neus_v10 * ex * (1.0f / (den * den));          // neus.py:153-156
                float *pa = a.ptaux + gp * kPtAux;
                f32x4v v0 = { sdf, rho, 0.f, Ops::get(act + (4 * tid + 1) * LD) };
                f32x4v v1 = { Ops::get(act + (4 * tid + 2) * LD), Ops::get(act + (4 * tid + 3) * LD), 0.f, 0.f };
                ((f32x4v *)pa)[0] = v0; ((f32x4v *)pa)[1] = v1;
                if (a.distance) a.distance[gp] = sdf;
                if (a.density) a.density[gp] = rho;
            }
        } else {
        // heads (neddf.py:220-230): ddf_out on all four rows, aux_out likewise (rows 1..3 feed aux_gg);
        // HSPLIT threads share one (row, head) dot product so that all 256 threads work on a 64-row tile
        constexpr int HSPLIT = (4 * ROWS <= THREADS) ? 2 : 1, KQ = WID / 4 / HSPLIT;
        for (int idx = tid; idx < HSPLIT * 2 * ROWS; idx += THREADS) {
            int part = idx / (2 * ROWS), pr = idx - part * 2 * ROWS;
            int head = pr / ROWS, row = pr - head * ROWS;
            const f32x4v *w = (const f32x4v *)(head ? a.w_aux_out : a.w_ddf_out) + part * KQ;
            const act_t *ar = act + row * LD + 4 * part * KQ;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
            for (int k = 0; k < KQ; ++k) {
                float x[4];
                Ops::load4(ar + 4 * k, x);
                f32x4v ww = w[k];
                s0 = fmaf(x[0], ww[0], s0); s1 = fmaf(x[1], ww[1], s1);
                s2 = fmaf(x[2], ww[2], s2); s3 = fmaf(x[3], ww[3], s3);
            }
            hd[idx] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
        if (HSPLIT == 2) {
            if (tid < 2 * ROWS) hd[tid] += hd[2 * ROWS + tid];
            __syncthreads();
        }
        if (tid < P && p0 + tid < a.n_points) {
            const int64_t gp = p0 + tid;
            float z = hd[4 * tid] + a.b_ddf_out;
            float az = hd[ROWS + 4 * tid] + a.b_aux_out;
            float sp, dsp, t, dsg;
            softplus_grad(z, sp, dsp);               // softplus.py:38-49
            float D = sp + a.d_near;
            float dg0 = dsp * hd[4 * tid + 1], dg1 = dsp * hd[4 * tid + 2], dg2 = dsp * hd[4 * tid + 3];
            sigmoid_grad(az, t, dsg);                // sigmoid.py:38-43
            float aux = a.aux_grad_scale * t;
            float q2 = dg0 * dg0 + dg1 * dg1 + dg2 * dg2;
            float dgn = sqrtf(q2);
            float dDdt = sqrtf(q2 + aux * aux);              // neddf.py:234-238
            float Dinv = 1.0f / D;
            float rho = act_val_rt(a.density_activation, Dinv * (1 - dDdt));   // :239-240
            float ninv = 1.0f / (dgn + 1e-7f);               // :241
            float *pa = a.ptaux + gp * kPtAux;
            f32x4v v0 = { D, rho, aux, ninv * dg0 };
            f32x4v v1 = { ninv * dg1, ninv * dg2, z, az };
            f32x4v v2 = { dg0, dg1, dg2, a.aux_grad_scale * (dsg * hd[ROWS + 4 * tid + 1]) };
            f32x4v v3 = { a.aux_grad_scale * (dsg * hd[ROWS + 4 * tid + 2]),
                          a.aux_grad_scale * (dsg * hd[ROWS + 4 * tid + 3]), dgn, dDdt };
            ((f32x4v *)pa)[0] = v0; ((f32x4v *)pa)[1] = v1; ((f32x4v *)pa)[2] = v2; ((f32x4v *)pa)[3] = v3;
            if (a.distance) a.distance[gp] = D;
            if (a.density) a.density[gp] = rho;
            if (a.aux_grad) a.aux_grad[gp] = aux;
        }
        }
        // hand the trunk features to the colour kernel (value row, or all four rows in full mode)
        {
            // 16-byte chunks; the feature matrix has the element type (and the planes) of the activations, planes packed densely
            constexpr int CE = 16 / sizeof(act_t), CPP = WID / CE, CPR = Ops::kPlanes * CPP;
            act_t *features = (act_t *)a.features;
            const int fr = a.feat_rows;
            for (int idx = tid; features && idx < P * fr * CPR; idx += THREADS) {
                int r = (unsigned)idx / CPR, c4 = (unsigned)idx % CPR;
                int p = r / fr, rr = r - p * fr;
                if (p0 + p < a.n_points) {
                    f32x4v v = *(const f32x4v *)(act + (4 * p + rr) * LD + (c4 / CPP) * Ops::kPlane + CE * (c4 % CPP));
                    *(f32x4v *)(features + ((size_t)(p0 + p) * fr + rr) * (Ops::kPlanes * WID) + CE * c4) = v;
                }
            }
        }
        if (tid == 0) ctl[0] = next_tile;
        __syncthreads();
        tile = ctl[0];
    }
}

// ----------------------------------------------------------------------------
// NeDDF distance trunk, eval-minimal, with the distance gradient in REVERSE mode.
//
// The reference carries the Jacobian forward through every layer (three extra rows per sample point, neddf.py:206-230) because
// its training penalties need Jacobians of several outputs.  Rendering needs the gradient of ONE scalar -- the raw distance
// z_D = w_ddf . h_L -- with respect to the position: density = a((1 - |(grad D, aux)|) / D), normal = grad D / |grad D|
// (neddf.py:234-241).  For one scalar, reverse mode is the cheaper direction:
//     forward   h_l = a(z_l), z_l = h_{l-1} W_l + b_l                      value rows only, keep y'_l = a'(z_l)
//     backward  g_L = w_ddf * y'_L;  g_{l-1} = (g_l W_l^T) * y'_{l-1}        one row per point again
//     encoding  g_pe = g_0 W_0^T + g_skip (encoding rows of W_skip)^T;  grad_x z_D = sum_c g_pe[c] dPE_c/dx   (20 terms per axis)
// i.e. two rows of matrix work per point and layer instead of four, on tiles of 64 POINTS (all M-tile rows useful) instead of
// 16 points x 4 rows: 13 layer-equivalents of MFMA work per 64 points against 29.6.  Same function, same weights; the
// result differs from the forward-mode Jacobian only in rounding (checked against the oracle and the reference goldens by the
// same gates).  What it costs: y'_l of the tile (64 KB per layer) does not fit in LDS next to the activations, so it makes a round
// trip through a per-workgroup scratch in global memory (written by the forward epilogue, read back by the backward one right
// after its product); and the activation is evaluated per element instead of once per four accumulator rows.
// Used when the caller wants no penalties / Jacobian outputs (render_image, render_rays without fields_penalty); the
// forward-mode kernel above serves the training-mode outputs.
// Tile shape: MT M-tiles = 32 MT points per tile (64 at engine width 128 / 256, 32 at 384 / 512), four waves, two workgroups per CU
// under every operand policy.  The alternatives were built and measured in rounds 2-5 (docs/lab_notebook.md R4.2, R5.3-R5.8): 128-point
// tiles on eight waves or in two column passes per wave, three / four workgroups per CU, twin teams one barrier apart, the colour
// trunk fused onto this tile -- equal or slower, removed from the product tree in round 6.
// Per-workgroup scratch: y' of every layer + [encoding Jacobian factors | encoding copy | parked encoding gradient] (64 columns each)
size_t ddf_rev_scratch_floats_per_wg(int n_layers, int points, int width) { return (size_t)n_layers * points * width + (size_t)points * 192; }

// The barrier of the tile's per-layer phases: LDS operations complete (lgkmcnt(0)), VECTOR-MEMORY operations stay in flight.
// __syncthreads() is a workgroup fence + barrier: hipcc drains vmcnt too, so every per-layer barrier waited for the next layer's
// prefetched weight fragments, the y' stores of the epilogue before it or the y' loads just requested -- an L2 / HBM round trip
// exposed ~30 times per tile.  What crosses these barriers between waves is the LDS tile only; the global scratch a phase writes is
// read back by the same lanes (y', the parked encoding gradient), and the arrays other threads read (encoding factors, staged
// inputs) are fenced by the full barriers that stay around them.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#ifndef NEDDF_BF16_REV_OPS
#define NEDDF_BF16_REV_OPS OpsBF16RT        // -DNEDDF_BF16_REV_OPS=OpsBF16T: the straight-product kernel of rounds 2-5 (A/B partner)
#endif

// The 16 values a lane holds of one 32 x 32 accumulator block (M-tile `mt` of the points, column tile `ct` of the features) -> the LDS
// tile act[point][feature].  Straight products (rows = points): register q is point 8 (q >> 2) + 4 h + (q & 3), feature j: sixteen
// 2-byte / 4-byte stores, pairs sharing one packed conversion.  Transposed products (Ops::kTransposed: rows = features): register q is
// feature 8 (q >> 2) + 4 h + (q & 3) of point j: four 8-byte stores.
template <class Ops>
__device__ __forceinline__ void block_to_tile(typename Ops::act_t *act, int mt, int ct, const float (&v)[16], int lane)
{
    constexpr int LD = Ops::kLd;
    const int j = lane & 31, h = lane >> 5;
    if constexpr (Ops::kTransposed) {
        typename Ops::act_t *o = act + (mt * 32 + j) * LD + ct * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) Ops::put4(o + 8 * g, v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
    } else {
        typename Ops::act_t *o = act + (mt * 32 + 4 * h) * LD + ct * 32 + j;
#pragma unroll
        for (int q = 0; q < 16; q += 2) Ops::put2(o + (8 * (q >> 2) + (q & 3)) * LD, o + (8 * (q >> 2) + (q & 3) + 1) * LD, v[q], v[q + 1]);
    }
}

constexpr int kPjLd = 65;          // row stride of the LDS-resident Jacobian factors: odd, so that lanes on consecutive points hit consecutive banks
constexpr int kRevSmallFloats = 3 * 512 + 16;      // head dots / lp / ctl / staged inputs and tail partials behind the tile (lds_bytes: small + 16)
template <int KIND, bool LAST, int MT, int NT, class Ops>
__device__ __forceinline__ void rev_forward_epilogue(f32x16 (&acc)[MT][NT], typename Ops::act_t *act, float *yp, const float *wseed, int wave,
                                                     int lane)
{
    constexpr bool MASK = KIND != 2 && !LAST;       // ReLU / LeakyReLU: y' leaves as one bit per element, built on the fly
    constexpr int NMW = (MT * NT + 1) / 2;
    unsigned mbits[MASK ? NMW : 1] = { 0 };
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int ct = wave * NT + t;
            // LAST: the seed of the reverse pass dz_D / dz_L = w_ddf[feature] * y'.  Straight products: this lane's feature is ct * 32 + j;
            // transposed: register q is feature ct * 32 + 8 (q >> 2) + 4 h + (q & 3) -- wseed is then the LDS-resident copy, read 16 bytes at a time
            float ws[Ops::kTransposed ? 16 : 1];
            if constexpr (LAST) {
                if constexpr (Ops::kTransposed) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4v w4 = *(const f32x4v *)(wseed + ct * 32 + 8 * g + 4 * h);
                        ws[4 * g] = w4[0]; ws[4 * g + 1] = w4[1]; ws[4 * g + 2] = w4[2]; ws[4 * g + 3] = w4[3];
                    }
                } else ws[0] = wseed[ct * 32 + j];
            }
            float y[16];
            constexpr int ILP = NEDDF_ACT_ILP;      // elements per activation batch (device_math.h act_grad_n: the same arithmetic, interleaved)
#pragma unroll
            for (int q0 = 0; q0 < 16; q0 += ILP) {
                float z[ILP], yy[ILP], dy[ILP];
#pragma unroll
                for (int i = 0; i < ILP; ++i) {
                    z[i] = acc[mt][t][q0 + i];
                    if constexpr (Ops::kWScale != 1.0f) z[i] *= (1.0f / Ops::kWScale);
                }
                act_grad_n<KIND, Ops::kActMode, ILP>(z, yy, dy);
#pragma unroll
                for (int i = 0; i < ILP; ++i) {
                    const int q = q0 + i;
                    y[q] = yy[i];
                    // y' replaces the accumulator (LAST: the seed).  The stashed copy carries the weight scale's inverse (a power of two:
                    // exact), so the reverse epilogue is one multiply per element
                    if constexpr (MASK) mbits[(mt * NT + t) / 2] |= (dy[i] == 1.0f ? 1u : 0u) << (16 * ((mt * NT + t) & 1) + q);
                    else acc[mt][t][q] = LAST ? ws[Ops::kTransposed ? q : 0] * dy[i] : dy[i] * (1.0f / Ops::kWScale);
                }
            }
            block_to_tile<Ops>(act, mt, ct, y, lane);
        }
    // y' leaves in the accumulators' own fragment order (16 bytes per lane, 1 KiB per wave and store): only this workgroup's
    // same lanes read it back, so nothing needs it row-major.  Under the bf16 policy it travels in 16 bits (the product it enters
    // is rounded to bf16 anyway): at that policy's speed the fp32 round trip (26 GB per launch) would be the kernel's bound
    if (!LAST) {
        if constexpr (KIND == 2) {
            if constexpr (Ops::kStashF16) {
#if !(NEDDF_PROBE_NOY & 1)
                stash_store_f16<MT, NT>(acc, yp, wave, lane);
#endif
            }
            else if constexpr (Ops::kStash16) stash_store16<MT, NT>(acc, yp, wave, lane);
            else stash_store<MT, NT>(acc, yp, wave, lane);
        } else {
            // ReLU / LeakyReLU: y' takes two values, so ONE BIT per element travels (16 per accumulator tile, two tiles per
            // dword: 8 bytes per lane and layer instead of 256) -- 12 KB per workgroup for six layers, 6 MB per launch grid: it
            // never leaves the L2, where the value-carrying round trip of tanhExp is 26 GB of HBM traffic per 2^21-point launch
            unsigned *dst = (unsigned *)yp + (size_t)wave * NMW * 64 + lane;
#pragma unroll
            for (int w = 0; w < NMW; ++w) dst[w * 64] = mbits[w];
        }
    }
}

// y' of a ReLU (slope 0) / LeakyReLU (slope 0.01) layer from its mask bit, with the weight scale's inverse folded in like the stashed values
template <class Ops>
__device__ __forceinline__ float mask_factor(unsigned bit, int kind)
{
    constexpr float inv = 1.0f / Ops::kWScale;
    return bit ? inv : (kind == 1 ? 0.01f * inv : 0.0f);
}

// MASKY kernels serve ReLU / LeakyReLU (y' as mask bits), the others tanhExp (y' as values): each carries only its own epilogues
template <bool LAST, bool MASKY, int MT, int NT, class Ops>
__device__ __forceinline__ void rev_forward_epilogue_rt(f32x16 (&acc)[MT][NT], typename Ops::act_t *act, float *yp, const float *wseed, int kind,
                                                        int wave, int lane)
{
    if constexpr (MASKY) {
        if (kind == 0) rev_forward_epilogue<0, LAST, MT, NT, Ops>(acc, act, yp, wseed, wave, lane);
        else rev_forward_epilogue<1, LAST, MT, NT, Ops>(acc, act, yp, wseed, wave, lane);
    } else rev_forward_epilogue<2, LAST, MT, NT, Ops>(acc, act, yp, wseed, wave, lane);
}

template <int MT, class Ops, bool MASKY>
__global__ __launch_bounds__(kThreads, 2) void ddf_rev_kernel(const DdfArgs a)
{
    typedef typename Ops::act_t act_t;
    typedef typename Ops::bfrag frag;
    constexpr int NW = kWaves, WID = Ops::kWid, NT = WID / 32 / NW, THREADS = kThreads, ROWS = MT * 32, P = ROWS, LD = Ops::kLd;
    static_assert(NT >= 1 && NT * 32 * NW == WID, "engine width = four waves x NT column tiles of 32");
    static_assert(MT == 1 || MT == 2, "32- or 64-point tiles");
    // 32 x 32 blocks of the [P, 64] encoding gradient per wave; a 32-point tile has two blocks for four waves: the upper waves idle there
    constexpr int NBLK = 2 * MT, BPW = NBLK >= NW ? NBLK / NW : 1;
    static_assert(BPW * NW == NBLK || NBLK < NW, "the encoding gradient's blocks must divide over the waves");
    constexpr int PARTS = THREADS / ROWS;               // threads per point in the tail
    extern __shared__ __attribute__((aligned(16))) float smem[];
    act_t *act = (act_t *)smem;
    float *hd = (float *)(act + ROWS * LD);  // [2 k-halves][2 heads][ROWS] head dot products
    float *lp = hd + 8 * ROWS;               // (transposed policy: [4 waves][2 heads][ROWS] partial head products)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const act_t *act_lane = act_lane_ptr<Ops>(act, lane);
    float *yp = a.rev_scratch + (size_t)blockIdx.x * ((size_t)a.n_layers * P * WID + (size_t)P * 192);
    float *pj = yp + (size_t)a.n_layers * ROWS * WID;           // [ROWS][64] dPE/dx factors: [q] sine half, [32 + q] cosine half
    float *pv = pj + ROWS * 64;                                  // [ROWS][64] the encoding itself, for the skip layer
    float *pg = pv + ROWS * 64;                                  // [ROWS][64] the skip layers' share of the encoding gradient, parked
    float *stg = lp + 16;                                        // [6 ROWS] staged positions / variances, later [PARTS][3][ROWS] tail partials (behind lp / ctl)
    // Ops::kEncInLds (bf16: the tile is half the bytes): the encoding also lives in a narrow LDS tile of its own for the whole tile, so a
    // skip layer multiplies it from there -- no reload from the scratch, no extra barriers (14.1 k -> 6.5 k cycles for that layer)
    act_t *enc_tile = (act_t *)(hd + kRevSmallFloats);
    const act_t *enc_lane = act_lane_ptr<EncView<Ops>>(enc_tile, lane);
    // Ops::kTransposed: the layers' biases and the distance head's weights as LDS-resident vectors [n_layers + 1][WID] behind the encoding
    // tile, filled once per workgroup -- an accumulator's four consecutive features take theirs with one ds_read_b128 (rev_lds_bytes)
    float *vecs = (float *)(enc_tile + (Ops::kEncInLds ? ROWS * kEncLd : 0));
    // ... and behind them the factors of the encoding's Jacobian [ROWS][kPjLd] (written by the encoding loop, read by the tail: they made
    // a round trip through the global scratch -- four scattered 4-byte stores per item and an exposed L2 round trip in the tail)
    float *pjl = vecs + (a.n_layers + 1) * WID;
    // ... and last the tile's sample points as the rays gave them [9][ROWS] (direction, position, variance): computed once at the tile's start for
    // the encoding, they are also what the tail writes into the colour kernel's per-point record -- recomputed there, they were a 64-bit division
    // and three dependent memory round trips on one wave while the other three waited at the tile's last barrier
    float *rsv = Ops::kTransposed ? pjl + ROWS * kPjLd : vecs;
    if constexpr (Ops::kTransposed) {
        for (int i = tid; i < (a.n_layers + 1) * WID; i += THREADS) {
            const int l = i / WID, c = i - l * WID;
            vecs[i] = l < a.n_layers ? (a.layer[l].bias ? a.layer[l].bias[c] : 0.0f) : a.w_ddf_out[c];
        }
    }
    // Ops::kTransposed: the two heads (neddf.py:220-230, value rows) as ONE narrow product on the matrix pipe instead of 2 x 128-term dot
    // products per thread on the vector ALU: the A operand's rows 0..2 carry w_ddf as three bf16 terms (hi + mid + lo = the fp32 weight
    // to 2^-24: the heads stay fp32-exact on bf16 activations, as before), rows 4..6 w_aux likewise, the other rows are zero.  The
    // fragments of rows 0..7 are built once per workgroup into the (otherwise unused) Jacobian-factor slot of its global scratch.
    // super-steps of a 256-wide product, as a RUN-TIME value (the argument block carries it, DdfArgs::ks_hidden): with the compile-time constant hipcc
    // unrolls the reverse products completely, materialises one 64-bit address per weight fragment, spills them and reloads each
    // behind an s_waitcnt vmcnt(0) -- which serialises the operand prefetch of half of the kernel's matrix work
    const int KS = a.ks_hidden;
    u32x4 *hfrag = (u32x4 *)pj;
    if constexpr (Ops::kTransposed) {
        for (int f = tid; f < KS * 16; f += THREADS) {
            const int S = f >> 4, li = f & 15, row = li & 7, hh = li >> 3, term = row & 3;
            const float *wv = row < 4 ? a.w_ddf_out : a.w_aux_out;
            unsigned short bits[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float w = wv[Ops::kStep * S + 8 * hh + r];
                unsigned short t16 = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {       // peel hi, mid, lo: each the bf16 rounding of what the previous terms left
                    t16 = Ops::cvt(w);
                    if (k == term) break;
                    w -= __builtin_bit_cast(float, (unsigned int)t16 << 16);
                }
                bits[r] = term < 3 ? t16 : (unsigned short)0;
            }
            hfrag[f] = (u32x4){ bits[0] | ((unsigned)bits[1] << 16), bits[2] | ((unsigned)bits[3] << 16), bits[4] | ((unsigned)bits[5] << 16),
                                bits[6] | ((unsigned)bits[7] << 16) };
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int e = 0; e < 10; ++e) lp[e] = a.enc.lowpass[e];
    }
    const int kin = Ops::kStep * a.layer[0].ksteps;
    const int K3 = 3 * a.enc.E, KH = a.enc.KH;
    const unsigned k3magic = (1u << 20) / (unsigned)K3 + 1u;
    const int64_t ntiles = (a.n_points + P - 1) / P;
    const int h = lane >> 5;

    int *ctl = (int *)(lp + 12);
    NEDDF_STAMP_DECL;
    int64_t tile = sched_begin(a.sched, a.sched_flags, ctl, tid);
    while (tile < ntiles) {
        NEDDF_STAMP_TILE();
        STAMP_WALL(0);
        STAMP();                                    // 0: tile start
        const int64_t p0 = tile * P;
        LayerPre<NT, Ops> pre;
        layer_prefetch<NT, Ops>(pre, a.layer[0].wp, a.layer[0].bias, a.layer[0].ksteps, wave, lane);
        // the tile's positions / variances in ONE coalesced request per array into LDS (the tail's partial-sum area, free until the end of
        // the tile): the encoding loop below then waits on LDS only -- per-item global loads were a chain of dependent L2 round trips
        // (8 per thread and tile: 14 k of a bf16 tile's 191 k cycles, profiles/r04_stamp_timeline_bf16.txt)
        if (a.rays.rd) {        // the points come from the rays: cone / point moments here, no sampling tensors (kernels.h RaySrc)
            if (tid < ROWS) {
                const int64_t gp = p0 + tid < a.n_points ? p0 + tid : a.n_points - 1;
                float ps[3], vr[3], dr[3];
                ray_point(a.rays, a.rays.base + gp, ps, vr, dr);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    stg[tid * 3 + k] = ps[k]; stg[3 * ROWS + tid * 3 + k] = a.neus ? 0.0f : vr[k];
                    rsv[k * ROWS + tid] = dr[k]; rsv[(3 + k) * ROWS + tid] = ps[k]; rsv[(6 + k) * ROWS + tid] = vr[k];
                }
            }
        } else
        for (int idx = tid; idx < 6 * ROWS; idx += THREADS) {
            const int k = idx < 3 * ROWS ? idx : idx - 3 * ROWS, p = (k * 43691) >> 17, d = k - 3 * p;      // k / 3 for k < 3 * 128
            const int64_t gp = p0 + p < a.n_points ? p0 + p : a.n_points - 1;
            stg[idx] = idx < 3 * ROWS ? a.pos[gp * 3 + d] : (a.neus ? 0.0f : a.var[gp * 3 + d]);
        }
        // the encoding loop below writes columns [0, K3) and [KH, KH + K3) of every row: only the padding columns need zeros
        {
            const int g1 = KH - K3, npad = kin - 2 * K3;            // [K3, KH) and [KH + K3, kin)
            for (int i = tid; i < ROWS * npad; i += THREADS) {
                const int r = i / npad, k = i - r * npad;
                Ops::zero(act + r * LD + (k < g1 ? K3 + k : KH + K3 + (k - g1)));
            }
            if constexpr (Ops::kEncInLds) {
                const int npe = 64 - 2 * K3;
                for (int i = tid; i < ROWS * npe; i += THREADS) {
                    const int r = i / npe, k = i - r * npe;
                    Ops::zero(enc_tile + r * kEncLd + (k < g1 ? K3 + k : KH + K3 + (k - g1)));
                }
            }
        }
        __syncthreads();
        int next_tile = 0;
        if (tid == 0) next_tile = sched_next(a.sched, a.sched_flags, tile);
        // scaled integrated encoding (neddf.py:193-204), value rows; the factors of its Jacobian go to the scratch (or stay in LDS)
        // one (point, pair) item = the pair's sine / cosine value into the tile (and the encoding's own tile), its Jacobian factors into pjl / pj
        auto encode_item = [&](int p, int q, int e, float px, float vx) {
            float vs, vc, js, jc;
            if (a.neus) pe_pair<false, Ops::kFast>(e, px, 0.0f, lp[e], vs, vc, js, jc);      // plain PE (neus.py:118)
            else pe_pair<true, Ops::kFast>(e, px, vx, lp[e], vs, vc, js, jc);
            Ops::put(act + p * LD + q, vs);
            Ops::put(act + p * LD + KH + q, vc);
            if constexpr (Ops::kEncInLds) {
                Ops::put(enc_tile + p * kEncLd + q, vs);
                Ops::put(enc_tile + p * kEncLd + KH + q, vc);
            } else {
                pv[p * 64 + q] = vs;               // same column order as the LDS tile: [sine half (KH) | cosine half (KH)]
                pv[p * 64 + KH + q] = vc;
            }
            if constexpr (Ops::kTransposed) {
                pjl[p * kPjLd + q] = js;
                pjl[p * kPjLd + 32 + q] = jc;
            } else {
                pj[p * 64 + q] = js;
                pj[p * 64 + 32 + q] = jc;
            }
        };
        if constexpr (P % 64 == 0 && THREADS % P == 0 && Ops::kTransposed && Ops::kEncInLds) {
            // a thread keeps ITS point (p = tid mod P) and walks the pairs q = tid / P, + THREADS / P, ...: q, the frequency, the axis and the
            // low-pass factor are wave-uniform scalars, the point's six inputs are read from the staging area once.  Only where every output of
            // an item stays in LDS: with the Jacobian factors / the encoding in the global scratch (fp32, split fp16) consecutive lanes on
            // consecutive POINTS make their stores uncoalesced (fp32 distance kernel +0.7 %)
            const int p = tid % P;
            float x3[3], v3[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) { x3[d] = stg[p * 3 + d]; v3[d] = stg[3 * ROWS + p * 3 + d]; }
            for (int q = __builtin_amdgcn_readfirstlane(tid / P); q < K3; q += THREADS / P) {
                const int e = q / 3, d = q - 3 * e;
                encode_item(p, q, e, d == 0 ? x3[0] : d == 1 ? x3[1] : x3[2], d == 0 ? v3[0] : d == 1 ? v3[1] : v3[2]);
            }
        } else
        for (int item = tid; item < P * K3; item += THREADS) {
            // item / K3 for item < 4096, K3 <= 30: (item * magic) >> 20 is exact; q / 3 for q < 30: (q * 11) >> 5
            const int p = (int)(((unsigned)item * k3magic) >> 20), q = item - p * K3;
            const int e = (q * 11) >> 5, d = q - 3 * e;
            encode_item(p, q, e, stg[p * 3 + d], stg[3 * ROWS + p * 3 + d]);
        }
        STAMP();                                    // 1: encoding done
        __syncthreads();
        STAMP();                                    // 2: encoding barrier passed

        f32x16 acc[MT][NT];
        // ---- forward, value rows
        for (int l = 0; l < a.n_layers; ++l) {
            const LayerW &L = a.layer[l];
            if constexpr (Ops::kTransposed) {       // bias of the four consecutive features of every register group: LDS -> accumulator registers
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4v b4 = *(const f32x4v *)(vecs + l * WID + (wave * NT + t) * 32 + 8 * g + 4 * h);
                            acc[mt][t][4 * g] = b4[0]; acc[mt][t][4 * g + 1] = b4[1]; acc[mt][t][4 * g + 2] = b4[2]; acc[mt][t][4 * g + 3] = b4[3];
                        }
            } else acc_init_pre<MT, NT, false, Ops>(acc, pre);
            dense_pre<MT, NT, Ops>(acc, act_lane, (const frag *)L.wp + (size_t)wave * NT * L.ksteps * 64 + lane, L.ksteps, pre);
            if (L.stash >= 0 && Ops::kEncInLds) {       // cat([encoding, h]) (neddf.py:217-219): the encoding's own LDS tile
                const StashW &sw = a.stash[L.stash];
                dense<MT, NT, EncView<Ops>>(acc, enc_lane + sw.col0, (const frag *)sw.wp + (size_t)wave * NT * sw.ksteps * 64 + lane, sw.ksteps);
            } else if (L.stash >= 0) {  // ... or the encoding comes back from the scratch into the tile's first columns
                const StashW &sw = a.stash[L.stash];
                __syncthreads();                        // every wave finished reading the hidden state
                for (int i = tid; i < ROWS * (kin / 4); i += THREADS) {
                    const int r = i / (kin / 4), c = i - r * (kin / 4);
                    f32x4v v = { 0.f, 0.f, 0.f, 0.f };
                    if (4 * c < 64) v = *(const f32x4v *)(pv + r * 64 + 4 * c);
                    // columns [K3, KH) and [KH + K3, 2 KH) of the scratch rows were never written: mask them
                    float x[4] = { v[0], v[1], v[2], v[3] };
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int cc = 4 * c + u, qq = cc < KH ? cc : cc - KH;
                        Ops::put(act + r * LD + cc, (cc < 2 * KH && qq < K3) ? x[u] : 0.f);
                    }
                }
                __syncthreads();
                dense<MT, NT, Ops>(acc, act_lane + sw.col0, (const frag *)sw.wp + (size_t)wave * NT * sw.ksteps * 64 + lane, sw.ksteps);
            }
            if (l + 1 < a.n_layers) layer_prefetch<NT, Ops>(pre, a.layer[l + 1].wp, a.layer[l + 1].bias, a.layer[l + 1].ksteps, wave, lane);
            STAMP();                                // forward layer l: 3 + 4l product done
            lds_barrier();
            STAMP();                                //                  4 + 4l barrier passed
            if (l + 1 < a.n_layers) rev_forward_epilogue_rt<false, MASKY, MT, NT, Ops>(acc, act, yp + (size_t)((NEDDF_PROBE_NOY & 4) ? 0 : l) * ROWS * WID, nullptr, a.activation, wave, lane);
            else rev_forward_epilogue_rt<true, MASKY, MT, NT, Ops>(acc, act, nullptr, Ops::kTransposed ? vecs + a.n_layers * WID : a.w_ddf_out, a.activation, wave, lane);
            STAMP();                                //                  5 + 4l epilogue done
            lds_barrier();
            STAMP();                                //                  6 + 4l barrier passed
        }
        // ---- heads on the features (value only: the distance gradient comes from the reverse pass, the aux gradient's own
        // Jacobian is not an eval output), and the feature hand-off to the colour kernel
        if constexpr (Ops::kTransposed) {
            // wave w multiplies its quarter of k for both point tiles: rows = head terms, columns = points (8 MFMAs per wave and tile)
            f32x16 ha[MT];
            const int kq = KS / NW;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 16; ++q) ha[mt][q] = 0.f;
            const int row = lane & 31;
            for (int S = wave * kq; S < (wave + 1) * kq; ++S) {
                u32x4 wf = { 0u, 0u, 0u, 0u };
                if (row < 8) wf = hfrag[S * 16 + h * 8 + row];
                const frag wfr = __builtin_bit_cast(frag, wf);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) ha[mt] = Ops::mfma(Ops::load_a(act_lane + mt * 32 * LD + S * Ops::kStep), wfr, ha[mt], 0);
            }
            // lane (j, h): registers 0..2 = the three terms of head h for point j
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) hd[(wave * 2 + h) * ROWS + mt * 32 + (lane & 31)] = (ha[mt][0] + ha[mt][1]) + ha[mt][2];
        } else
        for (int item = tid; item < 4 * ROWS; item += THREADS) {       // 2 k-halves x 2 heads x ROWS rows
            const int part = item / (2 * ROWS), head = (item / ROWS) & 1, row = item % ROWS;
            const f32x4v *w = (const f32x4v *)(head ? a.w_aux_out : a.w_ddf_out) + part * (WID / 8);
            const act_t *ar = act + row * LD + part * (WID / 2);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
            for (int k = 0; k < WID / 8; ++k) {
                float x[4];
                Ops::load4(ar + 4 * k, x);
                f32x4v ww = w[k];
                s0 = fmaf(x[0], ww[0], s0); s1 = fmaf(x[1], ww[1], s1);
                s2 = fmaf(x[2], ww[2], s2); s3 = fmaf(x[3], ww[3], s3);
            }
            hd[item] = (s0 + s1) + (s2 + s3);
        }
        {
            constexpr int CE = 16 / sizeof(act_t), CPP = WID / CE, CPR = Ops::kPlanes * CPP;
            act_t *features = (act_t *)a.features;
            constexpr bool AFF = Ops::kTransposed && THREADS % CPR == 0 && (P * CPR) % (4 * THREADS) == 0;      // (fp32: measured 0.6 % slower with it)
            if (AFF && features && p0 + P <= a.n_points) {
                // a full tile: the thread keeps its column, chunk i is row r0 + i * RSTEP.  Four chunks in four register sets per batch: written
                // as one loop, every store waits for the previous one's acknowledgement before its data registers are loaded again
                // (s_waitcnt vmcnt(0) per chunk: eight L2 round trips in a row in the bf16 tile, sixteen in the fp32 one)
                constexpr int RSTEP = AFF ? THREADS / CPR : 1, NCH = P * CPR / THREADS;
                const int r0 = tid / CPR, c4 = tid % CPR;
                const act_t *src = act + r0 * LD + (c4 / CPP) * Ops::kPlane + CE * (c4 % CPP);
                act_t *dst = features + (size_t)(p0 + r0) * (Ops::kPlanes * WID) + CE * c4;
#pragma unroll
                for (int b = 0; b < NCH; b += 4) {
                    f32x4v v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = *(const f32x4v *)(src + (b + i) * RSTEP * LD);
#pragma unroll
                    for (int i = 0; i < 4; ++i) *(f32x4v *)(dst + (size_t)(b + i) * RSTEP * (Ops::kPlanes * WID)) = v[i];
                }
            } else
            for (int idx = tid; features && idx < P * CPR; idx += THREADS) {
                const int p = (unsigned)idx / CPR, c4 = (unsigned)idx % CPR;
                if (p0 + p < a.n_points) {
                    f32x4v v = *(const f32x4v *)(act + p * LD + (c4 / CPP) * Ops::kPlane + CE * (c4 % CPP));
                    *(f32x4v *)(features + (size_t)(p0 + p) * (Ops::kPlanes * WID) + CE * c4) = v;
                }
            }
        }
        STAMP();                        // F: heads + feature hand-off done
        __syncthreads();                // the features are consumed: the tile now carries gradients
        STAMP();                        // F + 1
        // ---- reverse pass: g_L (held in the accumulators since the last epilogue) -> LDS
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = acc[mt][t][q];
                block_to_tile<Ops>(act, mt, wave * NT + t, v, lane);
            }
        STAMP();                        // F + 2: g_L stored
        __syncthreads();
        STAMP();                        // F + 3
        // the [P, 64] encoding gradient in 32 x 32 blocks, BPW per wave: block b = wave * BPW + i is M-tile b >> 1, N-tile b & 1.  The
        // skip layer's share waits in the scratch, not in registers, while the remaining layers run (the product loop needs them)
        f32x4v *gpe_park = (f32x4v *)pg + (size_t)wave * BPW * 4 * 64 + lane;
        bool parked = false;
        for (int l = a.n_layers - 1; l >= 1; --l) {
            if (a.layer[l].stash >= 0) {    // cat([encoding, h]): the encoding rows of W_l take their share of g_l (every skip layer adds its own)
#pragma unroll
                for (int i = 0; i < BPW; ++i) {
                    const int b = wave * BPW + i;
                    if (NBLK < NW && b >= NBLK) break;
                    f32x16 gs[1][1];
                    if (parked) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4v v = gpe_park[(i * 4 + g) * 64];
                            gs[0][0][4 * g] = v[0]; gs[0][0][4 * g + 1] = v[1]; gs[0][0][4 * g + 2] = v[2]; gs[0][0][4 * g + 3] = v[3];
                        }
                        dense<1, 1, Ops>(gs, act_lane + (b >> 1) * 32 * LD, (const frag *)a.wT_pe_skip[a.layer[l].stash] + (size_t)(b & 1) * KS * 64 + lane, KS);
                    } else dense_from_zero<1, 1, Ops>(gs, act_lane + (b >> 1) * 32 * LD, (const frag *)a.wT_pe_skip[a.layer[l].stash] + (size_t)(b & 1) * KS * 64 + lane, KS);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4v v = { gs[0][0][4 * g], gs[0][0][4 * g + 1], gs[0][0][4 * g + 2], gs[0][0][4 * g + 3] };
                        gpe_park[(i * 4 + g) * 64] = v;
                    }
                }
                parked = true;
            }
            // y'_{l-1} of this lane's accumulator positions comes back M-tile by M-tile through two register sets, requested after the
            // product (requesting them before it -- 64 more live registers -- measured no faster: the CU's other waves cover the latency)
            f32x16 yb[2][NT];
            const f32x4v *ysrc = (const f32x4v *)(yp + (size_t)((NEDDF_PROBE_NOY & 4) ? 0 : l - 1) * ROWS * WID) + (size_t)wave * (MT * NT * (Ops::kStash16 ? 2 : 4)) * 64 + lane;
            auto load_y = [&](f32x16 (&dst)[NT], int mt) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if constexpr (Ops::kStash16) stash_load16(dst[t], (const u32x4 *)ysrc + ((mt * NT + t) * 2) * 64);
                    else
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4v v = ysrc[((mt * NT + t) * 4 + g) * 64];
                            dst[t][4 * g] = v[0]; dst[t][4 * g + 1] = v[1]; dst[t][4 * g + 2] = v[2]; dst[t][4 * g + 3] = v[3];
                        }
                }
            };
            constexpr int NMW = (MT * NT + 1) / 2;
            unsigned mw[NMW];
            if constexpr (MASKY) {      // ReLU / LeakyReLU: the layer's mask bits (rev_forward_epilogue), requested ahead of the product
                const unsigned *msrc = (const unsigned *)(yp + (size_t)(l - 1) * ROWS * WID) + (size_t)wave * NMW * 64 + lane;
#pragma unroll
                for (int w = 0; w < NMW; ++w) mw[w] = msrc[w * 64];
            }
            STAMP();                    // reverse layer: +0 skip share / setup done
            dense_from_zero<MT, NT, Ops>(acc, act_lane, (const frag *)a.wT[l] + (size_t)wave * NT * KS * 64 + lane, KS);
            STAMP();                    //                +1 product done
            if constexpr (MASKY) {
                lds_barrier();          // every wave finished reading g_l
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const unsigned bits = mw[(mt * NT + t) / 2] >> (16 * ((mt * NT + t) & 1));
                        float v[16];
#pragma unroll
                        for (int q = 0; q < 16; ++q) v[q] = acc[mt][t][q] * mask_factor<Ops>((bits >> q) & 1u, a.activation);
                        block_to_tile<Ops>(act, mt, wave * NT + t, v, lane);
                    }
                lds_barrier();
            } else if constexpr (Ops::kStashF16) {
                // y' as fp16 pairs, packed until the multiply: one mixed-precision multiply per element (v_fma_mix_f32), no unpacking
                u32x4 yraw[MT][NT][2];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
#if NEDDF_PROBE_NOY & 2      // timing probes (results invalid): NEDDF_PROBE_NOY bit 0 = no y' stores, bit 1 = no y' loads, bit 2 = every layer's y' in ONE slot (L2-resident)
                        for (int c = 0; c < 2; ++c) yraw[mt][t][c] = (u32x4){ 0x3c003c00u + (unsigned)lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u };
#else
                        for (int c = 0; c < 2; ++c) yraw[mt][t][c] = ((const u32x4 *)ysrc)[((mt * NT + t) * 2 + c) * 64];
#endif
                lds_barrier();              // every wave finished reading g_l (the y' just requested stays in flight)
                STAMP();                    //                +2 barrier passed
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        float v[16];
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const unsigned w = yraw[mt][t][q >> 3][(q & 7) >> 1];
                            v[q] = (q & 1) ? mul_f16_half<1>(w, acc[mt][t][q]) : mul_f16_half<0>(w, acc[mt][t][q]);
                        }
                        block_to_tile<Ops>(act, mt, wave * NT + t, v, lane);
                    }
                STAMP();                    //                +3 y' multiply + store done
                lds_barrier();
                STAMP();                    //                +4 barrier passed
            } else {
                load_y(yb[0], 0);
                if (MT > 1) load_y(yb[1], 1);
                lds_barrier();              // every wave finished reading g_l (the y' just requested stays in flight)
                STAMP();                    //                +2 barrier passed
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        float v[16];
#pragma unroll
                        for (int q = 0; q < 16; ++q) v[q] = acc[mt][t][q] * yb[mt & 1][t][q];
                        block_to_tile<Ops>(act, mt, wave * NT + t, v, lane);
                    }
                STAMP();                    //                +3 y' multiply + store done
                lds_barrier();
                STAMP();                    //                +4 barrier passed
            }
        }
        f32x16 gpe[BPW][1];
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const int b = wave * BPW + i;
            if (NBLK < NW && b >= NBLK) break;
            f32x16 one[1][1];
            if (parked) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4v v = gpe_park[(i * 4 + g) * 64];
                    one[0][0][4 * g] = v[0]; one[0][0][4 * g + 1] = v[1]; one[0][0][4 * g + 2] = v[2]; one[0][0][4 * g + 3] = v[3];
                }
                dense<1, 1, Ops>(one, act_lane + (b >> 1) * 32 * LD, (const frag *)a.wT_pe0 + (size_t)(b & 1) * KS * 64 + lane, KS);       // g_0 W_0^T
            } else dense_from_zero<1, 1, Ops>(one, act_lane + (b >> 1) * 32 * LD, (const frag *)a.wT_pe0 + (size_t)(b & 1) * KS * 64 + lane, KS);
            gpe[i][0] = one[0][0];
        }
        __syncthreads();                // every wave finished reading g_0
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            const int b = wave * BPW + i;
            if (NBLK < NW && b >= NBLK) break;
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = Ops::kWScale != 1.0f ? gpe[i][0][q] * (1.0f / Ops::kWScale) : gpe[i][0][q];
            block_to_tile<Ops>(act, b >> 1, b & 1, v, lane);
        }
        __syncthreads();
        // ---- per point: grad_x z_D = sum over the encoding channels of g_pe * dPE/dx (PARTS threads per point, every PARTS-th frequency
        // each: the 60-term contraction was one wave's work while three idled), then the head arithmetic (neddf.py:220-241)
        {
            const int part = tid / ROWS, p = tid - part * ROWS;
            const act_t *gr = act + p * LD;
            const float *pjr = Ops::kTransposed ? pjl + p * kPjLd : pj + p * 64;
            float gp_[3] = { 0.f, 0.f, 0.f };
            for (int e = part; e < a.enc.E; e += PARTS)
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int q = 3 * e + d;
                    gp_[d] = fmaf(Ops::get(gr + q), pjr[q], gp_[d]);
                    gp_[d] = fmaf(Ops::get(gr + KH + q), pjr[32 + q], gp_[d]);
                }
#pragma unroll
            for (int d = 0; d < 3; ++d) stg[(part * 3 + d) * ROWS + p] = gp_[d];
            __syncthreads();
        }
        if (tid < P && p0 + tid < a.n_points) {
            const int64_t gp = p0 + tid;
            float gz[3] = { 0.f, 0.f, 0.f };
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int q = 0; q < PARTS; ++q) gz[d] += stg[(q * 3 + d) * ROWS + tid];
            float zs, azs;
            if constexpr (Ops::kTransposed) {       // four waves' partial products
                zs = (hd[tid] + hd[2 * ROWS + tid]) + (hd[4 * ROWS + tid] + hd[6 * ROWS + tid]);
                azs = (hd[ROWS + tid] + hd[3 * ROWS + tid]) + (hd[5 * ROWS + tid] + hd[7 * ROWS + tid]);
            } else { zs = hd[tid] + hd[2 * ROWS + tid]; azs = hd[ROWS + tid] + hd[3 * ROWS + tid]; }
            const float z = zs + a.b_ddf_out;
            const float az = azs + a.b_aux_out;
            if (a.neus) {       // NeuS: the "distance head" is e_0, so z is the sdf and gz its position gradient (neus.py:132-156)
                const float ex = expf(-a.neus_v10 * z), den = 1 + ex;
                const float rho = a.neus_v10 * ex * (1.0f / (den * den));
                float *pa = a.ptaux + gp * kPtAux;
                f32x4v v0 = { z, rho, 0.f, gz[0] };
                f32x4v v1 = { gz[1], gz[2], 0.f, 0.f };
                ((f32x4v *)pa)[0] = v0; ((f32x4v *)pa)[1] = v1;
                if (a.distance) a.distance[gp] = z;
                if (a.density) a.density[gp] = rho;
            } else {
                float sp, dsp, t, dsg;
                if constexpr (Ops::kFast) softplus_grad_fast(z, sp, dsp);
                else softplus_grad(z, sp, dsp);          // softplus.py:38-49
                const float D = sp + a.d_near;
                const float dg0 = dsp * gz[0], dg1 = dsp * gz[1], dg2 = dsp * gz[2];
                if constexpr (Ops::kFast) sigmoid_grad_fast(az, t, dsg);
                else sigmoid_grad(az, t, dsg);           // sigmoid.py:38-43
                const float aux = a.aux_grad_scale * t;
                const float q2 = dg0 * dg0 + dg1 * dg1 + dg2 * dg2;
                const float dgn = sqrtf(q2);
                const float dDdt = sqrtf(q2 + aux * aux);              // neddf.py:234-238
                const float Dinv = 1.0f / D;
                const float rho = act_val_rt(a.density_activation, Dinv * (1 - dDdt));   // :239-240
                const float ninv = 1.0f / (dgn + 1e-7f);               // :241
                float *pa = a.ptaux + gp * kPtAux;
                f32x4v v0 = { D, rho, aux, ninv * dg0 };
                f32x4v v1 = { ninv * dg1, ninv * dg2, z, az };
                f32x4v v2 = { dg0, dg1, dg2, 0.f };
                f32x4v v3 = { 0.f, 0.f, dgn, dDdt };
                if (a.rays.rd) {    // the colour kernel's inputs ride in the slots it does not read in eval-minimal mode (PA_R_*)
                    float ps[3], vr[3], dr[3];        // (this thread's own values from the tile's start: rsv)
#pragma unroll
                    for (int k = 0; k < 3; ++k) { dr[k] = rsv[k * ROWS + tid]; ps[k] = rsv[(3 + k) * ROWS + tid]; vr[k] = rsv[(6 + k) * ROWS + tid]; }
                    v0[0] = dr[0]; v0[1] = dr[1]; v0[2] = dr[2];
                    v2[0] = ps[0]; v2[1] = ps[1]; v2[2] = ps[2]; v2[3] = vr[0];
                    v3[0] = vr[1]; v3[1] = vr[2];
                }
                ((f32x4v *)pa)[0] = v0; ((f32x4v *)pa)[1] = v1; ((f32x4v *)pa)[2] = v2; ((f32x4v *)pa)[3] = v3;
                if (a.distance) a.distance[gp] = D;
                if (a.density) a.density[gp] = rho;
                if (a.aux_grad) a.aux_grad[gp] = aux;
            }
        }
        STAMP();                        // tail: encoding gradient, head arithmetic, outputs done
        if (tid == 0) ctl[0] = next_tile;
        __syncthreads();
        tile = ctl[0];
        STAMP();                        // tile end
        STAMP_WALL(1);
    }
    NEDDF_STAMP_EXIT();
}

// ----------------------------------------------------------------------------
// NeDDF colour trunk.  ROWS4 = false: eval-minimal, one row per point (the
// colour Jacobian is dead code in eval); ROWS4 = true:
// full mode with Jacobian rows + field penalties (neddf.py:244-300).
template <bool ROWS4, int MT, int WPS, class Ops, int NW = kWaves>
__global__ __launch_bounds__(64 * NW, WPS * NW / 4) void col_trunk_kernel(const ColArgs a)
{
    typedef typename Ops::act_t act_t;
    typedef typename Ops::bfrag frag;            // weight fragments
    constexpr int WID = Ops::kWid, NT = WID / 32 / NW, THREADS = 64 * NW;        // geometry: see ddf_trunk_kernel
    constexpr int ROWS = MT * 32, P = ROWS4 ? MT * 8 : ROWS, RPP = ROWS4 ? 4 : 1, LD = Ops::kLd;
    constexpr int REG_BUDGET = 512 / (WPS * NW / 4);
    constexpr int OPREGS = 2 * (MT * (int)sizeof(typename Ops::afrag) / 4 + NT * (int)sizeof(typename Ops::bfrag) / 4);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    act_t *act = (act_t *)smem;
    float *hd = (float *)(act + ROWS * LD);  // [THREADS / ROWS][ROWS][3] partial colour dots
    float *lp = hd + THREADS * 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const act_t *act_lane = act_lane_ptr<Ops>(act, lane);
    if (tid == 0) {
#pragma unroll
        for (int e = 0; e < 10; ++e) lp[e] = a.enc.lowpass[e];
    }
    // the colour head's weights [WID][3], LDS-resident behind the small scratch (col_lds_bytes): as global loads they were twelve vector loads per
    // sixteen features, requested and awaited inside the head's loop -- 9.5 % of the bf16 tile of rounds 4-5 (eight waves, 128 points; profiles/r06_stamp_col_bf16_before.txt)
    float *wo = lp + 16;
    for (int i = tid; i < WID * 3; i += THREADS) wo[i] = a.w_out[i];
    const int ka = Ops::kStep * a.ksteps_a;
    const int c_dir = 2 * a.enc.KH, c_n = c_dir + 2 * a.enc.KD;
    const int64_t ntiles = (a.n_points + P - 1) / P;

    int *ctl = (int *)(lp + 12);
    int64_t tile = sched_begin(a.sched, a.sched_flags, ctl, tid);
    NEDDF_STAMP_DECL;           // (stamp builds: tools/stamp_timeline_col.py names the phases)
    while (tile < ntiles) {
        const int64_t p0 = tile * P;
        NEDDF_STAMP_TILE();
        STAMP_WALL(0);
        STAMP();                                    // 0: tile start
        // layer 0, small-input segment: [embed_pos | embed_dir | norm_dir] (neddf.py:243)
        zero_cols<Ops, THREADS>(act, ROWS, ka, tid);
        __syncthreads();
        STAMP();                                    // 1: columns zeroed + barrier
        int next_tile = 0;
        if (tid == 0) next_tile = sched_next(a.sched, a.sched_flags, tile);
        if (a.mode == 1) {      // NeuS: [pos | gradient | pad | embed_dir] (neus.py:146-149)
            for (int i = tid; i < P * 3; i += THREADS) {
                int p = i / 3, d = i - 3 * p;
                int64_t gp = p0 + p < a.n_points ? p0 + p : a.n_points - 1;
                Ops::put(act + (RPP * p) * LD + d, a.pos[gp * 3 + d]);
                Ops::put(act + (RPP * p) * LD + 3 + d, a.ptaux[gp * kPtAux + PA_N0 + d]);
            }
            encode_dir<ROWS4, Ops, THREADS>(act, 8, a.enc, a.dir, p0, a.n_points, P, tid);
        } else if (!ROWS4 && a.rays) {     // the sample point rides in the per-point record (kernels.h PA_R_*)
            encode_pos<ROWS4, false, Ops, THREADS>(act, 0, a.enc, lp, a.ptaux + PA_R_POS, a.ptaux + PA_R_VAR, p0, a.n_points, P, tid, true, kPtAux);
            encode_dir<ROWS4, Ops, THREADS>(act, c_dir, a.enc, a.ptaux + PA_R_DIR, p0, a.n_points, P, tid, kPtAux);
            for (int i = tid; i < P * 3; i += THREADS) {
                int p = i / 3, d = i - 3 * p;
                int64_t gp = p0 + p < a.n_points ? p0 + p : a.n_points - 1;
                Ops::put(act + (RPP * p) * LD + c_n + d, a.ptaux[gp * kPtAux + PA_N0 + d]);
            }
        } else {
            encode_pos<ROWS4, false, Ops, THREADS>(act, 0, a.enc, lp, a.pos, a.var, p0, a.n_points, P, tid);
            encode_dir<ROWS4, Ops, THREADS>(act, c_dir, a.enc, a.dir, p0, a.n_points, P, tid);
            for (int i = tid; i < P * 3; i += THREADS) {
                int p = i / 3, d = i - 3 * p;
                int64_t gp = p0 + p < a.n_points ? p0 + p : a.n_points - 1;
                Ops::put(act + (RPP * p) * LD + c_n + d, a.ptaux[gp * kPtAux + PA_N0 + d]);
            }
        }
        STAMP();                                    // 2: encodings done
        __syncthreads();
        STAMP();                                    // 3: barrier passed
        f32x16 acc[MT][NT];
        // layer 0, feature segment: in the eval-minimal 64-row tile the trunk features are requested now
        // (global -> VGPR) and land in LDS after the small-input dense; the other variants lack the registers
        constexpr int CE = 16 / sizeof(act_t), CPP = WID / CE, CPR = Ops::kPlanes * CPP;     // 16-byte chunks per feature row
        constexpr int NF = ROWS * CPR / THREADS;
        static_assert(NF * THREADS == ROWS * CPR, "a tile's feature chunks divide evenly over the threads");
        auto lds_chunk = [&](int idx) {     // chunk idx of the tile -> its place in LDS (planes are kPlane elements apart)
            const int r = (unsigned)idx / CPR, c4 = (unsigned)idx % CPR;
            return (f32x4v *)(act + r * LD + (c4 / CPP) * Ops::kPlane + CE * (c4 % CPP));
        };
        constexpr bool FPRE = NW == 4 ? ((MT == 2 && WPS <= 2) || (MT == 1 && WPS <= 4)) && !ROWS4
                                      : !ROWS4 && (MT * NT * 16 + OPREGS + NF * 4 + 64 <= REG_BUDGET);
        f32x4v fpre[FPRE ? NF : 1];
        auto feature_src = [&](int idx) {
            int r = (unsigned)idx / CPR, c4 = (unsigned)idx % CPR;
            int64_t grow = p0 * RPP + r;
            int64_t last = a.n_points * RPP - 1;
            if (grow > last) grow = last;
            int64_t src = ROWS4 ? grow : grow * a.feat_rows;      // value row of [n][feat_rows][256]
            return (const f32x4v *)((const act_t *)a.features + (size_t)src * (Ops::kPlanes * WID) + CE * c4);
        };
        // A full tile's chunks are affine in the chunk index: chunk tid + i * THREADS is row r0 + i * RSTEP at the thread's own column, so one
        // pointer per thread and a uniform stride replace a clamped 64-bit address per chunk (the general form stays for the launch's last tile).
        constexpr bool AFF = THREADS % CPR == 0;        // a thread keeps its column over the tile's chunks (every shape but the 384-wide ones)
        constexpr int RSTEP = AFF ? THREADS / CPR : 1;
        const bool full_tile = AFF && (p0 + P) <= a.n_points;
        const int fr0 = tid / CPR, fc4 = tid % CPR;
        const char *fsrc = (const char *)a.features + ((size_t)(p0 * RPP + fr0) * (ROWS4 ? 1 : a.feat_rows) * (Ops::kPlanes * WID) + CE * fc4) * sizeof(act_t);
        const size_t fstep = (size_t)RSTEP * (ROWS4 ? 1 : a.feat_rows) * (Ops::kPlanes * WID) * sizeof(act_t);
        act_t *fdst = act + fr0 * LD + (fc4 / CPP) * Ops::kPlane + CE * (fc4 % CPP);
        if constexpr (FPRE) {
            if (full_tile) {
#pragma unroll
                for (int i = 0; i < NF; ++i) fpre[i] = *(const f32x4v *)(fsrc + i * fstep);
            } else {
#pragma unroll
                for (int i = 0; i < NF; ++i) fpre[i] = *feature_src(tid + i * THREADS);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        acc_init<MT, NT, ROWS4>(acc, a.layer[0].bias, wave, lane, Ops::kWScale);
        dense<MT, NT, Ops>(acc, act_lane, (const frag *)a.wp_a + (size_t)wave * NT * a.ksteps_a * 64 + lane, a.ksteps_a);
        LayerPre<NT, Ops> pre;
        layer_prefetch<NT, Ops>(pre, a.layer[0].wp, nullptr, a.layer[0].ksteps, wave, lane);
        STAMP();                                    // 4: features requested, small-input product done
        __syncthreads();
        STAMP();                                    // 5: barrier passed
        if constexpr (FPRE) {
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                if constexpr (AFF) *(f32x4v *)(fdst + i * RSTEP * LD) = fpre[i];
                else *lds_chunk(tid + i * THREADS) = fpre[i];
            }
        } else {
            // in batches of four chunks per thread (as a rolled loop hipcc waits for every chunk before it requests the next: eight round trips
            // in a row per tile of the shapes without FPRE)
            constexpr int FB = NF % 4 == 0 ? 4 : 1;
#pragma unroll
            for (int b = 0; b < NF; b += FB) {
                f32x4v f[FB];
                if (full_tile) {
#pragma unroll
                    for (int i = 0; i < FB; ++i) f[i] = *(const f32x4v *)(fsrc + (b + i) * fstep);
                } else {
#pragma unroll
                    for (int i = 0; i < FB; ++i) f[i] = *feature_src(tid + (b + i) * THREADS);
                }
#pragma unroll
                for (int i = 0; i < FB; ++i) {
                    if constexpr (AFF) *(f32x4v *)(fdst + (b + i) * RSTEP * LD) = f[i];
                    else *lds_chunk(tid + (b + i) * THREADS) = f[i];
                }
            }
        }
        STAMP();                                    // 6: features in LDS
        __syncthreads();
        STAMP();                                    // 7: barrier passed
        for (int l = 0; l < a.n_layers; ++l) {                     // neddf.py:254-256
            const LayerW &L = a.layer[l];
            if (l > 0) acc_init_pre<MT, NT, ROWS4, Ops>(acc, pre);
            dense_pre<MT, NT, Ops>(acc, act_lane, (const frag *)L.wp + (size_t)wave * NT * L.ksteps * 64 + lane, L.ksteps, pre);
            if (l + 1 < a.n_layers)
                layer_prefetch<NT, Ops>(pre, a.layer[l + 1].wp, a.layer[l + 1].bias, a.layer[l + 1].ksteps, wave, lane);
            STAMP();                                // layer l: 8 + 4l product done
            __syncthreads();
            STAMP();                                //          9 + 4l barrier passed
            epilogue_rt<MT, NT, ROWS4, Ops>(acc, act, a.activation, wave, lane);
            STAMP();                                //          10 + 4l epilogue done
            __syncthreads();
            STAMP();                                //          11 + 4l barrier passed
        }
        // layer_col_out 256 -> 3 (neddf.py:257), no output activation.  Every thread works: NPART = THREADS / ROWS threads share a
        // row, each over WID / NPART consecutive features.  With 64- and 128-row tiles a wave's lanes share the part index, so the weights are
        // wave-uniform (scalar loads, SGPR operands) instead of one vector load per multiply-add; the 128 idle threads and those
        // loads made this head 2.8 % of the kernel (profiles/r03_col_ablation.txt; the bf16 shape of rounds 4-5 (eight waves) still had them until
        // round 6: 9.5 % of its tile)
        constexpr int NPART = THREADS / ROWS, KPART = WID / NPART;
        static_assert(NPART >= 1 && NPART * ROWS == THREADS && KPART % 4 == 0, "the colour head splits a row over THREADS / ROWS threads");
        {
            const int part = ROWS % 64 == 0 ? __builtin_amdgcn_readfirstlane(tid / ROWS) : tid / ROWS, row = tid - (tid / ROWS) * ROWS;
            const act_t *ar = act + row * LD + part * KPART;
            const float *w = wo + part * KPART * 3;
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll 4
            for (int k = 0; k < KPART / 4; ++k) {
                float x[4];
                Ops::load4(ar + 4 * k, x);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c0 = fmaf(x[u], w[(4 * k + u) * 3 + 0], c0);
                    c1 = fmaf(x[u], w[(4 * k + u) * 3 + 1], c1);
                    c2 = fmaf(x[u], w[(4 * k + u) * 3 + 2], c2);
                }
            }
            hd[tid * 3 + 0] = c0; hd[tid * 3 + 1] = c1; hd[tid * 3 + 2] = c2;
        }
        STAMP();                                    // H: head dot products done
        __syncthreads();
        STAMP();                                    // H + 1: barrier passed
        if (tid < P && p0 + tid < a.n_points) {
            const int64_t gp = p0 + tid;
            float c[RPP][3];
#pragma unroll
            for (int r = 0; r < RPP; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float sum = (r == 0 ? a.b_out[k] : 0.f);
#pragma unroll
                    for (int q = 0; q < NPART; ++q) sum += hd[(q * ROWS + RPP * tid + r) * 3 + k];
                    c[r][k] = sum;
                }
            if (a.final_act >= 0)
#pragma unroll
                for (int k = 0; k < 3; ++k) c[0][k] = act_val_rt(a.final_act, c[0][k]);
            a.color[gp * 3 + 0] = c[0][0]; a.color[gp * 3 + 1] = c[0][1]; a.color[gp * 3 + 2] = c[0][2];
            if (ROWS4 && a.penalty) {                          // neddf.py:260-300
                const float *pa = a.ptaux + gp * kPtAux;
                float D = pa[PA_D], aux = pa[PA_AUX], dgn = pa[PA_DGN], dDdt = pa[PA_DDDT];
                float z = pa[PA_DDF_RAW], az = pa[PA_AUX_RAW];
                float Dinv = 1.0f / D;
                float pen[6];
                float d2 = pa[PA_AGG0] * pa[PA_N0] + pa[PA_AGG1] * pa[PA_N1] + pa[PA_AGG2] * pa[PA_N2];
                float rest = 3 * aux * Dinv;
                float sc = aux * dgn * D;
                pen[0] = sc * ((d2 - rest) * (d2 - rest));
                float t1 = fmaxf(-1.0f + dDdt, 0.f);
                pen[1] = t1 * t1;
                float a1 = fmaxf(-4.6f - z, 0.f), a2 = fmaxf(-a.distance_range_max + z, 0.f);
                pen[2] = (a1 + a2) * (a1 + a2);
                float b1 = fmaxf(-4.6f - az, 0.f), b2 = fmaxf(-4.6f + az, 0.f);
                pen[3] = (b1 + b2) * (b1 + b2);
                pen[4] = 0.f; pen[5] = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float c1 = fmaxf(-0.0f - c[0][k], 0.f), c2 = fmaxf(-1.0f + c[0][k], 0.f);
                    pen[4] += (c1 + c2) * (c1 + c2);
                    float s = c[RPP > 1 ? 1 : 0][k] * pa[PA_DG0] + c[RPP > 2 ? 2 : 0][k] * pa[PA_DG1] + c[RPP > 3 ? 3 : 0][k] * pa[PA_DG2];
                    pen[5] += s * s;
                }
                float tot = 0.f;
#pragma unroll
                for (int k = 0; k < 6; ++k) tot += a.penalty_has[k] ? pen[k] * a.penalty_weight[k] : pen[k];
                a.penalty[gp] = tot;
            }
        }
        STAMP();                                    // H + 2: outputs written
        if (tid == 0) ctl[0] = next_tile;
        __syncthreads();
        STAMP();                                    // H + 3: tile end
        STAMP_WALL(1);
        tile = ctl[0];
    }
    NEDDF_STAMP_EXIT();
}

// ----------------------------------------------------------------------------
// plain NeRF field (nerf.py:139-165): value rows only, 128 points per tile
template <int MT, int WPS, class Ops>
__global__ __launch_bounds__(kThreads, WPS) void nerf_kernel(const NerfArgs a)
{
    typedef typename Ops::act_t act_t;
    typedef typename Ops::bfrag frag;            // weight fragments
    // NTC: column tiles per wave of the colour head's hidden layer (layer_width / 2 outputs, padded to a multiple of 128)
    constexpr int WID = Ops::kWid, NT = WID / 128, NTC = (WID / 2 + 127) / 128, HC = NTC * 128, ROWS = MT * 32, P = ROWS, LD = Ops::kLd;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    act_t *act = (act_t *)smem;
    float *hd = (float *)(act + ROWS * LD);  // [2][ROWS][3]
    float *lp = hd + 2 * ROWS * 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const act_t *act_lane = act_lane_ptr<Ops>(act, lane);
    float *scratch = a.scratch + (size_t)blockIdx.x * kStashFloatsPerWg * (a.n_stash > 1 ? a.n_stash : 1);
    if (tid == 0) {
#pragma unroll
        for (int e = 0; e < 10; ++e) lp[e] = a.enc.lowpass[e];
    }
    const int c_dir = a.stash[a.col_stash].col0;                            // first column of the direction encoding
    const int kin = c_dir + Ops::kStep * a.stash[a.col_stash].ksteps;
    const int64_t ntiles = (a.n_points + P - 1) / P;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t p0 = tile * P;
        zero_cols<Ops>(act, ROWS, kin, tid);
        __syncthreads();
        encode_pos<false, false, Ops>(act, 0, a.enc, lp, a.pos, a.var, p0, a.n_points, P, tid);
        encode_dir<false, Ops>(act, c_dir, a.enc, a.dir, p0, a.n_points, P, tid);
        __syncthreads();
        f32x16 acc[MT][NT];
        // Early partials (products with the encodings, which only exist in LDS at tile start).  With the 64-row tile and the usual
        // one skip connection + the colour head's direction segment both stay in registers (64 + 32 per lane) until their layers;
        // otherwise they are parked in the per-workgroup global scratch (3 KB of HBM traffic per point).
        constexpr bool REG_STASH = (MT == 2) && (WPS <= 2);
        const bool in_regs = REG_STASH && a.n_stash == 2;
        f32x16 held[REG_STASH ? MT : 1][REG_STASH ? NT : 1], held1[REG_STASH ? MT : 1][REG_STASH ? NTC : 1];
        for (int s = 0; s < a.n_stash; ++s) {
            const frag *wl = (const frag *)a.stash[s].wp;
            float *slot = scratch + (size_t)s * kStashFloatsPerWg;
            if (s == a.col_stash) {         // colour head's view-direction segment: layer_width / 2 outputs (NTC tiles per wave)
                if constexpr (REG_STASH) {
                    if (in_regs) {
                        acc_init<MT, NTC, false>(held1, nullptr, wave, lane);
                        dense<MT, NTC, Ops>(held1, act_lane + a.stash[s].col0, wl + (size_t)wave * NTC * a.stash[s].ksteps * 64 + lane, a.stash[s].ksteps);
                        continue;
                    }
                }
                f32x16 acc1[MT][NTC];
                acc_init<MT, NTC, false>(acc1, nullptr, wave, lane);
                dense<MT, NTC, Ops>(acc1, act_lane + a.stash[s].col0, wl + (size_t)wave * NTC * a.stash[s].ksteps * 64 + lane, a.stash[s].ksteps);
                stash_store<MT, NTC>(acc1, slot, wave, lane);
            } else {                        // skip layers: cat([hx, embed_pos]) (nerf.py:154-155)
                if constexpr (REG_STASH) {
                    if (in_regs) {
                        acc_init<MT, NT, false>(held, nullptr, wave, lane);
                        dense<MT, NT, Ops>(held, act_lane + a.stash[s].col0, wl + (size_t)wave * NT * a.stash[s].ksteps * 64 + lane, a.stash[s].ksteps);
                        continue;
                    }
                }
                acc_init<MT, NT, false>(acc, nullptr, wave, lane);
                dense<MT, NT, Ops>(acc, act_lane + a.stash[s].col0, wl + (size_t)wave * NT * a.stash[s].ksteps * 64 + lane, a.stash[s].ksteps);
                stash_store<MT, NT>(acc, slot, wave, lane);
            }
        }
        for (int l = 0; l < a.n_layers; ++l) {
            const LayerW &L = a.layer[l];
            acc_init<MT, NT, false>(acc, L.bias, wave, lane, Ops::kWScale);
            if (L.stash >= 0) {
                bool done = false;
                if constexpr (REG_STASH) {
                    if (in_regs) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int t = 0; t < NT; ++t) acc[mt][t] += held[mt][t];
                        done = true;
                    }
                }
                if (!done) stash_add<MT, NT>(acc, scratch + (size_t)L.stash * kStashFloatsPerWg, wave, lane);
            }
            dense<MT, NT, Ops>(acc, act_lane, (const frag *)L.wp + (size_t)wave * NT * L.ksteps * 64 + lane, L.ksteps);
            __syncthreads();
            epilogue_rt<MT, NT, false, Ops>(acc, act, a.activation, wave, lane);
            __syncthreads();
        }
        // density head (nerf.py:156)
        if (tid < ROWS) {
            const act_t *ar = act + tid * LD;
            const f32x4v *w = (const f32x4v *)a.w_density;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
            for (int k = 0; k < WID / 4; ++k) {
                float x[4];
                Ops::load4(ar + 4 * k, x);
                f32x4v ww = w[k];
                s0 = fmaf(x[0], ww[0], s0); s1 = fmaf(x[1], ww[1], s1);
                s2 = fmaf(x[2], ww[2], s2); s3 = fmaf(x[3], ww[3], s3);
            }
            if (p0 + tid < a.n_points)
                a.density[p0 + tid] = act_val_rt(a.density_activation, (s0 + s1) + (s2 + s3) + a.b_density);
        }
        // colour head: Linear(256+dir, 128) -> ReLU -> Linear(128, 3) (nerf.py:99-103,158-159)
        {
            f32x16 acc1[MT][NTC];
            acc_init<MT, NTC, false>(acc1, a.col0.bias, wave, lane, Ops::kWScale);
            bool done = false;
            if constexpr (REG_STASH) {
                if (in_regs) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int t = 0; t < NTC; ++t) acc1[mt][t] += held1[mt][t];
                    done = true;
                }
            }
            if (!done) stash_add<MT, NTC>(acc1, scratch + (size_t)a.col_stash * kStashFloatsPerWg, wave, lane);
            dense<MT, NTC, Ops>(acc1, act_lane, (const frag *)a.col0.wp + (size_t)wave * NTC * a.col0.ksteps * 64 + lane, a.col0.ksteps);
            __syncthreads();
            epilogue<MT, NTC, false, 0, Ops>(acc1, act, wave, lane);
            __syncthreads();
        }
        for (int idx = tid; idx < 2 * ROWS; idx += kThreads) {
            int half = idx / ROWS, row = idx - half * ROWS;
            const act_t *ar = act + row * LD + half * (HC / 2);       // w_col1: [3][HC], zero beyond layer_width / 2
            float c[3] = { 0.f, 0.f, 0.f };
#pragma unroll 4
            for (int k = 0; k < HC / 8; ++k) {
                float x[4];
                Ops::load4(ar + 4 * k, x);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int o = 0; o < 3; ++o) c[o] = fmaf(x[u], a.w_col1[o * HC + half * (HC / 2) + 4 * k + u], c[o]);
            }
            hd[idx * 3 + 0] = c[0]; hd[idx * 3 + 1] = c[1]; hd[idx * 3 + 2] = c[2];
        }
        __syncthreads();
        if (tid < P && p0 + tid < a.n_points)
            for (int k = 0; k < 3; ++k)
                a.color[(p0 + tid) * 3 + k] = hd[tid * 3 + k] + hd[(ROWS + tid) * 3 + k] + a.b_col1[k];
        __syncthreads();
    }
}

// ----------------------------------------------------------------------------
// LinearGradFunction.forward (linear.py:40-46) as a stand-alone op on the same tile
// engine: y = xW + b, G = JW for N points; Cin <= 256 (zero-padded to 8), Cout = 128*NT.
template <int NT>
__global__ __launch_bounds__(kThreads, 1) void linear_grad_kernel(const float *x, const float *J, int64_t n_points, int cin, int ldx,
                                                                  int ksteps, const float *wp, const float *bias, float *y, float *G,
                                                                  int ldo, int nvalid, int accumulate)
{
    // one (K block, N block) of the layer: input columns [0, cin) of rows with stride ldx, output columns [0, nvalid) of rows with stride
    // ldo (the caller offsets the pointers); accumulate adds to what an earlier K block left
    constexpr int MT = 4, ROWS = MT * 32, P = MT * 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *act = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *act_lane = act + (lane & 31) * kActLd + 4 * (lane >> 5);
    const int64_t ntiles = (n_points + P - 1) / P;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t p0 = tile * P;
        zero_cols(act, ROWS, 8 * ksteps, tid);
        __syncthreads();
        for (int i = tid; i < ROWS * cin; i += kThreads) {
            int r = i / cin, c = i - r * cin;
            int64_t gp = p0 + (r >> 2);
            if (gp < n_points) act[r * kActLd + c] = (r & 3) == 0 ? x[gp * ldx + c] : J[(gp * 3 + (r & 3) - 1) * ldx + c];
        }
        __syncthreads();
        f32x16 acc[MT][NT];
        acc_init<MT, NT, true>(acc, bias, wave, lane);
        dense<MT, NT>(acc, act_lane, (const f32x4v *)wp + (size_t)wave * NT * ksteps * 64 + lane, ksteps);
        const int j = lane & 31, h = lane >> 5;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int64_t gp = p0 + mt * 8 + 2 * g + h;
                    int col = (wave * NT + t) * 32 + j;
                    if (gp < n_points && col < nvalid) {
                        float *yo = y + gp * ldo + col;
                        *yo = accumulate ? *yo + acc[mt][t][4 * g] : acc[mt][t][4 * g];
#pragma unroll
                        for (int r = 1; r < 4; ++r) {
                            float *go = G + (gp * 3 + r - 1) * ldo + col;
                            *go = accumulate ? *go + acc[mt][t][4 * g + r] : acc[mt][t][4 * g + r];
                        }
                    }
                }
        __syncthreads();
    }
}

void launch_linear_grad(const float *x, const float *J, int64_t n, int cin, int ldx, int cout_block, int ksteps, const float *wp,
                        const float *bias, float *y, float *G, int ldo, int nvalid, int accumulate, int grid, hipStream_t s)
{
    size_t lds = field_lds_bytes(4);
    static bool once = ((void)hipFuncSetAttribute((const void *)linear_grad_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)field_lds_bytes(4)),
                        (void)hipFuncSetAttribute((const void *)linear_grad_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)field_lds_bytes(4)), true);
    (void)once;
    if (cout_block == 256) hipLaunchKernelGGL((linear_grad_kernel<2>), dim3(grid), dim3(kThreads), lds, s, x, J, n, cin, ldx, ksteps, wp, bias, y, G, ldo, nvalid, accumulate);
    else hipLaunchKernelGGL((linear_grad_kernel<1>), dim3(grid), dim3(kThreads), lds, s, x, J, n, cin, ldx, ksteps, wp, bias, y, G, ldo, nvalid, accumulate);
}

// ----------------------------------------------------------------------------
template <class Ops>
static size_t lds_bytes(int mt)      // the tile + the kernels' small scratch behind it: 6 floats per row (head dot products) or 3 per thread (8 waves at most), + 16
{
    const size_t small = (size_t)(2 * mt * 32 * 3) > (size_t)(3 * 512) ? (size_t)(2 * mt * 32 * 3) : (size_t)(3 * 512);
    return (size_t)mt * 32 * Ops::kLd * sizeof(typename Ops::act_t) + (small + 16) * sizeof(float);
}
size_t field_lds_bytes(int mt) { return lds_bytes<OpsF32>(mt); }

// Tile geometry (MT, WPS, NW) per operand policy at engine width 256 -- see ddf_trunk_kernel.  One compiled shape per kernel and
// policy: the fastest of the sweeps of rounds 1-5 (profiles/r02_geometry_sweep.md, docs/lab_notebook.md R4.2, R5.3-R5.8; the other
// shapes and their switches left the product tree in round 6).
//   forward-mode distance trunk / colour trunk (Jacobian rows, training-mode outputs)
//     fp32        (2, 2, 4): 64-row tiles, two workgroups per CU (one workgroup's VALU epilogue beside the other's MFMA stream:
//                 134 TF against 118 TF for one 128-row workgroup per CU, 124 / 119 TF for 32-row tiles at three / four)
//     bf16        (4, 2, 4): 128-row tiles (half the LDS bytes of fp32): each fetched weight fragment feeds four M-tiles;
//                 its colour trunk (2, 3, 4): 64-row tiles, THREE workgroups per CU (43 KB of LDS and 168 registers each).  Rounds 4-5 ran it
//                 on (4, 2, 8); once round 6 had taken the waits out of the phases around the layers, three independent four-wave
//                 workgroups beat two eight-wave ones in lockstep: 55.2 -> 50.4 ms per C2 step ((4, 2, 4): 52.5; (2, 2, 4): 56.6)
//     split fp16  (2, 2, 4): two fp16 planes = the LDS bytes of fp32, same shape as fp32
//   reverse-mode distance kernel: 64-point tiles, four waves, two workgroups per CU under every policy (ddf_rev_kernel)
// Engine widths other than 256 (hidden width padded to 128 / 384 / 512; the reference's constructors take any width,
// neddf.py:52-66, nerf.py:34-44) have ONE shape per width under every operand policy: 64-row tiles at 128 columns, 32-row tiles
// at 384 / 512 (the LDS tile and the accumulators grow with the width: 32 x 516 floats and 4 column tiles per wave at 512 are
// the footprint of the 64 x 260 tile with 2 column tiles at 256) -- always two workgroups per CU.
struct Geo {
    int mt, wps, nw;
};
static Geo geo_w(int width) { return width == 128 ? Geo{ 2, 2, 4 } : Geo{ 1, 2, 4 }; }
static Geo geo(int operands, int width) { return width != 256 ? geo_w(width) : (operands == 1 ? Geo{ 4, 2, 4 } : Geo{ 2, 2, 4 }); }
static Geo geo_col(int operands, int width) { return width != 256 ? geo_w(width) : (operands == 1 ? Geo{ 2, 3, 4 } : Geo{ 2, 2, 4 }); }
static Geo geo_nerf(int width) { return width == 256 ? Geo{ 2, 2, 4 } : geo_w(width); }
int field_wgs_per_cu(int operands, int width) { return geo(operands, width).wps; }
int col_wgs_per_cu(int operands, int width) { return geo_col(operands, width).wps; }
int ddf_points_per_tile(int operands, int width) { return geo(operands, width).mt * 8; }
int col_points_per_tile(bool rows4, int operands, int width) { return rows4 ? geo_col(operands, width).mt * 8 : geo_col(operands, width).mt * 32; }
int nerf_points_per_tile(int width) { return geo_nerf(width).mt * 32; }
int nerf_wgs_per_cu(int width) { return geo_nerf(width).wps; }
int ddf_rev_points(int, int width) { return (width == 128 || width == 256) ? 64 : 32; }
int ddf_rev_wgs_per_cu(int, int) { return 2; }

static void set_lds(const void *fn, size_t bytes)
{
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int MT, int WPS, int NW, class Ops>
static void launch_ddf_g(const DdfArgs &a, int grid, hipStream_t s)
{
    static bool once = (set_lds((const void *)ddf_trunk_kernel<MT, WPS, Ops, NW>, lds_bytes<Ops>(MT)), true);
    (void)once;
    hipLaunchKernelGGL((ddf_trunk_kernel<MT, WPS, Ops, NW>), dim3(grid), dim3(64 * NW), lds_bytes<Ops>(MT), s, a);
}

template <int MT, int WPS, int NW, class Ops>
static void launch_col_g(const ColArgs &a, int grid, bool rows4, hipStream_t s)
{
    const size_t lds = lds_bytes<Ops>(MT) + (size_t)Ops::kWid * 3 * sizeof(float);      // + the colour head's weights (col_trunk_kernel: wo)
    static bool once = (set_lds((const void *)col_trunk_kernel<false, MT, WPS, Ops, NW>, lds),
                        set_lds((const void *)col_trunk_kernel<true, MT, WPS, Ops, NW>, lds), true);
    (void)once;
    if (rows4) hipLaunchKernelGGL((col_trunk_kernel<true, MT, WPS, Ops, NW>), dim3(grid), dim3(64 * NW), lds, s, a);
    else hipLaunchKernelGGL((col_trunk_kernel<false, MT, WPS, Ops, NW>), dim3(grid), dim3(64 * NW), lds, s, a);
}

// the three operand policies over one engine width; MT etc. follow geo() / geo_col() above
template <int WID>
static void launch_ddf_w(const DdfArgs &a, int grid, hipStream_t s)
{
    constexpr int MT = WID <= 256 ? 2 : 1;
    if (a.operands == 2) launch_ddf_g<MT, 2, 4, OpsF16SplitT<WID>>(a, grid, s);
    else if (a.operands == 1) launch_ddf_g<(WID == 256 ? 4 : MT), 2, 4, OpsBF16T<WID>>(a, grid, s);
    else launch_ddf_g<MT, 2, 4, OpsF32T<WID>>(a, grid, s);
}

void launch_ddf(const DdfArgs &a, int grid, hipStream_t s)
{
    if (a.width == 128) return launch_ddf_w<128>(a, grid, s);
    if (a.width == 384) return launch_ddf_w<384>(a, grid, s);
    if (a.width == 512) return launch_ddf_w<512>(a, grid, s);
    launch_ddf_w<256>(a, grid, s);
}

// reverse-mode kernel (see ddf_rev_kernel)
template <class Ops>
static size_t rev_lds_bytes(int mt, int n_layers)
{
    return lds_bytes<Ops>(mt) + (Ops::kEncInLds ? (size_t)mt * 32 * kEncLd * sizeof(typename Ops::act_t) : 0) +
           // the LDS-resident bias / head vectors and the encoding's Jacobian factors (ddf_rev_kernel: vecs, pjl)
           (Ops::kTransposed ? ((size_t)(n_layers + 1) * Ops::kWid + (size_t)mt * 32 * kPjLd) * sizeof(float) : 0) +
           (size_t)mt * 32 * 9 * sizeof(float);         // rsv: the tile's sample points
}

template <int MT, class Ops>
static void launch_ddf_rev_t(const DdfArgs &a, int grid, hipStream_t s)
{
    // tanhExp: y' round trip as values; ReLU / LeakyReLU: as mask bits (the kernel's header)
    static bool once = (set_lds((const void *)ddf_rev_kernel<MT, Ops, false>, rev_lds_bytes<Ops>(MT, kMaxLayers)),
                        set_lds((const void *)ddf_rev_kernel<MT, Ops, true>, rev_lds_bytes<Ops>(MT, kMaxLayers)), true);
    (void)once;
    const size_t lds = rev_lds_bytes<Ops>(MT, a.n_layers);
    if (a.activation == 2) hipLaunchKernelGGL((ddf_rev_kernel<MT, Ops, false>), dim3(grid), dim3(kThreads), lds, s, a);
    else hipLaunchKernelGGL((ddf_rev_kernel<MT, Ops, true>), dim3(grid), dim3(kThreads), lds, s, a);
}

template <int WID>
static void launch_ddf_rev_w(const DdfArgs &a, int grid, hipStream_t s)
{
    constexpr int MT = WID <= 256 ? 2 : 1;
    if (a.operands == 2) launch_ddf_rev_t<MT, OpsF16SplitT<WID>>(a, grid, s);
    else if (a.operands == 1) launch_ddf_rev_t<MT, NEDDF_BF16_REV_OPS<WID>>(a, grid, s);
    else launch_ddf_rev_t<MT, OpsF32T<WID>>(a, grid, s);
}

void launch_ddf_rev(const DdfArgs &a, int grid, hipStream_t s)
{
    if (a.width == 128) return launch_ddf_rev_w<128>(a, grid, s);
    if (a.width == 384) return launch_ddf_rev_w<384>(a, grid, s);
    if (a.width == 512) return launch_ddf_rev_w<512>(a, grid, s);
    launch_ddf_rev_w<256>(a, grid, s);
}

template <int WID>
static void launch_col_w(const ColArgs &a, int grid, bool rows4, hipStream_t s)
{
    constexpr int MT = WID <= 256 ? 2 : 1;
    if (a.operands == 2) launch_col_g<MT, 2, 4, OpsF16SplitT<WID>>(a, grid, rows4, s);
    else if (a.operands == 1) {
        if constexpr (WID == 256) launch_col_g<2, 3, 4, OpsBF16>(a, grid, rows4, s);
        else launch_col_g<MT, 2, 4, OpsBF16T<WID>>(a, grid, rows4, s);
    } else launch_col_g<MT, 2, 4, OpsF32T<WID>>(a, grid, rows4, s);
}

void launch_col(const ColArgs &a, int grid, bool rows4, hipStream_t s)
{
    if (a.width == 128) return launch_col_w<128>(a, grid, rows4, s);
    if (a.width == 384) return launch_col_w<384>(a, grid, rows4, s);
    if (a.width == 512) return launch_col_w<512>(a, grid, rows4, s);
    launch_col_w<256>(a, grid, rows4, s);
}

template <int MT, int WPS, class Ops>
static void launch_nerf_g(const NerfArgs &a, int grid, hipStream_t s)
{
    static bool once = (set_lds((const void *)nerf_kernel<MT, WPS, Ops>, lds_bytes<Ops>(MT)), true);
    (void)once;
    hipLaunchKernelGGL((nerf_kernel<MT, WPS, Ops>), dim3(grid), dim3(kThreads), lds_bytes<Ops>(MT), s, a);
}

template <int WID>
static void launch_nerf_w(const NerfArgs &a, int grid, hipStream_t s)
{
    constexpr int MT = WID <= 256 ? 2 : 1;
    if (a.operands == 2) launch_nerf_g<MT, 2, OpsF16SplitT<WID>>(a, grid, s);
    else if (a.operands == 1) launch_nerf_g<MT, 2, OpsBF16T<WID>>(a, grid, s);
    else launch_nerf_g<MT, 2, OpsF32T<WID>>(a, grid, s);
}

void launch_nerf(const NerfArgs &a, int grid, hipStream_t s)
{
    if (a.width == 128) return launch_nerf_w<128>(a, grid, s);
    if (a.width == 384) return launch_nerf_w<384>(a, grid, s);
    if (a.width == 512) return launch_nerf_w<512>(a, grid, s);
    launch_nerf_w<256>(a, grid, s);
}

}  // namespace neddf
