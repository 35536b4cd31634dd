// render_kernels.hip -- the HBM-bound stages around the field: ray generation,
// stratified / conical-frustum sampling, alpha compositing (one wavefront per
// ray, wave-level multiplicative scan in fp64 like torch-CPU's cumprod) and
// inverse-CDF importance resampling (one wavefront per ray: sequential-order
// L1 norm + fp64 cdf for bit-exact indices, binary search, bitonic sort in registers).
// Compiled with -ffp-contract=off: the elementwise stages follow the operation
// order of the reference's eager torch ops so that, given the same inputs,
// results are bit-identical to the oracle wherever only +,-,*,/,sqrt occur.
#include "kernels.h"
#include "device_math.h"
#include <math.h>

namespace neddf {

// ----------------------------------------------------------------------------
// Camera.create_rays camera.py:155-171 (+ :173-187, pinhole_calib.py:51-74)
template <typename T>
__global__ void raygen_kernel(const T *uv, int64_t n, CameraArg cam, float *ray_dir, float *ray_orig)
{
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const float fx = cam.calib[0], fy = cam.calib[1], cx = cam.calib[2], cy = cam.calib[3];
    float u = 0.5f + 1.0f * (float)uv[2 * b + 0];
    float v = 0.5f + 1.0f * (float)uv[2 * b + 1];
    float x = (1.0f / fx) * (u - cx);
    float y = (1.0f / fy) * (v - cy);
    float px = x, py = -y, pz = -1.0f;                  // rdf2rub = diag(1,-1,-1)
    float nrm = sqrtf(px * px + py * py + pz * pz);
    nrm = nrm < 1e-12f ? 1e-12f : nrm;                  // F.normalize eps
    px /= nrm; py /= nrm; pz /= nrm;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ray_dir[3 * b + i] = cam.R[3 * i + 0] * px + cam.R[3 * i + 1] * py + cam.R[3 * i + 2] * pz;
        ray_orig[3 * b + i] = cam.T[i];
    }
}

void launch_raygen(const void *uv, int uv_type, int64_t n, const CameraArg &cam, float *dir, float *orig, hipStream_t s)
{
    if (n <= 0) return;
    dim3 g((unsigned)((n + 255) / 256)), b(256);
    switch (uv_type) {
    case 0: hipLaunchKernelGGL(raygen_kernel<float>, g, b, 0, s, (const float *)uv, n, cam, dir, orig); break;
    case 1: hipLaunchKernelGGL(raygen_kernel<int64_t>, g, b, 0, s, (const int64_t *)uv, n, cam, dir, orig); break;
    case 2: hipLaunchKernelGGL(raygen_kernel<int32_t>, g, b, 0, s, (const int32_t *)uv, n, cam, dir, orig); break;
    default: hipLaunchKernelGGL(raygen_kernel<int16_t>, g, b, 0, s, (const int16_t *)uv, n, cam, dir, orig); break;
    }
}

// torch.linspace (float): symmetric evaluation around the midpoint
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i)
{
    if (steps == 1) return start;
    float step = (end - start) / (float)(steps - 1);
    return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

// stratified coarse distances nerf_render.py:131-140
__global__ void sample_coarse_kernel(const float *U, int64_t total, int S1, float near_, float far_, float step, float *dists)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int j = (int)(i % S1);
    dists[i] = linspace_at(near_, far_, S1, j) + U[i] * step;
}

void launch_sample_coarse(const float *U, int64_t n, int S1, float near_, float far_, float *dists, hipStream_t s)
{
    int64_t total = n * S1;
    if (total <= 0) return;
    float step = (float)(((double)far_ - (double)near_) / (double)(S1 - 1));
    hipLaunchKernelGGL(sample_coarse_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, U, total, S1, near_, far_, step, dists);
}

// Ray.get_sampling_cones ray.py:128-194 / get_sampling_points ray.py:88-126
template <bool CONE>
__global__ void sampling_kernel(const float *rd, const float *ro, const float *view, const float *dists, int64_t n, int S, float r2,
                                float *pos, float *dir, float *var)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * S) return;
    int64_t b = i / S;
    int j = (int)(i - b * S);
    const float *d = dists + b * S;
    float t_mu, t_var, r_var;
    sample_moments<CONE>(d, j, S, r2, t_mu, t_var, r_var);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float dd = rd[3 * b + k];
        sample_coord<CONE>(dd, ro[3 * b + k], t_mu, t_var, r_var, pos[3 * i + k], var[3 * i + k]);
        dir[3 * i + k] = view ? view[3 * b + k] : dd;      // NDC rays: the field sees the world-space viewing direction
    }
}

void launch_sampling(const float *rd, const float *ro, const float *view, const float *dists, int64_t n, int S, double radius,
                     float *pos, float *dir, float *var, hipStream_t s)
{
    int64_t total = n * S;
    if (total <= 0) return;
    dim3 g((unsigned)((total + 255) / 256)), b(256);
    if (radius >= 0.0)
        hipLaunchKernelGGL(sampling_kernel<true>, g, b, 0, s, rd, ro, view, dists, n, S, (float)(radius * radius), pos, dir, var);
    else
        hipLaunchKernelGGL(sampling_kernel<false>, g, b, 0, s, rd, ro, view, dists, n, S, 0.f, pos, dir, var);
}

// Normalised-device-coordinate rays for forward-facing captures (BASELINE.json configs[4]).  NOT in the reference
// (it has no NDC/LLFF code): this is the published construction of the original NeRF paper (Mildenhall et al. 2020,
// appendix C): shift the origin to the near plane z = -near, then map the frustum to the cube,
//   o' = (-fx/(W/2) ox/oz, -fy/(H/2) oy/oz, 1 + 2 near/oz),
//   d' = (-fx/(W/2) (dx/dz - ox/oz), -fy/(H/2) (dy/dz - oy/oz), -2 near/oz),
// so that o' + t' d', t' in [0, 1], sweeps the ray from the near plane to infinity.
__global__ void ndc_kernel(const float *rd, const float *ro, int64_t n, float sx, float sy, float near_, float *nd, float *no)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float dx = rd[3 * i], dy = rd[3 * i + 1], dz = rd[3 * i + 2];
    float ox = ro[3 * i], oy = ro[3 * i + 1], oz = ro[3 * i + 2];
    float t = -(near_ + oz) / dz;
    ox = ox + t * dx; oy = oy + t * dy; oz = oz + t * dz;
    float ozi = 1.0f / oz, dzi = 1.0f / dz;
    no[3 * i] = -sx * (ox * ozi);
    no[3 * i + 1] = -sy * (oy * ozi);
    no[3 * i + 2] = 1.0f + 2.0f * near_ * ozi;
    nd[3 * i] = -sx * (dx * dzi - ox * ozi);
    nd[3 * i + 1] = -sy * (dy * dzi - oy * ozi);
    nd[3 * i + 2] = -2.0f * near_ * ozi;
}

void launch_ndc(const float *rd, const float *ro, int64_t n, float width, float height, float fx, float fy, float near_, float *nd,
                float *no, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(ndc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rd, ro, n, fx / (0.5f * width), fy / (0.5f * height),
                       near_, nd, no);
}

// ----------------------------------------------------------------------------
// wave-level helpers (wave64)
__device__ __forceinline__ double wave_scan_mul(double v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double o = __shfl_up(v, off, 64);
        if (lane >= off) v *= o;
    }
    return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// integrate_volume_render base_neural_render.py:117-172.  One wavefront per
// ray; T_j = prod_{k<j}(1 - o_k + 1e-7) is a wave multiplicative scan carried in
// fp64 (torch-CPU cumprod accumulates in double and rounds each output, N2).
// col == NULL: weights only (the coarse pass of render_rays when nobody asked for
// its pixels: only the resampling weights are consumed); depth / color / trans may then be NULL too.
__global__ __launch_bounds__(256) void composite_kernel(const float *dists, const float *dens, const float *col, int64_t n,
                                                        int S, float max_dist, float *weight, float *depth, float *color,
                                                        float *trans, int *nan_flag)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= n) return;
    const float *d = dists + b * S, *r = dens + b * S, *c = col ? col + b * S * 3 : nullptr;
    double carry = 1.0;               // T entering this 64-sample chunk
    float sd = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
    bool bad = false;
    for (int base = 0; base < S - 1; base += 64) {
        int j = base + lane;
        bool on = j < S - 1;
        float dj = on ? d[j] : 0.f;
        float delta = on ? d[j + 1] - dj : 0.f;
        float o = on ? 1.0f - expf(-r[j] * delta) : 0.f;
        double aj = on ? (double)(1.0f - o + 1e-7f) : 1.0;
        double incl = wave_scan_mul(aj, lane) * carry;          // T after sample j
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = carry;
        float tprev = (float)excl;                              // t[:, j], rounded like cumprod's output
        float w = o * tprev;
        if (on) {
            if (w != w) bad = true;
            if (weight) weight[b * (S - 1) + j] = w;
            sd += w * dj;
            if (c) {
                s0 += w * c[3 * j + 0];
                s1 += w * c[3 * j + 1];
                s2 += w * c[3 * j + 2];
            }
        }
        carry = __shfl(incl, 63, 64);
    }
    sd = wave_sum(sd); s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    float tend = (float)carry;
    if (lane == 0) {
        if (depth) depth[b] = sd + tend * max_dist;             // black background, :163
        if (color) { color[3 * b + 0] = s0; color[3 * b + 1] = s1; color[3 * b + 2] = s2; }
        if (trans) trans[b] = tend;
    }
    if (nan_flag && __any(bad) && lane == 0) atomicOr(nan_flag, 1);
}

void launch_composite(const float *dists, const float *dens, const float *col, int64_t n, int S, float max_dist,
                      float *w, float *depth, float *color, float *trans, int *nan_flag, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, dists, dens, col, n, S, max_dist, w,
                       depth, color, trans, nan_flag);
}

// penalty line integral nerf_render.py:153-159
__global__ __launch_bounds__(256) void integrate_penalty_kernel(const float *dists, const float *pen, int64_t n, int S, float *out)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= n) return;
    float s = 0.f;
    for (int j = lane; j < S - 1; j += 64) s += (dists[b * S + j + 1] - dists[b * S + j]) * pen[b * S + j];
    s = wave_sum(s);
    if (lane == 0) out[b] = s;
}

void launch_integrate_penalty(const float *dists, const float *pen, int64_t n, int S, float *out, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(integrate_penalty_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, dists, pen, n, S, out);
}

// ----------------------------------------------------------------------------
// The ascending sort of sample_pdf's merged samples (torch.sort values, base_neural_render.py:97-100) as a bitonic network over 64 R values
// held in REGISTERS: element i = lane * R + r.  A compare-exchange at distance j < R is between two registers of a lane; at j >= R
// between the same register of lanes `lane` and `lane ^ (j / R)` (one cross-lane read per value, no LDS storage, no barrier).  The
// network and the exchange rule -- swap iff (x > y) == ascending, x the value at the lower index -- are those of the LDS loop it
// replaces (kept below for more than 512 values), so the output is the same bits, ties and signed zeros included.  That loop was
// 36 steps (256 values) x four rounds of two LDS reads, index arithmetic and two conditional LDS writes + a barrier: most of the
// kernel's instructions, and the kernel is bound by instruction issue (1 808 B per ray of traffic: 5 % of the HBM rate).
template <int R>
__device__ __forceinline__ void bitonic_in_registers(float *srt, int lane)
{
    constexpr int NP = 64 * R;
    float v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = srt[lane * R + r];
#pragma unroll
    for (int k = 2; k <= NP; k <<= 1) {
        // (i & k) == 0 for i = lane * R + r: the register index decides below R, the lane from R on
        const bool up_lane = ((lane * R) & k) == 0;
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < R) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r & j) continue;
                    const float x = v[r], y = v[r ^ j];
                    bool sw;
                    if (k < R) sw = (r & k) == 0 ? x > y : !(x > y);
                    else sw = (x > y) == up_lane;
                    v[r] = sw ? y : x;
                    v[r ^ j] = sw ? x : y;
                }
            } else {
                // all R partner values requested before the first is used; the value at the lower index of a pair is a, the other b,
                // on both lanes: one compare decides the same exchange on either side
                const int jl = j / R;
                const bool lower = (lane & jl) == 0;
                float y[R];
#pragma unroll
                for (int r = 0; r < R; ++r) y[r] = __shfl_xor(v[r], jl, 64);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float a = lower ? v[r] : y[r], b = lower ? y[r] : v[r];
                    const bool sw = (a > b) == up_lane;
                    v[r] = sw ? y[r] : v[r];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) srt[lane * R + r] = v[r];
}

// sample_pdf base_neural_render.py:27-115, one wavefront (= one workgroup) per ray.
// LDS: w[nw] | cdf[n] | pad to 16 bytes | sorted[max(npow2, n)]
// `group` rays share one NaN-fallback decision (the reference decides per sample_pdf call, i.e. per render_rays chunk).
// R = npow2 / 64 values per lane in the register sort; R = 0: the sort stays in LDS (npow2 > 512)
template <int R>
__global__ __launch_bounds__(64) void resample_kernel(const float *dists, float *weights, const float *U, int n, int nf, int cat,
                                                      int npow2, float *out, int64_t *ids, int *flag, int64_t group, int64_t offset)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int nw = n - 1, no = cat ? nf + n : nf;
    float *w = sm, *cdf = sm + nw, *srt = sm + ((nw + n + 3) & ~3);
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const float *d = dists + b * n;
    float *wrow = weights + b * nw;
    // sanitise in place (:52-55), then + 1e-2 (:58)
    for (int j = lane; j < nw; j += 64) {
        float x = wrow[j];
        if (x < 0.0f) x *= 0.0f;
        if (x != x) x = 0.0f;
        wrow[j] = x;
        w[j] = x + 1e-2f;
    }
    __syncthreads();
    if (!cat && nw >= 3) {             // :61-68 neighbour-max smoothing, computed from the un-smoothed values
        for (int j = lane; j < nw; j += 64) {
            float v = w[j];
            if (j >= 1 && j < nw - 1) v = 0.5f * (fmaxf(w[j + 1], w[j]) + fmaxf(w[j - 1], w[j]));
            srt[j] = v;
        }
        __syncthreads();
        for (int j = lane; j < nw; j += 64) w[j] = srt[j];
        __syncthreads();
    }
    // F.normalize(p=1): sequential fp32 sum of |w| (N3); every lane redoes it (LDS broadcast reads)
    float l1 = 0.f;
    for (int j = 0; j < nw; ++j) l1 += fabsf(w[j]);
    l1 = l1 < 1e-12f ? 1e-12f : l1;
    // cumsum in double, each output rounded to fp32 (N2), summed in index order.  The order is the result (double addition does not
    // associate), so the chain stays sequential: the quotients in parallel, then ONE lane adds them up -- n dependent additions per
    // ray (rounds 1-4: every lane re-summed its own prefixes, n^2 / 128 quotients and additions per lane: the same bits, but quadratic
    // in sample_coarse)
    for (int j = lane; j < nw; j += 64) srt[j] = w[j] / l1;
    __syncthreads();
    // ... unless the order cannot matter: with at most 64 quotients whose exponents lie within 10 of the first one's, every partial sum is
    // a multiple of 2^(emin - 23) below 2^(emax + 7) -- at most 51 bits -- so every double addition is EXACT in any order, and a wave-level
    // prefix sum (six steps) gives the bits of the sequential chain (64 x five dependent instructions on one lane: a fifth of the kernel's
    // instructions).  The sample weights of a render (w + 1e-2 with w in [0, 1]) always qualify; anything else takes the chain.
    bool exact_scan = false;
    if (nw <= 64) {
        const float q = lane < nw ? srt[lane] : 0.0f;
        const int e = (int)((__float_as_uint(q) >> 23) & 0xffu), e0 = __builtin_amdgcn_readfirstlane(e);
        const bool ok = lane >= nw || q == 0.0f || (e != 255 && e - e0 <= 10 && e0 - e <= 10);
        exact_scan = __all(ok);
        if (exact_scan) {
            double acc = (double)q;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) {
                const double t = __shfl_up(acc, dlt, 64);
                if (lane >= dlt) acc += t;
            }
            if (lane < nw) cdf[lane + 1] = (float)acc;
            if (lane == 0) cdf[0] = 0.0f;
        }
    }
    if (!exact_scan && lane == 0) {
        double acc = 0.0;
        cdf[0] = 0.0f;
        for (int k = 0; k < nw; ++k) {
            acc += (double)srt[k];
            cdf[k + 1] = (float)acc;
        }
    }
    __syncthreads();
    bool bad = false;
    int nsteps = 0;
    while ((1 << nsteps) < n + 1) ++nsteps;
    for (int s = lane; s < nf; s += 64) {
        float u = U[b * nf + s];
        int lo = 0, hi = n;                 // searchsorted(right=True): first index with cdf > u
        // the probes of `while (lo < hi)`, as a loop of the wave-uniform depth ceil(log2(n + 1)) without a divergent exit: a lane that
        // is done keeps lo == hi (its read of cdf[lo], at most one past the end and inside this workgroup's LDS, is not used)
        for (int it = 0; it < nsteps; ++it) {
            const int mid = (lo + hi) >> 1;
            const bool go = lo < hi, right = cdf[mid] > u;
            hi = go && right ? mid : hi;
            lo = go && !right ? mid + 1 : lo;
        }
        int id = lo;
        int below = id - 1 > 0 ? id - 1 : 0;
        int above = id < n - 1 ? id : n - 1;
        if (ids) ids[b * nf + s] = id;
        float denom = cdf[above] - cdf[below];
        if (denom < 1e-5f) denom = 1.0f;
        float t = (u - cdf[below]) / denom;
        float v = d[below] + t * (d[above] - d[below]);
        if (v != v) bad = true;
        srt[s] = v;
    }
    if (cat)
        for (int j = lane; j < n; j += 64) {
            float v = d[j];
            if (v != v) bad = true;
            srt[nf + j] = v;
        }
    for (int j = no + lane; j < npow2; j += 64) srt[j] = INFINITY;
    __syncthreads();
    // bitonic sort ascending (torch.sort values)
    if constexpr (R > 0) {
        bitonic_in_registers<R>(srt, lane);
        __syncthreads();
    } else
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < npow2; i += 64) {
                int p = i ^ j;
                if (p > i) {
                    float x = srt[i], y = srt[p];
                    bool up = (i & k) == 0;
                    if ((x > y) == up) { srt[i] = y; srt[p] = x; }
                }
            }
            __syncthreads();
        }
    for (int j = lane; j < no; j += 64) out[b * no + j] = srt[j];
    if (__any(bad) && lane == 0) atomicOr(flag + (b + offset) / group, 1);
}

// NaN fallback (:105-114): linspace(dists[0,0], dists[0,-1], no) for every ray of a group (= of a sample_pdf call of the
// reference) in which any sample came out NaN; row 0 is the group's first ray.  One wavefront per ray: it reads its group's flag and
// leaves (the regular case: one load per ray -- one thread per OUTPUT with a 64-bit division in front of the flag took 0.15 ms per
// 640 k rays, a ninth of the resampling); a flagged ray's lanes write its row.
__global__ __launch_bounds__(256) void resample_fallback_kernel(const float *dists, int n, int64_t n_rays, int no, float *out, const int *flag,
                                                                int64_t group, int64_t offset)
{
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_rays) return;
    const int64_t g = (b + offset) / group;
    if (!flag[g]) return;
    const int64_t r0 = g * group - offset;          // a leading partial group (offset > 0) starts before this batch
    const float *first = dists + (r0 > 0 ? r0 : 0) * n;
    const float lo = first[0], hi = first[n - 1];
    for (int j = threadIdx.x & 63; j < no; j += 64) out[b * no + j] = linspace_at(lo, hi, no, j);
}

void launch_resample(const float *dists, float *weights, const float *U, int64_t n_rays, int n, int nf, int cat,
                     float *out, int64_t *ids, int *flag, int64_t group, int64_t offset, hipStream_t s)
{
    if (n_rays <= 0) return;
    if (group <= 0) { group = n_rays; offset = 0; }
    const int64_t n_groups = (n_rays + offset + group - 1) / group;
    int no = cat ? nf + n : nf;
    int npow2 = 64;             // one value per lane at least (the padding is +inf)
    while (npow2 < no) npow2 <<= 1;
    size_t lds = sizeof(float) * (size_t)((((n - 1) + n + 3) & ~3) + (npow2 > n ? npow2 : n));
    (void)hipMemsetAsync(flag, 0, sizeof(int) * n_groups, s);
    auto kernel = npow2 == 64 ? resample_kernel<1> : npow2 == 128 ? resample_kernel<2> : npow2 == 256 ? resample_kernel<4> :
                  npow2 == 512 ? resample_kernel<8> : resample_kernel<0>;
    hipLaunchKernelGGL(kernel, dim3((unsigned)n_rays), dim3(64), lds, s, dists, weights, U, n, nf, cat, npow2, out, ids, flag, group, offset);
    hipLaunchKernelGGL(resample_fallback_kernel, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, s, dists, n, n_rays, no, out, flag, group, offset);
}

}  // namespace neddf
