// mfma16_ubench.hip -- ceiling of the dense-layer inner loop on the 16-bit matrix instructions (bf16 policy: one
// v_mfma_f32_32x32x16_bf16 per tile and k-step; split-fp16 policy: three v_mfma_f32_32x32x16_f16 on two operand planes),
// with the A fragments from LDS and the B fragments from global memory (L2-resident weights) as in tile_engine.h's
// dense_pipeline, at prefetch distances of 1..3 k-steps.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma16_ubench tools/mfma16_ubench.hip && tools/mfma16_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

struct PB {     // bf16 policy
    static constexpr int LD = 264, PLANES = 1, SUB = 1;
    struct frag { bf16x8 v; };
    static __device__ __forceinline__ frag lda(const u16 *p) { frag f; f.v = *(const bf16x8 *)p; return f; }
    static __device__ __forceinline__ f32x16 mfma(const frag &a, const frag &b, const f32x16 &c, int)
    { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0); }
};
struct PS {     // split-fp16 policy
    static constexpr int LD = 528, PLANES = 2, SUB = 3;
    struct frag { f16x8 h, m; };
    static __device__ __forceinline__ frag lda(const u16 *p) { frag f; f.h = *(const f16x8 *)p; f.m = *(const f16x8 *)(p + 264); return f; }
    static __device__ __forceinline__ f32x16 mfma(const frag &a, const frag &b, const f32x16 &c, int r)
    {
        if (r == 0) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.m, b.h, c, 0, 0, 0);
        if (r == 1) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.m, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, c, 0, 0, 0);
    }
};

// MODE bit 0: A from LDS, bit 1: B from global, bit 2: barrier per pass over the k range; D = prefetch distance (k-steps)
template <class P, int MT, int NT, int MODE, int WPS, int D>
__global__ __launch_bounds__(256, WPS) void k(const void *w, float *out, int iters, int ksteps)
{
    extern __shared__ __attribute__((aligned(16))) u16 act[];
    typedef typename P::frag frag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < MT * 32 * P::LD; i += 256) act[i] = (u16)(0x3c00 + (i % 7));
    __syncthreads();
    const u16 *act_lane = act + (lane & 31) * P::LD + 8 * (lane >> 5);
    const frag *wl = (const frag *)w + (size_t)wave * NT * ksteps * 64 + lane;
    f32x16 acc[MT][NT];
    for (int mt = 0; mt < MT; ++mt) for (int t = 0; t < NT; ++t) for (int q = 0; q < 16; ++q) acc[mt][t][q] = 0.f;
    frag a[D + 1][MT], b[D + 1][NT];
#pragma unroll
    for (int u = 0; u <= D; ++u) {
        for (int mt = 0; mt < MT; ++mt) a[u][mt] = P::lda(act_lane + mt * 32 * P::LD + 16 * u);
        for (int t = 0; t < NT; ++t) b[u][t] = wl[((size_t)t * ksteps + u) * 64];
    }
    for (int it = 0; it < iters; ++it) {
        for (int S = 0; S < ksteps; S += D + 1) {
#pragma unroll
            for (int u = 0; u <= D; ++u) {
                const int slot = (u + D) % (D + 1);
                int Sn = S + u + D;
                if (Sn >= ksteps) Sn -= ksteps;
                if (MODE & 1) for (int mt = 0; mt < MT; ++mt) a[slot][mt] = P::lda(act_lane + mt * 32 * P::LD + 16 * Sn);
                if (MODE & 2) for (int t = 0; t < NT; ++t) b[slot][t] = wl[((size_t)t * ksteps + Sn) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < P::SUB; ++r)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[mt][t] = P::mfma(a[u][mt], b[u][t], acc[mt][t], r);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE & 4) __syncthreads();
    }
    float s = 0.f;
    for (int mt = 0; mt < MT; ++mt) for (int t = 0; t < NT; ++t) for (int q = 0; q < 16; ++q) s += acc[mt][t][q];
    out[blockIdx.x * 256 + tid] = s;
}

// Shared weight fragments: ONE workgroup of 8 waves per CU, two row groups of 64 rows (waves 0-3 / 4-7); wave (g, q) owns output
// columns [64q, 64q+64) of its group's rows, so waves q and q+4 need the same B fragments.  They are fetched from global once per
// CU in chunks of CH k-steps (all 512 threads, 16 B each per k-step), parked in a two-deep LDS ring in fragment order and read by
// both row groups with ds_read_b128; one barrier per chunk.  bf16 policy.
template <int CH>
__global__ __launch_bounds__(512, 1) void kshared(const void *w, float *out, int iters, int ksteps)
{
    constexpr int MT = 2, NT = 2, LD = 264;
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    u16 *act = smem;                                  // [128][LD]
    bf16x8 *ring = (bf16x8 *)(smem + 128 * LD);       // [2][8 pairs][CH][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = wave >> 2, q = wave & 3;
    for (int i = tid; i < 128 * LD; i += 512) act[i] = (u16)(0x3c00 + (i % 7));
    const u16 *act_lane = act + (g * 64 + (lane & 31)) * LD + 8 * (lane >> 5);
    const bf16x8 *wg = (const bf16x8 *)w;             // global fragments [(pair * ksteps + S) * 64 + lane]
    const int pair_ld = wave;                         // this wave stages pair `wave` (8 waves <-> 8 (q, t) pairs)
    f32x16 acc[MT][NT];
    for (int mt = 0; mt < MT; ++mt) for (int t = 0; t < NT; ++t) for (int k = 0; k < 16; ++k) acc[mt][t][k] = 0.f;
    bf16x8 st[CH];
    auto fetch = [&](int S0) {
#pragma unroll
        for (int s = 0; s < CH; ++s) st[s] = wg[((size_t)pair_ld * ksteps + S0 + s) * 64 + lane];
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int s = 0; s < CH; ++s) ring[((buf * 8 + pair_ld) * CH + s) * 64 + lane] = st[s];
    };
    fetch(0);
    park(0);
    __syncthreads();
    const int chunks = ksteps / CH;
    for (int it = 0; it < iters; ++it) {
        for (int c = 0; c < chunks; ++c) {
            const int buf = c & 1;
            fetch(((c + 1) % chunks) * CH);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                bf16x8 a[MT], b[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = *(const bf16x8 *)(act_lane + mt * 32 * LD + 16 * ((c * CH + s) % 16));
#pragma unroll
                for (int t = 0; t < NT; ++t) b[t] = ring[((buf * 8 + q * NT + t) * CH + s) * 64 + lane];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], b[t], acc[mt][t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            park(buf ^ 1);
            __syncthreads();
        }
    }
    float sum = 0.f;
    for (int mt = 0; mt < MT; ++mt) for (int t = 0; t < NT; ++t) for (int k = 0; k < 16; ++k) sum += acc[mt][t][k];
    out[blockIdx.x * 512 + tid] = sum;
}

template <int CH>
void run_shared(const char *name, const void *w, float *out)
{
    const int ksteps = 48, iters = 300, grid = 256;
    size_t lds = (size_t)128 * 264 * sizeof(u16) + (size_t)2 * 8 * CH * 64 * 16;
    (void)hipFuncSetAttribute((const void *)kshared<CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kshared<CH>), dim3(grid), dim3(512), lds, 0, w, out, 10, ksteps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kshared<CH>), dim3(grid), dim3(512), lds, 0, w, out, iters, ksteps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double mfmas = (double)grid * 8 * iters * ksteps * 4.0;
    double tf = mfmas * (2.0 * 32 * 32 * 16) / ms / 1e9;
    printf("%-58s %8.2f ms  %7.1f TF of MFMA work = %4.1f %% of 2500\n", name, ms, tf, tf / 25.0);
    fflush(stdout);
}

template <class P, int MT, int NT, int MODE, int WPS, int D>
void run(const char *name, const void *w, float *out)
{
    const int ksteps = 48, iters = 300, grid = 256 * WPS;
    size_t lds = (size_t)MT * 32 * P::LD * sizeof(u16);
    (void)hipFuncSetAttribute((const void *)k<P, MT, NT, MODE, WPS, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<P, MT, NT, MODE, WPS, D>), dim3(grid), dim3(256), lds, 0, w, out, 10, ksteps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<P, MT, NT, MODE, WPS, D>), dim3(grid), dim3(256), lds, 0, w, out, iters, ksteps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double mfmas = (double)grid * 4 * iters * ksteps * (double)(P::SUB * MT * NT);
    double tf = mfmas * (2.0 * 32 * 32 * 16) / ms / 1e9;
    printf("%-58s %8.2f ms  %7.1f TF of MFMA work = %4.1f %% of 2500\n", name, ms, tf, tf / 25.0);
    fflush(stdout);
}

int main()
{
    void *w; float *out;
    hipMalloc(&w, 64 << 20); hipMemset(w, 0, 64 << 20);
    hipMalloc(&out, 4 << 20);
    printf("bf16 policy (4 MFMAs per k-step and wave at MT2 NT2)\n");
    run<PB, 2, 2, 0, 2, 1>("bf16 MT2 2wg/CU mfma only", w, out);
    run<PB, 2, 2, 1, 2, 1>("bf16 MT2 2wg/CU +A          distance 1", w, out);
    run<PB, 2, 2, 2, 2, 1>("bf16 MT2 2wg/CU +B          distance 1", w, out);
    run<PB, 2, 2, 3, 2, 1>("bf16 MT2 2wg/CU +A+B        distance 1", w, out);
    run<PB, 2, 2, 3, 2, 2>("bf16 MT2 2wg/CU +A+B        distance 2", w, out);
    run<PB, 2, 2, 3, 2, 3>("bf16 MT2 2wg/CU +A+B        distance 3", w, out);
    run<PB, 2, 2, 7, 2, 3>("bf16 MT2 2wg/CU +A+B+barrier distance 3", w, out);
    run<PB, 2, 2, 3, 3, 1>("bf16 MT2 3wg/CU +A+B        distance 1", w, out);
    run<PB, 2, 2, 3, 3, 3>("bf16 MT2 3wg/CU +A+B        distance 3", w, out);
    run<PB, 4, 2, 3, 1, 1>("bf16 MT4 1wg/CU +A+B        distance 1", w, out);
    run<PB, 4, 2, 3, 1, 3>("bf16 MT4 1wg/CU +A+B        distance 3", w, out);
    run<PB, 4, 1, 3, 1, 3>("bf16 MT4 NT1 1wg/CU +A+B    distance 3", w, out);
    run<PB, 4, 2, 0, 2, 1>("bf16 MT4 2wg/CU mfma only", w, out);
    run<PB, 4, 2, 3, 2, 1>("bf16 MT4 2wg/CU +A+B        distance 1", w, out);
    run<PB, 4, 2, 3, 2, 2>("bf16 MT4 2wg/CU +A+B        distance 2", w, out);
    run<PB, 4, 2, 7, 2, 1>("bf16 MT4 2wg/CU +A+B+barrier distance 1", w, out);
    run<PB, 3, 2, 3, 2, 1>("bf16 MT3 2wg/CU +A+B        distance 1", w, out);
    run_shared<2>("bf16 8 waves/CU, B shared through LDS, chunks of 2 k-steps", w, out);
    run_shared<4>("bf16 8 waves/CU, B shared through LDS, chunks of 4 k-steps", w, out);
    printf("split-fp16 policy (12 MFMAs per k-step and wave at MT2 NT2)\n");
    run<PS, 2, 2, 0, 2, 1>("split MT2 2wg/CU mfma only", w, out);
    run<PS, 2, 2, 1, 2, 1>("split MT2 2wg/CU +A          distance 1", w, out);
    run<PS, 2, 2, 2, 2, 1>("split MT2 2wg/CU +B          distance 1", w, out);
    run<PS, 2, 2, 3, 2, 1>("split MT2 2wg/CU +A+B        distance 1", w, out);
    run<PS, 2, 2, 3, 2, 2>("split MT2 2wg/CU +A+B        distance 2", w, out);
    run<PS, 2, 2, 7, 2, 1>("split MT2 2wg/CU +A+B+barrier distance 1", w, out);
    run<PS, 4, 2, 3, 1, 1>("split MT4 1wg/CU +A+B        distance 1", w, out);
    run<PS, 4, 2, 3, 1, 2>("split MT4 1wg/CU +A+B        distance 2", w, out);
    return 0;
}
