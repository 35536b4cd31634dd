// Product phase of the sized next 16-bit kernel (docs/lab_notebook.md R4.11), in isolation: ONE workgroup of eight waves per CU on a
// 128-row bf16 activation tile in LDS; waves w and w + 4 own the same 64 output columns for rows 0-63 / 64-127; the weight fragments
// of a 256 x 256 layer come from L2 ONCE per workgroup, by LDS-DMA (global_load_lds_dwordx4: 1 KB = one packed fragment per
// wave-instruction) into a two-slab ring (4 k-steps = 32 KB per slab), and every wave reads A and B fragments with ds_read_b128.
// Reports cycles per layer against the matrix time (2 waves x 64 MFMAs x 32 cycles = 4 096 per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/ldsb_ubench.hip -o tools/bin/ldsb_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kRows = 128, kW = 256, kLd = kW + 8;            // halves per activation row (528 B: 4 dwords mod 64)
constexpr int kSlabFrags = 32;                                // 4 k-steps x (4 column slices x 2 n-tiles) fragments of 1 KB
constexpr int kSlabBytes = kSlabFrags * 1024;

__device__ __forceinline__ void glds16(const void *g, void *lds)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

// MODE 0: B via the LDS ring.  MODE 1: B straight L2 -> VGPR per wave (today's engine, eight waves): the weight stream is NOT shared.
template <int MODE>
__global__ __launch_bounds__(512, 1) void product_kernel(const u32x4 *wfrag, float *out, long long *cyc, int layers)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short *act = (unsigned short *)smem;
    unsigned char *ring = smem + kRows * kLd * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cs = wave & 3, half = wave >> 2;
    for (int i = tid; i < kRows * kLd; i += 512) act[i] = (unsigned short)(0x3c00 + (i & 63));
    __syncthreads();
    const unsigned short *ap = act + (size_t)(half * 64 + (lane & 31)) * kLd + 8 * (lane >> 5);
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
    // fragment order of a layer: [k-step 16][column slice 4][n-tile 2] x 1 KB  (the ring holds 4 k-steps)
    auto issue = [&](int layer, int g, int buf) {          // this wave's 4 of the slab's 32 fragments
        const u32x4 *src = wfrag + ((size_t)(layer & 1) * 16 * 8 + (size_t)g * 32) * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = wave * 4 + q;
            glds16(src + (size_t)f * 64 + lane, ring + buf * kSlabBytes + f * 1024);
        }
    };
    const long long t0 = clock64();
    if (MODE == 0) { issue(0, 0, 0); __builtin_amdgcn_s_waitcnt(0x0f70); __builtin_amdgcn_s_barrier(); }
    for (int layer = 0; layer < layers; ++layer) {
        for (int g = 0; g < 4; ++g) {
            if (MODE == 0) {
                const int ng = g + 1 < 4 ? g + 1 : 0, nl = g + 1 < 4 ? layer : layer + 1;
                issue(nl, ng, (g + 1) & 1);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) a[m] = *(const bf16x8 *)(ap + (size_t)m * 32 * kLd + (g * 4 + ks) * 16);
                if (MODE == 0) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) b[t] = *(const bf16x8 *)(ring + (g & 1) * kSlabBytes + ((ks * 4 + cs) * 2 + t) * 1024 + lane * 16);
                } else {
                    const u32x4 *src = wfrag + ((size_t)(layer & 1) * 16 * 8 + (size_t)(g * 4 + ks) * 8 + cs * 2) * 64 + lane;
#pragma unroll
                    for (int t = 0; t < 2; ++t) { u32x4 v = src[t * 64]; b[t] = __builtin_bit_cast(bf16x8, v); }
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[t], acc[m][t], 0, 0, 0);
            }
            if (MODE == 0) {
                __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): this wave's DMA pieces of the next slab have landed
                __builtin_amdgcn_s_barrier();             // ... and every wave is done with the slab it just read
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[m][t][i];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// the same with a ring of NBUF slabs of SKS k-steps each, the DMA running NBUF - 1 slabs ahead, a counted vmcnt and a raw barrier per slab
template <int SKS, int NBUF>
__global__ __launch_bounds__(512, 1) void ring_kernel(const u32x4 *wfrag, float *out, long long *cyc, int layers)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short *act = (unsigned short *)smem;
    unsigned char *ring = smem + kRows * kLd * 2;
    constexpr int SLAB = SKS * 8 * 1024, SPL = 16 / SKS;          // bytes per slab, slabs per layer
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cs = wave & 3, half = wave >> 2;
    for (int i = tid; i < kRows * kLd; i += 512) act[i] = (unsigned short)(0x3c00 + (i & 63));
    __syncthreads();
    const unsigned short *ap = act + (size_t)(half * 64 + (lane & 31)) * kLd + 8 * (lane >> 5);
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
    auto issue = [&](int slab) {           // slab index over all layers; this wave's SKS of its SKS * 8 fragments
        const int layer = slab / SPL, g = slab - layer * SPL, buf = slab % NBUF;
        const u32x4 *src = wfrag + ((size_t)(layer & 1) * 16 * 8 + (size_t)g * SKS * 8) * 64;
#pragma unroll
        for (int q = 0; q < SKS; ++q) {
            const int f = wave * SKS + q;
            glds16(src + (size_t)f * 64 + lane, ring + buf * SLAB + f * 1024);
        }
    };
    const long long t0 = clock64();
#pragma unroll
    for (int s0 = 0; s0 < NBUF - 1; ++s0) issue(s0);
    if (NBUF == 2) __builtin_amdgcn_s_waitcnt(0x0f70);
    else __builtin_amdgcn_s_waitcnt(0x0f70 | (SKS * (NBUF - 2)));
    __builtin_amdgcn_s_barrier();
    const int nslabs = layers * SPL;
    for (int slab = 0; slab < nslabs; ++slab) {
        issue(slab + NBUF - 1);
        const int g = slab % SPL;
        const unsigned char *bb = ring + (slab % NBUF) * SLAB;
#pragma unroll
        for (int ks = 0; ks < SKS; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m] = *(const bf16x8 *)(ap + (size_t)m * 32 * kLd + (g * SKS + ks) * 16);
#pragma unroll
            for (int t = 0; t < 2; ++t) b[t] = *(const bf16x8 *)(bb + ((ks * 4 + cs) * 2 + t) * 1024 + lane * 16);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[t], acc[m][t], 0, 0, 0);
        }
        // the slab read next must have landed: everything but the NBUF - 2 newest slabs' pieces of this wave
        __builtin_amdgcn_s_waitcnt(0x0f70 | (SKS * (NBUF - 2)));
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[m][t][i];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// the ring kernel with the fragments of k-step ks + 1 read from LDS before the MFMAs of k-step ks (hand software pipeline inside a slab)
template <int SKS, int NBUF>
__global__ __launch_bounds__(512, 1) void ring_kernel_sp(const u32x4 *wfrag, float *out, long long *cyc, int layers)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short *act = (unsigned short *)smem;
    unsigned char *ring = smem + kRows * kLd * 2;
    constexpr int SLAB = SKS * 8 * 1024, SPL = 16 / SKS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cs = wave & 3, half = wave >> 2;
    for (int i = tid; i < kRows * kLd; i += 512) act[i] = (unsigned short)(0x3c00 + (i & 63));
    __syncthreads();
    const unsigned short *ap = act + (size_t)(half * 64 + (lane & 31)) * kLd + 8 * (lane >> 5);
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
    auto issue = [&](int slab) {
        const int layer = slab / SPL, g = slab - layer * SPL, buf = slab % NBUF;
        const u32x4 *src = wfrag + ((size_t)(layer & 1) * 16 * 8 + (size_t)g * SKS * 8) * 64;
#pragma unroll
        for (int q = 0; q < SKS; ++q) {
            const int f = wave * SKS + q;
            glds16(src + (size_t)f * 64 + lane, ring + buf * SLAB + f * 1024);
        }
    };
    const long long t0 = clock64();
#pragma unroll
    for (int s0 = 0; s0 < NBUF - 1; ++s0) issue(s0);
    __builtin_amdgcn_s_waitcnt(0x0f70 | (SKS * (NBUF - 2)));
    __builtin_amdgcn_s_barrier();
    const int nslabs = layers * SPL;
    for (int slab = 0; slab < nslabs; ++slab) {
        issue(slab + NBUF - 1);
        const int g = slab % SPL;
        const unsigned char *bb = ring + (slab % NBUF) * SLAB;
        bf16x8 a[2][2], b[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) a[0][m] = *(const bf16x8 *)(ap + (size_t)m * 32 * kLd + (g * SKS) * 16);
#pragma unroll
        for (int t = 0; t < 2; ++t) b[0][t] = *(const bf16x8 *)(bb + ((0 * 4 + cs) * 2 + t) * 1024 + lane * 16);
#pragma unroll
        for (int ks = 0; ks < SKS; ++ks) {
            if (ks + 1 < SKS) {
#pragma unroll
                for (int m = 0; m < 2; ++m) a[(ks + 1) & 1][m] = *(const bf16x8 *)(ap + (size_t)m * 32 * kLd + (g * SKS + ks + 1) * 16);
#pragma unroll
                for (int t = 0; t < 2; ++t) b[(ks + 1) & 1][t] = *(const bf16x8 *)(bb + (((ks + 1) * 4 + cs) * 2 + t) * 1024 + lane * 16);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][m], b[ks & 1][t], acc[m][t], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70 | (SKS * (NBUF - 2)));
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[m][t][i];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// ... and with the FIRST fragments of the next slab read before the barrier that ends this one (ring of four slabs: a slab is waited for
// one barrier earlier than it is consumed, so its data is visible a whole slab ahead)
template <int SKS>
__global__ __launch_bounds__(512, 1) void ring_kernel_xb(const u32x4 *wfrag, float *out, long long *cyc, int layers)
{
    constexpr int NBUF = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short *act = (unsigned short *)smem;
    unsigned char *ring = smem + kRows * kLd * 2;
    constexpr int SLAB = SKS * 8 * 1024, SPL = 16 / SKS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cs = wave & 3, half = wave >> 2;
    for (int i = tid; i < kRows * kLd; i += 512) act[i] = (unsigned short)(0x3c00 + (i & 63));
    __syncthreads();
    const unsigned short *ap = act + (size_t)(half * 64 + (lane & 31)) * kLd + 8 * (lane >> 5);
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
    auto issue = [&](int slab) {
        const int layer = slab / SPL, g = slab - layer * SPL, buf = slab % NBUF;
        const u32x4 *src = wfrag + ((size_t)(layer & 1) * 16 * 8 + (size_t)g * SKS * 8) * 64;
#pragma unroll
        for (int q = 0; q < SKS; ++q) {
            const int f = wave * SKS + q;
            glds16(src + (size_t)f * 64 + lane, ring + buf * SLAB + f * 1024);
        }
    };
    auto read = [&](int slab, int ks, bf16x8 (&a)[2], bf16x8 (&b)[2]) {
        const int g = slab % SPL;
        const unsigned char *bb = ring + (slab % NBUF) * SLAB;
#pragma unroll
        for (int m = 0; m < 2; ++m) a[m] = *(const bf16x8 *)(ap + (size_t)m * 32 * kLd + (g * SKS + ks) * 16);
#pragma unroll
        for (int t = 0; t < 2; ++t) b[t] = *(const bf16x8 *)(bb + ((ks * 4 + cs) * 2 + t) * 1024 + lane * 16);
    };
    const long long t0 = clock64();
    issue(0); issue(1); issue(2);
    __builtin_amdgcn_s_waitcnt(0x0f70 | SKS);          // slabs 0 and 1 landed
    __builtin_amdgcn_s_barrier();
    bf16x8 a[2][2], b[2][2];
    read(0, 0, a[0], b[0]);
    const int nslabs = layers * SPL;
    for (int slab = 0; slab < nslabs; ++slab) {
        issue(slab + 3);
#pragma unroll
        for (int ks = 0; ks < SKS; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;       // SKS is even: every slab starts in set 0
            if (ks + 1 < SKS) read(slab, ks + 1, a[nxt], b[nxt]);
            else read(slab + 1, 0, a[nxt], b[nxt]);     // visible since the barrier that ended slab - 1
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][m], b[cur][t], acc[m][t], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70 | SKS);      // everything but the slab issued in this iteration has landed: slab + 2 is complete
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[m][t][i];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SKS>
static void run_ring_xb(const u32x4 *w, float *out, long long *cyc, int cus, int layers)
{
    const size_t lds = (size_t)kRows * kLd * 2 + (size_t)4 * SKS * 8 * 1024;
    (void)hipFuncSetAttribute((const void *)ring_kernel_xb<SKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e9f;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((ring_kernel_xb<SKS>), dim3(cus), dim3(512), lds, 0, w, out, cyc, layers);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 2.0 * 128 * 256 * 256 * layers * cus;
    printf("ring of 4 + prefetch across the barrier: %d k-steps per slab (%3zu KB LDS)  %8.3f ms  %6.0f cycles per layer  %7.1f TF  (%.2f of 2.5 PF)\n",
           SKS, lds / 1024, best, best * 1e-3 * 2.4e9 / layers, flop / (best * 1e-3) * 1e-12, flop / (best * 1e-3) / 2.5e15);
}

template <int SKS, int NBUF>
static void run_ring_sp(const u32x4 *w, float *out, long long *cyc, int cus, int layers)
{
    const size_t lds = (size_t)kRows * kLd * 2 + (size_t)NBUF * SKS * 8 * 1024;
    (void)hipFuncSetAttribute((const void *)ring_kernel_sp<SKS, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e9f;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((ring_kernel_sp<SKS, NBUF>), dim3(cus), dim3(512), lds, 0, w, out, cyc, layers);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 2.0 * 128 * 256 * 256 * layers * cus;
    printf("ring + fragment prefetch: %d k-steps per slab x %d slabs  %8.3f ms  %6.0f cycles per layer  %7.1f TF  (%.2f of 2.5 PF)\n", SKS, NBUF,
           best, best * 1e-3 * 2.4e9 / layers, flop / (best * 1e-3) * 1e-12, flop / (best * 1e-3) / 2.5e15);
}

template <int SKS, int NBUF>
static void run_ring(const u32x4 *w, float *out, long long *cyc, int cus, int layers)
{
    const size_t lds = (size_t)kRows * kLd * 2 + (size_t)NBUF * SKS * 8 * 1024;
    (void)hipFuncSetAttribute((const void *)ring_kernel<SKS, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e9f;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((ring_kernel<SKS, NBUF>), dim3(cus), dim3(512), lds, 0, w, out, cyc, layers);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 2.0 * 128 * 256 * 256 * layers * cus;
    printf("ring: %d k-steps per slab x %d slabs (%3zu KB LDS)  %8.3f ms  %6.0f cycles per layer at 2.4 GHz  %7.1f TF  (%.2f of 2.5 PF)\n", SKS, NBUF,
           lds / 1024, best, best * 1e-3 * 2.4e9 / layers, flop / (best * 1e-3) * 1e-12, flop / (best * 1e-3) / 2.5e15);
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, layers = 200;
    u32x4 *w; float *out; long long *cyc;
    (void)hipMalloc(&w, 2 * 128 * 1024);
    (void)hipMalloc(&out, (size_t)cus * 512 * sizeof(float));
    (void)hipMalloc(&cyc, cus * sizeof(long long));
    (void)hipMemset(w, 0x3c, 2 * 128 * 1024);
    const size_t lds = (size_t)kRows * kLd * 2 + 2 * kSlabBytes;
    (void)hipFuncSetAttribute((const void *)product_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)product_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    long long *h = (long long *)malloc(cus * sizeof(long long));
    printf("# %d CUs; eight waves per CU on a 128-row tile, %d layers of 256 x 256 bf16; matrix time 4096 cycles per layer\n", cus, layers);
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(product_kernel<0>, dim3(cus), dim3(512), lds, 0, w, out, cyc, layers);
            else hipLaunchKernelGGL(product_kernel<1>, dim3(cus), dim3(512), lds, 0, w, out, cyc, layers);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        (void)hipMemcpy(h, cyc, cus * sizeof(long long), hipMemcpyDeviceToHost);
        double mean = 0; for (int i = 0; i < cus; ++i) mean += (double)h[i]; mean /= cus;
        const double flop = 2.0 * 128 * 256 * 256 * layers * cus;
        printf("%-34s %8.3f ms  %8.0f shader-clock ticks per layer  %7.1f TF  (%.2f of 2.5 PF)\n",
               mode == 0 ? "B via LDS-DMA ring (shared)" : "B L2 -> VGPR per wave (today)", best, mean / layers, flop / (best * 1e-3) * 1e-12,
               flop / (best * 1e-3) / 2.5e15);
    }
    run_ring<4, 2>(w, out, cyc, cus, layers);
    run_ring<2, 2>(w, out, cyc, cus, layers);
    run_ring<2, 3>(w, out, cyc, cus, layers);
    run_ring<2, 4>(w, out, cyc, cus, layers);
    run_ring<1, 4>(w, out, cyc, cus, layers);
    run_ring<1, 8>(w, out, cyc, cus, layers);
    run_ring_sp<4, 2>(w, out, cyc, cus, layers);
    run_ring_sp<2, 3>(w, out, cyc, cus, layers);
    run_ring_xb<2>(w, out, cyc, cus, layers);
    return 0;
}
