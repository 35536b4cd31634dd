// mfma_ubench.hip -- what does the dense-layer inner loop cost beyond its MFMAs?
// Same block shape as field_kernels.hip's dense(): MT*NT accumulator tiles of
// v_mfma_f32_32x32x2_f32, 16*MT MFMAs per super-step, optionally with the A
// operand ds_read_b128s and the B operand global_load_dwordx4s of the real loop.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_ubench tools/mfma_ubench.hip && ./mfma_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int LD = 260;

template <int MT, int NT, int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const float *w, float *out, int iters, int ksteps)
{
    extern __shared__ __attribute__((aligned(16))) float act[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < MT * 32 * LD; i += 256) act[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    const float *act_lane = act + (lane & 31) * LD + 4 * (lane >> 5);
    const f32x4v *wl = (const f32x4v *)w + (size_t)wave * NT * ksteps * 64 + lane;
    f32x16 acc[MT][NT];
    for (int mt = 0; mt < MT; ++mt) for (int t = 0; t < NT; ++t) for (int q = 0; q < 16; ++q) acc[mt][t][q] = 0.f;
    f32x4v a0[MT], b0[NT], a1[MT], b1[NT];
    for (int mt = 0; mt < MT; ++mt) a0[mt] = a1[mt] = *(const f32x4v *)(act_lane + mt * 32 * LD);
    for (int t = 0; t < NT; ++t) b0[t] = b1[t] = wl[(size_t)t * ksteps * 64];
    for (int it = 0; it < iters; ++it) {
        for (int S = 0; S < ksteps; S += 2) {
            if (MODE & 1) for (int mt = 0; mt < MT; ++mt) a1[mt] = *(const f32x4v *)(act_lane + mt * 32 * LD + 8 * (S + 1));
            if (MODE & 2) for (int t = 0; t < NT; ++t) b1[t] = wl[((size_t)t * ksteps + S + 1) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mt][r], b0[t][r], acc[mt][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            int Sn = S + 2 < ksteps ? S + 2 : 0;
            if (MODE & 1) for (int mt = 0; mt < MT; ++mt) a0[mt] = *(const f32x4v *)(act_lane + mt * 32 * LD + 8 * Sn);
            if (MODE & 2) for (int t = 0; t < NT; ++t) b0[t] = wl[((size_t)t * ksteps + Sn) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mt][r], b1[t][r], acc[mt][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE & 4) __syncthreads();
    }
    float s = 0.f;
    for (int mt = 0; mt < MT; ++mt) for (int t = 0; t < NT; ++t) for (int q = 0; q < 16; ++q) s += acc[mt][t][q];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MT, int NT, int MODE, int WPS>
void run(const char *name, const float *w, float *out)
{
    const int ksteps = 32, iters = 400, grid = 256 * WPS;
    size_t lds = MT * 32 * LD * sizeof(float);
    (void)hipFuncSetAttribute((const void *)k<MT, NT, MODE, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MT, NT, MODE, WPS>), dim3(grid), dim3(256), lds, 0, w, out, 10, ksteps);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MT, NT, MODE, WPS>), dim3(grid), dim3(256), lds, 0, w, out, iters, ksteps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double flops = (double)grid * 4 * iters * ksteps * (4.0 * MT * NT) * (2.0 * 32 * 32 * 2);
    printf("%-52s %8.2f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}

int main()
{
    float *w, *out;
    hipMalloc(&w, 64 << 20); hipMemset(w, 0, 64 << 20);
    hipMalloc(&out, 1 << 20);
    run<4, 2, 0, 1>("MT4 1wg/CU  mfma only", w, out);
    run<4, 2, 1, 1>("MT4 1wg/CU  + ds_read_b128 A", w, out);
    run<4, 2, 2, 1>("MT4 1wg/CU  + global B", w, out);
    run<4, 2, 3, 1>("MT4 1wg/CU  + A + B", w, out);
    run<4, 2, 7, 1>("MT4 1wg/CU  + A + B + barrier per 32 super-steps", w, out);
    run<2, 2, 0, 2>("MT2 2wg/CU  mfma only", w, out);
    run<2, 2, 3, 2>("MT2 2wg/CU  + A + B", w, out);
    run<2, 2, 7, 2>("MT2 2wg/CU  + A + B + barrier", w, out);
    run<2, 2, 3, 1>("MT2 1wg/CU  + A + B", w, out);
    return 0;
}
