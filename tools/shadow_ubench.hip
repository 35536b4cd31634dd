// How much VALU / LDS-store work hides in the shadow of the fp32 (and bf16) MFMA stream on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/shadow_ubench.hip -o /tmp/shadow_ubench && /tmp/shadow_ubench
// One workgroup of 4 waves per CU (one wave per SIMD) or 8 waves (two per SIMD).  Every variant issues the same MFMA
// stream (4 independent accumulator tiles, NMFMA instructions per wave); variants add K filler instructions per MFMA
//   same-wave:  fillers sit between the wave's own MFMAs in program order
//   partner:    waves 4..7 (second wave of each SIMD) run ONLY the fillers, waves 0..3 only the MFMAs
// Filler kinds: 0 v_fma_f32, 1 tanhExp-shaped mix (2 v_exp + 1 v_rcp + fma/mul), 2 ds_write_b32.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void filler(float (&x)[8], int j, float *lds, int lane)
{
    if (KIND == 0) {
        x[j & 7] = fmaf(x[j & 7], 1.0001f, 0.5f);
    } else if (KIND == 1) {      // one instruction of the tanhExp sequence per call, cycling through a 12-instruction pattern
        float &v = x[j & 7];
        switch (j % 12) {
        case 0: v = __builtin_amdgcn_exp2f(v * 1.44f); break;
        case 5: v = __builtin_amdgcn_exp2f(v + v); break;
        case 8: v = __builtin_amdgcn_rcpf(v + 1.0f); break;
        default: v = fmaf(v, 0.999f, 0.001f); break;
        }
    } else {
        lds[(j & 63) * 64 + lane] = x[j & 7];
    }
}

template <bool BF16, int K, int KIND, bool PARTNER>
__global__ __launch_bounds__(512, 1) void k(int nmfma, float *out)
{
    __shared__ float lds[64 * 64 * 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (lane + i);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    bf16x8 a16, b16;
    for (int i = 0; i < 8; ++i) { a16[i] = (__bf16)(float)(lane + i); b16[i] = (__bf16)(float)(lane - i); }
    float a32 = 1.0f + lane, b32 = 2.0f - lane;
    float *l = lds + (wave >> 2) * 4096;
    if (PARTNER && wave >= 4) {
        for (int it = 0; it < nmfma / 4; ++it) {
#pragma unroll
            for (int u = 0; u < 4 * K; ++u) filler<KIND>(x, u, l, lane);
        }
    } else if (wave < 4) {
        for (int it = 0; it < nmfma / 4; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (BF16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16, b16, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a32, b32, acc[i], 0, 0, 0);
                if (!PARTNER) {
#pragma unroll
                    for (int u = 0; u < K; ++u) filler<KIND>(x, i * K + u, l, lane);
                    if (K > 0) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(KIND == 2 ? 0x200 : 0x002, K, 0);          // K DS-write / VALU
                    }
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s + l[lane];
}

template <bool BF16, int K, int KIND, bool PARTNER>
static float run(int nmfma, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BF16, K, KIND, PARTNER>), dim3(256), dim3(512), 0, 0, nmfma, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BF16, K, KIND, PARTNER>), dim3(256), dim3(512), 0, 0, nmfma, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <bool BF16, int KIND>
static void sweep(const char *name, int nmfma, float *out)
{
    float base = run<BF16, 0, 0, false>(nmfma, out);
    printf("%s, filler %s: MFMA only %.3f ms | same wave K=1,2,4,8,12: %.3f %.3f %.3f %.3f %.3f | partner wave K=1,2,4,8: %.3f %.3f %.3f %.3f\n", name,
           KIND == 0 ? "v_fma" : KIND == 1 ? "tanhExp mix" : "ds_write_b32", base, run<BF16, 1, KIND, false>(nmfma, out),
           run<BF16, 2, KIND, false>(nmfma, out), run<BF16, 4, KIND, false>(nmfma, out), run<BF16, 8, KIND, false>(nmfma, out),
           run<BF16, 12, KIND, false>(nmfma, out), run<BF16, 1, KIND, true>(nmfma, out), run<BF16, 2, KIND, true>(nmfma, out),
           run<BF16, 4, KIND, true>(nmfma, out), run<BF16, 8, KIND, true>(nmfma, out));
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    sweep<false, 0>("fp32 32x32x2 (64 cyc)", 8192, out);
    sweep<false, 1>("fp32 32x32x2 (64 cyc)", 8192, out);
    sweep<false, 2>("fp32 32x32x2 (64 cyc)", 8192, out);
    sweep<true, 0>("bf16 32x32x16 (32 cyc)", 16384, out);
    sweep<true, 1>("bf16 32x32x16 (32 cyc)", 16384, out);
    sweep<true, 2>("bf16 32x32x16 (32 cyc)", 16384, out);
    return 0;
}
