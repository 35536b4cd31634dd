// Do MFMA (bf16 32x32x16) and VALU work of two different waves on one SIMD overlap on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/overlap_ubench.hip -o /tmp/overlap_ubench && /tmp/overlap_ubench
// Modes: 0 = 8 waves, waves 0..3 MFMA loop, 4..7 idle; 1 = waves 4..7 VALU loop (transcendental mix), 0..3 idle;
//        2 = both at once (one MFMA wave + one VALU wave per SIMD); 3 = 4 waves, each with the VALU work placed between its
//        MFMAs in program order; 4 = the same on 8 waves with half the work each.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mfma_loop(int n, float *out, int lane)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    out[0] = s;
}

__device__ __forceinline__ void valu_loop(int n, float *out, int lane)
{
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (lane + i);
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {       // per element: 2 exp2 + 1 rcp + ~10 fma/mul -- the shape of the tanhExp epilogue
            float e = __builtin_amdgcn_exp2f(x[i] * 1.44f);
            float e2 = __builtin_amdgcn_exp2f((e + e) * 1.44f);
            float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e2 + 1.0f);
            float y = x[i] * t;
            float d = fmaf(-(x[i] * e), fmaf(t, t, -1.0f), t);
            x[i] = fmaf(y, 0.5f, d * 0.25f) * 0.9f + 0.01f;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[0] = s;
}

// one wave: MFMAs with the VALU work placed between them in program order (2 elements per 4 MFMAs)
__device__ __forceinline__ void mixed_loop(int n, float *out, int lane)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (lane + i);
    for (int it = 0; it < n; it += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            if (i & 1) {
                const int j = 2 * u + (i >> 1);
                float e = __builtin_amdgcn_exp2f(x[j] * 1.44f);
                float e2 = __builtin_amdgcn_exp2f((e + e) * 1.44f);
                float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e2 + 1.0f);
                float y = x[j] * t;
                float d = fmaf(-(x[j] * e), fmaf(t, t, -1.0f), t);
                x[j] = fmaf(y, 0.5f, d * 0.25f) * 0.9f + 0.01f;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        }
      }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[0] = s;
}

__global__ __launch_bounds__(512, 1) void k(int mode, int nm, int nv, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *o = out + (size_t)blockIdx.x * 512 + threadIdx.x;
    if (mode == 3) { if (wave < 4) mixed_loop(nm, o, lane); return; }                 // 4 waves, interleaved in program order
    if (mode == 4) { mixed_loop(nm / 2, o, lane); return; }                           // 8 waves, each half the work
    if (wave < 4) { if (mode == 0 || mode == 2) mfma_loop(nm, o, lane); }
    else { if (mode == 1 || mode == 2) valu_loop(nv, o, lane); }
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int nm = 4096, nv = 1024;     // 16384 MFMAs (524288 pipe cycles) ; 8192 elements
    for (int mode = 0; mode < 5; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, nm, nv, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, nm, nv, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d: %.3f ms\n", mode, ms);
    }
    return 0;
}
