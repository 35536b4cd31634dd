// valu_rate_ubench.hip -- issue cost of the vector instructions the activation epilogues are made of, on gfx950: cycles per wave64 instruction
// with one and with two waves per SIMD (independent instruction streams, 16 registers in rotation, s_memtime around 4096 instructions).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_rate_ubench tools/valu_rate_ubench.hip && tools/bin/valu_rate_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int WHICH>
__global__ void k(unsigned long long *out, float seed)
{
    float r[16];
    f32x2 p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i] = seed + 0.001f * (float)(i + threadIdx.x); p[i] = (f32x2){ r[i], r[i] + 1.0f }; }
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 256; ++it) {
#define EXP32(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#define EXP16(i) asm volatile("v_exp_f16 %0, %0" : "+v"(r[i]));
#define RCP32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define RCP16(i) asm volatile("v_rcp_f16 %0, %0" : "+v"(r[i]));
#define FMA32(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
#define PKMUL16(i) asm volatile("v_pk_mul_f16 %0, %0, %0" : "+v"(r[i]));
#define PKFMA16(i) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(r[i]));
#define CVTBF(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(r[i]));
#define MIX(i) asm volatile("v_fma_mix_f32 %0, %0, %0, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r[i]));
#define MED3(i) asm volatile("v_med3_f32 %0, %0, %0, %0" : "+v"(r[i]));
#define LOG32(i) asm volatile("v_log_f32 %0, %0" : "+v"(r[i]));
#define SIN32(i) asm volatile("v_sin_f32 %0, %0" : "+v"(r[i]));
#define PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
#define PKMUL32(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
#define PKADD32(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
        if (WHICH == 12) { REP16(PKFMA32) }
        if (WHICH == 13) { REP16(PKMUL32) }
        if (WHICH == 14) { REP16(PKADD32) }
        if (WHICH == 0) { REP16(EXP32) }
        if (WHICH == 1) { REP16(EXP16) }
        if (WHICH == 2) { REP16(RCP32) }
        if (WHICH == 3) { REP16(RCP16) }
        if (WHICH == 4) { REP16(FMA32) }
        if (WHICH == 5) { REP16(PKMUL16) }
        if (WHICH == 6) { REP16(PKFMA16) }
        if (WHICH == 7) { REP16(CVTBF) }
        if (WHICH == 8) { REP16(MIX) }
        if (WHICH == 9) { REP16(MED3) }
        if (WHICH == 10) { REP16(LOG32) }
        if (WHICH == 11) { REP16(SIN32) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + p[i][0] + p[i][1];
    if (s == 1234.5f) out[1023] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[WHICH] = t1 - t0;
}

template <int WHICH>
static void run(const char *name, unsigned long long *d, int waves_per_simd)
{
    // one workgroup per CU slot: 256 threads = one wave per SIMD; 512 threads = two
    hipLaunchKernelGGL(k<WHICH>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, 0.5f);
    CHK(hipDeviceSynchronize());
    unsigned long long h[16];
    CHK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-22s %d wave(s) per SIMD: %6.2f cycles per instruction per wave  (%6.2f aggregate per SIMD)\n", name, waves_per_simd,
           (double)h[WHICH] / 4096.0, (double)h[WHICH] / 4096.0 / waves_per_simd);
}

int main()
{
    unsigned long long *d;
    CHK(hipMalloc((void **)&d, 1024 * sizeof(unsigned long long)));
    for (int w = 1; w <= 2; ++w) {
        run<0>("v_exp_f32", d, w); run<1>("v_exp_f16", d, w); run<2>("v_rcp_f32", d, w); run<3>("v_rcp_f16", d, w);
        run<10>("v_log_f32", d, w); run<11>("v_sin_f32", d, w);
        run<4>("v_fma_f32", d, w); run<9>("v_med3_f32", d, w); run<5>("v_pk_mul_f16", d, w); run<6>("v_pk_fma_f16", d, w);
        run<7>("v_cvt_pk_bf16_f32", d, w); run<8>("v_fma_mix_f32", d, w);
        run<12>("v_pk_fma_f32", d, w); run<13>("v_pk_mul_f32", d, w); run<14>("v_pk_add_f32", d, w);
    }
    return 0;
}
