// chain_ubench.hip -- ceiling of a REGISTER-RESIDENT layer chain on gfx950 (a candidate structure for the distance kernel).
//
// Today's tile engine keeps a 64-point tile's activations in LDS: every layer is product -> barrier -> epilogue (LDS write) -> barrier,
// a wave owns 64 of the 256 output features.  The alternative measured here: a wave owns 32 POINTS and ALL 256 features, and the MFMA
// operands are swapped (A = weights, B = activations), so that the accumulator layout of layer l -- lane = point, register q of tile
// kt = feature 32 kt + 8 (q >> 2) + 4 h + (q & 3) -- IS the B-operand layout of layer l + 1 once the weights' k order is permuted to
// match (a fixed permutation, done when the weights are packed): no LDS, no barrier, no transposition between layers.  Costs: the
// wave streams the whole 256 x 256 weight matrix per layer for its 32 points (twice today's L2 -> VGPR traffic per point), and
// 128 accumulators + 128 operand registers leave room for ONE wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o chain_ubench tools/chain_ubench.hip && ./chain_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int KS = 128;         // k-steps of a 256-wide layer (32x32x2: two k per step)
constexpr int MT = 8;           // output-feature tiles of 32

// weights of one layer: [KS][2][64 lanes][4]: a lane's dwordx4 = the A fragments of four output tiles at one k-step
template <int D, int EPI>
__global__ __launch_bounds__(256, 1) void chain(const float *w, float *out, int layers, int wlayers)
{
    const int lane = threadIdx.x & 63;
    f32x16 acc[MT], act[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[t][q] = 0.f; act[t][q] = 0.001f * (float)((lane + q + t) & 15); }
    for (int l = 0; l < layers; ++l) {
        const f32x4v *wl = (const f32x4v *)w + (size_t)(l % wlayers) * KS * 2 * 64 + lane;
        f32x4v ring[D][2];
#pragma unroll
        for (int d = 0; d < D; ++d) { ring[d][0] = wl[(d * 2 + 0) * 64]; ring[d][1] = wl[(d * 2 + 1) * 64]; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4v a0 = ring[s % D][0], a1 = ring[s % D][1];
            if (s + D < KS) { ring[s % D][0] = wl[((s + D) * 2 + 0) * 64]; ring[s % D][1] = wl[((s + D) * 2 + 1) * 64]; }
            __builtin_amdgcn_sched_barrier(0);      // (without it hipcc sinks the loads of the cheap-epilogue variants to their uses: vmcnt(0) per step)
            const float b = act[s >> 4][s & 15];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[4 + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b, acc[4 + t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: the accumulators become the next layer's operands (EPI 1: a multiply-add each; EPI 2: a tanhExp-shaped
        // activation: 2 exp2 + 1 rcp + 8 plain operations)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float x = acc[t][q];
                if (EPI == 2) {
                    float ex = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(x, -3.0e38f, 40.0f) * 1.442695f);
                    float e2 = __builtin_amdgcn_exp2f(ex * 2.88539f);
                    float tx = fmaf(-2.0f, __builtin_amdgcn_rcpf(e2 + 1.0f), 1.0f);
                    act[t][q] = x * tx * 0.01f + fmaf(-(x * ex), fmaf(tx, tx, -1.0f), tx) * 1e-6f;
                } else act[t][q] = fmaf(x, 0.01f, 0.001f);
                acc[t][q] = 0.f;
            }
    }
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += act[t][q];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int D, int EPI>
void run(const char *name, const float *w, float *out, int wlayers)
{
    const int layers = 448, grid = 256;         // 448 layers of 32 points x 4 waves x 256 workgroups
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((chain<D, EPI>), dim3(grid), dim3(256), 0, 0, w, out, 16, wlayers);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((chain<D, EPI>), dim3(grid), dim3(256), 0, 0, w, out, layers, wlayers);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double flop = (double)grid * 4 * layers * KS * MT * 4096.0;
    printf("%-64s %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 157.3)   weights from L2: %.2f TB/s\n", name, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3 * 100.0,
           (double)grid * 4 * layers * KS * 2 * 64 * 16 / ms * 1e-9);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main()
{
    const int wlayers = 14;     // 14 x 256 KB = 3.5 MB of packed weights, as one reverse-mode pass cycles through (forward + transposed)
    std::vector<float> hw((size_t)wlayers * KS * 2 * 64 * 4);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.01f * (float)((i * 2654435761u >> 20) & 15) - 0.07f;
    float *w, *out;
    hipMalloc(&w, hw.size() * sizeof(float));
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMemcpy(w, hw.data(), hw.size() * sizeof(float), hipMemcpyHostToDevice);
    run<2, 1>("ring 2 k-steps, multiply-add epilogue", w, out, wlayers);
    run<4, 1>("ring 4 k-steps, multiply-add epilogue", w, out, wlayers);
    run<8, 1>("ring 8 k-steps, multiply-add epilogue", w, out, wlayers);
    run<4, 2>("ring 4 k-steps, tanhExp-shaped epilogue", w, out, wlayers);
    run<8, 2>("ring 8 k-steps, tanhExp-shaped epilogue", w, out, wlayers);
    run<4, 1>("ring 4 k-steps, multiply-add epilogue, ONE weight layer (L2-hot)", w, out, 1);
    hipFree(w); hipFree(out);
    return 0;
}
