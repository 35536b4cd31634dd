// How fast can a CU pull a weight layer from L2?  Every workgroup streams the same 128 KB buffer (a 256 x 256 bf16 layer: L2-resident,
// each of its four waves a distinct 32 KB quarter, 1 KB per wave-instruction as the tile engine's buffer loads do), `iters` times.
// Reports bytes per clock per CU for 1, 2 and 4 workgroups per CU.   hipcc --offload-arch=gfx950 -O3 tools/l2_stream_ubench.hip -o tools/bin/l2_stream_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4 *src, u32x4 *out, int iters, int vec_per_wave)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32x4 *p = src + (size_t)wave * vec_per_wave + lane;
    u32x4 acc = { 0, 0, 0, 0 };
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < vec_per_wave / 64; i += UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)(i + u) * 64);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (acc[0] == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int UNROLL>
__global__ __launch_bounds__(256) void stream_kernel_plain(const u32x4 *src, u32x4 *out, int iters, int vec_per_wave)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32x4 *p = src + (size_t)wave * vec_per_wave + lane;
    u32x4 acc = { 0, 0, 0, 0 };
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < vec_per_wave / 64; i += UNROLL) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = p[(size_t)(i + u) * 64];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (acc[0] == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    const size_t bytes = 128 * 1024;
    u32x4 *src, *out;
    hipMalloc(&src, bytes);
    hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(u32x4));
    hipMemset(src, 1, bytes);
    const int iters = 400, vec_per_wave = (int)(bytes / 16 / 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# %d CUs, %.2f GHz nominal; every workgroup reads the same 128 KB %d times; 16 B per lane per load\n", cus, ghz, iters);
    printf("%-10s %-8s %8s %10s %14s\n", "kind", "wg/CU", "unroll", "ms", "B/clk/CU");
    for (int kind = 0; kind < 2; ++kind)
        for (int wpc : { 1, 2, 4 })
            for (int un : { 4, 8 }) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    if (kind == 0 && un == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(cus * wpc), dim3(256), 0, 0, src, out, iters, vec_per_wave);
                    if (kind == 0 && un == 8) hipLaunchKernelGGL(stream_kernel<8>, dim3(cus * wpc), dim3(256), 0, 0, src, out, iters, vec_per_wave);
                    if (kind == 1 && un == 4) hipLaunchKernelGGL(stream_kernel_plain<4>, dim3(cus * wpc), dim3(256), 0, 0, src, out, iters, vec_per_wave);
                    if (kind == 1 && un == 8) hipLaunchKernelGGL(stream_kernel_plain<8>, dim3(cus * wpc), dim3(256), 0, 0, src, out, iters, vec_per_wave);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                const double total = (double)bytes * iters * wpc;         // per CU
                printf("%-10s %-8d %8d %10.3f %14.1f\n", kind ? "plain" : "nontemp", wpc, un, best, total / (best * 1e-3 * ghz * 1e9));
            }
    return 0;
}
