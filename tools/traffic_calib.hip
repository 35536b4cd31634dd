// traffic_calib.hip -- known byte counts in the field kernels' own access patterns, for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on
// the box (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern before trusting an absolute").
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/traffic_calib tools/traffic_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- tools/bin/traffic_calib        (and a second pass with WRITE_SIZE)
//   python tools/traffic_calib.py <dir>     -> factor = known bytes / (counter x 1024) per kernel
//
// Kernels (each launched as a persistent grid of 512 workgroups x 256 threads, like the field kernels):
//   read_frag_plain<0> 16 B per lane, a wave reads 1 KiB contiguous per instruction (the weight-fragment / y' pattern); 4 GiB read once
//   read_frag_plain<1>  the same over a 128 MiB buffer, 16 passes (fits the 256 MiB Infinity Cache after the first pass)
//   write_frag<0> 16 B per lane stores, 4 GiB written once
//   write_frag<1> the same over 128 MiB, 16 passes
//   roundtrip  every workgroup writes a 400 KiB private slot and reads it back (the y' round trip), 512 slots = 200 MiB, 8 rounds
//   read4 / write4   4 B per lane, consecutive lanes consecutive dwords (the per-point records): 1 GiB each
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read_frag(const u32x4 *src, size_t n_vec, int passes, unsigned *sink)
{
    u32x4 acc = { 0u, 0u, 0u, 0u };
    const size_t stride = (size_t)gridDim.x * 256;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += stride) {
            const u32x4 v = __builtin_nontemporal_load(&src[i]) ;
            acc ^= v;
        }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;        // keeps the loads alive, practically never true
}

template <int TAG>
__global__ __launch_bounds__(256) void read_frag_plain(const u32x4 *src, size_t n_vec, int passes, unsigned *sink)
{
    u32x4 acc = { 0u, 0u, 0u, 0u };
    const size_t stride = (size_t)gridDim.x * 256;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += stride) acc ^= src[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int TAG>
__global__ __launch_bounds__(256) void write_frag(u32x4 *dst, size_t n_vec, int passes)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += stride) dst[i] = (u32x4){ (unsigned)i, (unsigned)p, 3u, 4u };
}

__global__ __launch_bounds__(256) void roundtrip(u32x4 *slots, size_t vec_per_slot, int rounds, unsigned *sink)
{
    u32x4 *mine = slots + (size_t)blockIdx.x * vec_per_slot;
    u32x4 acc = { 0u, 0u, 0u, 0u };
    for (int r = 0; r < rounds; ++r) {
        for (size_t i = threadIdx.x; i < vec_per_slot; i += 256) mine[i] = (u32x4){ (unsigned)i, (unsigned)r, acc[0], 7u };
        __syncthreads();
        for (size_t i = threadIdx.x; i < vec_per_slot; i += 256) acc ^= mine[i];
        __syncthreads();
    }
    if ((acc[0] ^ acc[1]) == 0x12345678u) sink[0] = 1;
}

__global__ __launch_bounds__(256) void read4(const unsigned *src, size_t n, unsigned *sink)
{
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc ^= src[i];
    if (acc == 0x12345678u) sink[0] = 1;
}

__global__ __launch_bounds__(256) void write4(unsigned *dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = (unsigned)i;
}

int main()
{
    const size_t big = (size_t)4 << 30, warm = (size_t)128 << 20, slot = (size_t)400 << 10, gib = (size_t)1 << 30;
    char *buf;
    unsigned *sink;
    CHK(hipMalloc((void **)&buf, big));
    CHK(hipMalloc((void **)&sink, 64));
    CHK(hipMemset(buf, 1, big));
    CHK(hipMemset(sink, 0, 64));
    CHK(hipDeviceSynchronize());
    const int grid = 512;
    // name, known read bytes, known written bytes (printed for tools/traffic_calib.py)
    hipLaunchKernelGGL(read_frag_plain<0>, dim3(grid), dim3(256), 0, 0, (const u32x4 *)buf, big / 16, 1, sink);
    printf("KNOWN read_frag_plain<0> read %zu write 0   # 4 GiB once, 16 B per lane\n", big);
    hipLaunchKernelGGL(read_frag, dim3(grid), dim3(256), 0, 0, (const u32x4 *)buf, big / 16, 1, sink);
    printf("KNOWN read_frag read %zu write 0   # 4 GiB once, 16 B per lane, non-temporal\n", big);
    hipLaunchKernelGGL(write_frag<0>, dim3(grid), dim3(256), 0, 0, (u32x4 *)buf, big / 16, 1);
    printf("KNOWN write_frag<0> read 0 write %zu   # 4 GiB once\n", big);
    CHK(hipDeviceSynchronize());
    // warm cases (template tag 1: a kernel name of their own in the trace)
    hipLaunchKernelGGL(read_frag_plain<1>, dim3(grid), dim3(256), 0, 0, (const u32x4 *)buf, warm / 16, 16, sink);
    printf("KNOWN read_frag_plain<1> read %zu write 0   # 128 MiB x 16 passes (Infinity-Cache resident after the first)\n", warm * 16);
    hipLaunchKernelGGL(write_frag<1>, dim3(grid), dim3(256), 0, 0, (u32x4 *)buf, warm / 16, 16);
    printf("KNOWN write_frag<1> read 0 write %zu   # 128 MiB x 16 passes\n", warm * 16);
    hipLaunchKernelGGL(roundtrip, dim3(grid), dim3(256), 0, 0, (u32x4 *)buf, slot / 16, 8, sink);
    printf("KNOWN roundtrip read %zu write %zu   # 512 slots x 400 KiB (200 MiB) x 8 rounds, written then read back by the same workgroup\n",
           slot * grid * 8, slot * grid * 8);
    hipLaunchKernelGGL(read4, dim3(grid), dim3(256), 0, 0, (const unsigned *)buf, gib / 4, sink);
    printf("KNOWN read4 read %zu write 0   # 1 GiB, 4 B per lane\n", gib);
    hipLaunchKernelGGL(write4, dim3(grid), dim3(256), 0, 0, (unsigned *)buf, gib / 4);
    printf("KNOWN write4 read 0 write %zu   # 1 GiB, 4 B per lane\n", gib);
    CHK(hipDeviceSynchronize());
    CHK(hipGetLastError());
    printf("done\n");
    return 0;
}
