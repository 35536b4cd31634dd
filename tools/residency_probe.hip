// residency_probe.hip -- how many workgroups of a persistent grid does an MI355X actually keep resident, and where?
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/residency_probe tools/residency_probe.hip && tools/bin/residency_probe
//
// A grid of G workgroups with the footprint of the field kernels (256 threads, 72.8 KB of dynamic LDS: two fit a CU) spins for a
// few milliseconds; every workgroup records the constant 100 MHz clock at its start and its end and where it ran (XCC_ID, HW_ID:
// shader engine, shader array, CU).  A persistent kernel with a tile queue only uses the workgroups that are resident from the
// start -- one that starts when another ends finds the queue empty -- so "started within 100 us of the first" is the number that
// matters.  Printed per grid size: workgroups resident from the start, CUs holding 0 / 1 / 2 of them, per XCD and shader engine.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <map>
#include <tuple>
#include <vector>

struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };

__global__ __launch_bounds__(256, 2) void spin(Rec *rec, unsigned long long ticks)
{
    extern __shared__ float smem[];
    const unsigned long long t0 = wall_clock64();
    smem[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        rec[blockIdx.x].t0 = t0;
        rec[blockIdx.x].t1 = wall_clock64();
        rec[blockIdx.x].hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        rec[blockIdx.x].xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    }
    if (smem[(threadIdx.x + 1) & 255] < 0.f) rec[0].t0 = 0;       // keep the LDS alive
}

int main(int argc, char **argv)
{
    const size_t lds = argc > 1 ? (size_t)atoi(argv[1]) : 72768;         // ddf_rev_kernel<2, 4, 2, OpsF32>: 64 x 260 floats + 6 208 B
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    printf("device: %s, %d CUs; workgroup = 256 threads + %zu B of LDS\n", prop.name, prop.multiProcessorCount, lds);
    (void)hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int per_cu = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, spin, 256, lds);
    printf("hipOccupancyMaxActiveBlocksPerMultiprocessor: %d\n", per_cu);
    const int cus = prop.multiProcessorCount;
    for (int G : { 2 * cus, 2 * cus + cus / 4, 3 * cus, 4 * cus }) {
        Rec *d;
        (void)hipMalloc(&d, G * sizeof(Rec));
        (void)hipMemset(d, 0, G * sizeof(Rec));
        hipLaunchKernelGGL(spin, dim3(G), dim3(256), lds, 0, d, 300000ull);        // 3 ms
        (void)hipDeviceSynchronize();
        std::vector<Rec> h(G);
        (void)hipMemcpy(h.data(), d, G * sizeof(Rec), hipMemcpyDeviceToHost);
        (void)hipFree(d);
        unsigned long long first = ~0ull;
        for (auto &r : h) first = std::min(first, r.t0);
        int early = 0;
        std::map<std::tuple<unsigned, unsigned, unsigned, unsigned>, int> per_cu_count;      // (xcc, se, sh, cu) -> workgroups resident from the start
        std::map<std::pair<unsigned, unsigned>, int> per_se;
        for (auto &r : h) {
            if (r.t0 - first > 10000) continue;       // started more than 100 us after the first: it waited for a slot
            ++early;
            const unsigned cu = (r.hw >> 8) & 15, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7;
            ++per_cu_count[{ r.xcc, se, sh, cu }];
            ++per_se[{ r.xcc, se }];
        }
        int hist[8] = { 0 };
        for (auto &kv : per_cu_count) ++hist[kv.second < 7 ? kv.second : 7];
        printf("grid %4d: resident from the start %4d (%.1f %% of %d slots); CUs seen %zu: holding 1: %d, 2: %d, 3+: %d\n", G, early,
               100.0 * early / (2 * cus), 2 * cus, per_cu_count.size(), hist[1], hist[2], hist[3] + hist[4] + hist[5] + hist[6] + hist[7]);
        if (G == 2 * cus || G == 4 * cus) {
            printf("   per (XCD, shader engine): ");
            for (auto &kv : per_se) printf("(%u,%u)=%d ", kv.first.first, kv.first.second, kv.second);
            printf("\n   CUs per (XCD, shader engine) seen: ");
            std::map<std::pair<unsigned, unsigned>, int> cus_se;
            for (auto &kv : per_cu_count) ++cus_se[{ std::get<0>(kv.first), std::get<1>(kv.first) }];
            for (auto &kv : cus_se) printf("(%u,%u)=%d ", kv.first.first, kv.first.second, kv.second);
            printf("\n");
        }
    }
    return 0;
}
