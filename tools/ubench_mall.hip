// Does the 256 MiB Infinity Cache add read bandwidth on top of an HBM stream?  (Feasibility of a schedule in which a
// table's fold pass follows its first pass while the table still sits in the memory-side cache.)
//   (1) read-only sweeps, repeated back to back inside one timed region, over buffers of 64 MiB .. 4 GiB: the
//       small ones are served by the cache after the first sweep;
//   (2) a mixed kernel: half of the blocks re-read a cache-resident region, the other half stream a cold 6 GiB
//       region -- if cache hits rode a separate path, the sum would exceed the HBM-only rate.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_mall.hip -o /tmp/ubench_mall
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_read_blocked(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, uint32_t C) {
    const size_t base = (size_t)blockIdx.x * C;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t t = threadIdx.x; t < C; t += 256) {
        const size_t i = base + t;
        if (i < n) { uint4 v = in[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    if (acc.x == 0x12345 && acc.y == 0x777) out[0] = acc;
}
// even blocks: chunk of the cold stream; odd blocks: chunk of the warm region (wrapping around it)
__global__ void __launch_bounds__(256) k_read_mixed(const uint4* __restrict__ cold, size_t n_cold, const uint4* __restrict__ warm, size_t n_warm,
                                                    uint4* __restrict__ out, uint32_t C) {
    const bool is_warm = blockIdx.x & 1u;
    const size_t chunk = blockIdx.x >> 1;
    const uint4* src = is_warm ? warm : cold;
    const size_t n = is_warm ? n_warm : n_cold;
    const size_t base = is_warm ? (chunk * C) % n_warm : chunk * C;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t t = threadIdx.x; t < C; t += 256) {
        const size_t i = base + t;
        if (i < n) { uint4 v = src[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    if (acc.x == 0x12345 && acc.y == 0x777) out[0] = acc;
}

int main() {
    const size_t big = (size_t)6 << 30;
    uint4 *a, *b; CK(hipMalloc(&a, big)); CK(hipMalloc(&b, 4096));
    CK(hipMemset(a, 1, big));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t C = 4096;   // 64 KiB per block
    for (size_t mib : {32, 64, 96, 128, 192, 224, 256, 320, 512, 4096}) {
        const size_t bytes = mib << 20, n = bytes / 16;
        const int reps = (int)(((size_t)8 << 30) / bytes);
        hipLaunchKernelGGL(k_read_blocked, dim3((unsigned)((n + C - 1) / C)), dim3(256), 0, 0, a, b, n, C);   // first touch
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read_blocked, dim3((unsigned)((n + C - 1) / C)), dim3(256), 0, 0, a, b, n, C);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("re-read %5zu MiB x %4d sweeps: %8.1f GB/s  (%.1f us per sweep)\n", mib, reps, (double)bytes * reps / ms / 1e6, ms * 1e3 / reps);
    }
    // one long kernel per buffer size instead of many launches: the sweep repeated inside the grid (blocks wrap around)
    for (size_t mib : {96, 192}) {
        const size_t n_warm = (mib << 20) / 16, n_cold = ((size_t)4 << 30) / 16;
        uint4* cold = a + n_warm;   // disjoint from the warm region
        const unsigned pairs = (unsigned)(n_cold / C);
        hipLaunchKernelGGL(k_read_blocked, dim3((unsigned)(n_warm / C)), dim3(256), 0, 0, a, b, n_warm, C);   // warm it
        float best = 1e9;
        for (int r = 0; r < 3; ++r) {
            hipLaunchKernelGGL(k_read_blocked, dim3((unsigned)(n_warm / C)), dim3(256), 0, 0, a, b, n_warm, C);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_read_mixed, dim3(2 * pairs), dim3(256), 0, 0, cold, n_cold, a, n_warm, b, C);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("mixed: 4 GiB cold stream + 4 GiB of re-reads of a %zu MiB warm region: %8.1f GB/s total (%.3f ms)\n", mib,
               2.0 * (double)n_cold * 16 / best / 1e6, best);
        float cold_only = 1e9;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_read_blocked, dim3(pairs), dim3(256), 0, 0, cold, b, n_cold, C);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < cold_only) cold_only = ms;
        }
        printf("       the cold 4 GiB stream alone: %8.1f GB/s (%.3f ms)\n", (double)n_cold * 16 / cold_only / 1e6, cold_only);
    }
    CK(hipGetLastError());
    return 0;
}
