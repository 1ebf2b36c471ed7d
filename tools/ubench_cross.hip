// k_prod_cross_mfma alone on random tables: us per launch at 2^m entries (m = 20, 22), both block shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gkr_amd/csrc tools/ubench_cross.hip -o /tmp/ubench_cross
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "mfma_cross.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_fill(uint32_t* p, size_t words) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + 12345u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (i & 7) == 7 ? (x & 0x0fffffffu) : x;   // (below r)
    }
}

int main() {
    using namespace gkr;
    for (uint32_t m : {17u, 18u, 20u, 22u}) {
        const size_t n = (size_t)1 << m;
        Fr *W, *X, *Y, *part;
        CK(hipMalloc(&W, n * 32)); CK(hipMalloc(&X, n * 32)); CK(hipMalloc(&Y, n * 32)); CK(hipMalloc(&part, (size_t)8192 * 72 * 32));
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)W, n * 8);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)X, n * 8);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)Y, n * 8);
        CK(hipDeviceSynchronize());
        const uint32_t S = 1u << (m - 3);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (uint32_t kc : {128u, 256u, 512u, 1024u, 2048u}) {
            if (S / kc < 16u || S / kc > 2048u) continue;
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (kc == 128u) hipLaunchKernelGGL(k_prod_cross_mfma<128>, dim3(S / kc, 1), dim3(512), 0, 0, W, X, Y, m, part, (uint32_t)n);
                else if (kc == 256u) hipLaunchKernelGGL(k_prod_cross_mfma<256>, dim3(S / kc, 1), dim3(512), 0, 0, W, X, Y, m, part, (uint32_t)n);
                else if (kc == 512u) hipLaunchKernelGGL(k_prod_cross_mfma<512>, dim3(S / kc, 1), dim3(512), 0, 0, W, X, Y, m, part, (uint32_t)n);
                else if (kc == 2048u) hipLaunchKernelGGL(k_prod_cross_mfma<2048>, dim3(S / kc, 1), dim3(512), 0, 0, W, X, Y, m, part, (uint32_t)n);
                else hipLaunchKernelGGL(k_prod_cross_mfma<1024>, dim3(S / kc, 1), dim3(512), 0, 0, W, X, Y, m, part, (uint32_t)n);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("m %u  kc %4u  blocks %5u  %8.1f us   (%.0f GB/s of 3 x 2^m x 32 B)\n", m, kc, S / kc, best * 1e3, 3.0 * n * 32 / (best * 1e-3) / 1e9);
        }
        CK(hipFree(W)); CK(hipFree(X)); CK(hipFree(Y)); CK(hipFree(part));
    }
    return 0;
}
