// Micro-benchmark: latency of the MiMC7 transcript on the device, one sumcheck per lane -- what a device-side
// transcript of the multi-round passes would pay per round (2-element multi_hash = 2 x 91 rounds x 4 Montgomery
// multiplications, all dependent).  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gkr_amd/csrc tools/ubench_mimc.hip -o tools/bin/ubench_mimc && tools/bin/ubench_mimc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "mimc7.h"
using namespace gkr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// `reps` chained 2-element hashes per lane (each hash's input is the previous output), CH independent chains per lane
template <int CH>
__global__ void __launch_bounds__(64) k_chain(const Fr* __restrict__ cts, Fr* __restrict__ io, int reps) {
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    Fr x[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = io[(size_t)g * CH + c];
    for (int r = 0; r < reps; ++r) {
        Fr acc[CH], a0[CH], a1[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            a0[c] = to_mont(x[c]);
            a1[c] = to_mont(fr_add(x[c], x[c]));
            acc[c] = fr_zero();
        }
        for (int e = 0; e < 2; ++e) {
            Fr h[CH], a[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) { a[c] = e ? a1[c] : a0[c]; h[c] = fr_zero(); }
            for (int i = 0; i < kMimcRounds; ++i) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    Fr t = (i == 0) ? fr_add(a[c], acc[c]) : fr_add(fr_add(h[c], acc[c]), cts[i]);
                    Fr t2 = mont_mul(t, t);
                    Fr t4 = mont_mul(t2, t2);
                    Fr t6 = mont_mul(t4, t2);
                    h[c] = mont_mul(t6, t);
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = fr_add(fr_add(acc[c], a[c]), fr_add(h[c], acc[c]));
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) x[c] = from_mont(acc[c]);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) io[(size_t)g * CH + c] = x[c];
}

template <int CH>
static void run(const Fr* d_cts, Fr* d_io, int waves, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k_chain<CH><<<waves, 64>>>(d_cts, d_io, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_chain<CH><<<waves, 64>>>(d_cts, d_io, reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("chains/lane %d  waves %5d  reps %d: %8.1f us per 2-element hash (per lane), %9.3e hashes/s\n", CH, waves, reps,
           ms * 1e3 / reps, (double)waves * 64 * CH * reps / (ms * 1e-3));
}

int main() {
    std::vector<Fr> cts(kMimcRounds);
    for (int i = 0; i < kMimcRounds; ++i) {   // any field elements do for timing
        for (int j = 0; j < 8; ++j) cts[i].l[j] = 0x9e3779b9u * (i * 8 + j + 1);
        cts[i].l[7] &= 0x0fffffffu;
    }
    const int max_lanes = 4096 * 64 * 2;
    std::vector<Fr> io(max_lanes);
    for (int i = 0; i < max_lanes; ++i) {
        for (int j = 0; j < 8; ++j) io[i].l[j] = 0x85ebca6bu * (i * 8 + j + 3);
        io[i].l[7] &= 0x0fffffffu;
    }
    Fr *d_cts, *d_io;
    CK(hipMalloc(&d_cts, sizeof(Fr) * kMimcRounds));
    CK(hipMalloc(&d_io, sizeof(Fr) * max_lanes));
    CK(hipMemcpy(d_cts, cts.data(), sizeof(Fr) * kMimcRounds, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_io, io.data(), sizeof(Fr) * max_lanes, hipMemcpyHostToDevice));
    for (int waves : {1, 4, 16, 256, 1024, 4096}) run<1>(d_cts, d_io, waves, 8);
    for (int waves : {1, 4, 16, 256, 1024}) run<2>(d_cts, d_io, waves, 8);
    return 0;
}
