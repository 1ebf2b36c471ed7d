// The FP64-pipe Montgomery product (tools/f64mont.h, an experiment: not part of the library) against the v_mad_u64_u32 one (fr32.h): exactness on random
// and edge operands, latency of a dependent chain, throughput.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gkr_amd/csrc -I tools tools/ubench_f64mont.hip -o tools/bin/ubench_f64mont && tools/bin/ubench_f64mont
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "f64mont.h"
using namespace gkr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// 16 * x mod p
__device__ __forceinline__ Fr times16(Fr x) {
    for (int i = 0; i < 4; ++i) x = fr_add(x, x);
    return x;
}

__global__ void k_check(const Fr* a, const Fr* b, uint32_t n, uint32_t* bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr x = a[i], y = b[i];
    const Fr want = mont_mul(x, y);                                   // x y 2^-256
    const Fr got = times16(f64m::mont_mul260(f64m::to_l52(x), f64m::to_l52(y)));   // (x y 2^-260) 16
    if (!fr_eq(want, got)) atomicAdd(bad, 1u);
}

template <int MODE>
__global__ void __launch_bounds__(64) k_chain(Fr* io, int reps) {
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    Fr x = io[g], y = io[(g + 1u) & 0xfffffu];   // (the buffer holds 2^20 elements)
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0)
            x = mont_mul(x, y);
        else
            x = f64m::mont_mul260(f64m::to_l52(x), f64m::to_l52(y));
    }
    io[g] = x;
}

template <int MODE>
static void run(Fr* d_io, int waves, int reps, const char* what) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k_chain<MODE><<<waves, 64>>>(d_io, 4);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_chain<MODE><<<waves, 64>>>(d_io, reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-18s waves %6d: %7.1f ns per product in a chain, %9.3e products/s\n", what, waves, ms * 1e6 / reps, (double)waves * 64 * reps / (ms * 1e-3));
}

int main() {
    const uint32_t n = 1 << 20;
    std::vector<Fr> a(n), b(n);
    const uint32_t p[8] = GKR_MOD_LIMBS;
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (uint32_t i = 0; i < n; ++i) {
        for (int j = 0; j < 8; ++j) { a[i].l[j] = rnd(); b[i].l[j] = rnd(); }
        a[i].l[7] &= 0x1fffffffu; b[i].l[7] &= 0x1fffffffu;   // < 2^253 < p
    }
    // edge operands: 0, 1, p - 1, limbs of all ones below p, powers of two around the 52-bit limb boundaries
    for (int j = 0; j < 8; ++j) { a[0].l[j] = 0; a[1].l[j] = j == 0; a[2].l[j] = p[j]; a[3].l[j] = 0xffffffffu; b[2].l[j] = p[j]; b[3].l[j] = 0xffffffffu; }
    a[2].l[0] -= 1; b[2].l[0] -= 1; a[3].l[7] = 0x2fffffffu; b[3].l[7] = 0x2fffffffu;
    for (int e = 0; e < 253; ++e) {
        for (int j = 0; j < 8; ++j) { a[4 + e].l[j] = 0; b[4 + e].l[j] = 0xffffffffu; }
        a[4 + e].l[e / 32] = 1u << (e % 32);
        b[4 + e].l[7] = 0x2fffffffu;
        a[300 + e] = a[4 + e];
        a[300 + e].l[0] -= (e > 0);   // 2^e - 1
        b[300 + e] = a[2];
    }
    Fr *d_a, *d_b;
    uint32_t* d_bad;
    CK(hipMalloc(&d_a, sizeof(Fr) * n));
    CK(hipMalloc(&d_b, sizeof(Fr) * n));
    CK(hipMalloc(&d_bad, 4));
    CK(hipMemset(d_bad, 0, 4));
    CK(hipMemcpy(d_a, a.data(), sizeof(Fr) * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, b.data(), sizeof(Fr) * n, hipMemcpyHostToDevice));
    k_check<<<n / 256, 256>>>(d_a, d_b, n, d_bad);
    uint32_t bad = 0;
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("exactness: %u of %u products differ from the integer path\n", bad, n);
    for (int waves : {1, 1024, 4096, 16384}) {
        run<0>(d_a, waves, 256, "v_mad_u64_u32");
        run<1>(d_a, waves, 256, "fp64 fma (+conv)");
    }
    return bad ? 1 : 0;
}
