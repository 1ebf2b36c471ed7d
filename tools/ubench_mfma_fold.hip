// Micro-benchmark + exactness check of the matrix-core fold pass (gkr_amd/csrc/mfma_fold.h) against the
// v_mad_u64_u32 form and a host reference.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gkr_amd/csrc tools/ubench_mfma_fold.hip -o /tmp/ubench_mfma_fold && /tmp/ubench_mfma_fold
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "fr32.h"
#include "mfma_fold.h"
#ifndef MF_STREAM_PAD
#define MF_STREAM_PAD 0
#define MF_NO_PAD 1
#endif
#ifndef ROT
#define ROT (blockIdx.x * 5u + blockIdx.y * 3u)
#endif
using namespace gkr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int JIN>
__global__ void __launch_bounds__(256) k_plan(const Fr* __restrict__ weights, MfmaFoldPlan* __restrict__ plans) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32 * 32 * (1 << JIN)];
    mfma_plan_block<JIN>(weights + (size_t)blockIdx.x * 32, plans + blockIdx.x, lds);
}

template <int JIN>
__global__ void __launch_bounds__(256) k_mfma(const Fr* __restrict__ src, size_t src_stride, Fr* __restrict__ dst, size_t dst_stride,
                                              uint32_t S, const MfmaFoldPlan* __restrict__ plans, const Fr* __restrict__ weights,
                                              Acc<9>* __restrict__ partials) {
    const Fr* s = src + (size_t)blockIdx.y * src_stride;
    Fr* d = dst + (size_t)blockIdx.y * dst_stride;
    const uint32_t chunk = S / gridDim.x, begin = blockIdx.x * chunk;
    Acc<9> acc = acc_zero<9>();
    __shared__ __attribute__((aligned(16))) unsigned char lds[JIN > 2 ? 32 * 32 * (1 << JIN) : 16];
    mfma_multifold_block<JIN>(s, d, S, plans + blockIdx.y, begin, begin + chunk, ROT, acc, lds);
    partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = acc;
}

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static Fr rand_fr(int kind) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    Fr f;
    if (kind == 1) { for (int i = 0; i < 8; ++i) f.l[i] = p[i]; f.l[0] -= 1; return f; }   // p - 1
    if (kind == 2) return fr_zero();
    if (kind == 3) { for (int i = 0; i < 8; ++i) f.l[i] = 0xffffffffu; f.l[7] = 0x2fffffffu; return f; }
    if (kind == 4) { for (int i = 0; i < 8; ++i) f.l[i] = 0x80808080u; f.l[7] = 0x20808080u; return f; }
    for (int i = 0; i < 8; ++i) f.l[i] = (uint32_t)rnd();
    f.l[7] &= 0x1fffffffu;
    return f;
}

template <int JIN>
static int run(uint32_t S, uint32_t batch, uint32_t nblk, int reps, bool check) {
    constexpr int NB = 1 << JIN;
    const size_t src_len = ((size_t)S + MF_STREAM_PAD) * NB;   // MF_STREAM_PAD: the stride experiment (0 = the real layout)
    std::vector<Fr> h_src(src_len * batch), h_w(32 * (size_t)batch), h_wm(32 * (size_t)batch);
    for (size_t i = 0; i < h_src.size(); ++i) h_src[i] = rand_fr(i < 64 ? (int)(i % 5) : (rnd() % 64 == 0 ? (int)(rnd() % 5) : 0));
    for (size_t i = 0; i < h_w.size(); ++i) {
        h_w[i] = rand_fr(i < 5 ? (int)i : 0);
        h_wm[i] = to_mont(h_w[i]);
    }
    Fr *d_src, *d_dst, *d_w;
    Acc<9>* d_part;
    MfmaFoldPlan* d_plan;
    CK(hipMalloc(&d_plan, sizeof(MfmaFoldPlan) * batch));
    if (getenv("UB_CONTIG")) CK(hipExtMallocWithFlags((void**)&d_src, sizeof(Fr) * h_src.size(), hipDeviceMallocContiguous));
    else CK(hipMalloc(&d_src, sizeof(Fr) * h_src.size()));
    CK(hipMalloc(&d_dst, sizeof(Fr) * (size_t)S * batch));
    CK(hipMalloc(&d_w, sizeof(Fr) * h_wm.size()));
    CK(hipMalloc(&d_part, sizeof(Acc<9>) * 256 * (size_t)nblk * batch));
    CK(hipMemcpy(d_src, h_src.data(), sizeof(Fr) * h_src.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, h_wm.data(), sizeof(Fr) * h_wm.size(), hipMemcpyHostToDevice));
    dim3 grid(nblk, batch);
    hipLaunchKernelGGL(k_plan<JIN>, dim3(batch), dim3(256), 0, 0, d_w, d_plan);
    hipLaunchKernelGGL(k_mfma<JIN>, grid, dim3(256), 0, 0, d_src, src_len, d_dst, (size_t)S, S, d_plan, d_w, d_part);
    CK(hipDeviceSynchronize());
    int bad = 0;
    if (check) {
        std::vector<Fr> h_dst((size_t)S * batch);
        CK(hipMemcpy(h_dst.data(), d_dst, sizeof(Fr) * h_dst.size(), hipMemcpyDeviceToHost));
        for (uint32_t t = 0; t < batch; ++t)
            for (uint32_t i = 0; i < S; ++i) {
                Fr y = fr_zero();
                for (int b = 0; b < NB; ++b) y = fr_add(y, mont_mul(h_src[(size_t)t * src_len + (size_t)b * (S + MF_STREAM_PAD) + i], h_wm[(size_t)t * 32 + b]));
                if (!fr_eq(y, h_dst[(size_t)t * S + i])) {
                    if (bad < 4) {
                        printf("  mismatch table %u entry %u: got", t, i);
                        for (int k = 7; k >= 0; --k) printf(" %08x", h_dst[(size_t)t * S + i].l[k]);
                        printf("\n                              want");
                        for (int k = 7; k >= 0; --k) printf(" %08x", y.l[k]);
                        printf("\n");
                    }
                    ++bad;
                }
            }
        printf("J=%d S=%u batch=%u nblk=%u: %d mismatches of %zu\n", JIN, S, batch, nblk, bad, h_dst.size());
    }
    if (reps > 0) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_plan<JIN>, dim3(batch), dim3(256), 0, 0, d_w, d_plan);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms_plan;
        CK(hipEventElapsedTime(&ms_plan, e0, e1));
        printf("  plan kernel: %.1f us\n", ms_plan / reps * 1e3);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(k_mfma<JIN>, grid, dim3(256), 0, 0, d_src, src_len, d_dst, (size_t)S, S, d_plan, d_w, d_part);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)batch * ((double)src_len + S) * 32.0;
        printf("J=%d S=%u batch=%u nblk=%u: %.3f ms per pass, %.2f TB/s\n", JIN, S, batch, nblk, ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
    }
    CK(hipFree(d_plan)); CK(hipFree(d_src)); CK(hipFree(d_dst)); CK(hipFree(d_w)); CK(hipFree(d_part));
    return bad;
}

int main(int argc, char** argv) {
    int bad = 0;
    bad += run<3>(1024, 3, 2, 0, true);
    bad += run<2>(1024, 2, 4, 0, true);
    bad += run<1>(512, 2, 2, 0, true);
    bad += run<3>(1u << 14, 4, 16, 0, true);
    bad += run<4>(1024, 3, 2, 0, true);
    bad += run<5>(2048, 3, 4, 0, true);
    bad += run<5>(1u << 13, 2, 32, 0, true);
    if (bad) { printf("FAILED\n"); return 1; }
    if (argc > 1 && !strcmp(argv[1], "check")) return 0;
    for (uint32_t nblk : {8u, 16u, 32u, 64u, 128u, 512u}) run<3>(1u << 17, 64, nblk, 5, false);
    for (uint32_t nblk : {8u, 16u, 32u}) run<3>(1u << 17, 256, nblk, 3, false);
    for (uint32_t nblk : {8u, 16u, 32u, 64u}) run<3>(1u << 14, 64, nblk, 5, false);
    for (uint32_t nblk : {8u, 16u, 32u}) run<3>(1u << 11, 64, nblk, 5, false);
    for (uint32_t nblk : {16u, 32u, 64u}) run<4>(1u << 16, 64, nblk, 5, false);
    for (uint32_t nblk : {32u, 64u}) run<5>(1u << 15, 64, nblk, 5, false);
    for (uint32_t nblk : {4u, 8u, 16u}) run<5>(1u << 10, 64, nblk, 5, false);
    run<2>(1u << 18, 64, 128, 5, false);
    run<1>(1u << 19, 64, 256, 5, false);
    return 0;
}
