// Micro-benchmark for the device-side transcript question (DESIGN.md "Transcript placement"):
// ONE 254-bit Montgomery product spread over EIGHT lanes -- lane j of a group holds limb j (32 bits) of every operand --
// so that the 91 x 4 dependent products of a MiMC7 permutation are chains of ~12 dependent VALU instructions per limb step
// instead of a 256-instruction chain on one lane.  Eight transcripts per wave.
//   product:  operand scanning, one limb of b per step: acc += a_j * b_i; q = acc_0 * (-p^-1); acc += p_j * q; then the
//             accumulators move one lane down (division by 2^32), carries deferred in a 96-bit per-lane accumulator and
//             resolved once at the end with the ballot carry-lookahead.  Cross-lane moves are DPP (row_share / row_shl
//             inside 16-lane rows), not LDS permutes: they sit on the dependent chain.
//   values stay in [0, 2p) between products (no final subtraction: a, b < 2p => a b / 2^256 + p < 2p as 4p < 2^256); a
//   MiMC round costs one 3-operand addition with carry resolution and one conditional subtraction of 2p.
// Checked here against the one-lane arithmetic of fr32.h, then timed: one group alone (latency), all lanes of many waves
// (throughput).  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gkr_amd/csrc tools/ubench_mimc_lanes.hip -o tools/bin/ubench_mimc_lanes && tools/bin/ubench_mimc_lanes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "mimc7.h"
using namespace gkr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __constant__ uint32_t c_mod[8] = GKR_MOD_LIMBS;

// lane (j of its group of eight) helpers.  DPP controls: row_shl:n = 0x100 + n, row_shr:n = 0x110 + n, row_share:n = 0x150 + n
__device__ __forceinline__ uint32_t dpp_row_share0(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x150, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_row_share8(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x158, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_row_shl1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_row_shr1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true); }
// limb 0 of the group, in every lane of the group
__device__ __forceinline__ uint32_t group_first(uint32_t x, bool upper) { return upper ? dpp_row_share8(x) : dpp_row_share0(x); }
// limb i of the group (i uniform), in every lane of the group
template <int I>
__device__ __forceinline__ uint32_t group_bcast(uint32_t x, bool upper) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 + I, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x158 + I, 0xf, 0xf, true);
    return upper ? hi : lo;
}

// per-lane value v = limb + 2^32 * extra (extra small) -> canonical 32-bit limbs of the same integer (< 2^256: no carry
// out of the group's top lane).  One shift of the deferred carries, then generate / propagate through a ballot.
__device__ __forceinline__ uint32_t resolve_carries(uint64_t v, uint32_t j) {
    const uint32_t limb = (uint32_t)v, extra = (uint32_t)(v >> 32);
    uint32_t from_below = dpp_row_shr1(extra);
    if (j == 0) from_below = 0;
    const uint32_t s = limb + from_below;
    const uint64_t g = __ballot(s < limb), p = __ballot(s == 0xffffffffu);
    const uint64_t gs = (g << 1) & 0xfefefefefefefefeull;       // a carry never leaves its group of eight
    const uint64_t cin = ((gs + p) ^ p);                        // lanes a carry arrives at (runs of all-ones limbs pass it on)
    return s + (uint32_t)((cin >> (threadIdx.x & 63u)) & 1u);
}

// Montgomery product of two values < 2p held one limb per lane: returns limb j of a b 2^-256 mod p, < 2p
__device__ __forceinline__ uint32_t lanes_mont_mul(uint32_t a, uint32_t b, uint32_t pj, uint32_t j, bool upper) {
    uint32_t bi[8];
    bi[0] = group_bcast<0>(b, upper);
    bi[1] = group_bcast<1>(b, upper);
    bi[2] = group_bcast<2>(b, upper);
    bi[3] = group_bcast<3>(b, upper);
    bi[4] = group_bcast<4>(b, upper);
    bi[5] = group_bcast<5>(b, upper);
    bi[6] = group_bcast<6>(b, upper);
    bi[7] = group_bcast<7>(b, upper);
    uint64_t acc = 0;
    uint32_t ex = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mac96(acc, ex, a, bi[i]);
        const uint32_t q = group_first((uint32_t)acc, upper) * GKR_INV32;
        mac96(acc, ex, pj, q);
        // one limb down: lane j takes the low word of lane j + 1 (the group's last lane takes nothing)
        uint32_t from_above = dpp_row_shl1((uint32_t)acc);
        if (j == 7) from_above = 0;
        acc = (acc >> 32) + ((uint64_t)ex << 32) + from_above;
        ex = 0;
    }
    return resolve_carries(acc, j);
}

// x + y + z (each < 2p or a constant < p; the sum < 2^256), one limb per lane
__device__ __forceinline__ uint32_t lanes_add3(uint32_t x, uint32_t y, uint32_t z, uint32_t j) {
    return resolve_carries((uint64_t)x + y + z, j);
}
// x - 2p if x >= 2p (x < 4p): borrow lookahead the same way
__device__ __forceinline__ uint32_t lanes_cond_sub_2p(uint32_t x, uint32_t two_p_j, uint32_t j) {
    const uint32_t d = x - two_p_j;
    const uint64_t g = __ballot(x < two_p_j), p = __ballot(d == 0u);
    const uint64_t gs = (g << 1) & 0xfefefefefefefefeull;
    const uint64_t bin = ((gs + p) ^ p);
    const uint32_t r = d - (uint32_t)((bin >> (threadIdx.x & 63u)) & 1u);
    // borrow out of the group's top lane: x < 2p, keep x.  Top lane's own outcome, broadcast to its group
    const uint64_t out = (g | (p & bin)) & 0x8080808080808080ull;   // lane 7 generated, or passed one on
    const uint32_t grp = (threadIdx.x & 63u) >> 3;
    return ((out >> (grp * 8u + 7u)) & 1u) ? x : r;
}

// one MiMC7 permutation (91 rounds), key k, input x, all Montgomery form, one limb per lane
__device__ __forceinline__ uint32_t lanes_mimc(uint32_t x, uint32_t k, const Fr* __restrict__ cts, uint32_t pj, uint32_t two_pj, uint32_t j, bool upper) {
    uint32_t h = 0;
    for (int i = 0; i < kMimcRounds; ++i) {
        uint32_t t = i == 0 ? lanes_add3(x, k, 0u, j) : lanes_add3(h, k, cts[i].l[j], j);
        t = lanes_cond_sub_2p(t, two_pj, j);
        const uint32_t t2 = lanes_mont_mul(t, t, pj, j, upper);
        const uint32_t t4 = lanes_mont_mul(t2, t2, pj, j, upper);
        const uint32_t t6 = lanes_mont_mul(t4, t2, pj, j, upper);
        h = lanes_mont_mul(t6, t, pj, j, upper);
    }
    return lanes_cond_sub_2p(lanes_add3(h, k, 0u, j), two_pj, j);
}

// check: every group multiplies its pair; lane 0 of the group also does it with the one-lane arithmetic
__global__ void k_check(const Fr* __restrict__ a, const Fr* __restrict__ b, Fr* __restrict__ out_lanes, Fr* __restrict__ out_ref) {
    const uint32_t lane = threadIdx.x & 63u, j = lane & 7u, grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const bool upper = (lane & 8u) != 0u;
    const uint32_t r = lanes_mont_mul(a[grp].l[j], b[grp].l[j], c_mod[j], j, upper);
    out_lanes[grp].l[j] = r;
    if (j == 0) out_ref[grp] = mont_mul(a[grp], b[grp]);
}

// timing: `reps` chained permutations per group
__global__ void __launch_bounds__(256) k_hash_chain(const Fr* __restrict__ cts, Fr* __restrict__ io, int reps) {
    const uint32_t lane = threadIdx.x & 63u, j = lane & 7u, grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const bool upper = (lane & 8u) != 0u;
    const uint32_t pj = c_mod[j];
    // 2p, limb j (p < 2^254: no overflow)
    const uint32_t two_pj = (pj << 1) | (j ? c_mod[j - 1] >> 31 : 0u);
    uint32_t x = io[grp].l[j], k = 0;
    for (int r = 0; r < reps; ++r) {
        const uint32_t h = lanes_mimc(x, k, cts, pj, two_pj, j, upper);
        k = lanes_cond_sub_2p(lanes_add3(k, x, h, j), two_pj, j);   // multi_hash: r += a + hash(a, r)
        x = h;
    }
    io[grp].l[j] = k;
}
__global__ void __launch_bounds__(256) k_mul_chain(Fr* __restrict__ io, int reps) {
    const uint32_t lane = threadIdx.x & 63u, j = lane & 7u, grp = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const bool upper = (lane & 8u) != 0u;
    const uint32_t pj = c_mod[j];
    uint32_t x = io[grp].l[j];
    for (int r = 0; r < reps; ++r) x = lanes_mont_mul(x, x, pj, j, upper);
    io[grp].l[j] = x;
}

static double time_kernel(void (*launch)(int, int), int blocks, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch(blocks, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch(blocks, reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

static Fr *g_cts, *g_io;
static int g_threads = 64;
static void launch_hash(int blocks, int reps) { k_hash_chain<<<blocks, g_threads>>>(g_cts, g_io, reps); }
static void launch_mul(int blocks, int reps) { k_mul_chain<<<blocks, g_threads>>>(g_io, reps); }

int main() {
    const int groups = 4096 * 32;
    std::vector<Fr> a(groups), b(groups), cts(kMimcRounds);
    uint64_t st = 0x1234567;
    auto rnd = [&] { st = st * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(st >> 32); };
    for (int i = 0; i < groups; ++i) {
        for (int j = 0; j < 8; ++j) a[i].l[j] = rnd(), b[i].l[j] = rnd();
        a[i].l[7] &= 0x1fffffffu;   // < 2^253 < p
        b[i].l[7] &= 0x1fffffffu;
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) a[i].l[j] = i == 0 ? 0u : (i == 1 ? 0xffffffffu : (i == 2 ? (j == 0) : 0x80000000u));
    a[1].l[7] = 0x1fffffffu;
    for (int i = 0; i < kMimcRounds; ++i) {
        for (int j = 0; j < 8; ++j) cts[i].l[j] = rnd();
        cts[i].l[7] &= 0x1fffffffu;
    }
    Fr *d_a, *d_b, *d_o, *d_r;
    CK(hipMalloc(&d_a, sizeof(Fr) * groups));
    CK(hipMalloc(&d_b, sizeof(Fr) * groups));
    CK(hipMalloc(&d_o, sizeof(Fr) * groups));
    CK(hipMalloc(&d_r, sizeof(Fr) * groups));
    CK(hipMalloc(&g_cts, sizeof(Fr) * kMimcRounds));
    CK(hipMemcpy(d_a, a.data(), sizeof(Fr) * groups, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b, b.data(), sizeof(Fr) * groups, hipMemcpyHostToDevice));
    CK(hipMemcpy(g_cts, cts.data(), sizeof(Fr) * kMimcRounds, hipMemcpyHostToDevice));
    // ---- the product against the one-lane arithmetic (mod p: the lane form returns a value < 2p)
    const int check_groups = 4096;
    k_check<<<check_groups * 8 / 64, 64>>>(d_a, d_b, d_o, d_r);
    CK(hipDeviceSynchronize());
    std::vector<Fr> got(check_groups), ref(check_groups);
    CK(hipMemcpy(got.data(), d_o, sizeof(Fr) * check_groups, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ref.data(), d_r, sizeof(Fr) * check_groups, hipMemcpyDeviceToHost));
    int bad = 0;
    const uint32_t mod[8] = GKR_MOD_LIMBS;
    for (int i = 0; i < check_groups; ++i) {
        Fr g = got[i];
        for (int pass = 0; pass < 2; ++pass) {   // subtract p while >= p
            bool ge = true;
            for (int j = 7; j >= 0; --j)
                if (g.l[j] != mod[j]) { ge = g.l[j] > mod[j]; break; }
            if (!ge) break;
            uint64_t br = 0;
            for (int j = 0; j < 8; ++j) {
                const uint64_t d = (uint64_t)g.l[j] - mod[j] - br;
                g.l[j] = (uint32_t)d;
                br = (d >> 63) & 1;
            }
        }
        bool same = true;
        for (int j = 0; j < 8; ++j) same &= g.l[j] == ref[i].l[j];
        if (!same && bad++ < 4) printf("MISMATCH group %d\n", i);
    }
    printf("8-lane Montgomery product against the one-lane product: %d of %d groups differ\n", bad, check_groups);
    // ---- timing
    g_io = d_a;
    for (int threads : {64, 256}) {
        g_threads = threads;
        for (int blocks : {1, 256, 1024, 4096}) {
            const int reps_mul = 4096, reps_hash = 16;
            double ms = time_kernel(launch_mul, blocks, reps_mul);
            const double n_groups = (double)blocks * threads / 8;
            printf("block %3d x %5d blocks: product chain  %7.3f us per product (per group), %9.3e products/s chip-wide\n", threads, blocks, ms * 1e3 / reps_mul,
                   n_groups * reps_mul / (ms * 1e-3));
            ms = time_kernel(launch_hash, blocks, reps_hash);
            printf("block %3d x %5d blocks: MiMC7 permutation %7.1f us each (per group), %9.3e permutations/s chip-wide; a 2-element round vector = 2 permutations\n",
                   threads, blocks, ms * 1e3 / reps_hash, n_groups * reps_hash / (ms * 1e-3));
        }
    }
    return bad ? 1 : 0;
}
