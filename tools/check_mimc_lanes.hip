// The eight-lanes-per-element MiMC7 (gkr_amd/csrc/mimc_lanes.h) against the one-lane code of mimc7.h, with the real constants:
// random and edge round vectors of 1..3 elements.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gkr_amd/csrc tools/check_mimc_lanes.hip gkr_amd/csrc/keccak.cpp -o tools/bin/check_mimc_lanes && tools/bin/check_mimc_lanes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "keccak.h"
#include "mimc_lanes.h"
using namespace gkr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// vectors: [groups][3] canonical; len[g] in 1..3 (the hash takes the LAST len elements, like the round vectors)
__global__ void __launch_bounds__(64) k_lanes(const Fr* __restrict__ vec, const uint32_t* __restrict__ len, const Fr* __restrict__ cts, Fr* __restrict__ out) {
    const uint32_t g = (blockIdx.x * 64u + threadIdx.x) >> 3;
    const lanes::Ctx c = lanes::make_ctx();
    const Fr* v = vec + (size_t)g * 3;
    const uint32_t n = len[g];
    // groups of one wave may have different lengths: the wave runs the longest, shorter groups repeat their last element and
    // drop the result (every lane stays active: the ballots and DPP moves of the lane arithmetic need the whole wave)
    uint32_t longest = n;
    for (int off = 32; off >= 8; off >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, off, 64));
    uint32_t r = 0, result = 0;
    {
        // multi_hash written out so that a group can stop early
        for (uint32_t i = 0; i < longest; ++i) {
            const uint32_t ii = i < n ? i : n - 1u;
            const uint32_t a = lanes::cond_sub(lanes::mont_mul(v[3 - n + ii].l[c.j], c.r2j, c), c.pj, c);
            const uint32_t h = lanes::permutation(a, r, cts, c);
            uint32_t nr = lanes::add3(r, a, h, c);
            nr = lanes::cond_sub(lanes::cond_sub(nr, c.two_pj, c), c.pj, c);
            if (i < n) r = nr;
        }
        const uint32_t one = c.j == 0 ? 1u : 0u;
        result = lanes::cond_sub(lanes::mont_mul(r, one, c), c.pj, c);
    }
    out[g].l[c.j] = result;
}
__global__ void __launch_bounds__(64) k_one(const Fr* __restrict__ vec, const uint32_t* __restrict__ len, const Fr* __restrict__ cts, Fr* __restrict__ out, uint32_t groups) {
    const uint32_t g = blockIdx.x * 64u + threadIdx.x;
    if (g >= groups) return;
    const uint32_t n = len[g];
    out[g] = mimc7_multi_hash(vec + (size_t)g * 3 + (3 - n), (int)n, cts);
}

int main() {
    const uint32_t groups = 8192;
    std::vector<Fr> vec(groups * 3), cts(kMimcRounds);
    std::vector<uint32_t> len(groups);
    mimc7_make_constants(cts.data());
    const uint32_t mod[8] = GKR_MOD_LIMBS;
    uint64_t st = 0x9876543;
    auto rnd = [&] { st = st * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(st >> 32); };
    for (uint32_t g = 0; g < groups; ++g) {
        len[g] = 1 + rnd() % 3;
        for (int e = 0; e < 3; ++e) {
            Fr& x = vec[g * 3 + e];
            const uint32_t kind = rnd() % 8;
            for (int j = 0; j < 8; ++j) x.l[j] = kind == 0 ? 0u : (kind == 1 ? mod[j] : rnd());
            if (kind == 1) x.l[0] -= 1 + (rnd() % 3);     // p - 1, p - 2, p - 3
            else if (kind == 2) { for (int j = 1; j < 8; ++j) x.l[j] = 0; }   // small
            else x.l[7] &= 0x1fffffffu;                   // < 2^253 < p
        }
    }
    Fr *d_vec, *d_cts, *d_a, *d_b;
    uint32_t* d_len;
    CK(hipMalloc(&d_vec, sizeof(Fr) * vec.size()));
    CK(hipMalloc(&d_cts, sizeof(Fr) * kMimcRounds));
    CK(hipMalloc(&d_a, sizeof(Fr) * groups));
    CK(hipMalloc(&d_b, sizeof(Fr) * groups));
    CK(hipMalloc(&d_len, 4 * groups));
    CK(hipMemcpy(d_vec, vec.data(), sizeof(Fr) * vec.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cts, cts.data(), sizeof(Fr) * kMimcRounds, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_len, len.data(), 4 * groups, hipMemcpyHostToDevice));
    k_lanes<<<groups / 8, 64>>>(d_vec, d_len, d_cts, d_a);
    k_one<<<groups / 64, 64>>>(d_vec, d_len, d_cts, d_b, groups);
    CK(hipDeviceSynchronize());
    std::vector<Fr> a(groups), b(groups);
    CK(hipMemcpy(a.data(), d_a, sizeof(Fr) * groups, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d_b, sizeof(Fr) * groups, hipMemcpyDeviceToHost));
    int bad = 0;
    for (uint32_t g = 0; g < groups; ++g) {
        bool same = true;
        for (int j = 0; j < 8; ++j) same &= a[g].l[j] == b[g].l[j];
        if (!same && bad++ < 5) printf("MISMATCH group %u (len %u)\n", g, len[g]);
    }
    printf("eight-lane multi_hash against the one-lane code: %d of %u round vectors differ\n", bad, groups);
    return bad ? 1 : 0;
}
