// MiMC7-91 with EIGHT lanes per field element (device only): lane j of an aligned group of eight holds limb j (32 bits) of
// every operand, so a permutation's 364 dependent Montgomery products are chains of ~12 dependent VALU instructions per
// limb step instead of a 256-instruction chain on one lane: 0.44 us per product, 165 us per permutation on MI355X
// (tools/ubench_mimc_lanes.hip, profiles/r04/c_*; one lane: 1.25 us / 455 us).  The host's cores hash a transcript in
// 3.6 us (mimc_adx.cpp), so this is no path for a lone proof; it is the path for transcripts the host has NO core for
// (a rank with two host threads, tables so small that the step is the host's hashing): their chains run beside the
// streaming passes of other sumchecks, on ALUs those passes leave idle.
//
//   product: operand scanning, one limb of b per step: acc += a_j * b_i; q = acc_0 * (-p^-1); acc += p_j * q; then the
//            accumulators move one lane down (division by 2^32); carries deferred in a 96-bit per-lane accumulator and
//            resolved once at the end with a ballot carry-lookahead.  Cross-lane moves are DPP (row_share / row_shl inside
//            16-lane rows): they sit on the dependent chain.
//   bounds:  operands below 3p give a product below 2.7p (a b / 2^256 + p) and no intermediate reaches 2^256
//            (4p = 0.756 * 2^256); sums are brought back with conditional subtractions of 2p or p (borrow lookahead).
// Reference call sites replaced: Mimc7::new(91) rust/src/gkr/sumcheck.rs:45; multi_hash sumcheck.rs:84,129,152
// (mimc-rs: r = key; for a in arr { r += a + hash(a, r) }).  Checked against mimc7.h's one-lane code on the device
// (gkr_selftest_lanes_hash) and through the sumcheck parity tests with the device-hashed groups forced on.
#pragma once
#include "mimc7.h"

namespace gkr {
namespace lanes {

// DPP controls: row_shl:n = 0x100 + n, row_shr:n = 0x110 + n, row_share:n = 0x150 + n
__device__ __forceinline__ uint32_t dpp_row_shl1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x101, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_row_shr1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true); }
// limb I of the group (I a constant), in every lane of the group; `upper`: the group is the upper half of its 16-lane row
template <int I>
__device__ __forceinline__ uint32_t group_bcast(uint32_t x, bool upper) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 + I, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x158 + I, 0xf, 0xf, true);
    return upper ? hi : lo;
}

struct Ctx {
    uint32_t j;        // this lane's limb index
    bool upper;        // the group is the upper half of its 16-lane row
    uint32_t pj;       // limb j of p
    uint32_t two_pj;   // limb j of 2p
    uint32_t r2j;      // limb j of 2^512 mod p (to Montgomery form)
};
__device__ __forceinline__ Ctx make_ctx() {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    constexpr uint32_t r2[8] = GKR_R2_LIMBS;
    Ctx c;
    const uint32_t lane = threadIdx.x & 63u;
    c.j = lane & 7u;
    c.upper = (lane & 8u) != 0u;
    uint32_t pj = 0, pjm = 0, r2j = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        pj = c.j == (uint32_t)i ? p[i] : pj;
        r2j = c.j == (uint32_t)i ? r2[i] : r2j;
        if (i) pjm = c.j == (uint32_t)i ? p[i - 1] : pjm;
    }
    c.pj = pj;
    c.two_pj = (pj << 1) | (c.j ? pjm >> 31 : 0u);   // p < 2^254: no overflow
    c.r2j = r2j;
    return c;
}

// per-lane value v = limb + 2^32 * extra (extra small) -> 32-bit limbs of the same integer (< 2^256: no carry out of the
// group's top lane).  One shift of the deferred carries, then generate / propagate through a ballot.
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

// a b 2^-256 mod p for a, b < 3p, one limb per lane: < 2.7p
__device__ __forceinline__ uint32_t mont_mul(uint32_t a, uint32_t b, const Ctx& c) {
    uint32_t bi[8];
    bi[0] = group_bcast<0>(b, c.upper);
    bi[1] = group_bcast<1>(b, c.upper);
    bi[2] = group_bcast<2>(b, c.upper);
    bi[3] = group_bcast<3>(b, c.upper);
    bi[4] = group_bcast<4>(b, c.upper);
    bi[5] = group_bcast<5>(b, c.upper);
    bi[6] = group_bcast<6>(b, c.upper);
    bi[7] = group_bcast<7>(b, c.upper);
    uint64_t acc = 0;
    uint32_t ex = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mac96(acc, ex, a, bi[i]);
        const uint32_t q = group_bcast<0>((uint32_t)acc, c.upper) * GKR_INV32;
        mac96(acc, ex, c.pj, q);
        // one limb down: lane j takes the low word of lane j + 1 (the group's last lane takes nothing)
        uint32_t from_above = dpp_row_shl1((uint32_t)acc);
        if (c.j == 7) from_above = 0;
        acc = (acc >> 32) + ((uint64_t)ex << 32) + from_above;
        ex = 0;
    }
    return resolve_carries(acc, c.j);
}

// x + y + z (the sum < 2^256), one limb per lane
__device__ __forceinline__ uint32_t add3(uint32_t x, uint32_t y, uint32_t z, const Ctx& c) { return resolve_carries((uint64_t)x + y + z, c.j); }

// x - m if x >= m, else x (m's limb j in mj): borrow lookahead the way resolve_carries does carries
__device__ __forceinline__ uint32_t cond_sub(uint32_t x, uint32_t mj, const Ctx& c) {
    const uint32_t d = x - mj;
    const uint64_t g = __ballot(x < mj), p = __ballot(d == 0u);
    const uint64_t gs = (g << 1) & 0xfefefefefefefefeull;
    const uint64_t bin = ((gs + p) ^ p);                        // lanes a borrow arrives at
    const uint32_t r = d - (uint32_t)((bin >> (threadIdx.x & 63u)) & 1u);
    // a borrow out of the group's top lane: x < m, keep x.  The top lane generated one, or passed one on
    const uint64_t out = (g | (p & bin)) & 0x8080808080808080ull;
    const uint32_t grp = (threadIdx.x & 63u) >> 3;
    return ((out >> (grp * 8u + 7u)) & 1u) ? x : r;
}

// is the group's value zero (every lane of the group gets the answer)
__device__ __forceinline__ bool is_zero(uint32_t x) {
    const uint64_t nz = __ballot(x != 0u);
    const uint32_t grp = (threadIdx.x & 63u) >> 3;
    return ((nz >> (grp * 8u)) & 0xffull) == 0ull;
}

// hash(x, k) of mimc7.h: x, k Montgomery form and below p; cts in Montgomery form (global memory); result below 2p
__device__ __forceinline__ uint32_t permutation(uint32_t x, uint32_t k, const Fr* __restrict__ cts, const Ctx& c) {
    uint32_t h = 0;
    for (int i = 0; i < kMimcRounds; ++i) {
        // t = h + k + c_i < 2.25p + p + p: once minus 2p leaves it below 2.25p
        uint32_t t = i == 0 ? add3(x, k, 0u, c) : add3(h, k, cts[i].l[c.j], c);
        t = cond_sub(t, c.two_pj, c);
        const uint32_t t2 = mont_mul(t, t, c);     // < 2.7p (t < 3p)
        const uint32_t t4 = mont_mul(t2, t2, c);   // < 2.38p
        const uint32_t t6 = mont_mul(t4, t2, c);   // < 2.22p
        h = mont_mul(t6, t, c);                    // < 2.25p
    }
    return cond_sub(add3(h, k, 0u, c), c.two_pj, c);   // h + k < 3.25p -> below 2p
}

// multi_hash(arr, 0) of mimc7.h for one group: elem(i) gives limb j of the i-th canonical element; returns limb j of the
// canonical result
template <typename Elem>
__device__ __forceinline__ uint32_t multi_hash(int n, Elem elem, const Fr* __restrict__ cts, const Ctx& c) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) {
        const uint32_t a = cond_sub(mont_mul(elem(i), c.r2j, c), c.pj, c);   // to Montgomery form: < 2p -> below p
        const uint32_t h = permutation(a, r, cts, c);                       // < 2p
        r = add3(r, a, h, c);                                                // < 4p
        r = cond_sub(cond_sub(r, c.two_pj, c), c.pj, c);                     // below p
    }
    const uint32_t one = c.j == 0 ? 1u : 0u;
    return cond_sub(mont_mul(r, one, c), c.pj, c);   // out of Montgomery form: r / 2^256 + p < 2p -> canonical
}

// limb j of a value every lane of the group holds in full
__device__ __forceinline__ uint32_t limb_of(const Fr& x, uint32_t j) {
    uint32_t v = x.l[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) v = j == (uint32_t)i ? x.l[i] : v;
    return v;
}
// the full value in every lane of the group from one limb per lane
__device__ __forceinline__ Fr gather(uint32_t mine, bool upper) {
    Fr x;
    x.l[0] = group_bcast<0>(mine, upper);
    x.l[1] = group_bcast<1>(mine, upper);
    x.l[2] = group_bcast<2>(mine, upper);
    x.l[3] = group_bcast<3>(mine, upper);
    x.l[4] = group_bcast<4>(mine, upper);
    x.l[5] = group_bcast<5>(mine, upper);
    x.l[6] = group_bcast<6>(mine, upper);
    x.l[7] = group_bcast<7>(mine, upper);
    return x;
}

}  // namespace lanes
}  // namespace gkr
