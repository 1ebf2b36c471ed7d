// The host's share of a multi-round pass of the plain sumcheck (host_pass_scalar / gkr_ifma_pass: rust/src/gkr/sumcheck.rs
// :158-214 with the Fiat-Shamir challenge of :84,129,152) ON THE DEVICE, for sumchecks the host has no core for.
//
// Eight lanes per sumcheck, eight sumchecks per wave.  The MiMC7 chain -- what the time goes into: 2 permutations per
// round, 165 us each -- runs one limb per lane (mimc_lanes.h); the bookkeeping around it (the halves' sums, the fold of the
// 2^J sums by the challenge, the 2^J weights of the next fold pass) is ordinary one-lane arithmetic with the vectors in
// LDS, their entries dealt over the group's eight lanes.  Same field elements as the host's pass: every value is reduced
// to its canonical form where the host reduces it.
//
// The records (sums in, written by k_mle_sub_reduce / k_mle_multifold_small / a fused publish) and the weights (out, read by
// k_mle_fold_plan / the fold kernels) are where the host path has them -- pinned host memory --, so no other kernel
// changes and a sumcheck can be hashed on either side; the round outputs go to a pinned staging copy of the caller's arrays.
#include <hip/hip_runtime.h>

#include "dev_util.h"
#include "kernels.h"
#include "mimc_lanes.h"

namespace gkr {

// xor-shuffle tree over the eight lanes of a group: every lane ends with the modular sum of the eight values
__device__ __forceinline__ Fr group_sum(Fr x) {
#pragma unroll
    for (int off = 1; off <= 4; off <<= 1) {
        Fr o;
#pragma unroll
        for (int l = 0; l < 8; ++l) o.l[l] = (uint32_t)__shfl_xor((int)x.l[l], off, 64);
        x = fr_add(x, o);
    }
    return x;
}

// grid = ceil(count / 8), block = 64.  rec / weights / dep_last / out_*: indexed by sumcheck (the caller passes the
// pointers of the launch's first sumcheck); round0: global index of the pass's first round, n_out: rounds per sumcheck in
// the output arrays; final_pass: the pass's last round is the sumcheck's last (length rule of sumcheck.rs:206-207:
// two coefficients iff the table depends on x_n -- dep_last, written here from pass 0's record)
__global__ void __launch_bounds__(64) k_mle_pass_hash_lanes(const MleHostRecSub* __restrict__ rec, uint32_t count, uint32_t J, uint32_t round0,
                                                            uint32_t n_out, uint32_t final_pass, uint32_t first_pass, const Fr* __restrict__ cts,
                                                            uint32_t* __restrict__ dep_last, Fr* __restrict__ weights, Fr* __restrict__ out_coeffs,
                                                            uint32_t* __restrict__ out_len, Fr* __restrict__ out_r) {
    __shared__ Fr s_S[8][kMleMaxSub];
    __shared__ Fr s_W[2][8][kMleMaxSub];
    const lanes::Ctx c = lanes::make_ctx();
    const uint32_t grp = (threadIdx.x & 63u) >> 3, j = c.j;
    const uint32_t b_raw = blockIdx.x * 8u + grp;
    const bool live = b_raw < count;
    const uint32_t b = live ? b_raw : count - 1u;   // (a group past the end repeats the last sumcheck and stores nothing)
    const uint32_t nsub = 1u << J;
    for (uint32_t e = j; e < nsub; e += 8u) s_S[grp][e] = load_fr(&rec[b].sums[e]);
    uint32_t dep = 0;
    if (final_pass) dep = first_pass ? rec[b].dep : dep_last[b];
    if (first_pass && live && j == 0) dep_last[b] = rec[b].dep;
    __syncthreads();
    Fr rm[kMlePassMaxRounds];
    for (uint32_t t = 0; t < J; ++t) {
        const uint32_t half = 1u << (J - t - 1u);
        Fr lo = fr_zero(), hi = fr_zero();
        for (uint32_t e = j; e < half; e += 8u) {
            lo = fr_add(lo, s_S[grp][e]);
            hi = fr_add(hi, s_S[grp][half + e]);
        }
        lo = group_sum(lo);
        hi = group_sum(hi);
        const Fr d = fr_sub(hi, lo);
        const uint32_t ln = (final_pass && t == J - 1u) ? (dep ? 2u : 1u) : (fr_is_zero(d) ? 1u : 2u);
        // the round vector's hash: [d, lo] or [lo]; the wave runs the longest vector of its groups, a group with the
        // shorter one repeats its element and keeps the state it had (every lane stays active for the ballots and DPP moves)
        uint32_t longest = ln;
#pragma unroll
        for (int off = 8; off <= 32; off <<= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)longest, off, 64);
            longest = o > longest ? o : longest;
        }
        const uint32_t e_first = lanes::limb_of(ln == 2u ? d : lo, j), e_lo = lanes::limb_of(lo, j);
        uint32_t r = 0;
        for (uint32_t i = 0; i < longest; ++i) {
            const uint32_t elem = i == 0 ? e_first : e_lo;
            const uint32_t a = lanes::cond_sub(lanes::mont_mul(elem, c.r2j, c), c.pj, c);
            const uint32_t h = lanes::permutation(a, r, cts, c);
            uint32_t nr = lanes::add3(r, a, h, c);
            nr = lanes::cond_sub(lanes::cond_sub(nr, c.two_pj, c), c.pj, c);
            r = i < ln ? nr : r;
        }
        const uint32_t one = j == 0 ? 1u : 0u;
        const Fr rc = lanes::gather(lanes::cond_sub(lanes::mont_mul(r, one, c), c.pj, c), c.upper);
        if (live && j == 0) {
            const size_t at = (size_t)b * n_out + round0 + t;
            store_fr(out_coeffs + at * 2, ln == 2u ? d : fr_zero());
            store_fr(out_coeffs + at * 2 + 1, lo);
            out_len[at] = ln;
            store_fr(out_r + at, rc);
        }
        rm[t] = to_mont(rc);
        for (uint32_t e = j; e < half; e += 8u) {
            const Fr x = s_S[grp][e];
            s_S[grp][e] = fr_add(x, mont_mul(fr_sub(s_S[grp][half + e], x), rm[t]));
        }
        __syncthreads();
    }
    if (!weights) return;
    // w_b = prod_t (bit_t(b) ? r_t : 1 - r_t), bit_0 = most significant, Montgomery form
    Fr one_c = fr_zero();
    one_c.l[0] = 1u;
    const Fr one_m = to_mont(one_c);
    if (j == 0) s_W[0][grp][0] = one_m;
    __syncthreads();
    uint32_t cur = 1, src = 0;
    for (uint32_t t = 0; t < J; ++t) {
        const Fr nr = fr_sub(one_m, rm[t]);
        for (uint32_t e = j; e < cur; e += 8u) {
            const Fr x = s_W[src][grp][e];
            s_W[src ^ 1u][grp][2u * e + 1u] = mont_mul(x, rm[t]);
            s_W[src ^ 1u][grp][2u * e] = mont_mul(x, nr);
        }
        __syncthreads();
        cur <<= 1;
        src ^= 1u;
    }
    if (live)
        for (uint32_t e = j; e < nsub; e += 8u) store_fr(weights + (size_t)b * kMleMaxSub + e, s_W[src][grp][e]);
}

void launch_mle_pass_hash_lanes(const MleHostRecSub* rec, uint32_t count, uint32_t J, uint32_t round0, uint32_t n_out, bool final_pass,
                                bool first_pass, const Fr* cts, uint32_t* dep_last, Fr* weights, Fr* out_coeffs, uint32_t* out_len, Fr* out_r,
                                hipStream_t s) {
    hipLaunchKernelGGL(k_mle_pass_hash_lanes, dim3((count + 7u) / 8u), dim3(64), 0, s, rec, count, J, round0, n_out, final_pass ? 1u : 0u,
                       first_pass ? 1u : 0u, cts, dep_last, weights, out_coeffs, out_len, out_r);
}

}  // namespace gkr
