// BN254-Fr products on the FP64 pipe (device only) -- an EXPERIMENT, not part of the library (tools/ubench_f64mont.hip).
// Result on MI355X (profiles/r02/n_ubench_f64mont.txt): bit-exact on 2^20 random and edge operand pairs; a dependent
// chain on one wave runs 1.47x faster than the v_mad_u64_u32 product (834 against 1223 ns), but with the chip full the
// integer product is the faster one (1.16e11 against 7.7e10 products/s): its cost on one wave is the latency of its
// carry chains, which other waves hide, not its issue rate.  So this is no lever for the throughput kernels.
//
// The idea: v_fma_f64 issues every 4 cycles.  With 52-bit limbs held in doubles, one FMA pair splits a limb product
// exactly:  hi = fma(a, b, 2^104)  rounds a b to a multiple of 2^52 (H = the rounded quotient, in hi's mantissa),
// lo = fma(a, b, (2^104 + 3 2^51) - hi) = a b - H 2^52 + 3 2^51  is exact and lies in [2^52, 2^53): its mantissa is
// L + 2^51 with L = a b - H 2^52 in [-2^51, 2^51].  IEEE bit patterns are linear in the mantissa within and across
// the binade, so the products are accumulated as INTEGER sums of the bit patterns and the constants come off at the
// end (Emmart et al.'s "DPF" scheme, in round-to-nearest form: no rounding-mode switch).  A Montgomery product in
// radix 2^52 (5 limbs, R = 2^260 -- the radix of the host's IFMA code, mimc_ifma.cpp) is then 55 such splits:
// 110 FMAs + 110 64-bit adds instead of 128 + 128 quarter-rate multiply-adds with their carry chains.
#pragma once
#include "fr32.h"

namespace gkr {
namespace f64m {

struct L52 {   // five 52-bit limbs of a value below 2^260, as exactly representable doubles
    double l[5];
};

__device__ __forceinline__ double u52_to_double(uint64_t x) {   // x < 2^52, exact
    return __longlong_as_double((long long)(x | 0x4330000000000000ull)) - 4503599627370496.0;
}

// canonical 8 x 32-bit -> 5 x 52-bit doubles
__device__ __forceinline__ L52 to_l52(const Fr& a) {
    const uint64_t w0 = a.l[0] | ((uint64_t)a.l[1] << 32), w1 = a.l[2] | ((uint64_t)a.l[3] << 32);
    const uint64_t w2 = a.l[4] | ((uint64_t)a.l[5] << 32), w3 = a.l[6] | ((uint64_t)a.l[7] << 32);
    constexpr uint64_t M = (1ull << 52) - 1;
    L52 r;
    r.l[0] = u52_to_double(w0 & M);
    r.l[1] = u52_to_double(((w0 >> 52) | (w1 << 12)) & M);
    r.l[2] = u52_to_double(((w1 >> 40) | (w2 << 24)) & M);
    r.l[3] = u52_to_double(((w2 >> 28) | (w3 << 36)) & M);
    r.l[4] = u52_to_double(w3 >> 16);
    return r;
}

constexpr double kC1 = 20282409603651670423947251286016.0;                       // 2^104
// bit patterns of 2^104 and of 2^52 (the binade the low parts land in)
constexpr uint64_t kC1Bits = 0x4670000000000000ull;
constexpr uint64_t kLoBits = 0x4330000000000000ull;

// the raw bit patterns of the split of a * b: add them up, take the constants off at the end
__device__ __forceinline__ void split_bits(double a, double b, uint64_t& hi_bits, uint64_t& lo_bits) {
    const double hi = __fma_rn(a, b, kC1);
    const double lo = __fma_rn(a, b, (kC1 - hi) + 6755399441055744.0);
    hi_bits = (uint64_t)__double_as_longlong(hi);
    lo_bits = (uint64_t)__double_as_longlong(lo);
}

// 2^52-radix constants of p
__device__ __forceinline__ void p52(double (&p)[5], double& pinv) {
    // p = 0x30644e72e131a029 b85045b68181585d 2833e84879b97091 43e1f593f0000001
    p[0] = u52_to_double(0x1f593f0000001ull);
    p[1] = u52_to_double(0x4879b9709143eull);
    p[2] = u52_to_double(0x181585d2833e8ull);
    p[3] = u52_to_double(0xa029b85045b68ull);
    p[4] = u52_to_double(0x30644e72e131ull);
    pinv = u52_to_double(0x1f593efffffffull);   // -p^-1 mod 2^52
}

// a b 2^-260 mod p, canonical, as five 52-bit integer limbs (a, b < 2^260 with 52-bit limbs; a b < 2^260 p)
__device__ __forceinline__ void mont_mul_limbs(const L52& a, const L52& b, uint64_t (&out)[5]) {
    constexpr uint64_t M = (1ull << 52) - 1;
    uint64_t c[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) c[k] = 0;
    // columns of a b: c[k] += L(i, j) for i + j = k, c[k + 1] += H(i, j); bit patterns, constants taken off below
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            uint64_t hb, lb;
            split_bits(a.l[i], b.l[j], hb, lb);
            c[i + j] += lb;
            c[i + j + 1] += hb;
        }
    // column k holds n_lo(k) low parts and n_hi(k) = n_lo(k - 1) high parts; L = mantissa - 2^51
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const int nlo = k < 5 ? k + 1 : (k < 9 ? 9 - k : 0), nhi = k >= 1 ? (k - 1 < 5 ? k : 10 - k) : 0;
        c[k] -= (uint64_t)nlo * (kLoBits + (1ull << 51)) + (uint64_t)nhi * kC1Bits;
    }
    double p[5], pinv;
    p52(p, pinv);
    // five Montgomery steps: m = c[k] pinv mod 2^52, c += m p 2^(52 k), the carry moves up
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        uint64_t hb, lb;
        split_bits(u52_to_double(c[k] & M), pinv, hb, lb);
        const uint64_t m = (lb - kLoBits - (1ull << 51)) & M;
        const double md = u52_to_double(m);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            split_bits(md, p[j], hb, lb);
            c[k + j] += lb - kLoBits - (1ull << 51);
            c[k + j + 1] += hb - kC1Bits;
        }
        c[k + 1] += (uint64_t)((long long)c[k] >> 52);   // c[k] is a multiple of 2^52 now (possibly negative before the add)
    }
    // c[5..9]: the result, limbs not yet normalised (signed excess): carry up, then one conditional subtraction
    uint64_t t[5];
    long long carry = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long v = (long long)c[5 + k] + carry;
        t[k] = (uint64_t)v & M;
        carry = v >> 52;
    }
    t[4] = (uint64_t)((long long)c[9] + carry);   // value < 2 p < 2^255: the top limb holds what is left (< 2^47)
    constexpr uint64_t P[5] = {0x1f593f0000001ull, 0x4879b9709143eull, 0x181585d2833e8ull, 0xa029b85045b68ull, 0x30644e72e131ull};
    uint64_t d[5];
    long long br = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long v = (long long)t[k] - (long long)P[k] + br;
        d[k] = (uint64_t)v & M;
        br = v >> 52;
    }
    const long long top = (long long)t[4] - (long long)P[4] + br;
    d[4] = (uint64_t)top;
    const bool keep = top < 0;   // t < p
#pragma unroll
    for (int k = 0; k < 5; ++k) out[k] = keep ? t[k] : d[k];
}

__device__ __forceinline__ Fr limbs_to_fr(const uint64_t (&l)[5]) {
    const uint64_t w0 = l[0] | (l[1] << 52), w1 = (l[1] >> 12) | (l[2] << 40), w2 = (l[2] >> 24) | (l[3] << 28), w3 = (l[3] >> 36) | (l[4] << 16);
    Fr r;
    r.l[0] = (uint32_t)w0; r.l[1] = (uint32_t)(w0 >> 32); r.l[2] = (uint32_t)w1; r.l[3] = (uint32_t)(w1 >> 32);
    r.l[4] = (uint32_t)w2; r.l[5] = (uint32_t)(w2 >> 32); r.l[6] = (uint32_t)w3; r.l[7] = (uint32_t)(w3 >> 32);
    return r;
}

// a b 2^-260 mod p
__device__ __forceinline__ Fr mont_mul260(const L52& a, const L52& b) {
    uint64_t l[5];
    mont_mul_limbs(a, b, l);
    return limbs_to_fr(l);
}

}  // namespace f64m
}  // namespace gkr
