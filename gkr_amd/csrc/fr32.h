// BN254 scalar field on 8 x 32-bit limbs for CDNA4 (and, for unit tests, the host).
//
// The reference's field is halo2curves bn256::Fr (rust/src/aggregator.rs:9,
// rust/Cargo.toml:21); values cross every boundary as 32-byte little-endian
// canonical integers (rust/src/gkr/sumcheck.rs:10-22).  On the device a value is
// eight little-endian u32 limbs -- the same 32 bytes -- so tables are loaded with
// two global_load_dwordx4 per element and never re-encoded.
//
// Arithmetic convention used by every kernel: tables stay CANONICAL; only
// multipliers (challenges r, the W copies of the layer kernel, MiMC state) are
// kept in Montgomery form xR, R = 2^256, because
//     mont_mul(canonical a, Montgomery bR) = a * b   (canonical).
// Sums are accumulated unreduced in 288-bit (9-limb) accumulators -- 2^32 canonical
// values fit -- and reduced once per kernel.
#pragma once
#include <stdint.h>

#if defined(__HIP__)
#define GKR_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define GKR_HD inline
#endif

namespace gkr {

struct alignas(16) Fr {
    uint32_t l[8];
};

// r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
#define GKR_MOD_LIMBS                                                                             \
    { 0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u,  \
      0x30644e72u }
// R^2 mod r, R = 2^256
#define GKR_R2_LIMBS                                                                              \
    { 0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u,  \
      0x0216d0b1u }
// R mod r (Montgomery form of 1)
#define GKR_R1_LIMBS                                                                              \
    { 0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu,  \
      0x0e0a77c1u }
#define GKR_INV32 0xefffffffu  // -r^{-1} mod 2^32

GKR_HD uint32_t mod_limb(int i) {
    constexpr uint32_t m[8] = GKR_MOD_LIMBS;
    return m[i];
}

GKR_HD Fr fr_zero() {
    Fr z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.l[i] = 0;
    return z;
}

GKR_HD Fr fr_r2() {
    constexpr uint32_t m[8] = GKR_R2_LIMBS;
    Fr z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.l[i] = m[i];
    return z;
}

GKR_HD Fr fr_mont_one() {
    constexpr uint32_t m[8] = GKR_R1_LIMBS;
    Fr z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.l[i] = m[i];
    return z;
}

GKR_HD bool fr_is_zero(const Fr& a) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc |= a.l[i];
    return acc == 0;
}

GKR_HD bool fr_eq(const Fr& a, const Fr& b) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc |= a.l[i] ^ b.l[i];
    return acc == 0;
}

// out = a - r; returns the final borrow (1 iff a < r)
GKR_HD uint32_t sub_mod_raw(const Fr& a, Fr& out) {
    constexpr uint32_t m[8] = GKR_MOD_LIMBS;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t d = (uint64_t)a.l[i] - m[i] - borrow;
        out.l[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    return (uint32_t)borrow;
}

// a in [0, 2r) -> [0, r)
GKR_HD Fr fr_reduce_once(const Fr& a) {
    Fr d;
    uint32_t borrow = sub_mod_raw(a, d);
    Fr out;
#pragma unroll
    for (int i = 0; i < 8; ++i) out.l[i] = borrow ? a.l[i] : d.l[i];
    return out;
}

GKR_HD bool fr_is_canonical(const Fr& a) {
    Fr d;
    return sub_mod_raw(a, d) != 0;
}

GKR_HD Fr fr_add(const Fr& a, const Fr& b) {
    Fr s;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.l[i] + b.l[i] + carry;
        s.l[i] = (uint32_t)t;
        carry = t >> 32;
    }
    return fr_reduce_once(s);  // a, b < r < 2^254: no carry out of 256 bits
}

GKR_HD Fr fr_sub(const Fr& a, const Fr& b) {
    constexpr uint32_t m[8] = GKR_MOD_LIMBS;
    Fr d;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.l[i] - b.l[i] - borrow;
        d.l[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    const uint32_t mask = borrow ? 0xffffffffu : 0u;
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)d.l[i] + (m[i] & mask) + carry;
        d.l[i] = (uint32_t)t;
        carry = t >> 32;
    }
    return d;
}

// Montgomery product a * b * 2^-256 mod r; a, b < r.  Coarsely integrated operand
// scanning on 32-bit limbs; every (u64)x*y + c maps to one v_mad_u64_u32.
GKR_HD Fr mont_mul(const Fr& a, const Fr& b) {
    constexpr uint32_t m[8] = GKR_MOD_LIMBS;
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint64_t p = (uint64_t)a.l[j] * b.l[i] + t[j] + carry;
            t[j] = (uint32_t)p;
            carry = p >> 32;
        }
        uint64_t s = (uint64_t)t[8] + carry;
        t[8] = (uint32_t)s;
        t[9] = (uint32_t)(s >> 32);
        const uint32_t q = t[0] * GKR_INV32;
        uint64_t p = (uint64_t)q * m[0] + t[0];
        carry = p >> 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            p = (uint64_t)q * m[j] + t[j] + carry;
            t[j - 1] = (uint32_t)p;
            carry = p >> 32;
        }
        s = (uint64_t)t[8] + carry;
        t[7] = (uint32_t)s;
        t[8] = t[9] + (uint32_t)(s >> 32);
    }
    Fr out;
#pragma unroll
    for (int i = 0; i < 8; ++i) out.l[i] = t[i];
    // result < 2r and r < 2^254, so t[8] == 0 here
    return fr_reduce_once(out);
}

GKR_HD Fr to_mont(const Fr& a) { return mont_mul(a, fr_r2()); }

GKR_HD Fr from_mont(const Fr& a) {
    Fr one = fr_zero();
    one.l[0] = 1;
    return mont_mul(a, one);
}

// canonical * canonical -> canonical (two Montgomery products)
GKR_HD Fr fr_mul(const Fr& a, const Fr& b) { return mont_mul(to_mont(a), b); }

// T + r (H - T) with r in Montgomery form, T and H canonical: the table fold of
// partial_eval_i (rust/src/gkr/poly.rs:160-179) on a multilinear table.
GKR_HD Fr fr_fold(const Fr& lo, const Fr& hi, const Fr& r_mont) {
    return fr_add(lo, mont_mul(fr_sub(hi, lo), r_mont));
}

// ---------------------------------------------------------------- wide sums
// Unreduced accumulator: NL 32-bit limbs.  Acc<9> takes 2^32 canonical addends,
// Acc<10> takes 2^32 Acc<9> values.
template <int NL>
struct Acc {
    uint32_t l[NL];
};

template <int NL>
GKR_HD Acc<NL> acc_zero() {
    Acc<NL> a;
#pragma unroll
    for (int i = 0; i < NL; ++i) a.l[i] = 0;
    return a;
}

template <int NL>
GKR_HD void acc_add_fr(Acc<NL>& a, const Fr& x) {
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t t = (uint64_t)a.l[i] + x.l[i] + carry;
        a.l[i] = (uint32_t)t;
        carry = t >> 32;
    }
#pragma unroll
    for (int i = 8; i < NL; ++i) {
        uint64_t t = (uint64_t)a.l[i] + carry;
        a.l[i] = (uint32_t)t;
        carry = t >> 32;
    }
}

template <int NL, int NS>
GKR_HD void acc_add_acc(Acc<NL>& a, const Acc<NS>& x) {
    static_assert(NS <= NL, "source wider than destination");
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        uint64_t t = (uint64_t)a.l[i] + (i < NS ? x.l[i] : 0u) + carry;
        a.l[i] = (uint32_t)t;
        carry = t >> 32;
    }
}

// value of a wide accumulator mod r, canonical.  x = lo + hi * 2^256 with
// lo < 2^256 < 6r and hi < 2^64: lo by <= 5 subtractions, hi * 2^256 by one
// Montgomery product with R^2 (hi * R^2 * R^-1 = hi * R).
template <int NL>
GKR_HD Fr acc_reduce(const Acc<NL>& a) {
    static_assert(NL >= 8 && NL <= 10, "acc_reduce handles up to 64 overflow bits");
    Fr lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) lo.l[i] = a.l[i];
#pragma unroll
    for (int k = 0; k < 5; ++k) lo = fr_reduce_once(lo);
    Fr hi = fr_zero();
#pragma unroll
    for (int i = 8; i < NL; ++i) hi.l[i - 8] = a.l[i];
    return fr_add(lo, mont_mul(hi, fr_r2()));
}

}  // namespace gkr
