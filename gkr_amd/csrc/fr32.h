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

GKR_HD Fr fr_add_portable(const Fr& a, const Fr& b) {
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

GKR_HD Fr fr_sub_portable(const Fr& a, const Fr& b) {
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

// Montgomery product a * b * 2^-256 mod r; a, b < r.
//
// Portable form (host, and the reference for the device form): coarsely
// integrated operand scanning on 32-bit limbs.
GKR_HD Fr mont_mul_portable(const Fr& a, const Fr& b) {
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

#if defined(__HIP_DEVICE_COMPILE__)
// ---------------------------------------------------------------------------
// gfx950 forms.  Two facts shape them (measured, tools/ubench_fold.hip):
//  * hipcc turns the portable 64-bit-carry C into ~690 instructions per Montgomery
//    product (zero-extending moves around every 64-bit add);
//  * on gfx950 a VALU that reads a carry (VCC or an SGPR pair) needs two wait
//    states after the VALU that wrote it.  The pads are written into the asm
//    strings (s_nop 1); with several waves per SIMD they cost no throughput.
// Product scanning: one column at a time into a 96-bit accumulator (lo:hi in a
// VGPR pair, ex in a third VGPR); each partial product is ONE v_mad_u64_u32 whose
// carry-out feeds ex through ONE v_addc_co_u32.
// ---------------------------------------------------------------------------
__device__ inline __attribute__((always_inline)) void mac96(uint64_t& acc, uint32_t& ex, uint32_t x, uint32_t y) {
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2"
        : "+v"(acc), "+v"(ex), "=&s"(carry)
        : "v"(x), "v"(y));
}
// the first product of a column: ex is SET to the carry (saves clearing it between columns)
__device__ inline __attribute__((always_inline)) void mac96_first(uint64_t& acc, uint32_t& ex, uint32_t x, uint32_t y) {
    uint64_t carry;
    uint32_t e;
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, %2, 0, 0, %2"
        : "+v"(acc), "=v"(e), "=&s"(carry)
        : "v"(x), "v"(y));
    ex = e;
}
// the start of a column: {hi, lo} = {k1, k0} + a -- what the previous column carried on (its high word and its carry
// count) plus the accumulator's limb.  Two full-rate adds; written as a 64-bit sum the compiler assembles both operands
// in register pairs first (two moves) and adds them with v_lshl_add_u64 (half rate): four issue slots per column.
__device__ inline __attribute__((always_inline)) uint64_t column_start(uint32_t k0, uint32_t k1, uint32_t a) {
    uint32_t lo, hi;
    asm("v_add_co_u32_e32 %0, vcc, %2, %3\n\ts_nop 1\n\tv_addc_co_u32_e32 %1, vcc, 0, %4, vcc"
        : "=&v"(lo), "=v"(hi)
        : "v"(k0), "v"(a), "v"(k1)
        : "vcc");
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
// y wave-uniform (an SGPR or a constant the compiler puts in one)
__device__ inline __attribute__((always_inline)) void mac96_s(uint64_t& acc, uint32_t& ex, uint32_t x, uint32_t y_uniform) {
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2"
        : "+v"(acc), "+v"(ex), "=&s"(carry)
        : "v"(x), "s"(y_uniform));
}

// Interleaved forms: two / three INDEPENDENT accumulators advanced together.  The other chains'
// instructions fill the two wait states a carry needs (no pad for three chains, one s_nop 0 for
// two) and give the in-order wave independent work while a v_mad_u64_u32 result is in flight
// (measured: a dependent mad chain issues every ~10 cycles per wave, independent ones every ~5).
__device__ inline __attribute__((always_inline)) void mac96x2_s(uint64_t& accA, uint32_t& exA, uint32_t xA, uint32_t yA,
                                                                uint64_t& accB, uint32_t& exB, uint32_t xB, uint32_t yB) {
    uint64_t cA, cB;
    asm("v_mad_u64_u32 %0, %4, %6, %7, %0\n\t"
        "v_mad_u64_u32 %2, %5, %8, %9, %2\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %1, %4, 0, %1, %4\n\t"
        "v_addc_co_u32_e64 %3, %5, 0, %3, %5"
        : "+v"(accA), "+v"(exA), "+v"(accB), "+v"(exB), "=&s"(cA), "=&s"(cB)
        : "v"(xA), "s"(yA), "v"(xB), "s"(yB));
}
__device__ inline __attribute__((always_inline)) void mac96x3_s(uint64_t& accA, uint32_t& exA, uint32_t xA, uint32_t yA,
                                                                uint64_t& accB, uint32_t& exB, uint32_t xB, uint32_t yB,
                                                                uint64_t& accC, uint32_t& exC, uint32_t xC, uint32_t yC) {
    uint64_t cA, cB, cC;
    asm("v_mad_u64_u32 %0, %6, %9, %10, %0\n\t"
        "v_mad_u64_u32 %2, %7, %11, %12, %2\n\t"
        "v_mad_u64_u32 %4, %8, %13, %14, %4\n\t"
        "v_addc_co_u32_e64 %1, %6, 0, %1, %6\n\t"
        "v_addc_co_u32_e64 %3, %7, 0, %3, %7\n\t"
        "v_addc_co_u32_e64 %5, %8, 0, %5, %8"
        : "+v"(accA), "+v"(exA), "+v"(accB), "+v"(exB), "+v"(accC), "+v"(exC), "=&s"(cA), "=&s"(cB), "=&s"(cC)
        : "v"(xA), "s"(yA), "v"(xB), "s"(yB), "v"(xC), "s"(yC));
}

__device__ inline __attribute__((always_inline)) void mac96x4_s(uint64_t& accA, uint32_t& exA, uint32_t xA, uint32_t yA,
                                                                uint64_t& accB, uint32_t& exB, uint32_t xB, uint32_t yB,
                                                                uint64_t& accC, uint32_t& exC, uint32_t xC, uint32_t yC,
                                                                uint64_t& accD, uint32_t& exD, uint32_t xD, uint32_t yD) {
    uint64_t cA, cB, cC, cD;
    asm("v_mad_u64_u32 %0, %8, %12, %13, %0\n\t"
        "v_mad_u64_u32 %2, %9, %14, %15, %2\n\t"
        "v_mad_u64_u32 %4, %10, %16, %17, %4\n\t"
        "v_mad_u64_u32 %6, %11, %18, %19, %6\n\t"
        "v_addc_co_u32_e64 %1, %8, 0, %1, %8\n\t"
        "v_addc_co_u32_e64 %3, %9, 0, %3, %9\n\t"
        "v_addc_co_u32_e64 %5, %10, 0, %5, %10\n\t"
        "v_addc_co_u32_e64 %7, %11, 0, %7, %11"
        : "+v"(accA), "+v"(exA), "+v"(accB), "+v"(exB), "+v"(accC), "+v"(exC), "+v"(accD), "+v"(exD), "=&s"(cA), "=&s"(cB),
          "=&s"(cC), "=&s"(cD)
        : "v"(xA), "s"(yA), "v"(xB), "s"(yB), "v"(xC), "s"(yC), "v"(xD), "s"(yD));
}

// o = a + b over 256 bits (carry out dropped: callers keep sums below 2^256)
__device__ inline __attribute__((always_inline)) void add256(uint32_t (&o)[8], const uint32_t (&a)[8], const uint32_t (&b)[8]) {
    asm("v_add_co_u32_e32 %0, vcc, %8, %16\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %1, vcc, %9, %17, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %2, vcc, %10, %18, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %3, vcc, %11, %19, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %4, vcc, %12, %20, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %5, vcc, %13, %21, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %6, vcc, %14, %22, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %7, vcc, %15, %23, vcc"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
        : "vcc");
}
// o = a - b over 256 bits; borrow_mask = all ones iff a < b
__device__ inline __attribute__((always_inline)) void sub256(uint32_t (&o)[8], uint32_t& borrow_mask, const uint32_t (&a)[8],
                                       const uint32_t (&b)[8]) {
    uint32_t bm;
    asm("v_sub_co_u32_e32 %0, vcc, %9, %17\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %1, vcc, %10, %18, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %2, vcc, %11, %19, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %3, vcc, %12, %20, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %4, vcc, %13, %21, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %5, vcc, %14, %22, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %6, vcc, %15, %23, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %7, vcc, %16, %24, vcc\n\ts_nop 1\n\t"
        "v_cndmask_b32_e64 %8, 0, -1, vcc"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
          "=&v"(bm)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
        : "vcc");
    borrow_mask = bm;
}

#else
// Portable twins of the four primitives (host builds and the host pass of hipcc);
// everything built on them below is shared, so the CPU unit tests exercise the
// same column schedule the device runs.
GKR_HD void mac96(uint64_t& acc, uint32_t& ex, uint32_t x, uint32_t y) {
    const uint64_t p = (uint64_t)x * y, s = acc + p;
    ex += (s < acc) ? 1u : 0u;
    acc = s;
}
GKR_HD void mac96_first(uint64_t& acc, uint32_t& ex, uint32_t x, uint32_t y) {
    const uint64_t p = (uint64_t)x * y, s = acc + p;
    ex = (s < acc) ? 1u : 0u;
    acc = s;
}
GKR_HD uint64_t column_start(uint32_t k0, uint32_t k1, uint32_t a) { return ((uint64_t)k0 | ((uint64_t)k1 << 32)) + a; }
GKR_HD void mac96_s(uint64_t& acc, uint32_t& ex, uint32_t x, uint32_t y) { mac96(acc, ex, x, y); }
GKR_HD void mac96x2_s(uint64_t& accA, uint32_t& exA, uint32_t xA, uint32_t yA, uint64_t& accB, uint32_t& exB, uint32_t xB,
                      uint32_t yB) {
    mac96(accA, exA, xA, yA);
    mac96(accB, exB, xB, yB);
}
GKR_HD void mac96x3_s(uint64_t& accA, uint32_t& exA, uint32_t xA, uint32_t yA, uint64_t& accB, uint32_t& exB, uint32_t xB,
                      uint32_t yB, uint64_t& accC, uint32_t& exC, uint32_t xC, uint32_t yC) {
    mac96(accA, exA, xA, yA);
    mac96(accB, exB, xB, yB);
    mac96(accC, exC, xC, yC);
}
GKR_HD void mac96x4_s(uint64_t& accA, uint32_t& exA, uint32_t xA, uint32_t yA, uint64_t& accB, uint32_t& exB, uint32_t xB,
                      uint32_t yB, uint64_t& accC, uint32_t& exC, uint32_t xC, uint32_t yC, uint64_t& accD, uint32_t& exD,
                      uint32_t xD, uint32_t yD) {
    mac96(accA, exA, xA, yA);
    mac96(accB, exB, xB, yB);
    mac96(accC, exC, xC, yC);
    mac96(accD, exD, xD, yD);
}
GKR_HD void add256(uint32_t (&o)[8], const uint32_t (&a)[8], const uint32_t (&b)[8]) {
    uint64_t carry = 0;
    for (int i = 0; i < 8; ++i) {
        const uint64_t t = (uint64_t)a[i] + b[i] + carry;
        o[i] = (uint32_t)t;
        carry = t >> 32;
    }
}
GKR_HD void sub256(uint32_t (&o)[8], uint32_t& borrow_mask, const uint32_t (&a)[8], const uint32_t (&b)[8]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 8; ++i) {
        const uint64_t t = (uint64_t)a[i] - b[i] - borrow;
        o[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    borrow_mask = borrow ? 0xffffffffu : 0u;
}
#endif

// s in [0, 2r) -> [0, r)
GKR_HD Fr cond_sub_mod(const uint32_t (&s)[8]) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    uint32_t pl[8], d[8], bm;
#pragma unroll
    for (int i = 0; i < 8; ++i) pl[i] = p[i];
    sub256(d, bm, s, pl);
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = bm ? s[i] : d[i];
    return r;
}

GKR_HD Fr fr_add(const Fr& a, const Fr& b) {
    uint32_t s[8];
    add256(s, a.l, b.l);
    return cond_sub_mod(s);
}

GKR_HD Fr fr_sub(const Fr& a, const Fr& b) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    uint32_t d[8], bm, pm[8];
    sub256(d, bm, a.l, b.l);
#pragma unroll
    for (int i = 0; i < 8; ++i) pm[i] = p[i] & bm;
    Fr r;
    add256(r.l, d, pm);
    return r;
}

GKR_HD Fr mont_mul(const Fr& a, const Fr& b) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    uint32_t m[8], t[8];
    uint64_t acc = 0;
    uint32_t ex = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int i = 0; i <= c; ++i) mac96(acc, ex, a.l[i], b.l[c - i]);
#pragma unroll
        for (int i = 0; i < c; ++i) mac96_s(acc, ex, m[i], p[c - i]);
        m[c] = (uint32_t)acc * GKR_INV32;
        mac96_s(acc, ex, m[c], p[0]);     // low word becomes 0
        acc = (acc >> 32) | ((uint64_t)ex << 32);
        ex = 0;
    }
#pragma unroll
    for (int c = 8; c < 15; ++c) {
#pragma unroll
        for (int i = c - 7; i < 8; ++i) mac96(acc, ex, a.l[i], b.l[c - i]);
#pragma unroll
        for (int i = c - 7; i < 8; ++i) mac96_s(acc, ex, m[i], p[c - i]);
        t[c - 8] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)ex << 32);
        ex = 0;
    }
    t[7] = (uint32_t)acc;   // the total is < 2r < 2^255: nothing above t[7]
    return cond_sub_mod(t);
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

// ---------------------------------------------------------------- fixed multiplier
// Every fold of a round multiplies by the SAME challenge r.  With the table
//     R_i = r * 2^(32 i) * 2^64 mod p      (i = 0..7, canonical, wave-uniform)
// d * r = (sum_i d_i R_i) * 2^-64: 64 partial products for the sum (< 2^291) and
// two 32-bit Montgomery steps (16 more) leave a value < 2p -- 80 v_mad_u64_u32
// instead of the 128 of a general Montgomery product, and the table rides in
// SGPRs.  Works on canonical and on Montgomery-form d alike (it is a plain
// modular product by r).
struct FixedMul {
    uint32_t w[8][8];
};

// host: build the table from canonical r
GKR_HD FixedMul make_fixed_mul(const Fr& r_canonical) {
    Fr two32 = fr_zero(), two64 = fr_zero();
    two32.l[1] = 1;
    two64.l[2] = 1;
    FixedMul T;
    Fr cur = mont_mul_portable(mont_mul_portable(r_canonical, fr_r2()), two64);   // r * 2^64
    const Fr two32_m = mont_mul_portable(two32, fr_r2());
    for (int i = 0; i < 8; ++i) {
        for (int c = 0; c < 8; ++c) T.w[i][c] = cur.l[c];
        cur = mont_mul_portable(cur, two32_m);
    }
    return T;
}

GKR_HD Fr mul_fixed(const Fr& d, const FixedMul& T) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    uint32_t s[11];
    uint64_t acc = 0;
    uint32_t ex = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) mac96_s(acc, ex, d.l[i], T.w[i][c]);
        s[c] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)ex << 32);
        ex = 0;
    }
    s[8] = (uint32_t)acc;
    s[9] = (uint32_t)(acc >> 32);
    s[10] = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {   // two 32-bit Montgomery steps
        const uint32_t m = s[k] * GKR_INV32;
        uint64_t a2 = s[k];
        uint32_t e2 = 0;
        mac96_s(a2, e2, m, p[0]);
        a2 = (a2 >> 32) | ((uint64_t)e2 << 32);
        e2 = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            a2 += s[k + j];   // a2 < 2^33 here: no overflow
            mac96_s(a2, e2, m, p[j]);
            s[k + j] = (uint32_t)a2;
            a2 = (a2 >> 32) | ((uint64_t)e2 << 32);
            e2 = 0;
        }
#pragma unroll
        for (int j = k + 8; j < 11; ++j) {
            a2 += s[j];
            s[j] = (uint32_t)a2;
            a2 >>= 32;
        }
    }
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = s[i + 2];
    return cond_sub_mod(t);   // (2^35 + 2^64) p / 2^64 < 2p and s[10] == 0
}

// T + r (H - T) through the round's fixed-multiplier table
GKR_HD Fr fr_fold_fixed(const Fr& lo, const Fr& hi, const FixedMul& T) {
    return fr_add(lo, mul_fixed(fr_sub(hi, lo), T));
}

// two fixed-multiplier products advanced together (same table, independent operands)
GKR_HD void mul_fixed2(const Fr& dA, const Fr& dB, const FixedMul& T, Fr& outA, Fr& outB) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    uint32_t sA[11], sB[11];
    uint64_t aA = 0, aB = 0;
    uint32_t eA = 0, eB = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) mac96x2_s(aA, eA, dA.l[i], T.w[i][c], aB, eB, dB.l[i], T.w[i][c]);
        sA[c] = (uint32_t)aA;
        sB[c] = (uint32_t)aB;
        aA = (aA >> 32) | ((uint64_t)eA << 32);
        aB = (aB >> 32) | ((uint64_t)eB << 32);
        eA = 0;
        eB = 0;
    }
    sA[8] = (uint32_t)aA; sA[9] = (uint32_t)(aA >> 32); sA[10] = 0;
    sB[8] = (uint32_t)aB; sB[9] = (uint32_t)(aB >> 32); sB[10] = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t mA = sA[k] * GKR_INV32, mB = sB[k] * GKR_INV32;
        uint64_t xA = sA[k], xB = sB[k];
        uint32_t fA = 0, fB = 0;
        mac96x2_s(xA, fA, mA, p[0], xB, fB, mB, p[0]);
        xA = (xA >> 32) | ((uint64_t)fA << 32);
        xB = (xB >> 32) | ((uint64_t)fB << 32);
        fA = 0;
        fB = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            xA += sA[k + j];
            xB += sB[k + j];
            mac96x2_s(xA, fA, mA, p[j], xB, fB, mB, p[j]);
            sA[k + j] = (uint32_t)xA;
            sB[k + j] = (uint32_t)xB;
            xA = (xA >> 32) | ((uint64_t)fA << 32);
            xB = (xB >> 32) | ((uint64_t)fB << 32);
            fA = 0;
            fB = 0;
        }
#pragma unroll
        for (int j = k + 8; j < 11; ++j) {
            xA += sA[j]; sA[j] = (uint32_t)xA; xA >>= 32;
            xB += sB[j]; sB[j] = (uint32_t)xB; xB >>= 32;
        }
    }
    uint32_t tA[8], tB[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        tA[i] = sA[i + 2];
        tB[i] = sB[i + 2];
    }
    outA = cond_sub_mod(tA);
    outB = cond_sub_mod(tB);
}

// the two folds of one fold-and-sum step together
GKR_HD void fr_fold_fixed2(const Fr& lo0, const Fr& hi0, const Fr& lo1, const Fr& hi1, const FixedMul& T, Fr& y0, Fr& y1) {
    Fr t0, t1;
    mul_fixed2(fr_sub(hi0, lo0), fr_sub(hi1, lo1), T, t0, t1);
    y0 = fr_add(lo0, t0);
    y1 = fr_add(lo1, t1);
}

// ---------------------------------------------------------------- lazy dot products
// sum_i a_i * b_i with ONE modular reduction at the end: each product is added as a full
// 512-bit integer into a 544-bit (17-limb) accumulator -- 64 partial products instead of the
// 128 of a reduced Montgomery product; 2^36 products fit.  b is wave-uniform (SGPR operands).
struct Lazy17 {
    uint32_t l[17];
};

GKR_HD Lazy17 lazy_zero() {
    Lazy17 z;
#pragma unroll
    for (int i = 0; i < 17; ++i) z.l[i] = 0;
    return z;
}

GKR_HD void lazy_mac_s(Lazy17& acc, const Fr& a, const Fr& b_uniform) {
    uint64_t col = acc.l[0];
    uint32_t ex = 0;
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        if (c) {
            col = column_start((uint32_t)(col >> 32), ex, acc.l[c]);   // < 2^35 + 2^32: cannot overflow
            ex = 0;
        }
#pragma unroll
        for (int i = (c > 7 ? c - 7 : 0); i <= (c < 7 ? c : 7); ++i) mac96_s(col, ex, a.l[i], b_uniform.l[c - i]);
        acc.l[c] = (uint32_t)col;
    }
    col = (col >> 32) | ((uint64_t)ex << 32);
    col += acc.l[15];
    acc.l[15] = (uint32_t)col;
    col >>= 32;
    col += acc.l[16];
    acc.l[16] = (uint32_t)col;
}

// the same with a per-lane multiplier (both operands in vector registers)
GKR_HD void lazy_mac_v(Lazy17& acc, const Fr& a, const Fr& b) {
    uint64_t col = acc.l[0];
    uint32_t ex = 0;
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        if (c) col = column_start((uint32_t)(col >> 32), ex, acc.l[c]);   // < 2^35 + 2^32: cannot overflow
        const int i0 = c > 7 ? c - 7 : 0, i1 = c < 7 ? c : 7;
        mac96_first(col, ex, a.l[i0], b.l[c - i0]);
#pragma unroll
        for (int i = i0 + 1; i <= i1; ++i) mac96(col, ex, a.l[i], b.l[c - i]);
        acc.l[c] = (uint32_t)col;
    }
    col = (col >> 32) | ((uint64_t)ex << 32);
    col += acc.l[15];
    acc.l[15] = (uint32_t)col;
    col >>= 32;
    col += acc.l[16];
    acc.l[16] = (uint32_t)col;
}

// One product, added to ONE of two accumulators chosen per lane -- without a branch: a wave whose lanes pick different
// accumulators would otherwise run the 64 multiply-adds twice (once per side of the branch).  Three selects per column.
GKR_HD void lazy_mac_sel(Lazy17& A, Lazy17& B, bool toA, const Fr& a, const Fr& b) {
    uint64_t col = toA ? A.l[0] : B.l[0];
    uint32_t ex = 0;
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        if (c) col = column_start((uint32_t)(col >> 32), ex, toA ? A.l[c] : B.l[c]);
        const int i0 = c > 7 ? c - 7 : 0, i1 = c < 7 ? c : 7;
        mac96_first(col, ex, a.l[i0], b.l[c - i0]);
#pragma unroll
        for (int i = i0 + 1; i <= i1; ++i) mac96(col, ex, a.l[i], b.l[c - i]);
        const uint32_t lo = (uint32_t)col;
        A.l[c] = toA ? lo : A.l[c];
        B.l[c] = toA ? B.l[c] : lo;
    }
    col = (col >> 32) | ((uint64_t)ex << 32);
#pragma unroll
    for (int c = 15; c < 17; ++c) {
        col += toA ? A.l[c] : B.l[c];
        const uint32_t lo = (uint32_t)col;
        A.l[c] = toA ? lo : A.l[c];
        B.l[c] = toA ? B.l[c] : lo;
        col >>= 32;
    }
}

// acc += x * 2^256 (x where `on`, else nothing): after the Montgomery reduction of the accumulator this is "+ x" --
// how a term that carries no second factor joins a lazy sum of products for nine additions instead of a product by one.
#if defined(__HIP_DEVICE_COMPILE__)
// (device: the addend masked with eight selects, then one padded carry chain over limbs 8..16 -- the portable form
// below compiles to ~45 instructions, and the segment pass is bound by its instruction count)
__device__ inline __attribute__((always_inline)) void lazy_add_hi(Lazy17& acc, const Fr& x, bool on) {
    uint32_t m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = on ? x.l[i] : 0u;
    asm("v_add_co_u32_e32 %0, vcc, %0, %9\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %1, vcc, %1, %10, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %11, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %3, vcc, %3, %12, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %4, vcc, %4, %13, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %5, vcc, %5, %14, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %6, vcc, %6, %15, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %7, vcc, %7, %16, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %8, vcc, 0, %8, vcc"
        : "+v"(acc.l[8]), "+v"(acc.l[9]), "+v"(acc.l[10]), "+v"(acc.l[11]), "+v"(acc.l[12]), "+v"(acc.l[13]), "+v"(acc.l[14]),
          "+v"(acc.l[15]), "+v"(acc.l[16])
        : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7])
        : "vcc");
}
#else
GKR_HD void lazy_add_hi(Lazy17& acc, const Fr& x, bool on) {
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint64_t t = (uint64_t)acc.l[8 + i] + (on ? x.l[i] : 0u) + carry;
        acc.l[8 + i] = (uint32_t)t;
        carry = t >> 32;
    }
    acc.l[16] += (uint32_t)carry;
}
#endif

// two independent dot products advanced together (see mac96x2_s)
GKR_HD void lazy_mac2_s(Lazy17& A, const Fr& a, const Fr& ua, Lazy17& B, const Fr& b, const Fr& ub) {
    uint64_t cA = 0, cB = 0;
    uint32_t eA = 0, eB = 0;
#pragma unroll
    for (int col = 0; col < 15; ++col) {
        cA += A.l[col];
        cB += B.l[col];
#pragma unroll
        for (int i = (col > 7 ? col - 7 : 0); i <= (col < 7 ? col : 7); ++i)
            mac96x2_s(cA, eA, a.l[i], ua.l[col - i], cB, eB, b.l[i], ub.l[col - i]);
        A.l[col] = (uint32_t)cA;
        B.l[col] = (uint32_t)cB;
        cA = (cA >> 32) | ((uint64_t)eA << 32);
        cB = (cB >> 32) | ((uint64_t)eB << 32);
        eA = 0;
        eB = 0;
    }
    cA += A.l[15]; A.l[15] = (uint32_t)cA; cA >>= 32; cA += A.l[16]; A.l[16] = (uint32_t)cA;
    cB += B.l[15]; B.l[15] = (uint32_t)cB; cB >>= 32; cB += B.l[16]; B.l[16] = (uint32_t)cB;
}

// four independent dot products advanced together: two instructions per partial product, no pads
GKR_HD void lazy_mac4_s(Lazy17& A, const Fr& a, const Fr& ua, Lazy17& B, const Fr& b, const Fr& ub, Lazy17& C, const Fr& c,
                        const Fr& uc, Lazy17& D, const Fr& d, const Fr& ud) {
    uint64_t cA = 0, cB = 0, cC = 0, cD = 0;
    uint32_t eA = 0, eB = 0, eC = 0, eD = 0;
#pragma unroll
    for (int col = 0; col < 15; ++col) {
        cA += A.l[col];
        cB += B.l[col];
        cC += C.l[col];
        cD += D.l[col];
#pragma unroll
        for (int i = (col > 7 ? col - 7 : 0); i <= (col < 7 ? col : 7); ++i)
            mac96x4_s(cA, eA, a.l[i], ua.l[col - i], cB, eB, b.l[i], ub.l[col - i], cC, eC, c.l[i], uc.l[col - i], cD, eD,
                      d.l[i], ud.l[col - i]);
        A.l[col] = (uint32_t)cA;
        B.l[col] = (uint32_t)cB;
        C.l[col] = (uint32_t)cC;
        D.l[col] = (uint32_t)cD;
        cA = (cA >> 32) | ((uint64_t)eA << 32);
        cB = (cB >> 32) | ((uint64_t)eB << 32);
        cC = (cC >> 32) | ((uint64_t)eC << 32);
        cD = (cD >> 32) | ((uint64_t)eD << 32);
        eA = 0;
        eB = 0;
        eC = 0;
        eD = 0;
    }
    cA += A.l[15]; A.l[15] = (uint32_t)cA; cA >>= 32; cA += A.l[16]; A.l[16] = (uint32_t)cA;
    cB += B.l[15]; B.l[15] = (uint32_t)cB; cB >>= 32; cB += B.l[16]; B.l[16] = (uint32_t)cB;
    cC += C.l[15]; C.l[15] = (uint32_t)cC; cC >>= 32; cC += C.l[16]; C.l[16] = (uint32_t)cC;
    cD += D.l[15]; D.l[15] = (uint32_t)cD; cD >>= 32; cD += D.l[16]; D.l[16] = (uint32_t)cD;
}

// sum_{p < NP} x_p * u_p as ONE 16-limb integer, column by column: four chains (product p goes to
// chain p mod 4) advance together inside a column, their 96-bit totals are merged at the column's
// end, and the merged carry seeds chain 0 of the next column.  No per-product accumulator state:
// NP = 8 needs ~1400 instructions where two lazy_mac4_s calls plus merges need ~2000.
template <int NP>
GKR_HD void weighted_sum_s(const Fr (&x)[NP], const Fr* __restrict__ u, Lazy17& out) {
    static_assert(NP == 4 || NP == 8, "four chains");
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
#pragma unroll
    for (int col = 0; col < 15; ++col) {
#pragma unroll
        for (int g = 0; g < NP; g += 4)
#pragma unroll
            for (int i = (col > 7 ? col - 7 : 0); i <= (col < 7 ? col : 7); ++i)
                mac96x4_s(c0, e0, x[g].l[i], u[g].l[col - i], c1, e1, x[g + 1].l[i], u[g + 1].l[col - i], c2, e2,
                          x[g + 2].l[i], u[g + 2].l[col - i], c3, e3, x[g + 3].l[i], u[g + 3].l[col - i]);
        // merge the four 96-bit chain totals (low, high, overflow words summed with carries)
        const uint64_t lo = (c0 & 0xffffffffull) + (c1 & 0xffffffffull) + (c2 & 0xffffffffull) + (c3 & 0xffffffffull);
        const uint64_t hi = (c0 >> 32) + (c1 >> 32) + (c2 >> 32) + (c3 >> 32) + (lo >> 32);
        const uint64_t ex = (uint64_t)e0 + e1 + e2 + e3 + (hi >> 32);
        out.l[col] = (uint32_t)lo;
        c0 = (hi & 0xffffffffull) | (ex << 32);   // carry into the next column (ex < 2^32: column sums stay far below 2^96)
        c1 = c2 = c3 = 0;
        e0 = e1 = e2 = e3 = 0;
    }
    out.l[15] = (uint32_t)c0;
    out.l[16] = (uint32_t)(c0 >> 32);
}

// A += B (both unreduced)
GKR_HD void lazy_add(Lazy17& A, const Lazy17& B) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 17; ++i) {
        c += (uint64_t)A.l[i] + B.l[i];
        A.l[i] = (uint32_t)c;
        c >>= 32;
    }
}

// three independent dot products advanced together (see mac96x3_s)
GKR_HD void lazy_mac3_s(Lazy17& A, const Fr& a, const Fr& ua, Lazy17& B, const Fr& b, const Fr& ub, Lazy17& C, const Fr& c,
                        const Fr& uc) {
    uint64_t cA = 0, cB = 0, cC = 0;
    uint32_t eA = 0, eB = 0, eC = 0;
#pragma unroll
    for (int col = 0; col < 15; ++col) {
        cA += A.l[col];
        cB += B.l[col];
        cC += C.l[col];
#pragma unroll
        for (int i = (col > 7 ? col - 7 : 0); i <= (col < 7 ? col : 7); ++i)
            mac96x3_s(cA, eA, a.l[i], ua.l[col - i], cB, eB, b.l[i], ub.l[col - i], cC, eC, c.l[i], uc.l[col - i]);
        A.l[col] = (uint32_t)cA;
        B.l[col] = (uint32_t)cB;
        C.l[col] = (uint32_t)cC;
        cA = (cA >> 32) | ((uint64_t)eA << 32);
        cB = (cB >> 32) | ((uint64_t)eB << 32);
        cC = (cC >> 32) | ((uint64_t)eC << 32);
        eA = 0;
        eB = 0;
        eC = 0;
    }
    cA += A.l[15]; A.l[15] = (uint32_t)cA; cA >>= 32; cA += A.l[16]; A.l[16] = (uint32_t)cA;
    cB += B.l[15]; B.l[15] = (uint32_t)cB; cB >>= 32; cB += B.l[16]; B.l[16] = (uint32_t)cB;
    cC += C.l[15]; C.l[15] = (uint32_t)cC; cC >>= 32; cC += C.l[16]; C.l[16] = (uint32_t)cC;
}

// X * 2^-256 mod r, canonical: eight 32-bit Montgomery steps leave (X + M p) / 2^256 < 2^288 + p
// in ten limbs, which the wide-sum reduction below brings under r.
template <int NL>
struct Acc;
template <int NL>
GKR_HD Fr acc_reduce(const Acc<NL>& a);

GKR_HD Fr lazy_reduce(const Lazy17& x);

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

#if defined(__HIP_DEVICE_COMPILE__)
// 288-bit accumulate as one padded carry chain (see the hazard note above)
__device__ inline __attribute__((always_inline)) void acc_add_fr(Acc<9>& a, const Fr& x) {
    asm("v_add_co_u32_e32 %0, vcc, %0, %9\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %1, vcc, %1, %10, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %11, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %3, vcc, %3, %12, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %4, vcc, %4, %13, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %5, vcc, %5, %14, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %6, vcc, %6, %15, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %7, vcc, %7, %16, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %8, vcc, 0, %8, vcc"
        : "+v"(a.l[0]), "+v"(a.l[1]), "+v"(a.l[2]), "+v"(a.l[3]), "+v"(a.l[4]), "+v"(a.l[5]), "+v"(a.l[6]),
          "+v"(a.l[7]), "+v"(a.l[8])
        : "v"(x.l[0]), "v"(x.l[1]), "v"(x.l[2]), "v"(x.l[3]), "v"(x.l[4]), "v"(x.l[5]), "v"(x.l[6]), "v"(x.l[7])
        : "vcc");
}
#endif

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

// the same for a sum of at most EIGHT products of values below r: X < 8 r^2, so the eight
// Montgomery steps leave (X + M p) / 2^256 < 8 r^2 / 2^256 + r < 2.6 r -- two conditional
// subtractions finish the job (no wide reduction, no extra product).
GKR_HD Fr lazy_reduce_k8(const Lazy17& x) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    uint32_t t[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) t[i] = x.l[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t m = t[k] * GKR_INV32;
        uint64_t a2 = t[k];
        uint32_t e2 = 0;
        mac96_s(a2, e2, m, p[0]);
        a2 = (a2 >> 32) | ((uint64_t)e2 << 32);
        e2 = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            a2 += t[k + j];
            mac96_s(a2, e2, m, p[j]);
            t[k + j] = (uint32_t)a2;
            a2 = (a2 >> 32) | ((uint64_t)e2 << 32);
            e2 = 0;
        }
#pragma unroll
        for (int j = k + 8; j < 17; ++j) {
            a2 += t[j];
            t[j] = (uint32_t)a2;
            a2 >>= 32;
        }
    }
    uint32_t r8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r8[i] = t[8 + i];   // < 2.6 r < 2^256: t[16] == 0
    const Fr once = cond_sub_mod(r8);
    return cond_sub_mod(once.l);
}

// The same eight Montgomery steps for a sum of at most 32 terms, each a product of two values below r or a value
// below r times 2^256 (lazy_add_hi), WITHOUT the final canonical reduction: the result is some 256-bit representative
// of x / 2^256 mod r, which is all a value needs to be that goes on as an operand of another lazy product.
// (x + M p) / 2^256 < 32 r + p < 7.2 * 2^256, so the ninth limb is at most 7; since 2^256 = R1 (mod r) with
// R1 < 2^252, replacing top * 2^256 by top * R1 twice brings the value below 2^256.
GKR_HD Fr lazy_reduce_partial32(const Lazy17& x) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    constexpr uint32_t r1[8] = GKR_R1_LIMBS;
    uint32_t t[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) t[i] = x.l[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t m = t[k] * GKR_INV32;
        uint64_t a2 = t[k];
        uint32_t e2 = 0;
        mac96_s(a2, e2, m, p[0]);
        a2 = (a2 >> 32) | ((uint64_t)e2 << 32);
        e2 = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            a2 += t[k + j];
            mac96_s(a2, e2, m, p[j]);
            t[k + j] = (uint32_t)a2;
            a2 = (a2 >> 32) | ((uint64_t)e2 << 32);
            e2 = 0;
        }
#pragma unroll
        for (int j = k + 8; j < 17; ++j) {
            a2 += t[j];
            t[j] = (uint32_t)a2;
            a2 >>= 32;
        }
    }
    uint32_t top = t[16];
    Fr out;
#pragma unroll
    for (int i = 0; i < 8; ++i) out.l[i] = t[8 + i];
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c += (uint64_t)out.l[i] + (uint64_t)top * r1[i];   // top <= 7: no overflow of the 64-bit column
            out.l[i] = (uint32_t)c;
            c >>= 32;
        }
        top = (uint32_t)c;
    }
    return out;   // top == 0 now
}

GKR_HD Fr lazy_reduce(const Lazy17& x) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    uint32_t t[19];
#pragma unroll
    for (int i = 0; i < 17; ++i) t[i] = x.l[i];
    t[17] = 0;
    t[18] = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t m = t[k] * GKR_INV32;
        uint64_t a2 = t[k];
        uint32_t e2 = 0;
        mac96_s(a2, e2, m, p[0]);
        a2 = (a2 >> 32) | ((uint64_t)e2 << 32);
        e2 = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            a2 += t[k + j];
            mac96_s(a2, e2, m, p[j]);
            t[k + j] = (uint32_t)a2;
            a2 = (a2 >> 32) | ((uint64_t)e2 << 32);
            e2 = 0;
        }
#pragma unroll
        for (int j = k + 8; j < 19; ++j) {
            a2 += t[j];
            t[j] = (uint32_t)a2;
            a2 >>= 32;
        }
    }
    Acc<10> r;
#pragma unroll
    for (int i = 0; i < 10; ++i) r.l[i] = t[8 + i];
    return acc_reduce(r);
}

}  // namespace gkr
