// Host-side BN254 Fr on 4 x 64-bit limbs (Montgomery, R = 2^256) for the
// transcript: MiMC7 is a 728-deep serial chain of modular products per round
// vector, which a 5 GHz host core finishes ~100x sooner than one GPU lane.
// Same Montgomery radix as the device's 8 x 32-bit form (fr32.h), so a value in
// Montgomery form is the same 32 bytes on both sides.
#pragma once
#include <stdint.h>
#include <string.h>

namespace gkr {
namespace h64 {

typedef unsigned __int128 u128;

struct F {
    uint64_t l[4];
};

static const uint64_t kMod[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
                                 0x30644e72e131a029ULL};
static const uint64_t kInv = 0xc2e1f593efffffffULL;  // -r^{-1} mod 2^64
static const F kR2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};

inline bool is_zero(const F& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }

inline bool geq_mod(const F& a) {
    for (int i = 3; i >= 0; --i) {
        if (a.l[i] > kMod[i]) return true;
        if (a.l[i] < kMod[i]) return false;
    }
    return true;
}

inline void sub_mod(F& a) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - kMod[i] - borrow;
        a.l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
}

inline F add(const F& a, const F& b) {
    F s;
    uint64_t carry = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a.l[i] + b.l[i] + carry;
        s.l[i] = (uint64_t)t;
        carry = (uint64_t)(t >> 64);
    }
    if (geq_mod(s)) sub_mod(s);
    return s;
}

inline F sub(const F& a, const F& b) {
    F d;
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a.l[i] - b.l[i] - borrow;
        d.l[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1;
    }
    if (borrow) {
        uint64_t carry = 0;
        for (int i = 0; i < 4; ++i) {
            u128 t = (u128)d.l[i] + kMod[i] + carry;
            d.l[i] = (uint64_t)t;
            carry = (uint64_t)(t >> 64);
        }
    }
    return d;
}

// a * b * 2^-256 mod r
inline F mont_mul(const F& a, const F& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; ++j) {
            u128 p = (u128)a.l[j] * b.l[i] + t[j] + carry;
            t[j] = (uint64_t)p;
            carry = (uint64_t)(p >> 64);
        }
        u128 s = (u128)t[4] + carry;
        t[4] = (uint64_t)s;
        t[5] = (uint64_t)(s >> 64);
        const uint64_t m = t[0] * kInv;
        u128 p = (u128)m * kMod[0] + t[0];
        carry = (uint64_t)(p >> 64);
        for (int j = 1; j < 4; ++j) {
            p = (u128)m * kMod[j] + t[j] + carry;
            t[j - 1] = (uint64_t)p;
            carry = (uint64_t)(p >> 64);
        }
        s = (u128)t[4] + carry;
        t[3] = (uint64_t)s;
        t[4] = t[5] + (uint64_t)(s >> 64);
    }
    F out = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_mod(out)) sub_mod(out);
    return out;
}

// Sums of products with ONE reduction (the device's lazy_mac_v / lazy_reduce on 64-bit limbs): a product of two values below
// 2^256 is added as a 512-bit integer into a 576-bit accumulator -- 2^64 products fit -- and the sum is reduced once:
// four Montgomery steps, then a quotient estimate from the top bits (the sum over 2^256 is below 2^274 for up to 2^18
// products of values below r) and one product with r.  wide_reduce(sum of a_i b_i) == the modular sum of mont_mul(a_i, b_i).
struct Wide {
    uint64_t l[9];
};
inline Wide wide_zero() { return Wide{{0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
inline void wide_mac(Wide& acc, const F& a, const F& b) {
    for (int i = 0; i < 4; ++i) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; ++j) {
            const u128 p = (u128)a.l[j] * b.l[i] + acc.l[i + j] + carry;
            acc.l[i + j] = (uint64_t)p;
            carry = (uint64_t)(p >> 64);
        }
        for (int j = i + 4; j < 9 && carry; ++j) {
            const u128 s = (u128)acc.l[j] + carry;
            acc.l[j] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
    }
}
inline F wide_reduce(const Wide& x) {
    uint64_t t[9];
    for (int i = 0; i < 9; ++i) t[i] = x.l[i];
    for (int i = 0; i < 4; ++i) {
        const uint64_t m = t[i] * kInv;
        uint64_t carry = 0;
        for (int j = 0; j < 4; ++j) {
            const u128 p = (u128)m * kMod[j] + t[i + j] + carry;
            t[i + j] = (uint64_t)p;
            carry = (uint64_t)(p >> 64);
        }
        for (int j = i + 4; j < 9 && carry; ++j) {
            const u128 s = (u128)t[j] + carry;
            t[j] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
    }
    // y = t[4 .. 8] < 2^274: q = floor((y >> 224) * mu / 2^61) with mu = floor(2^61 / ((r >> 224) + 1)) never exceeds y / r and
    // falls short of it by less than 2 (mfma_fold.h, mf_reduce_274): y - q r is in [0, 2r)
    const uint64_t top = (t[8] << 32) | (t[7] >> 32);
    const uint64_t q = (uint64_t)(((u128)top * 0xa948e8c4ull) >> 61);
    F out;
    uint64_t carry = 0, borrow = 0;
    for (int i = 0; i < 4; ++i) {
        const u128 p = (u128)q * kMod[i] + carry;
        carry = (uint64_t)(p >> 64);
        const u128 d = (u128)t[4 + i] - (uint64_t)p - borrow;
        out.l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    if (geq_mod(out)) sub_mod(out);
    return out;
}

inline F to_mont(const F& a) { return mont_mul(a, kR2); }
inline F from_mont(const F& a) {
    const F one = {{1, 0, 0, 0}};
    return mont_mul(a, one);
}

// MiMC7-91 (see mimc7.h for the reference call sites).  cts: 91 Montgomery constants.
inline F mimc7_hash_mont(const F& x, const F& k, const F* cts) {
    F h = {{0, 0, 0, 0}};
    for (int i = 0; i < 91; ++i) {
        F t = (i == 0) ? add(x, k) : add(add(h, k), cts[i]);
        F t2 = mont_mul(t, t);
        F t4 = mont_mul(t2, t2);
        F t6 = mont_mul(t4, t2);
        h = mont_mul(t6, t);
    }
    return add(h, k);
}

// arr canonical -> canonical multi_hash(arr, key = 0); also returns r in Montgomery form
inline F mimc7_multi_hash(const F* arr, int n, const F* cts, F* r_mont_out) {
    F r = {{0, 0, 0, 0}};
    for (int i = 0; i < n; ++i) {
        F a = to_mont(arr[i]);
        F h = mimc7_hash_mont(a, r, cts);
        r = add(add(r, a), h);
    }
    F canon = from_mont(r);
    if (r_mont_out) *r_mont_out = r;   // the running state is already the Montgomery form of the result
    return canon;
}

// the round's fixed-multiplier table (fr32.h FixedMul): R_i = r * 2^(32 i) * 2^64 mod p, canonical,
// written as 8 x 8 little-endian 32-bit limbs
inline void make_fixed_mul(const F& r_canonical, uint32_t (*w)[8]) {
    const F two64 = {{0, 1, 0, 0}}, two32 = {{1ULL << 32, 0, 0, 0}};
    F cur = mont_mul(to_mont(r_canonical), two64);   // r * 2^64
    const F two32_m = to_mont(two32);
    for (int i = 0; i < 8; ++i) {
        memcpy(w[i], cur.l, 32);
        cur = mont_mul(cur, two32_m);
    }
}

}  // namespace h64
}  // namespace gkr
