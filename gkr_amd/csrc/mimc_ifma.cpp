// MiMC7 multi_hash for EIGHT transcripts at once on AVX-512 IFMA (vpmadd52luq / vpmadd52huq).
//
// The host transcript needs one hash per sumcheck per round; on the GPU box the process gets 16
// CPUs (cgroup quota), so hashing throughput is what bounds a batch in its late rounds.  The
// hashes of different sumchecks are independent: this file runs eight of them in the eight 64-bit
// lanes of a zmm register, radix 2^52 (5 limbs, Montgomery radix 2^260), ~135 vector
// instructions per eight modular products.  Selected at run time when the CPU has
// avx512ifma (EPYC Zen 4/5, Xeon Ice Lake+); otherwise the scalar 4x64-bit code in fr64.h runs.
// Same function as the reference's Mimc7::multi_hash (call sites rust/src/gkr/sumcheck.rs:84,129,152).
//
// Built with -mavx512f -mavx512ifma -mavx512vl for this translation unit only; nothing here is
// executed unless gkr_ifma_available() says so.
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include "mimc_ifma.h"

namespace gkr {
namespace ifma {

typedef unsigned __int128 u128;
static const uint64_t M52 = (1ULL << 52) - 1;

struct V {   // eight field elements, limb-major
    __m512i l[5];
};

// ---- scalar helpers for constant set-up (canonical 4x64 <-> 5x52)
static void to52(const uint64_t x[4], uint64_t o[5]) {
    o[0] = x[0] & M52;
    o[1] = ((x[0] >> 52) | (x[1] << 12)) & M52;
    o[2] = ((x[1] >> 40) | (x[2] << 24)) & M52;
    o[3] = ((x[2] >> 28) | (x[3] << 36)) & M52;
    o[4] = x[3] >> 16;
}
static void from52(const uint64_t l[5], uint64_t x[4]) {
    x[0] = l[0] | (l[1] << 52);
    x[1] = (l[1] >> 12) | (l[2] << 40);
    x[2] = (l[2] >> 24) | (l[3] << 28);
    x[3] = (l[3] >> 36) | (l[4] << 16);
}

static const uint64_t kP64[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};

// small big-int helpers on 64-bit words for the one-time constants
static int geq(const uint64_t* a, const uint64_t* b, int n) {
    for (int i = n - 1; i >= 0; --i) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static void subn(uint64_t* a, const uint64_t* b, int n) {
    uint64_t br = 0;
    for (int i = 0; i < n; ++i) {
        u128 d = (u128)a[i] - b[i] - br;
        a[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
}
// x = 2 x mod p (x < p, 4 words + overflow word)
static void dbl_mod(uint64_t x[4]) {
    uint64_t t[5];
    uint64_t c = 0;
    for (int i = 0; i < 4; ++i) {
        t[i] = (x[i] << 1) | c;
        c = x[i] >> 63;
    }
    t[4] = c;
    uint64_t p5[5] = {kP64[0], kP64[1], kP64[2], kP64[3], 0};
    if (geq(t, p5, 5)) subn(t, p5, 5);
    memcpy(x, t, 32);
}

static uint64_t g_r264_52[5];   // 2^264 mod p: a 4x64-Montgomery operand (x 2^256) times this, through mont_mul (/ 2^260), is that operand's value x 2^260
static uint64_t g_p52[5], g_pinv52, g_r2_52[5], g_one52[5], g_r256_52[5];   // g_r256: 2^256 mod p (the 4x64-bit code's Montgomery one)
static uint64_t g_cts52[91][5];   // MiMC constants, Montgomery (radix 2^260)
static bool g_ready = false;

static inline __m512i bc(uint64_t x) { return _mm512_set1_epi64((long long)x); }

// Montgomery product of eight pairs: a b 2^-260 mod p, inputs and output < p with 52-bit limbs
static inline V mont_mul(const V& a, const V& b) {
    const __m512i zero = _mm512_setzero_si512(), mask = bc(M52), pinv = bc(g_pinv52);
    __m512i t0 = zero, t1 = zero, t2 = zero, t3 = zero, t4 = zero, t5 = zero;
    const __m512i p0 = bc(g_p52[0]), p1 = bc(g_p52[1]), p2 = bc(g_p52[2]), p3 = bc(g_p52[3]), p4 = bc(g_p52[4]);
#define GKR_IFMA_ROUND(bi)                                        \
    {                                                             \
        t0 = _mm512_madd52lo_epu64(t0, a.l[0], bi);               \
        t1 = _mm512_madd52hi_epu64(t1, a.l[0], bi);               \
        t1 = _mm512_madd52lo_epu64(t1, a.l[1], bi);               \
        t2 = _mm512_madd52hi_epu64(t2, a.l[1], bi);               \
        t2 = _mm512_madd52lo_epu64(t2, a.l[2], bi);               \
        t3 = _mm512_madd52hi_epu64(t3, a.l[2], bi);               \
        t3 = _mm512_madd52lo_epu64(t3, a.l[3], bi);               \
        t4 = _mm512_madd52hi_epu64(t4, a.l[3], bi);               \
        t4 = _mm512_madd52lo_epu64(t4, a.l[4], bi);               \
        t5 = _mm512_madd52hi_epu64(t5, a.l[4], bi);               \
        const __m512i m = _mm512_and_si512(_mm512_madd52lo_epu64(zero, t0, pinv), mask); \
        t0 = _mm512_madd52lo_epu64(t0, m, p0);                    \
        t1 = _mm512_madd52hi_epu64(t1, m, p0);                    \
        t1 = _mm512_madd52lo_epu64(t1, m, p1);                    \
        t2 = _mm512_madd52hi_epu64(t2, m, p1);                    \
        t2 = _mm512_madd52lo_epu64(t2, m, p2);                    \
        t3 = _mm512_madd52hi_epu64(t3, m, p2);                    \
        t3 = _mm512_madd52lo_epu64(t3, m, p3);                    \
        t4 = _mm512_madd52hi_epu64(t4, m, p3);                    \
        t4 = _mm512_madd52lo_epu64(t4, m, p4);                    \
        t5 = _mm512_madd52hi_epu64(t5, m, p4);                    \
        /* low 52 bits of t0 are now zero: shift the window down one limb */ \
        t0 = _mm512_add_epi64(t1, _mm512_srli_epi64(t0, 52));     \
        t1 = t2; t2 = t3; t3 = t4; t4 = t5; t5 = zero;            \
    }
    GKR_IFMA_ROUND(b.l[0]) GKR_IFMA_ROUND(b.l[1]) GKR_IFMA_ROUND(b.l[2]) GKR_IFMA_ROUND(b.l[3]) GKR_IFMA_ROUND(b.l[4])
#undef GKR_IFMA_ROUND
    // carry-normalise to 52-bit limbs (value < 2p < 2^255)
    t1 = _mm512_add_epi64(t1, _mm512_srli_epi64(t0, 52)); t0 = _mm512_and_si512(t0, mask);
    t2 = _mm512_add_epi64(t2, _mm512_srli_epi64(t1, 52)); t1 = _mm512_and_si512(t1, mask);
    t3 = _mm512_add_epi64(t3, _mm512_srli_epi64(t2, 52)); t2 = _mm512_and_si512(t2, mask);
    t4 = _mm512_add_epi64(t4, _mm512_srli_epi64(t3, 52)); t3 = _mm512_and_si512(t3, mask);
    // conditional subtract p: d = t - p with a borrow chain; keep d where no final borrow
    __m512i d0 = _mm512_sub_epi64(t0, p0);
    __m512i d1 = _mm512_sub_epi64(_mm512_sub_epi64(t1, p1), _mm512_srli_epi64(d0, 63));
    __m512i d2 = _mm512_sub_epi64(_mm512_sub_epi64(t2, p2), _mm512_srli_epi64(d1, 63));
    __m512i d3 = _mm512_sub_epi64(_mm512_sub_epi64(t3, p3), _mm512_srli_epi64(d2, 63));
    __m512i d4 = _mm512_sub_epi64(_mm512_sub_epi64(t4, p4), _mm512_srli_epi64(d3, 63));
    const __mmask8 neg = _mm512_cmplt_epi64_mask(d4, _mm512_setzero_si512());   // sign bit set: t < p, keep t
    V r;
    r.l[0] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d0, mask), t0);
    r.l[1] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d1, mask), t1);
    r.l[2] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d2, mask), t2);
    r.l[3] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d3, mask), t3);
    r.l[4] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d4, mask), t4);
    return r;
}

// (a + b) mod p, inputs < p with 52-bit limbs
static inline V add_mod(const V& a, const V& b) {
    const __m512i mask = bc(M52);
    const __m512i p0 = bc(g_p52[0]), p1 = bc(g_p52[1]), p2 = bc(g_p52[2]), p3 = bc(g_p52[3]), p4 = bc(g_p52[4]);
    __m512i t0 = _mm512_add_epi64(a.l[0], b.l[0]);
    __m512i t1 = _mm512_add_epi64(a.l[1], b.l[1]);
    __m512i t2 = _mm512_add_epi64(a.l[2], b.l[2]);
    __m512i t3 = _mm512_add_epi64(a.l[3], b.l[3]);
    __m512i t4 = _mm512_add_epi64(a.l[4], b.l[4]);
    t1 = _mm512_add_epi64(t1, _mm512_srli_epi64(t0, 52)); t0 = _mm512_and_si512(t0, mask);
    t2 = _mm512_add_epi64(t2, _mm512_srli_epi64(t1, 52)); t1 = _mm512_and_si512(t1, mask);
    t3 = _mm512_add_epi64(t3, _mm512_srli_epi64(t2, 52)); t2 = _mm512_and_si512(t2, mask);
    t4 = _mm512_add_epi64(t4, _mm512_srli_epi64(t3, 52)); t3 = _mm512_and_si512(t3, mask);
    __m512i d0 = _mm512_sub_epi64(t0, p0);
    __m512i d1 = _mm512_sub_epi64(_mm512_sub_epi64(t1, p1), _mm512_srli_epi64(d0, 63));
    __m512i d2 = _mm512_sub_epi64(_mm512_sub_epi64(t2, p2), _mm512_srli_epi64(d1, 63));
    __m512i d3 = _mm512_sub_epi64(_mm512_sub_epi64(t3, p3), _mm512_srli_epi64(d2, 63));
    __m512i d4 = _mm512_sub_epi64(_mm512_sub_epi64(t4, p4), _mm512_srli_epi64(d3, 63));
    const __mmask8 neg = _mm512_cmplt_epi64_mask(d4, _mm512_setzero_si512());
    V r;
    r.l[0] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d0, mask), t0);
    r.l[1] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d1, mask), t1);
    r.l[2] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d2, mask), t2);
    r.l[3] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d3, mask), t3);
    r.l[4] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d4, mask), t4);
    return r;
}

// Lazy forms for the inside of the MiMC7 round chain: no conditional subtraction.  With R = 2^260 and
// p < 2^254 a Montgomery product of operands below 2^257 is below 2^254 + p < 2^255, and a sum of three
// such values stays below 2^257, so the chain never needs a canonical value until the very end.
static inline V mont_mul_lazy(const V& a, const V& b) {
    const __m512i zero = _mm512_setzero_si512(), mask = bc(M52), pinv = bc(g_pinv52);
    __m512i t0 = zero, t1 = zero, t2 = zero, t3 = zero, t4 = zero, t5 = zero;
    const __m512i p0 = bc(g_p52[0]), p1 = bc(g_p52[1]), p2 = bc(g_p52[2]), p3 = bc(g_p52[3]), p4 = bc(g_p52[4]);
#define GKR_IFMA_ROUND(bi)                                        \
    {                                                             \
        t0 = _mm512_madd52lo_epu64(t0, a.l[0], bi);               \
        t1 = _mm512_madd52hi_epu64(t1, a.l[0], bi);               \
        t1 = _mm512_madd52lo_epu64(t1, a.l[1], bi);               \
        t2 = _mm512_madd52hi_epu64(t2, a.l[1], bi);               \
        t2 = _mm512_madd52lo_epu64(t2, a.l[2], bi);               \
        t3 = _mm512_madd52hi_epu64(t3, a.l[2], bi);               \
        t3 = _mm512_madd52lo_epu64(t3, a.l[3], bi);               \
        t4 = _mm512_madd52hi_epu64(t4, a.l[3], bi);               \
        t4 = _mm512_madd52lo_epu64(t4, a.l[4], bi);               \
        t5 = _mm512_madd52hi_epu64(t5, a.l[4], bi);               \
        const __m512i m = _mm512_and_si512(_mm512_madd52lo_epu64(zero, t0, pinv), mask); \
        t0 = _mm512_madd52lo_epu64(t0, m, p0);                    \
        t1 = _mm512_madd52hi_epu64(t1, m, p0);                    \
        t1 = _mm512_madd52lo_epu64(t1, m, p1);                    \
        t2 = _mm512_madd52hi_epu64(t2, m, p1);                    \
        t2 = _mm512_madd52lo_epu64(t2, m, p2);                    \
        t3 = _mm512_madd52hi_epu64(t3, m, p2);                    \
        t3 = _mm512_madd52lo_epu64(t3, m, p3);                    \
        t4 = _mm512_madd52hi_epu64(t4, m, p3);                    \
        t4 = _mm512_madd52lo_epu64(t4, m, p4);                    \
        t5 = _mm512_madd52hi_epu64(t5, m, p4);                    \
        t0 = _mm512_add_epi64(t1, _mm512_srli_epi64(t0, 52));     \
        t1 = t2; t2 = t3; t3 = t4; t4 = t5; t5 = zero;            \
    }
    GKR_IFMA_ROUND(b.l[0]) GKR_IFMA_ROUND(b.l[1]) GKR_IFMA_ROUND(b.l[2]) GKR_IFMA_ROUND(b.l[3]) GKR_IFMA_ROUND(b.l[4])
#undef GKR_IFMA_ROUND
    V r;
    t1 = _mm512_add_epi64(t1, _mm512_srli_epi64(t0, 52)); r.l[0] = _mm512_and_si512(t0, mask);
    t2 = _mm512_add_epi64(t2, _mm512_srli_epi64(t1, 52)); r.l[1] = _mm512_and_si512(t1, mask);
    t3 = _mm512_add_epi64(t3, _mm512_srli_epi64(t2, 52)); r.l[2] = _mm512_and_si512(t2, mask);
    t4 = _mm512_add_epi64(t4, _mm512_srli_epi64(t3, 52)); r.l[3] = _mm512_and_si512(t3, mask);
    r.l[4] = t4;
    return r;
}

// a + b + c with carry normalisation only (limbs back to 52 bits; value may exceed p)
static inline V add3_lazy(const V& a, const V& b, const V& c) {
    const __m512i mask = bc(M52);
    __m512i t0 = _mm512_add_epi64(_mm512_add_epi64(a.l[0], b.l[0]), c.l[0]);
    __m512i t1 = _mm512_add_epi64(_mm512_add_epi64(a.l[1], b.l[1]), c.l[1]);
    __m512i t2 = _mm512_add_epi64(_mm512_add_epi64(a.l[2], b.l[2]), c.l[2]);
    __m512i t3 = _mm512_add_epi64(_mm512_add_epi64(a.l[3], b.l[3]), c.l[3]);
    __m512i t4 = _mm512_add_epi64(_mm512_add_epi64(a.l[4], b.l[4]), c.l[4]);
    V r;
    t1 = _mm512_add_epi64(t1, _mm512_srli_epi64(t0, 52)); r.l[0] = _mm512_and_si512(t0, mask);
    t2 = _mm512_add_epi64(t2, _mm512_srli_epi64(t1, 52)); r.l[1] = _mm512_and_si512(t1, mask);
    t3 = _mm512_add_epi64(t3, _mm512_srli_epi64(t2, 52)); r.l[2] = _mm512_and_si512(t2, mask);
    t4 = _mm512_add_epi64(t4, _mm512_srli_epi64(t3, 52)); r.l[3] = _mm512_and_si512(t3, mask);
    r.l[4] = t4;
    return r;
}

// subtract p where the value is >= p (limbs normalised, value < 2^260)
static inline V cond_sub(const V& t) {
    const __m512i mask = bc(M52);
    const __m512i p0 = bc(g_p52[0]), p1 = bc(g_p52[1]), p2 = bc(g_p52[2]), p3 = bc(g_p52[3]), p4 = bc(g_p52[4]);
    __m512i d0 = _mm512_sub_epi64(t.l[0], p0);
    __m512i d1 = _mm512_sub_epi64(_mm512_sub_epi64(t.l[1], p1), _mm512_srli_epi64(d0, 63));
    __m512i d2 = _mm512_sub_epi64(_mm512_sub_epi64(t.l[2], p2), _mm512_srli_epi64(d1, 63));
    __m512i d3 = _mm512_sub_epi64(_mm512_sub_epi64(t.l[3], p3), _mm512_srli_epi64(d2, 63));
    __m512i d4 = _mm512_sub_epi64(_mm512_sub_epi64(t.l[4], p4), _mm512_srli_epi64(d3, 63));
    const __mmask8 neg = _mm512_cmplt_epi64_mask(d4, _mm512_setzero_si512());
    V r;
    r.l[0] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d0, mask), t.l[0]);
    r.l[1] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d1, mask), t.l[1]);
    r.l[2] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d2, mask), t.l[2]);
    r.l[3] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d3, mask), t.l[3]);
    r.l[4] = _mm512_mask_blend_epi64(neg, _mm512_and_si512(d4, mask), t.l[4]);
    return r;
}

static inline V splat(const uint64_t l[5]) {
    V v;
    for (int i = 0; i < 5; ++i) v.l[i] = bc(l[i]);
    return v;
}

static inline V load8(const uint64_t (*x)[4]) {   // eight canonical 4x64 values -> limb-major 5x52
    uint64_t t[5][8];
    for (int k = 0; k < 8; ++k) {
        uint64_t o[5];
        to52(x[k], o);
        for (int i = 0; i < 5; ++i) t[i][k] = o[i];
    }
    V v;
    for (int i = 0; i < 5; ++i) v.l[i] = _mm512_loadu_si512(t[i]);
    return v;
}

static inline void store8(const V& v, uint64_t (*x)[4]) {
    uint64_t t[5][8];
    for (int i = 0; i < 5; ++i) _mm512_storeu_si512(t[i], v.l[i]);
    for (int k = 0; k < 8; ++k) {
        uint64_t l[5] = {t[0][k], t[1][k], t[2][k], t[3][k], t[4][k]};
        from52(l, x[k]);
    }
}

static void init_constants(const uint64_t (*cts_canonical)[4]) {
    to52(kP64, g_p52);
    // -p^-1 mod 2^52 by Newton iteration on the low word
    uint64_t inv = 1;
    for (int i = 0; i < 6; ++i) inv *= 2 - kP64[0] * inv;
    g_pinv52 = (0 - inv) & M52;
    // R^2 = 2^520 mod p by repeated doubling of 1
    uint64_t x[4] = {1, 0, 0, 0};
    for (int i = 0; i < 520; ++i) dbl_mod(x);
    to52(x, g_r2_52);
    uint64_t one[4] = {1, 0, 0, 0};
    to52(one, g_one52);
    uint64_t r256[4] = {1, 0, 0, 0};
    for (int i = 0; i < 256; ++i) dbl_mod(r256);
    to52(r256, g_r256_52);
    uint64_t r264[4];
    memcpy(r264, r256, sizeof r264);
    for (int i = 0; i < 8; ++i) dbl_mod(r264);
    to52(r264, g_r264_52);
    g_ready = true;   // mont_mul usable from here
    const V r2 = splat(g_r2_52);
    for (int base = 0; base < 91; base += 8) {
        uint64_t in[8][4];
        memset(in, 0, sizeof in);
        for (int k = 0; k < 8 && base + k < 91; ++k) memcpy(in[k], cts_canonical[base + k], 32);
        V m = mont_mul(load8(in), r2);
        uint64_t t[5][8];
        for (int i = 0; i < 5; ++i) _mm512_storeu_si512(t[i], m.l[i]);
        for (int k = 0; k < 8 && base + k < 91; ++k)
            for (int i = 0; i < 5; ++i) g_cts52[base + k][i] = t[i][k];
    }
}

// x, k Montgomery and canonical -> hash(x, k) Montgomery and canonical, eight lanes.  Inside the chain the
// values are only kept below 2^257 (see mont_mul_lazy); the last sum is canonicalised.
static inline V mimc7_hash(const V& x, const V& k) {
    V zero;
    for (int i = 0; i < 5; ++i) zero.l[i] = _mm512_setzero_si512();
    V h = zero;
    for (int i = 0; i < 91; ++i) {
        const V t = (i == 0) ? add3_lazy(x, k, zero) : add3_lazy(h, k, splat(g_cts52[i]));
        const V t2 = mont_mul_lazy(t, t);
        const V t4 = mont_mul_lazy(t2, t2);
        const V t6 = mont_mul_lazy(t4, t2);
        h = mont_mul_lazy(t6, t);
    }
    // h < 2^255, k < p: the sum is below 6p
    V s = add3_lazy(h, k, zero);
    for (int i = 0; i < 5; ++i) s = cond_sub(s);
    return s;
}

// W independent eight-lane products advanced together: one product is a chain of dependent madd52
// (4-cycle latency each), two fill the FMA pipes.
template <int W>
static inline void mont_mul_lazy_w(const V (&a)[W], const V (&b)[W], V (&r)[W]) {
    const __m512i zero = _mm512_setzero_si512(), mask = bc(M52), pinv = bc(g_pinv52);
    const __m512i p0 = bc(g_p52[0]), p1 = bc(g_p52[1]), p2 = bc(g_p52[2]), p3 = bc(g_p52[3]), p4 = bc(g_p52[4]);
    __m512i t[W][6];
#pragma GCC unroll 8
    for (int w = 0; w < W; ++w)
        for (int i = 0; i < 6; ++i) t[w][i] = zero;
#pragma GCC unroll 8
    for (int j = 0; j < 5; ++j) {
#pragma GCC unroll 8
        for (int w = 0; w < W; ++w) {
            const __m512i bi = b[w].l[j];
            t[w][0] = _mm512_madd52lo_epu64(t[w][0], a[w].l[0], bi);
            t[w][1] = _mm512_madd52hi_epu64(t[w][1], a[w].l[0], bi);
            t[w][1] = _mm512_madd52lo_epu64(t[w][1], a[w].l[1], bi);
            t[w][2] = _mm512_madd52hi_epu64(t[w][2], a[w].l[1], bi);
            t[w][2] = _mm512_madd52lo_epu64(t[w][2], a[w].l[2], bi);
            t[w][3] = _mm512_madd52hi_epu64(t[w][3], a[w].l[2], bi);
            t[w][3] = _mm512_madd52lo_epu64(t[w][3], a[w].l[3], bi);
            t[w][4] = _mm512_madd52hi_epu64(t[w][4], a[w].l[3], bi);
            t[w][4] = _mm512_madd52lo_epu64(t[w][4], a[w].l[4], bi);
            t[w][5] = _mm512_madd52hi_epu64(t[w][5], a[w].l[4], bi);
        }
#pragma GCC unroll 8
        for (int w = 0; w < W; ++w) {
            const __m512i m = _mm512_and_si512(_mm512_madd52lo_epu64(zero, t[w][0], pinv), mask);
            t[w][0] = _mm512_madd52lo_epu64(t[w][0], m, p0);
            t[w][1] = _mm512_madd52hi_epu64(t[w][1], m, p0);
            t[w][1] = _mm512_madd52lo_epu64(t[w][1], m, p1);
            t[w][2] = _mm512_madd52hi_epu64(t[w][2], m, p1);
            t[w][2] = _mm512_madd52lo_epu64(t[w][2], m, p2);
            t[w][3] = _mm512_madd52hi_epu64(t[w][3], m, p2);
            t[w][3] = _mm512_madd52lo_epu64(t[w][3], m, p3);
            t[w][4] = _mm512_madd52hi_epu64(t[w][4], m, p3);
            t[w][4] = _mm512_madd52lo_epu64(t[w][4], m, p4);
            t[w][5] = _mm512_madd52hi_epu64(t[w][5], m, p4);
            t[w][0] = _mm512_add_epi64(t[w][1], _mm512_srli_epi64(t[w][0], 52));
            t[w][1] = t[w][2]; t[w][2] = t[w][3]; t[w][3] = t[w][4]; t[w][4] = t[w][5]; t[w][5] = zero;
        }
    }
#pragma GCC unroll 8
    for (int w = 0; w < W; ++w) {
        t[w][1] = _mm512_add_epi64(t[w][1], _mm512_srli_epi64(t[w][0], 52)); r[w].l[0] = _mm512_and_si512(t[w][0], mask);
        t[w][2] = _mm512_add_epi64(t[w][2], _mm512_srli_epi64(t[w][1], 52)); r[w].l[1] = _mm512_and_si512(t[w][1], mask);
        t[w][3] = _mm512_add_epi64(t[w][3], _mm512_srli_epi64(t[w][2], 52)); r[w].l[2] = _mm512_and_si512(t[w][2], mask);
        t[w][4] = _mm512_add_epi64(t[w][4], _mm512_srli_epi64(t[w][3], 52)); r[w].l[3] = _mm512_and_si512(t[w][3], mask);
        r[w].l[4] = t[w][4];
    }
}

// W independent eight-lane SQUARES, same contract and same result as mont_mul_lazy_w(a, a): the ten distinct
// off-diagonal limb products once and doubled, the five squares, then the five reduction rounds on the 10-limb
// product -- 85 madd52 instead of 105 (half of the products of x^7 are squares).
template <int W>
static inline void mont_sqr_lazy_w(const V (&a)[W], V (&r)[W]) {
    const __m512i zero = _mm512_setzero_si512(), mask = bc(M52), pinv = bc(g_pinv52);
    const __m512i p0 = bc(g_p52[0]), p1 = bc(g_p52[1]), p2 = bc(g_p52[2]), p3 = bc(g_p52[3]), p4 = bc(g_p52[4]);
    __m512i c[W][10];
#pragma GCC unroll 8
    for (int w = 0; w < W; ++w) {
        const __m512i a0 = a[w].l[0], a1 = a[w].l[1], a2 = a[w].l[2], a3 = a[w].l[3], a4 = a[w].l[4];
        __m512i* t = c[w];
        t[1] = _mm512_madd52lo_epu64(zero, a0, a1);
        t[2] = _mm512_madd52hi_epu64(zero, a0, a1);
        t[2] = _mm512_madd52lo_epu64(t[2], a0, a2);
        t[3] = _mm512_madd52hi_epu64(zero, a0, a2);
        t[3] = _mm512_madd52lo_epu64(t[3], a0, a3);
        t[4] = _mm512_madd52hi_epu64(zero, a0, a3);
        t[4] = _mm512_madd52lo_epu64(t[4], a0, a4);
        t[5] = _mm512_madd52hi_epu64(zero, a0, a4);
        t[3] = _mm512_madd52lo_epu64(t[3], a1, a2);
        t[4] = _mm512_madd52hi_epu64(t[4], a1, a2);
        t[4] = _mm512_madd52lo_epu64(t[4], a1, a3);
        t[5] = _mm512_madd52hi_epu64(t[5], a1, a3);
        t[5] = _mm512_madd52lo_epu64(t[5], a1, a4);
        t[6] = _mm512_madd52hi_epu64(zero, a1, a4);
        t[5] = _mm512_madd52lo_epu64(t[5], a2, a3);
        t[6] = _mm512_madd52hi_epu64(t[6], a2, a3);
        t[6] = _mm512_madd52lo_epu64(t[6], a2, a4);
        t[7] = _mm512_madd52hi_epu64(zero, a2, a4);
        t[7] = _mm512_madd52lo_epu64(t[7], a3, a4);
        t[8] = _mm512_madd52hi_epu64(zero, a3, a4);
        for (int i = 1; i <= 8; ++i) t[i] = _mm512_add_epi64(t[i], t[i]);
        t[0] = _mm512_madd52lo_epu64(zero, a0, a0);
        t[1] = _mm512_madd52hi_epu64(t[1], a0, a0);
        t[2] = _mm512_madd52lo_epu64(t[2], a1, a1);
        t[3] = _mm512_madd52hi_epu64(t[3], a1, a1);
        t[4] = _mm512_madd52lo_epu64(t[4], a2, a2);
        t[5] = _mm512_madd52hi_epu64(t[5], a2, a2);
        t[6] = _mm512_madd52lo_epu64(t[6], a3, a3);
        t[7] = _mm512_madd52hi_epu64(t[7], a3, a3);
        t[8] = _mm512_madd52lo_epu64(t[8], a4, a4);
        t[9] = _mm512_madd52hi_epu64(zero, a4, a4);
    }
#pragma GCC unroll 8
    for (int i = 0; i < 5; ++i) {
#pragma GCC unroll 8
        for (int w = 0; w < W; ++w) {
            __m512i* t = c[w] + i;
            const __m512i m = _mm512_and_si512(_mm512_madd52lo_epu64(zero, t[0], pinv), mask);
            t[0] = _mm512_madd52lo_epu64(t[0], m, p0);
            t[1] = _mm512_madd52hi_epu64(t[1], m, p0);
            t[1] = _mm512_madd52lo_epu64(t[1], m, p1);
            t[2] = _mm512_madd52hi_epu64(t[2], m, p1);
            t[2] = _mm512_madd52lo_epu64(t[2], m, p2);
            t[3] = _mm512_madd52hi_epu64(t[3], m, p2);
            t[3] = _mm512_madd52lo_epu64(t[3], m, p3);
            t[4] = _mm512_madd52hi_epu64(t[4], m, p3);
            t[4] = _mm512_madd52lo_epu64(t[4], m, p4);
            t[5] = _mm512_madd52hi_epu64(t[5], m, p4);
            t[1] = _mm512_add_epi64(t[1], _mm512_srli_epi64(t[0], 52));   // the low 52 bits of t[0] are zero now
        }
    }
#pragma GCC unroll 8
    for (int w = 0; w < W; ++w) {
        __m512i* t = c[w] + 5;
        t[1] = _mm512_add_epi64(t[1], _mm512_srli_epi64(t[0], 52)); r[w].l[0] = _mm512_and_si512(t[0], mask);
        t[2] = _mm512_add_epi64(t[2], _mm512_srli_epi64(t[1], 52)); r[w].l[1] = _mm512_and_si512(t[1], mask);
        t[3] = _mm512_add_epi64(t[3], _mm512_srli_epi64(t[2], 52)); r[w].l[2] = _mm512_and_si512(t[2], mask);
        t[4] = _mm512_add_epi64(t[4], _mm512_srli_epi64(t[3], 52)); r[w].l[3] = _mm512_and_si512(t[3], mask);
        r[w].l[4] = t[4];
    }
}

template <int W>
static inline void mimc7_hash_w(const V (&x)[W], const V (&k)[W], V (&out)[W]) {
    V zero;
    for (int i = 0; i < 5; ++i) zero.l[i] = _mm512_setzero_si512();
    V h[W], t[W], t2[W], t4[W], t6[W];
    for (int w = 0; w < W; ++w) h[w] = zero;
    for (int i = 0; i < 91; ++i) {
        const V c = (i == 0) ? zero : splat(g_cts52[i]);
        for (int w = 0; w < W; ++w) t[w] = (i == 0) ? add3_lazy(x[w], k[w], zero) : add3_lazy(h[w], k[w], c);
#ifndef GKR_IFMA_NO_SQUARE   // (A/B build switch for tools/hash_timing.cpp)
        mont_sqr_lazy_w<W>(t, t2);
        mont_sqr_lazy_w<W>(t2, t4);
#else
        mont_mul_lazy_w<W>(t, t, t2);
        mont_mul_lazy_w<W>(t2, t2, t4);
#endif
        mont_mul_lazy_w<W>(t4, t2, t6);
        mont_mul_lazy_w<W>(t6, t, h);
    }
    for (int w = 0; w < W; ++w) {
        V s = add3_lazy(h[w], k[w], zero);
        for (int i = 0; i < 5; ++i) s = cond_sub(s);
        out[w] = s;
    }
}

template <int W>
static void multi_hash_w(const uint64_t (*vec)[3][4], const uint32_t* len, int slots, uint64_t (*out)[4]) {
    const V r2 = splat(g_r2_52), one = splat(g_one52);
    V r[W], a[W], h[W];
    __mmask8 active[W];
    for (int w = 0; w < W; ++w)
        for (int i = 0; i < 5; ++i) r[w].l[i] = _mm512_setzero_si512();
    for (int s = 0; s < slots; ++s) {
        unsigned any = 0;
        for (int w = 0; w < W; ++w) {
            uint64_t in[8][4];
            active[w] = 0;
            for (int k = 0; k < 8; ++k) {
                memcpy(in[k], vec[8 * w + k][s], 32);
                if ((uint32_t)(slots - s) <= len[8 * w + k]) active[w] |= (__mmask8)(1u << k);
            }
            any |= active[w];
            a[w] = mont_mul(load8(in), r2);
        }
        if (!any) continue;
        mimc7_hash_w<W>(a, r, h);
        for (int w = 0; w < W; ++w) {
            const V nr = add_mod(add_mod(r[w], a[w]), h[w]);
            for (int i = 0; i < 5; ++i) r[w].l[i] = _mm512_mask_blend_epi64(active[w], r[w].l[i], nr.l[i]);
        }
    }
    for (int w = 0; w < W; ++w) store8(mont_mul(r[w], one), out + 8 * w);
}

// (a - b) mod p, inputs < p with 52-bit limbs: a + (p - b), the difference formed with a borrow chain
static inline V sub_mod(const V& a, const V& b) {
    const __m512i mask = bc(M52);
    const __m512i p0 = bc(g_p52[0]), p1 = bc(g_p52[1]), p2 = bc(g_p52[2]), p3 = bc(g_p52[3]), p4 = bc(g_p52[4]);
    V n;
    __m512i d0 = _mm512_sub_epi64(p0, b.l[0]);
    __m512i d1 = _mm512_sub_epi64(_mm512_sub_epi64(p1, b.l[1]), _mm512_srli_epi64(d0, 63));
    __m512i d2 = _mm512_sub_epi64(_mm512_sub_epi64(p2, b.l[2]), _mm512_srli_epi64(d1, 63));
    __m512i d3 = _mm512_sub_epi64(_mm512_sub_epi64(p3, b.l[3]), _mm512_srli_epi64(d2, 63));
    __m512i d4 = _mm512_sub_epi64(_mm512_sub_epi64(p4, b.l[4]), _mm512_srli_epi64(d3, 63));
    n.l[0] = _mm512_and_si512(d0, mask);
    n.l[1] = _mm512_and_si512(d1, mask);
    n.l[2] = _mm512_and_si512(d2, mask);
    n.l[3] = _mm512_and_si512(d3, mask);
    n.l[4] = _mm512_and_si512(d4, mask);   // p - b in [1, p]; add_mod folds a + p back below p
    return add_mod(a, n);
}

// eight canonical 4x64 values at base + k * stride_words (lanes >= count read lane 0's) <-> limb-major 5x52, all in
// vector registers: four gathers / scatters of the 64-bit words and shifts between the two radices
static inline __m512i lane_offsets(size_t stride_words, int count) {
    alignas(64) uint64_t o[8];
    for (int k = 0; k < 8; ++k) o[k] = (k < count ? (uint64_t)k : 0) * stride_words;
    return _mm512_load_si512(o);
}
static inline V gather8(const uint64_t* base, __m512i off) {
    const __m512i mask = bc(M52);
    const __m512i w0 = _mm512_i64gather_epi64(off, base + 0, 8), w1 = _mm512_i64gather_epi64(off, base + 1, 8);
    const __m512i w2 = _mm512_i64gather_epi64(off, base + 2, 8), w3 = _mm512_i64gather_epi64(off, base + 3, 8);
    V v;
    v.l[0] = _mm512_and_si512(w0, mask);
    v.l[1] = _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(w0, 52), _mm512_slli_epi64(w1, 12)), mask);
    v.l[2] = _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(w1, 40), _mm512_slli_epi64(w2, 24)), mask);
    v.l[3] = _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(w2, 28), _mm512_slli_epi64(w3, 36)), mask);
    v.l[4] = _mm512_srli_epi64(w3, 16);
    return v;
}
static inline void scatter8(const V& v, uint64_t* base, __m512i off, __mmask8 lanes) {
    const __m512i w0 = _mm512_or_si512(v.l[0], _mm512_slli_epi64(v.l[1], 52));
    const __m512i w1 = _mm512_or_si512(_mm512_srli_epi64(v.l[1], 12), _mm512_slli_epi64(v.l[2], 40));
    const __m512i w2 = _mm512_or_si512(_mm512_srli_epi64(v.l[2], 24), _mm512_slli_epi64(v.l[3], 28));
    const __m512i w3 = _mm512_or_si512(_mm512_srli_epi64(v.l[3], 36), _mm512_slli_epi64(v.l[4], 16));
    _mm512_mask_i64scatter_epi64(base + 0, lanes, off, w0, 8);
    _mm512_mask_i64scatter_epi64(base + 1, lanes, off, w1, 8);
    _mm512_mask_i64scatter_epi64(base + 2, lanes, off, w2, 8);
    _mm512_mask_i64scatter_epi64(base + 3, lanes, off, w3, 8);
}
static inline __mmask8 nonzero_lanes(const V& v) {
    const __m512i any = _mm512_or_si512(_mm512_or_si512(v.l[0], v.l[1]), _mm512_or_si512(_mm512_or_si512(v.l[2], v.l[3]), v.l[4]));
    return _mm512_test_epi64_mask(any, any);
}

// The host's whole share of one multi-round pass for 8 W sumchecks side by side (gkr_ifma_pass below): the 2^J
// sub-block sums stay in vector registers / L1 from the load to the last binding, the round vectors go into the
// hash without passing through memory.
template <int W>
static void pass_w(const uint64_t* sums, size_t sums_row_words, int count, int J, const uint32_t* final_len, uint64_t (*c0)[16][4],
                   uint64_t (*c1)[16][4], uint64_t (*r)[16][4], uint32_t (*len)[16], uint64_t* weights, size_t w_row_words) {
    const V r2 = splat(g_r2_52), one = splat(g_one52);
    V zero;
    for (int i = 0; i < 5; ++i) zero.l[i] = _mm512_setzero_si512();
    V S[W][32], rm[5][W];
    __m512i off_in[W], off_out[W], off_w[W];
    __mmask8 valid[W];
    int cnt[W];
    for (int w = 0; w < W; ++w) {
        cnt[w] = count - 8 * w < 0 ? 0 : (count - 8 * w > 8 ? 8 : count - 8 * w);
        valid[w] = (__mmask8)((1u << cnt[w]) - 1);
        off_in[w] = lane_offsets(sums_row_words, cnt[w]);
        off_out[w] = lane_offsets(4, cnt[w]);
        off_w[w] = lane_offsets(w_row_words, cnt[w]);
        const uint64_t* base = sums + (size_t)(cnt[w] ? 8 * w : 0) * sums_row_words;
        for (int b = 0; b < (1 << J); ++b) S[w][b] = gather8(base + 4 * (size_t)b, off_in[w]);
    }
    for (int t = 0; t < J; ++t) {
        const int half = 1 << (J - t - 1);
        V lo[W], d[W], a[W], acc[W], h[W];
        __mmask8 two[W];
        unsigned any_two = 0;
        for (int w = 0; w < W; ++w) {
            V l = S[w][0], u = S[w][half];
            for (int b = 1; b < half; ++b) {
                l = add_mod(l, S[w][b]);
                u = add_mod(u, S[w][half + b]);
            }
            lo[w] = l;
            d[w] = sub_mod(u, l);
            if (final_len && t == J - 1) {
                two[w] = 0;
                for (int k = 0; k < cnt[w]; ++k)
                    if (final_len[8 * w + k] == 2) two[w] |= (__mmask8)(1u << k);
            } else {
                two[w] = nonzero_lanes(d[w]) & valid[w];
            }
            any_two |= two[w];
            alignas(32) uint32_t ln[8];
            for (int k = 0; k < 8; ++k) ln[k] = (two[w] >> k) & 1 ? 2u : 1u;
            memcpy(&len[t][8 * w], ln, sizeof ln);
            scatter8(l, &c0[t][8 * w][0], off_out[w], valid[w]);
            scatter8(d[w], &c1[t][8 * w][0], off_out[w], valid[w]);
            acc[w] = zero;
        }
        // multi_hash([c1, c0]) where the vector has two entries, multi_hash([c0]) where it has one
        if (any_two) {
            for (int w = 0; w < W; ++w) a[w] = mont_mul(d[w], r2);
            mimc7_hash_w<W>(a, acc, h);
            for (int w = 0; w < W; ++w) {
                const V nr = add_mod(add_mod(acc[w], a[w]), h[w]);
                for (int i = 0; i < 5; ++i) acc[w].l[i] = _mm512_mask_blend_epi64(two[w], acc[w].l[i], nr.l[i]);
            }
        }
        for (int w = 0; w < W; ++w) a[w] = mont_mul(lo[w], r2);
        mimc7_hash_w<W>(a, acc, h);
        for (int w = 0; w < W; ++w) {
            rm[t][w] = add_mod(add_mod(acc[w], a[w]), h[w]);   // the challenge, Montgomery form (radix 2^260)
            scatter8(mont_mul(rm[t][w], one), &r[t][8 * w][0], off_out[w], valid[w]);
            for (int b = 0; b < half; ++b) S[w][b] = add_mod(S[w][b], mont_mul(sub_mod(S[w][half + b], S[w][b]), rm[t][w]));
        }
    }
    if (!weights) return;
    for (int w = 0; w < W; ++w) {
        if (!cnt[w]) continue;
        V* wv = S[w];   // the sums are spent
        wv[0] = splat(g_r256_52);
        int cur = 1;
        for (int t = 0; t < J; ++t) {
            for (int b = cur; b-- > 0;) {
                const V hi = mont_mul(wv[b], rm[t][w]);
                wv[2 * b + 1] = hi;
                wv[2 * b] = sub_mod(wv[b], hi);
            }
            cur <<= 1;
        }
        uint64_t* base = weights + (size_t)8 * w * w_row_words;
        for (int b = 0; b < cur; ++b) scatter8(wv[b], base + 4 * (size_t)b, off_w[w], valid[w]);
    }
}

// The host's share of one product pass (kernels.hip, "product passes") for 8 W layer sumchecks side by side: the
// 8 x 8 cross-sum matrix and the eight Y sums of every lane stay in vector registers / L1 from the gather to the
// last fold; per round the three coefficients, the hash of [c2, lin, c0] (without c2 where the vector has two
// entries), the matrix folded along both indices with the challenge.
template <int W>
static void prod_pass_w(const uint64_t* recs, size_t rec_row_words, int count, int J, const uint32_t (*vec_len)[16], uint64_t (*c2)[16][4],
                        uint64_t (*lin)[16][4], uint64_t (*c0)[16][4], uint64_t (*r)[16][4], uint64_t* weights, size_t w_row_words) {
    const V r2 = splat(g_r2_52), one = splat(g_one52);
    V zero;
    for (int i = 0; i < 5; ++i) zero.l[i] = _mm512_setzero_si512();
    V M[W][64], SY[W][8], rm[3][W];
    __m512i off_in[W], off_out[W], off_w[W];
    __mmask8 valid[W];
    int cnt[W];
    const int n = 1 << J;
    for (int w = 0; w < W; ++w) {
        cnt[w] = count - 8 * w < 0 ? 0 : (count - 8 * w > 8 ? 8 : count - 8 * w);
        valid[w] = (__mmask8)((1u << cnt[w]) - 1);
        off_in[w] = lane_offsets(rec_row_words, cnt[w]);
        off_out[w] = lane_offsets(4, cnt[w]);
        off_w[w] = lane_offsets(w_row_words, cnt[w]);
        const uint64_t* base = recs + (size_t)(cnt[w] ? 8 * w : 0) * rec_row_words;
        for (int a = 0; a < n; ++a) {
            for (int b = 0; b < n; ++b) M[w][a * 8 + b] = gather8(base + 4 * (size_t)(a * 8 + b), off_in[w]);
            SY[w][a] = gather8(base + 4 * (size_t)(64 + a), off_in[w]);
        }
    }
    for (int t = 0; t < J; ++t) {
        const int half = 1 << (J - t - 1);
        V vc2[W], vlin[W], vc0[W], a[W], acc[W], h[W];
        __mmask8 three[W];
        unsigned any_three = 0;
        for (int w = 0; w < W; ++w) {
            V p00 = M[w][0], p01 = M[w][half], p10 = M[w][half * 8], p11 = M[w][half * 8 + half], s0 = SY[w][0], s1 = SY[w][half];
            for (int x = 1; x < half; ++x) {
                p00 = add_mod(p00, M[w][x * 8 + x]);
                p01 = add_mod(p01, M[w][x * 8 + half + x]);
                p10 = add_mod(p10, M[w][(half + x) * 8 + x]);
                p11 = add_mod(p11, M[w][(half + x) * 8 + half + x]);
                s0 = add_mod(s0, SY[w][x]);
                s1 = add_mod(s1, SY[w][half + x]);
            }
            vc0[w] = add_mod(p00, s0);
            const V g1 = add_mod(p11, s1);
            vc2[w] = sub_mod(add_mod(p11, p00), add_mod(p10, p01));
            vlin[w] = sub_mod(sub_mod(g1, vc0[w]), vc2[w]);
            three[w] = 0;
            for (int k = 0; k < cnt[w]; ++k)
                if (vec_len[t][8 * w + k] == 3) three[w] |= (__mmask8)(1u << k);
            any_three |= three[w];
            scatter8(vc2[w], &c2[t][8 * w][0], off_out[w], valid[w]);
            scatter8(vlin[w], &lin[t][8 * w][0], off_out[w], valid[w]);
            scatter8(vc0[w], &c0[t][8 * w][0], off_out[w], valid[w]);
            acc[w] = zero;
        }
        // multi_hash([c2, lin, c0]) or multi_hash([lin, c0])
        if (any_three) {
            for (int w = 0; w < W; ++w) a[w] = mont_mul(vc2[w], r2);
            mimc7_hash_w<W>(a, acc, h);
            for (int w = 0; w < W; ++w) {
                const V nr = add_mod(add_mod(acc[w], a[w]), h[w]);
                for (int i = 0; i < 5; ++i) acc[w].l[i] = _mm512_mask_blend_epi64(three[w], acc[w].l[i], nr.l[i]);
            }
        }
        for (int w = 0; w < W; ++w) a[w] = mont_mul(vlin[w], r2);
        mimc7_hash_w<W>(a, acc, h);
        for (int w = 0; w < W; ++w) acc[w] = add_mod(add_mod(acc[w], a[w]), h[w]);
        for (int w = 0; w < W; ++w) a[w] = mont_mul(vc0[w], r2);
        mimc7_hash_w<W>(a, acc, h);
        for (int w = 0; w < W; ++w) {
            rm[t][w] = add_mod(add_mod(acc[w], a[w]), h[w]);   // the challenge, Montgomery form (radix 2^260)
            scatter8(mont_mul(rm[t][w], one), &r[t][8 * w][0], off_out[w], valid[w]);
            const V x = rm[t][w];
            for (int ra = 0; ra < half; ++ra)        // rows: the W index
                for (int cb = 0; cb < 2 * half; ++cb)
                    M[w][ra * 8 + cb] = add_mod(M[w][ra * 8 + cb], mont_mul(sub_mod(M[w][(half + ra) * 8 + cb], M[w][ra * 8 + cb]), x));
            for (int ra = 0; ra < half; ++ra)        // columns: the X index
                for (int cb = 0; cb < half; ++cb)
                    M[w][ra * 8 + cb] = add_mod(M[w][ra * 8 + cb], mont_mul(sub_mod(M[w][ra * 8 + half + cb], M[w][ra * 8 + cb]), x));
            for (int ra = 0; ra < half; ++ra) SY[w][ra] = add_mod(SY[w][ra], mont_mul(sub_mod(SY[w][half + ra], SY[w][ra]), x));
        }
    }
    if (!weights) return;
    for (int w = 0; w < W; ++w) {
        if (!cnt[w]) continue;
        V* wv = M[w];   // the matrix is spent
        wv[0] = splat(g_r256_52);
        int cur = 1;
        for (int t = 0; t < J; ++t) {
            for (int b = cur; b-- > 0;) {
                const V hi = mont_mul(wv[b], rm[t][w]);
                wv[2 * b + 1] = hi;
                wv[2 * b] = sub_mod(wv[b], hi);
            }
            cur <<= 1;
        }
        uint64_t* base = weights + (size_t)8 * w * w_row_words;
        for (int b = 0; b < cur; ++b) scatter8(wv[b], base + 4 * (size_t)b, off_w[w], valid[w]);
    }
}

}  // namespace ifma

// The host's share of one product pass of `count` <= 16 layer sumchecks (lanes), J <= 3 rounds:
//   in   recs[k * rec_row_words + 4 (a * 8 + b) ..]: lane k's cross sums m[a][b], a, b < 2^J, and at entry 64 + a its Y sums
//        vec_len[t][k]: 2 or 3, the length of lane k's round-t vector (3: W depends on the variable, poly.rs:388-420)
//   out  per round t < J and lane k the coefficients c2, lin, c0 of the round polynomial c2 x^2 + lin x + c0 and
//        r = multi_hash([c2, lin, c0] or [lin, c0], key 0); all canonical
//        weights (may be null): lane k's 2^J weights eq((r_0..r_{J-1}), b) times 2^256 at weights[k * w_row_words + 4 b]
void gkr_ifma_prod_pass(const uint64_t* recs, size_t rec_row_words, int count, int J, const uint32_t (*vec_len)[16], uint64_t (*c2)[16][4],
                        uint64_t (*lin)[16][4], uint64_t (*c0)[16][4], uint64_t (*r)[16][4], uint64_t* weights, size_t w_row_words) {
    if (count > 8)
        ifma::prod_pass_w<2>(recs, rec_row_words, count, J, vec_len, c2, lin, c0, r, weights, w_row_words);
    else
        ifma::prod_pass_w<1>(recs, rec_row_words, count, J, vec_len, c2, lin, c0, r, weights, w_row_words);
}

// The host tail of a phase's product passes (capi_layer.hip, host_tail_pass), eight field products per instruction group: the
// fold of the previous pass's 2^jp variables over eight consecutive entries at a time, the cross sums with the eight sub-blocks of
// X in the lanes.  tables: W (x 2^256), X, Y of 2^m entries each, `stride` elements apart, folded in place to 2^(m - jp) entries;
// weights: 2^jp values x 2^256; rec: 72 values (m[a * 8 + b], the sub-block sums of Y at 64 + a).  All canonical 4 x 64-bit.
// mont_mul divides by 2^260, the 4x64-bit code's Montgomery form carries 2^256: a weight goes in times 2^264 / 2^260 (then
// weight x entry is the entry's fold term in the entry's own form), and a cross sum of (W x 2^256) X / 2^260 comes out times
// 2^264 / 2^260 again.
void gkr_ifma_tail_pass(uint64_t* tables, size_t stride, uint32_t m, uint32_t jp, const uint64_t* weights, uint32_t J, uint64_t* rec) {
    using namespace ifma;
    const uint32_t mf = m - jp, len = 1u << mf, nsub = 1u << J, S = len >> J;
    const V c264 = splat(g_r264_52);
    if (jp) {
        V w[8];
        for (uint32_t b = 0; b < (1u << jp); ++b) {
            uint64_t l[5];
            to52(weights + 4 * b, l);
            w[b] = mont_mul(splat(l), c264);
        }
        for (int t = 0; t < 3; ++t) {
            uint64_t* T = tables + (size_t)t * stride * 4;
            for (uint32_t i = 0; i < len; i += 8) {
                const int count = len - i < 8 ? (int)(len - i) : 8;
                const __m512i off = lane_offsets(4, count);
                V acc = mont_mul(gather8(T + (size_t)i * 4, off), w[0]);
                for (uint32_t b = 1; b < (1u << jp); ++b) acc = add_mod(acc, mont_mul(gather8(T + (((size_t)b << mf) + i) * 4, off), w[b]));
                scatter8(acc, T + (size_t)i * 4, off, (__mmask8)((1u << count) - 1u));
            }
        }
    }
    const uint64_t *W = tables, *X = tables + stride * 4, *Y = tables + 2 * stride * 4;
    const __m512i sub_off = lane_offsets((size_t)S * 4, (int)nsub);     // lane b: sub-block b (lanes beyond nsub read sub-block 0's)
    const __mmask8 sub_lanes = (__mmask8)((1u << nsub) - 1u);
    const __m512i rec_off = lane_offsets(4, (int)nsub);
    uint64_t zero52[5] = {0, 0, 0, 0, 0};
    for (uint32_t a = 0; a < nsub; ++a) {
        V acc = splat(zero52);
        for (uint32_t i = 0; i < S; ++i) {
            uint64_t l[5];
            to52(W + ((size_t)a * S + i) * 4, l);
            acc = add_mod(acc, mont_mul(splat(l), gather8(X + (size_t)i * 4, sub_off)));
        }
        scatter8(mont_mul(acc, c264), rec + (size_t)a * 8 * 4, rec_off, sub_lanes);
    }
    V ysum = splat(zero52);
    for (uint32_t i = 0; i < S; ++i) ysum = add_mod(ysum, gather8(Y + (size_t)i * 4, sub_off));
    scatter8(ysum, rec + 64 * 4, rec_off, sub_lanes);
}

// The host's share of one multi-round pass of `count` <= 16 plain sumchecks (lanes), J <= 5 rounds:
//   in   sums[k * sums_row_words + 4 b .. +4), b < 2^J: lane k's sub-block sums (canonical)
//        final_len: null, or the vector length (1 | 2) of each lane's LAST round where that round is the sumcheck's
//        final one (its length follows the table's dependence on the last variable, not the coefficient)
//   out  per round t < J and lane k: c0[t][k] = sum of the lower half, c1[t][k] = upper - lower (the round polynomial
//        c1 x + c0), len[t][k] = 1 if c1 == 0 else 2, r[t][k] = multi_hash(the vector, key 0); all canonical
//        weights (may be null): lane k's 2^J weights eq((r_0..r_{J-1}), b) times 2^256 at weights[k * w_row_words + 4 b]
// Round t+1's sums come from binding round t's variable in the sub-block sums (linear in the table).
void gkr_ifma_pass(const uint64_t* sums, size_t sums_row_words, int count, int J, const uint32_t* final_len, uint64_t (*c0)[16][4],
                   uint64_t (*c1)[16][4], uint64_t (*r)[16][4], uint32_t (*len)[16], uint64_t* weights, size_t w_row_words) {
    if (count > 8)
        ifma::pass_w<2>(sums, sums_row_words, count, J, final_len, c0, c1, r, len, weights, w_row_words);
    else
        ifma::pass_w<1>(sums, sums_row_words, count, J, final_len, c0, c1, r, len, weights, w_row_words);
}

bool gkr_ifma_available() {
#if defined(__x86_64__)
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma");
#else
    return false;
#endif
}

void gkr_ifma_init(const uint64_t (*cts_canonical)[4]) { ifma::init_constants(cts_canonical); }

// Eight transcripts: lane k hashes the LAST len[k] of the `slots` elements vec[k][0..slots)
// (round vectors are right-aligned, highest degree first); out[k] = multi_hash(.., key 0), canonical.
void gkr_ifma_multi_hash8(const uint64_t (*vec)[3][4], const uint32_t* len, int slots, uint64_t (*out)[4]) {
    using namespace ifma;
    const V r2 = splat(g_r2_52), one = splat(g_one52);
    V r;
    for (int i = 0; i < 5; ++i) r.l[i] = _mm512_setzero_si512();
    for (int s = 0; s < slots; ++s) {
        uint64_t in[8][4];
        __mmask8 active = 0;
        for (int k = 0; k < 8; ++k) {
            memcpy(in[k], vec[k][s], 32);
            if ((uint32_t)(slots - s) <= len[k]) active |= (__mmask8)(1u << k);
        }
        if (!active) continue;
        const V a = mont_mul(load8(in), r2);
        const V h = mimc7_hash(a, r);
        const V nr = add_mod(add_mod(r, a), h);
        for (int i = 0; i < 5; ++i) r.l[i] = _mm512_mask_blend_epi64(active, r.l[i], nr.l[i]);
    }
    store8(mont_mul(r, one), out);
}

// Sixteen transcripts, as two interleaved groups of eight (same conventions; vec, len, out hold 16 entries).
void gkr_ifma_multi_hash16(const uint64_t (*vec)[3][4], const uint32_t* len, int slots, uint64_t (*out)[4]) {
    ifma::multi_hash_w<2>(vec, len, slots, out);
}

}  // namespace gkr
