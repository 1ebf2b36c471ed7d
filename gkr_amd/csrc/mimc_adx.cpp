// MiMC7-91 for ONE transcript at the lowest latency the host offers.
//
// A round vector's hash is a serial chain of 91 * 4 modular products per element; the sixteen-lane IFMA form
// (mimc_ifma.cpp) has the best throughput, but a lone hash (one sumcheck in flight: latency_ms_batch1, the 2^24-gate
// layer, a proof chain of a few inputs) wants the shortest chain.  Here: 4 x 64-bit CIOS Montgomery products with
// mulx and the two independent carry chains of adcx / adox, no final subtraction (values stay below 3r between
// products, one conditional subtraction of 2r per round), and x^7 as x^2 -> (x^3, x^4) -> x^7 so that two of the
// four products overlap.  EPYC 9575F: 3.6 us per permutation against 5.4 us for the portable code of fr64.h.
//
// Reference call sites: Mimc7::new(91) rust/src/gkr/sumcheck.rs:45, prover.rs:10; multi_hash sumcheck.rs:84,129,152,
// prover.rs:78 (mimc-rs: r = key; for a in arr { r += a + hash(a, r) }).
#include "mimc_adx.h"
#include "fr64.h"

#define GKR_ADX __attribute__((target("bmi2,adx")))

namespace gkr {

bool gkr_adx_available() { return __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx"); }

namespace {

using h64::F;
typedef unsigned __int128 u128;

const uint64_t kQ[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
const uint64_t kQ2[4] = {0x87c3eb27e0000002ULL, 0x5067d090f372e122ULL, 0x70a08b6d0302b0baULL, 0x60c89ce5c2634053ULL};   // 2r
const uint64_t kQInv = 0xc2e1f593efffffffULL;   // -r^{-1} mod 2^64

// t += x * y_i: the low halves ride the OF chain, the high halves the CF chain
#define GKR_ACC(YI)                                                                                   \
    "xorl %k[lo], %k[lo]\n\t"                                                                         \
    "movq " YI ", %%rdx\n\t"                                                                          \
    "mulxq %[x0], %[lo], %[A]\n\t" "adoxq %[lo], %[t0]\n\t"                                           \
    "adcxq %[A], %[t1]\n\t" "mulxq %[x1], %[lo], %[A]\n\t" "adoxq %[lo], %[t1]\n\t"                   \
    "adcxq %[A], %[t2]\n\t" "mulxq %[x2], %[lo], %[A]\n\t" "adoxq %[lo], %[t2]\n\t"                   \
    "adcxq %[A], %[t3]\n\t" "mulxq %[x3], %[lo], %[A]\n\t" "adoxq %[lo], %[t3]\n\t"                   \
    "movl $0, %k[lo]\n\t" "adcxq %[lo], %[A]\n\t" "adoxq %[lo], %[A]\n\t"
// t = (t + m r) / 2^64 with m = t0 * (-1/r); the limb that arrives on top is A
#define GKR_RED                                                                                       \
    "movq %[qinv], %%rdx\n\t" "imulq %[t0], %%rdx\n\t"                                                \
    "xorl %k[lo], %k[lo]\n\t"                                                                         \
    "mulxq %[q0], %[lo], %[C]\n\t" "adcxq %[t0], %[lo]\n\t" "movq %[C], %[t0]\n\t"                    \
    "adcxq %[t1], %[t0]\n\t" "mulxq %[q1], %[lo], %[t1]\n\t" "adoxq %[lo], %[t0]\n\t"                 \
    "adcxq %[t2], %[t1]\n\t" "mulxq %[q2], %[lo], %[t2]\n\t" "adoxq %[lo], %[t1]\n\t"                 \
    "adcxq %[t3], %[t2]\n\t" "mulxq %[q3], %[lo], %[t3]\n\t" "adoxq %[lo], %[t2]\n\t"                 \
    "movl $0, %k[lo]\n\t" "adcxq %[lo], %[t3]\n\t" "adoxq %[A], %[t3]\n\t"

// x y / 2^256 mod r, not reduced.  For x, y < 4r every outer step leaves t' = (t + x y_i + m r) / 2^64 <
// t / 2^64 + x + r, so t < 5r (1 + 2^-63) < 2^256 (5r = 0.945 * 2^256): the top limb never carries out, which is what
// lets the last limb of GKR_RED be a plain sum.  The result is < x y / 2^256 + r.
GKR_ADX inline F mul_lazy(const F& x, const F& y) {
    uint64_t t0, t1, t2, t3, A, C, lo;
    asm("movq %[y0], %%rdx\n\t"
        "mulxq %[x0], %[t0], %[t1]\n\t"
        "mulxq %[x1], %[lo], %[t2]\n\t" "addq %[lo], %[t1]\n\t"
        "mulxq %[x2], %[lo], %[t3]\n\t" "adcq %[lo], %[t2]\n\t"
        "mulxq %[x3], %[lo], %[A]\n\t"  "adcq %[lo], %[t3]\n\t"
        "adcq $0, %[A]\n\t"
        GKR_RED GKR_ACC("%[y1]") GKR_RED GKR_ACC("%[y2]") GKR_RED GKR_ACC("%[y3]") GKR_RED
        : [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3), [A] "=&r"(A), [C] "=&r"(C), [lo] "=&r"(lo)
        : [x0] "m"(x.l[0]), [x1] "m"(x.l[1]), [x2] "m"(x.l[2]), [x3] "m"(x.l[3]),
          [y0] "m"(y.l[0]), [y1] "m"(y.l[1]), [y2] "m"(y.l[2]), [y3] "m"(y.l[3]),
          [q0] "m"(kQ[0]), [q1] "m"(kQ[1]), [q2] "m"(kQ[2]), [q3] "m"(kQ[3]), [qinv] "m"(kQInv)
        : "rdx", "cc");
    return F{{t0, t1, t2, t3}};
}

// a - m if a >= m (no branch)
GKR_ADX inline F cond_sub(const F& a, const uint64_t (&m)[4]) {
    uint64_t d[4], borrow = 0;
    for (int i = 0; i < 4; ++i) {
        const u128 x = (u128)a.l[i] - m[i] - borrow;
        d[i] = (uint64_t)x;
        borrow = (uint64_t)(x >> 64) & 1;
    }
    F o;
    for (int i = 0; i < 4; ++i) o.l[i] = borrow ? a.l[i] : d[i];
    return o;
}

GKR_ADX inline F add_plain(const F& a, const F& b) {   // no reduction: the caller knows a + b < 2^256
    F s;
    uint64_t c = 0;
    for (int i = 0; i < 4; ++i) {
        const u128 t = (u128)a.l[i] + b.l[i] + c;
        s.l[i] = (uint64_t)t;
        c = (uint64_t)(t >> 64);
    }
    return s;
}

// hash(x, k) of mimc7.h; x, k canonical Montgomery; result canonical Montgomery.
// Bounds in units of r (r / 2^256 = 0.1891): t < 3 -> t^2 < 2.71, t^3 < 2.54, t^4 < 2.39, t^7 < 2.15; minus 2r if
// >= 2r: < 2; plus (k + c_i) < 1: t < 3 again.
GKR_ADX F permutation(const F& x, const F& k, const F* cts) {
    F kc[91];
    for (int i = 1; i < 91; ++i) kc[i] = h64::add(k, cts[i]);   // off the chain
    F t = add_plain(x, k);
    F h;
    for (int i = 0;; ++i) {
        const F t2 = mul_lazy(t, t);
        const F t3 = mul_lazy(t2, t);
        const F t4 = mul_lazy(t2, t2);
        h = cond_sub(mul_lazy(t3, t4), kQ2);
        if (i == 90) break;
        t = add_plain(h, kc[i + 1]);
    }
    return h64::add(cond_sub(h, kQ), k);
}

}  // namespace

GKR_ADX void gkr_adx_multi_hash(const uint64_t (*arr)[4], int n, const uint64_t (*cts_mont)[4], uint64_t* out) {
    const F* cts = reinterpret_cast<const F*>(cts_mont);
    const F one = {{1, 0, 0, 0}};
    F r = {{0, 0, 0, 0}};
    for (int i = 0; i < n; ++i) {
        F a;
        for (int j = 0; j < 4; ++j) a.l[j] = arr[i][j];
        a = cond_sub(mul_lazy(a, h64::kR2), kQ);   // < 0.19 + 1
        const F h = permutation(a, r, cts);
        r = h64::add(h64::add(r, a), h);
    }
    const F canon = cond_sub(mul_lazy(r, one), kQ);
    for (int j = 0; j < 4; ++j) out[j] = canon.l[j];
}

}  // namespace gkr
