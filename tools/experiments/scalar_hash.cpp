// One MiMC7-91 permutation on one host thread, six forms (see README.md here); the last one became gkr_amd/csrc/mimc_adx.cpp.
//   clang++ -O3 -std=c++17 -DGKR_HD= -I../../gkr_amd/csrc '-DTGT=__attribute__((target("bmi2,adx")))' scalar_hash.cpp \
//       ../../gkr_amd/csrc/keccak.o -o /tmp/scalar_hash && /tmp/scalar_hash        (-DXREG: left operand in registers)
#include "fr64.h"
#include "keccak.h"
#include "mimc7.h"
#include <chrono>
#include <cstdio>
#include <cstring>
using namespace gkr;
using h64::F; using h64::u128;

#ifndef TGT
#define TGT
#endif

// depth-3 ordering with the stock product
static inline F hash_d3(const F& x, const F& k, const F* cts) {
    F h = {{0, 0, 0, 0}};
    for (int i = 0; i < 91; ++i) {
        F t = (i == 0) ? h64::add(x, k) : h64::add(h64::add(h, k), cts[i]);
        F t2 = h64::mont_mul(t, t);
        F t3 = h64::mont_mul(t2, t);
        F t4 = h64::mont_mul(t2, t2);
        h = h64::mont_mul(t3, t4);
    }
    return h64::add(h, k);
}

// product first (16 independent 64x64), then four reduction steps; result < 2r kept lazily? -- here canonical
TGT static inline F mul2(const F& a, const F& b) {
    uint64_t t[8];
    {
        u128 p = (u128)a.l[0] * b.l[0]; t[0] = (uint64_t)p; uint64_t c = (uint64_t)(p >> 64);
        p = (u128)a.l[1] * b.l[0] + c; t[1] = (uint64_t)p; c = (uint64_t)(p >> 64);
        p = (u128)a.l[2] * b.l[0] + c; t[2] = (uint64_t)p; c = (uint64_t)(p >> 64);
        p = (u128)a.l[3] * b.l[0] + c; t[3] = (uint64_t)p; t[4] = (uint64_t)(p >> 64);
    }
    for (int i = 1; i < 4; ++i) {
        uint64_t c = 0;
        for (int j = 0; j < 4; ++j) {
            u128 p = (u128)a.l[j] * b.l[i] + t[i + j] + c;
            t[i + j] = (uint64_t)p; c = (uint64_t)(p >> 64);
        }
        t[i + 4] = c;
    }
    uint64_t top = 0;
    for (int i = 0; i < 4; ++i) {
        const uint64_t m = t[i] * h64::kInv;
        u128 p = (u128)m * h64::kMod[0] + t[i];
        uint64_t c = (uint64_t)(p >> 64);
        for (int j = 1; j < 4; ++j) {
            p = (u128)m * h64::kMod[j] + t[i + j] + c;
            t[i + j] = (uint64_t)p; c = (uint64_t)(p >> 64);
        }
        // propagate c into t[i+4..]
        u128 s = (u128)t[i + 4] + c + top;
        t[i + 4] = (uint64_t)s; top = (uint64_t)(s >> 64);
        // note: top carries to the next limb; handle by adding into t[i+5] next iteration
        if (i < 3) { /* carry moves one limb up with the next iteration's t[i+5] add */ }
        // emulate: add top to t[i+5] now
        if (i < 3 && top) { /* rare */ }
    }
    F out = {{t[4], t[5], t[6], t[7]}};
    if (top || h64::geq_mod(out)) h64::sub_mod(out);
    return out;
}

TGT static F hash_c(const F& x, const F& k, const F* cts) {
    F h = {{0, 0, 0, 0}};
    for (int i = 0; i < 91; ++i) {
        F t = (i == 0) ? h64::add(x, k) : h64::add(h64::add(h, k), cts[i]);
        F t2 = mul2(t, t);
        F t3 = mul2(t2, t);
        F t4 = mul2(t2, t2);
        h = mul2(t3, t4);
    }
    return h64::add(h, k);
}

// lazy domain: values < 3r held unreduced; product without the final subtraction
TGT static inline F mul_lazy(const F& a, const F& b) {
    uint64_t t[8];
    {
        u128 p = (u128)a.l[0] * b.l[0]; t[0] = (uint64_t)p; uint64_t c = (uint64_t)(p >> 64);
        p = (u128)a.l[1] * b.l[0] + c; t[1] = (uint64_t)p; c = (uint64_t)(p >> 64);
        p = (u128)a.l[2] * b.l[0] + c; t[2] = (uint64_t)p; c = (uint64_t)(p >> 64);
        p = (u128)a.l[3] * b.l[0] + c; t[3] = (uint64_t)p; t[4] = (uint64_t)(p >> 64);
    }
    for (int i = 1; i < 4; ++i) {
        uint64_t c = 0;
        for (int j = 0; j < 4; ++j) {
            u128 p = (u128)a.l[j] * b.l[i] + t[i + j] + c;
            t[i + j] = (uint64_t)p; c = (uint64_t)(p >> 64);
        }
        t[i + 4] = c;
    }
    uint64_t top = 0;
    for (int i = 0; i < 4; ++i) {
        const uint64_t m = t[i] * h64::kInv;
        u128 p = (u128)m * h64::kMod[0] + t[i];
        uint64_t c = (uint64_t)(p >> 64);
        for (int j = 1; j < 4; ++j) {
            p = (u128)m * h64::kMod[j] + t[i + j] + c;
            t[i + j] = (uint64_t)p; c = (uint64_t)(p >> 64);
        }
        u128 s = (u128)t[i + 4] + c + top;
        t[i + 4] = (uint64_t)s; top = (uint64_t)(s >> 64);
    }
    return F{{t[4], t[5], t[6], t[7]}};
}
static const uint64_t k2Mod[4] = {0x87c3eb27e0000002ULL, 0x5067d090f372e122ULL, 0x70a08b6d0302b0baULL, 0x60c89ce5c2634053ULL};
// a - 2r if a >= 2r (branch-free)
TGT static inline F csub2(const F& a) {
    uint64_t d[4]; uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) { u128 x = (u128)a.l[i] - k2Mod[i] - borrow; d[i] = (uint64_t)x; borrow = (uint64_t)(x >> 64) & 1; }
    F o; for (int i = 0; i < 4; ++i) o.l[i] = borrow ? a.l[i] : d[i];
    return o;
}
TGT static inline F add_nored(const F& a, const F& b) {
    F s; uint64_t c = 0;
    for (int i = 0; i < 4; ++i) { u128 t = (u128)a.l[i] + b.l[i] + c; s.l[i] = (uint64_t)t; c = (uint64_t)(t >> 64); }
    return s;
}
TGT static F hash_lazy(const F& x, const F& k, const F* cts) {
    F kc[91];
    for (int i = 1; i < 91; ++i) kc[i] = h64::add(k, cts[i]);
    F t = h64::add(x, k);
    F h;
    for (int i = 0;; ++i) {
        F t2 = mul_lazy(t, t);
        F t3 = mul_lazy(t2, t);
        F t4 = mul_lazy(t2, t2);
        h = csub2(mul_lazy(t3, t4));
        if (i == 90) break;
        t = add_nored(h, kc[i + 1]);
    }
    if (h64::geq_mod(h)) h64::sub_mod(h);
    return h64::add(h, k);
}

// 4 x 64 CIOS with mulx and the two carry chains of adcx / adox (BN254's modulus leaves the top bits free, so
// the last limb of every outer step cannot carry out); operands < 4r, result < (a b / (r 2^256) ... ) not reduced
#ifdef XREG
#define XC "r"
#else
#define XC "m"
#endif
#define ACC(YI) \
    "xorl %k[lo], %k[lo]\n\t" \
    "movq " YI ", %%rdx\n\t" \
    "mulxq %[x0], %[lo], %[A]\n\t" "adoxq %[lo], %[t0]\n\t" \
    "adcxq %[A], %[t1]\n\t" "mulxq %[x1], %[lo], %[A]\n\t" "adoxq %[lo], %[t1]\n\t" \
    "adcxq %[A], %[t2]\n\t" "mulxq %[x2], %[lo], %[A]\n\t" "adoxq %[lo], %[t2]\n\t" \
    "adcxq %[A], %[t3]\n\t" "mulxq %[x3], %[lo], %[A]\n\t" "adoxq %[lo], %[t3]\n\t" \
    "movl $0, %k[lo]\n\t" "adcxq %[lo], %[A]\n\t" "adoxq %[lo], %[A]\n\t"
#define RED \
    "movq %[qinv], %%rdx\n\t" "imulq %[t0], %%rdx\n\t" \
    "xorl %k[lo], %k[lo]\n\t" \
    "mulxq %[q0], %[lo], %[C]\n\t" "adcxq %[t0], %[lo]\n\t" "movq %[C], %[t0]\n\t" \
    "adcxq %[t1], %[t0]\n\t" "mulxq %[q1], %[lo], %[t1]\n\t" "adoxq %[lo], %[t0]\n\t" \
    "adcxq %[t2], %[t1]\n\t" "mulxq %[q2], %[lo], %[t2]\n\t" "adoxq %[lo], %[t1]\n\t" \
    "adcxq %[t3], %[t2]\n\t" "mulxq %[q3], %[lo], %[t3]\n\t" "adoxq %[lo], %[t2]\n\t" \
    "movl $0, %k[lo]\n\t" "adcxq %[lo], %[t3]\n\t" "adoxq %[A], %[t3]\n\t"
static const uint64_t kModA[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t kInvA = 0xc2e1f593efffffffULL;
TGT static inline F mul_adx(const F& x, const F& y) {
    uint64_t t0, t1, t2, t3, A, C, lo;
    asm("movq %[y0], %%rdx\n\t"
        "mulxq %[x0], %[t0], %[t1]\n\t"
        "mulxq %[x1], %[lo], %[t2]\n\t" "addq %[lo], %[t1]\n\t"
        "mulxq %[x2], %[lo], %[t3]\n\t" "adcq %[lo], %[t2]\n\t"
        "mulxq %[x3], %[lo], %[A]\n\t"  "adcq %[lo], %[t3]\n\t"
        "adcq $0, %[A]\n\t"
        RED ACC("%[y1]") RED ACC("%[y2]") RED ACC("%[y3]") RED
        : [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3), [A] "=&r"(A), [C] "=&r"(C), [lo] "=&r"(lo)
        : [x0] XC(x.l[0]), [x1] XC(x.l[1]), [x2] XC(x.l[2]), [x3] XC(x.l[3]),
          [y0] "m"(y.l[0]), [y1] "m"(y.l[1]), [y2] "m"(y.l[2]), [y3] "m"(y.l[3]),
          [q0] "m"(kModA[0]), [q1] "m"(kModA[1]), [q2] "m"(kModA[2]), [q3] "m"(kModA[3]), [qinv] "m"(kInvA)
        : "rdx", "cc");
    return F{{t0, t1, t2, t3}};
}
TGT static F hash_adx(const F& x, const F& k, const F* cts) {
    F kc[91];
    for (int i = 1; i < 91; ++i) kc[i] = h64::add(k, cts[i]);
    F t = h64::add(x, k);
    F h;
    for (int i = 0;; ++i) {
        F t2 = mul_adx(t, t);
        F t3 = mul_adx(t2, t);
        F t4 = mul_adx(t2, t2);
        h = csub2(mul_adx(t3, t4));
        if (i == 90) break;
        t = add_nored(h, kc[i + 1]);
    }
    if (h64::geq_mod(h)) h64::sub_mod(h);
    return h64::add(h, k);
}
TGT static F hash_adx_d4(const F& x, const F& k, const F* cts) {
    F kc[91];
    for (int i = 1; i < 91; ++i) kc[i] = h64::add(k, cts[i]);
    F t = h64::add(x, k);
    F h;
    for (int i = 0;; ++i) {
        F t2 = mul_adx(t, t);
        F t4 = mul_adx(t2, t2);
        F t6 = mul_adx(t4, t2);
        h = csub2(mul_adx(t6, t));
        if (i == 90) break;
        t = add_nored(h, kc[i + 1]);
    }
    if (h64::geq_mod(h)) h64::sub_mod(h);
    return h64::add(h, k);
}
TGT static F hash_a_tgt(const F& x, const F& k, const F* cts) { return h64::mimc7_hash_mont(x, k, cts); }
TGT static F hash_d3_tgt(const F& x, const F& k, const F* cts) { return hash_d3(x, k, cts); }

template <class H> static void run(const char* name, H hfn, const F* cts) {
    F x = {{5, 6, 7, 8}}, k = {{1, 2, 3, 4}};
    const int N = 3000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) { x = hfn(x, k, cts); }
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    printf("%-28s %6.2f us per permutation  (%.1f ns per product)  check %016llx\n", name, us, us * 1000 / 364, (unsigned long long)x.l[0]);
}

int main() {
    Fr cts32[91];
    mimc7_make_constants(cts32);
    F cts64[91];
    memcpy(cts64, cts32, sizeof cts64);
    for (int rep = 0; rep < 2; ++rep) {
        run("stock", [](const F& x, const F& k, const F* c) { return h64::mimc7_hash_mont(x, k, c); }, cts64);
        run("stock, depth 3", hash_d3, cts64);
        run("stock +bmi2/adx", hash_a_tgt, cts64);
        run("depth 3 +bmi2/adx", hash_d3_tgt, cts64);
        run("product-then-reduce, d3", hash_c, cts64);
        run("lazy, d3", hash_lazy, cts64);
        run("mulx/adx lazy, d3", hash_adx, cts64);
        run("mulx/adx lazy, d4", hash_adx_d4, cts64);
    }
}
