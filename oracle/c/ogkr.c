/* Plain-C dense oracle -- see ogkr.h.  TEST INFRASTRUCTURE ONLY. */
#include "ogkr.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* BN254 scalar modulus r (halo2curves bn256::Fr, rust/Cargo.toml:21) */
static const u64 MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
                           0x30644e72e131a029ULL};
static const u64 INV = 0xc2e1f593efffffffULL;                       /* -r^{-1} mod 2^64 */
static const u64 R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL,
                          0x0216d0b17f4e44a5ULL};                   /* 2^512 mod r */
static const u64 ONE[4] = {1, 0, 0, 0};

static inline int geq_mod(const u64 a[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > MOD[i]) return 1;
        if (a[i] < MOD[i]) return 0;
    }
    return 1;
}

static inline void sub_mod_inplace(u64 a[4]) {
    u64 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - MOD[i] - borrow;
        a[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
}

static inline void fr_add(const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 carry = 0;
    for (int i = 0; i < 4; ++i) {
        u128 s = (u128)a[i] + b[i] + carry;
        out[i] = (u64)s;
        carry = (u64)(s >> 64);
    }
    /* a, b < r < 2^254 so no carry out of 256 bits */
    if (geq_mod(out)) sub_mod_inplace(out);
}

static inline void fr_sub(const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        out[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
    if (borrow) {
        u64 carry = 0;
        for (int i = 0; i < 4; ++i) {
            u128 s = (u128)out[i] + MOD[i] + carry;
            out[i] = (u64)s;
            carry = (u64)(s >> 64);
        }
    }
}

/* Montgomery product a*b*2^-256 mod r (coarsely integrated operand scanning) */
static inline void mont_mul(const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u64 carry = 0;
        for (int j = 0; j < 4; ++j) {
            u128 p = (u128)a[j] * b[i] + t[j] + carry;
            t[j] = (u64)p;
            carry = (u64)(p >> 64);
        }
        u128 s = (u128)t[4] + carry;
        t[4] = (u64)s;
        t[5] = (u64)(s >> 64);
        u64 m = t[0] * INV;
        u128 p = (u128)m * MOD[0] + t[0];
        carry = (u64)(p >> 64);
        for (int j = 1; j < 4; ++j) {
            p = (u128)m * MOD[j] + t[j] + carry;
            t[j - 1] = (u64)p;
            carry = (u64)(p >> 64);
        }
        s = (u128)t[4] + carry;
        t[3] = (u64)s;
        t[4] = t[5] + (u64)(s >> 64);
    }
    memcpy(out, t, 32);
    if (t[4] || geq_mod(out)) sub_mod_inplace(out);
}

static inline void to_mont(const u64 a[4], u64 out[4]) { mont_mul(a, R2, out); }
static inline void from_mont(const u64 a[4], u64 out[4]) { mont_mul(a, ONE, out); }

/* canonical * canonical -> canonical */
static inline void fr_mul(const u64 a[4], const u64 b[4], u64 out[4]) {
    u64 am[4];
    to_mont(a, am);          /* a R */
    mont_mul(am, b, out);    /* a R b R^-1 = a b */
}

void ogkr_fr_add(const ogkr_fr *a, const ogkr_fr *b, ogkr_fr *out) { fr_add(a->l, b->l, out->l); }
void ogkr_fr_sub(const ogkr_fr *a, const ogkr_fr *b, ogkr_fr *out) { fr_sub(a->l, b->l, out->l); }
void ogkr_fr_mul(const ogkr_fr *a, const ogkr_fr *b, ogkr_fr *out) { fr_mul(a->l, b->l, out->l); }
int ogkr_fr_is_canonical(const ogkr_fr *a) { return !geq_mod(a->l); }

void ogkr_set_threads(int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
}

int ogkr_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------- keccak-256 */

static const u64 KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int KPIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};

static inline u64 rol64(u64 x, int n) { return (x << n) | (x >> (64 - n)); }

static void keccak_f(u64 st[25]) {
    for (int round = 0; round < 24; ++round) {
        u64 bc[5];
        for (int i = 0; i < 5; ++i) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; ++i) {
            u64 t = bc[(i + 4) % 5] ^ rol64(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        u64 t = st[1];
        for (int i = 0; i < 24; ++i) {
            int j = KPIL[i];
            u64 b = st[j];
            st[j] = rol64(t, KROT[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; ++i) bc[i] = st[j + i];
            for (int i = 0; i < 5; ++i) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= KRC[round];
    }
}

void ogkr_keccak256(const uint8_t *data, size_t len, uint8_t out[32]) {
    const size_t rate = 136;
    u64 st[25];
    memset(st, 0, sizeof st);
    uint8_t block[136];
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; ++i) {
            u64 w;
            memcpy(&w, data + 8 * i, 8);
            st[i] ^= w;
        }
        keccak_f(st);
        data += rate;
        len -= rate;
    }
    memset(block, 0, rate);
    memcpy(block, data, len);
    block[len] ^= 0x01;           /* original Keccak padding, not SHA-3's 0x06 */
    block[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; ++i) {
        u64 w;
        memcpy(&w, block + 8 * i, 8);
        st[i] ^= w;
    }
    keccak_f(st);
    memcpy(out, st, 32);
}

/* -------------------------------------------------------------------- MiMC7 */

#define MIMC_ROUNDS 91
static u64 CTS_MONT[MIMC_ROUNDS][4];
static int cts_ready = 0;

static void reduce_be32(const uint8_t be[32], u64 out[4]) {
    /* int_big_endian(be) mod r; 2^256 < 6 r so at most five subtractions */
    for (int i = 0; i < 4; ++i) {
        u64 w = 0;
        for (int j = 0; j < 8; ++j) w = (w << 8) | be[8 * (3 - i) + j];
        out[i] = w;
    }
    while (geq_mod(out)) sub_mod_inplace(out);
}

static void init_constants(void) {
#pragma omp critical(ogkr_cts)
    {
        if (!cts_ready) {
            uint8_t h[32];
            ogkr_keccak256((const uint8_t *)"mimc", 4, h);
            memset(CTS_MONT[0], 0, 32);
            for (int i = 1; i < MIMC_ROUNDS; ++i) {
                uint8_t nh[32];
                ogkr_keccak256(h, 32, nh);
                memcpy(h, nh, 32);
                u64 c[4];
                reduce_be32(h, c);
                to_mont(c, CTS_MONT[i]);
            }
            cts_ready = 1;
        }
    }
}

void ogkr_mimc7_constant(int i, ogkr_fr *out) {
    if (!cts_ready) init_constants();
    from_mont(CTS_MONT[i], out->l);
}

/* x, k in Montgomery form; out in Montgomery form */
static void mimc7_hash_mont(const u64 x[4], const u64 k[4], u64 out[4]) {
    u64 h[4] = {0, 0, 0, 0}, t[4], t2[4], t4[4], t6[4];
    for (int i = 0; i < MIMC_ROUNDS; ++i) {
        if (i == 0) {
            fr_add(x, k, t);
        } else {
            fr_add(h, k, t);
            fr_add(t, CTS_MONT[i], t);
        }
        mont_mul(t, t, t2);
        mont_mul(t2, t2, t4);
        mont_mul(t4, t2, t6);
        mont_mul(t6, t, h);
    }
    fr_add(h, k, out);
}

void ogkr_mimc7_hash(const ogkr_fr *x, const ogkr_fr *k, ogkr_fr *out) {
    if (!cts_ready) init_constants();
    u64 xm[4], km[4], hm[4];
    to_mont(x->l, xm);
    to_mont(k->l, km);
    mimc7_hash_mont(xm, km, hm);
    from_mont(hm, out->l);
}

void ogkr_multi_hash(const ogkr_fr *arr, size_t n, const ogkr_fr *key, ogkr_fr *out) {
    if (!cts_ready) init_constants();
    u64 r[4], a[4], h[4];
    to_mont(key->l, r);
    for (size_t i = 0; i < n; ++i) {
        to_mont(arr[i].l, a);
        mimc7_hash_mont(a, r, h);
        fr_add(r, a, r);
        fr_add(r, h, r);
    }
    from_mont(r, out->l);
}

/* ------------------------------------------------------- synthetic workload */

static inline u64 mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void ogkr_fill_table(ogkr_fr *table, size_t count, uint64_t seed) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < count; ++i) {
        for (int j = 0; j < 4; ++j)
            table[i].l[j] = mix64(seed + (4 * (u64)i + (u64)j + 1) * 0x9E3779B97F4A7C15ULL);
        table[i].l[3] &= 0x1FFFFFFFFFFFFFFFULL;
    }
}

/* --------------------------------------------------------- dense sumchecks */

static int set_threads(int threads) {
#ifdef _OPENMP
    int t = threads > 0 ? threads : omp_get_max_threads();
    return t;
#else
    (void)threads;
    return 1;
#endif
}

static void fold_table(u64 (*t)[4], size_t h, const u64 r_mont[4], int nt) {
    /* T[i] <- T[i] + r (T[i+h] - T[i]) */
    (void)nt;
#pragma omp parallel for schedule(static) num_threads(nt)
    for (size_t i = 0; i < h; ++i) {
        u64 d[4], p[4];
        fr_sub(t[i + h], t[i], d);
        mont_mul(d, r_mont, p);
        fr_add(t[i], p, t[i]);
    }
}

static int sumcheck_mle_on(u64 (*t)[4], int n, ogkr_fr *out_coeffs, uint32_t *out_len, ogkr_fr *out_r, int threads);

int ogkr_sumcheck_mle(const ogkr_fr *table, int n, ogkr_fr *out_coeffs, uint32_t *out_len,
                      ogkr_fr *out_r, int threads) {
    if (n < 2 || n > 40) return -1;
    size_t len = (size_t)1 << n;
    u64(*t)[4] = malloc(len * 32);
    if (!t) return -2;
    memcpy(t, table, len * 32);
    int rc = sumcheck_mle_on(t, n, out_coeffs, out_len, out_r, threads);
    free(t);
    return rc;
}

/* the same on the caller's table, which is overwritten (a 2^30-entry table is 32 GiB: no room for a copy) */
int ogkr_sumcheck_mle_inplace(ogkr_fr *table, int n, ogkr_fr *out_coeffs, uint32_t *out_len, ogkr_fr *out_r, int threads) {
    if (n < 2 || n > 40) return -1;
    return sumcheck_mle_on((u64(*)[4])table, n, out_coeffs, out_len, out_r, threads);
}

static int sumcheck_mle_on(u64 (*t)[4], int n, ogkr_fr *out_coeffs, uint32_t *out_len, ogkr_fr *out_r, int threads) {
    if (!cts_ready) init_constants();
    int nt = set_threads(threads);
    size_t len = (size_t)1 << n;
    int dep_last = 0;
#pragma omp parallel for schedule(static) num_threads(nt) reduction(| : dep_last)
    for (size_t i = 0; i < len / 2; ++i)
        dep_last |= memcmp(t[2 * i], t[2 * i + 1], 32) != 0;
    const ogkr_fr zero = {{0, 0, 0, 0}};
    for (int j = 0; j < n; ++j) {
        size_t h = len >> (j + 1);
        u64 c0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
#pragma omp parallel num_threads(nt)
        {
            u64 l0[4] = {0, 0, 0, 0}, l1[4] = {0, 0, 0, 0};
#pragma omp for schedule(static) nowait
            for (size_t i = 0; i < h; ++i) {
                fr_add(l0, t[i], l0);
                fr_add(l1, t[i + h], l1);
            }
#pragma omp critical(ogkr_acc)
            {
                fr_add(c0, l0, c0);
                fr_add(s1, l1, s1);
            }
        }
        u64 c1[4];
        fr_sub(s1, c0, c1);
        int c1_zero = !(c1[0] | c1[1] | c1[2] | c1[3]);
        int length = (j < n - 1) ? (c1_zero ? 1 : 2) : (dep_last ? 2 : 1);
        out_len[j] = (uint32_t)length;
        out_coeffs[2 * j] = zero;
        memcpy(out_coeffs[2 * j + 1].l, c0, 32);
        if (length == 2) memcpy(out_coeffs[2 * j].l, c1, 32);
        ogkr_multi_hash(&out_coeffs[2 * j + (2 - length)], (size_t)length, &zero, &out_r[j]);
        u64 rm[4];
        to_mont(out_r[j].l, rm);
        fold_table(t, h, rm, nt);
    }
    return 0;
}

static void eq_table(int k, const ogkr_fr *z, u64 (*e)[4]) {
    /* e[g] = prod_i (bit_i(g) ? z_i : 1 - z_i), variable 1 = most significant bit */
    memcpy(e[0], ONE, 32);
    size_t cur = 1;
    for (int i = 0; i < k; ++i) {
        u64 zm[4];
        to_mont(z[i].l, zm);
        for (size_t g = cur; g-- > 0;) {
            u64 hi[4], lo[4];
            mont_mul(e[g], zm, hi);      /* e * z (canonical) */
            fr_sub(e[g], hi, lo);        /* e * (1 - z) */
            memcpy(e[2 * g], lo, 32);
            memcpy(e[2 * g + 1], hi, 32);
        }
        cur <<= 1;
    }
}

int ogkr_predicate_tables(int k_i, int k_next, const uint8_t *gate_type, const uint32_t *left,
                          const uint32_t *right, const ogkr_fr *z, ogkr_fr *A, ogkr_fr *M) {
    if (k_i < 0 || k_i > 30 || k_next < 1 || k_next > 15) return -1;
    size_t G = (size_t)1 << k_i, N = (size_t)1 << (2 * k_next);
    u64(*e)[4] = malloc(G * 32);
    if (!e) return -2;
    eq_table(k_i, z, e);
    memset(A, 0, N * 32);
    memset(M, 0, N * 32);
    for (size_t g = 0; g < G; ++g) {
        if (left[g] >> k_next || right[g] >> k_next) {
            free(e);
            return -1;
        }
        size_t idx = ((size_t)left[g] << k_next) | right[g];
        ogkr_fr *dst = gate_type[g] ? &M[idx] : &A[idx];
        fr_add(dst->l, e[g], dst->l);
    }
    free(e);
    return 0;
}

static void depends_on(int k, const ogkr_fr *W, int *dep) {
    size_t n = (size_t)1 << k;
    for (int b = 0; b < k; ++b) {
        size_t bit = (size_t)1 << (k - 1 - b);
        dep[b] = 0;
        for (size_t i = 0; i < n && !dep[b]; ++i)
            if (!(i & bit) && memcmp(W[i].l, W[i ^ bit].l, 32)) dep[b] = 1;
    }
}

int ogkr_sumcheck_layer(int k_i, int k_next, const uint8_t *gate_type, const uint32_t *left,
                        const uint32_t *right, const ogkr_fr *z, const ogkr_fr *W,
                        ogkr_fr *out_coeffs, uint32_t *out_len, ogkr_fr *out_r, int threads) {
    if (k_i < 0 || k_i > 30 || k_next < 1 || k_next > 15) return -1;
    if (!cts_ready) init_constants();
    int nt = set_threads(threads);
    const int k = k_next;
    const size_t N = (size_t)1 << (2 * k), mask = ((size_t)1 << k) - 1;
    ogkr_fr *A = malloc(N * 32), *M = malloc(N * 32);
    u64(*F1)[4] = malloc(N * 32), (*F2)[4] = malloc(N * 32);
    if (!A || !M || !F1 || !F2) return -2;
    int rc = ogkr_predicate_tables(k_i, k, gate_type, left, right, z, A, M);
    if (rc) return rc;
    /* F1, F2 hold W in Montgomery form so that mont_mul(canonical, F) is canonical */
#pragma omp parallel for schedule(static) num_threads(nt)
    for (size_t i = 0; i < N; ++i) {
        to_mont(W[i >> k].l, F1[i]);
        to_mont(W[i & mask].l, F2[i]);
    }
    int dep[16];
    depends_on(k, W, dep);
    u64(*a)[4] = (u64(*)[4])A, (*m)[4] = (u64(*)[4])M;
    const ogkr_fr zero = {{0, 0, 0, 0}};
    for (int j = 0; j < 2 * k; ++j) {
        size_t h = N >> (j + 1);
        u64 c0[4] = {0}, c1[4] = {0}, c2[4] = {0};
#pragma omp parallel num_threads(nt)
        {
            u64 l0[4] = {0}, l1[4] = {0}, l2[4] = {0};
#pragma omp for schedule(static) nowait
            for (size_t i = 0; i < h; ++i) {
                u64 da[4], dm[4], dp[4], dq[4], s0[4], ds[4], pq0[4], pq1[4], pq2[4], t0[4], t1[4];
                fr_sub(a[i + h], a[i], da);
                fr_sub(m[i + h], m[i], dm);
                fr_sub(F1[i + h], F1[i], dp);
                fr_sub(F2[i + h], F2[i], dq);
                fr_add(F1[i], F2[i], s0);
                fr_add(dp, dq, ds);
                mont_mul(F1[i], F2[i], pq0);          /* (pR)(qR)/R = pq R */
                mont_mul(F1[i], dq, t0);
                mont_mul(dp, F2[i], t1);
                fr_add(t0, t1, pq1);
                mont_mul(dp, dq, pq2);
                /* c0 += a0 s0 + m0 pq0 */
                mont_mul(a[i], s0, t0);
                mont_mul(m[i], pq0, t1);
                fr_add(l0, t0, l0);
                fr_add(l0, t1, l0);
                /* c1 += a0 ds + da s0 + m0 pq1 + dm pq0 */
                mont_mul(a[i], ds, t0);
                fr_add(l1, t0, l1);
                mont_mul(da, s0, t0);
                fr_add(l1, t0, l1);
                mont_mul(m[i], pq1, t0);
                fr_add(l1, t0, l1);
                mont_mul(dm, pq0, t0);
                fr_add(l1, t0, l1);
                /* c2 += da ds + m0 pq2 + dm pq1 */
                mont_mul(da, ds, t0);
                fr_add(l2, t0, l2);
                mont_mul(m[i], pq2, t0);
                fr_add(l2, t0, l2);
                mont_mul(dm, pq1, t0);
                fr_add(l2, t0, l2);
            }
#pragma omp critical(ogkr_acc3)
            {
                fr_add(c0, l0, c0);
                fr_add(c1, l1, c1);
                fr_add(c2, l2, c2);
            }
        }
        int length = 2 + (dep[j % k] ? 1 : 0);
        out_len[j] = (uint32_t)length;
        memcpy(out_coeffs[3 * j].l, c2, 32);
        memcpy(out_coeffs[3 * j + 1].l, c1, 32);
        memcpy(out_coeffs[3 * j + 2].l, c0, 32);
        if (length == 2) out_coeffs[3 * j] = zero;   /* provably zero; kept zero for a stable layout */
        ogkr_multi_hash(&out_coeffs[3 * j + (3 - length)], (size_t)length, &zero, &out_r[j]);
        u64 rm[4];
        to_mont(out_r[j].l, rm);
        fold_table(a, h, rm, nt);
        fold_table(m, h, rm, nt);
        fold_table(F1, h, rm, nt);
        fold_table(F2, h, rm, nt);
    }
    free(A);
    free(M);
    free(F1);
    free(F2);
    return 0;
}

/* ---------------------------------------- the layer sumcheck in linear time
 * The same transcript as ogkr_sumcheck_layer (prove_sumcheck_opt, rust/src/gkr/sumcheck.rs:36-156) without the
 * 2^{2k}-entry tables, so that it can follow layers of 2^15 .. 2^24 values: the C twin of oracle/gatesum.py.
 *     sum_c f(b, c) = W(b) U(b) + V(b),  U(b) = sum_c [a(b,c) + m(b,c) W(c)],  V(b) = sum_c a(b,c) W(c)
 * are sums over the gate list (sumcheck.rs:50-63 reduces over the same list); with b bound to u the rounds over c
 * run on the single row a_u(c) = sum_b eq(u,b) a(b,c), m_u(c) (sumcheck.rs:97-124).  Field arithmetic is exact, so
 * the round vectors are the field elements the dense form gives: tests/test_oracle_c.py holds the two against each
 * other for every k the dense form can reach. */
static void depends_on_wide(int k, const ogkr_fr *W, int *dep, int nt) {
    size_t n = (size_t)1 << k;
    (void)nt;
    for (int b = 0; b < k; ++b) {
        size_t bit = (size_t)1 << (k - 1 - b);
        int d = 0;
#pragma omp parallel for schedule(static) num_threads(nt) reduction(| : d)
        for (size_t i = 0; i < n; ++i)
            if (!(i & bit) && memcmp(W[i].l, W[i ^ bit].l, 32)) d |= 1;
        dep[b] = d;
    }
}

static void publish_round(int j, int dep_j, const u64 c2[4], const u64 g1[4], const u64 c0[4], ogkr_fr *out_coeffs,
                          uint32_t *out_len, ogkr_fr *out_r) {
    const ogkr_fr zero = {{0, 0, 0, 0}};
    u64 c1[4];
    fr_sub(g1, c0, c1);
    fr_sub(c1, c2, c1);
    int length = 2 + (dep_j ? 1 : 0);
    out_len[j] = (uint32_t)length;
    memcpy(out_coeffs[3 * j].l, c2, 32);
    memcpy(out_coeffs[3 * j + 1].l, c1, 32);
    memcpy(out_coeffs[3 * j + 2].l, c0, 32);
    if (length == 2) out_coeffs[3 * j] = zero;
    ogkr_multi_hash(&out_coeffs[3 * j + (3 - length)], (size_t)length, &zero, &out_r[j]);
}

int ogkr_sumcheck_layer_lin(int k_i, int k_next, const uint8_t *gate_type, const uint32_t *left,
                            const uint32_t *right, const ogkr_fr *z, const ogkr_fr *W,
                            ogkr_fr *out_coeffs, uint32_t *out_len, ogkr_fr *out_r, int threads) {
    if (k_i < 0 || k_i > 30 || k_next < 1 || k_next > 28) return -1;
    if (!cts_ready) init_constants();
    int nt = set_threads(threads);
    const int k = k_next;
    const size_t G = (size_t)1 << k_i, n = (size_t)1 << k;
    for (size_t g = 0; g < G; ++g)
        if (left[g] >> k || right[g] >> k || gate_type[g] > 1) return -1;
    u64(*e)[4] = malloc(G * 32), (*eu)[4] = malloc(n * 32);
    u64(*Wm)[4] = malloc(n * 32), (*Wt)[4] = malloc(n * 32);
    u64(*X)[4] = calloc(n, 32), (*Y)[4] = calloc(n, 32);
    if (!e || !eu || !Wm || !Wt || !X || !Y) return -2;
    eq_table(k_i, z, e);
#pragma omp parallel for schedule(static) num_threads(nt)
    for (size_t i = 0; i < n; ++i) {
        to_mont(W[i].l, Wm[i]);
        memcpy(Wt[i], Wm[i], 32);
    }
    int dep[32];
    depends_on_wide(k, W, dep, nt);
    /* U = X, V = Y: one pass over the gates (serial: the cells collide) */
    for (size_t g = 0; g < G; ++g) {
        u64 t[4];
        mont_mul(e[g], Wm[right[g]], t);                 /* eq(z, g) W(right) */
        if (gate_type[g]) {
            fr_add(X[left[g]], t, X[left[g]]);
        } else {
            fr_add(X[left[g]], e[g], X[left[g]]);
            fr_add(Y[left[g]], t, Y[left[g]]);
        }
    }
    /* the k rounds that bind b: g(x) = sum_i W_i(x) U_i(x) + V_i(x) */
    for (int j = 0; j < k; ++j) {
        size_t h = n >> (j + 1);
        u64 c0[4] = {0}, g1[4] = {0}, c2[4] = {0};
#pragma omp parallel num_threads(nt)
        {
            u64 l0[4] = {0}, l1[4] = {0}, l2[4] = {0};
#pragma omp for schedule(static) nowait
            for (size_t i = 0; i < h; ++i) {
                u64 t[4], dw[4], du[4];
                mont_mul(X[i], Wt[i], t);
                fr_add(l0, t, l0);
                fr_add(l0, Y[i], l0);
                mont_mul(X[i + h], Wt[i + h], t);
                fr_add(l1, t, l1);
                fr_add(l1, Y[i + h], l1);
                fr_sub(Wt[i + h], Wt[i], dw);
                fr_sub(X[i + h], X[i], du);
                mont_mul(du, dw, t);
                fr_add(l2, t, l2);
            }
#pragma omp critical(ogkr_lin_b)
            {
                fr_add(c0, l0, c0);
                fr_add(g1, l1, g1);
                fr_add(c2, l2, c2);
            }
        }
        publish_round(j, dep[j], c2, g1, c0, out_coeffs, out_len, out_r);
        u64 rm[4];
        to_mont(out_r[j].l, rm);
        fold_table(X, h, rm, nt);
        fold_table(Y, h, rm, nt);
        fold_table(Wt, h, rm, nt);
    }
    u64 wu[4];                       /* W(u), Montgomery form */
    memcpy(wu, Wt[0], 32);
    /* the rows a_u (X), m_u (Y) */
    eq_table(k, out_r, eu);
#pragma omp parallel for schedule(static) num_threads(nt)
    for (size_t i = 0; i < n; ++i) {
        to_mont(eu[i], eu[i]);
        memset(X[i], 0, 32);
        memset(Y[i], 0, 32);
        memcpy(Wt[i], Wm[i], 32);
    }
    for (size_t g = 0; g < G; ++g) {
        u64 t[4];
        mont_mul(e[g], eu[left[g]], t);                  /* eq(z, g) eq(u, left) */
        u64 *dst = gate_type[g] ? Y[right[g]] : X[right[g]];
        fr_add(dst, t, dst);
    }
    /* the k rounds that bind c on the row at b = u */
    for (int j = 0; j < k; ++j) {
        size_t h = n >> (j + 1);
        u64 c0[4] = {0}, g1[4] = {0}, c2[4] = {0};
#pragma omp parallel num_threads(nt)
        {
            u64 l0[4] = {0}, l1[4] = {0}, l2[4] = {0};
#pragma omp for schedule(static) nowait
            for (size_t i = 0; i < h; ++i) {
                u64 s0[4], s1[4], pq0[4], pq1[4], da[4], dm[4], ds[4], dpq[4], t[4];
                fr_add(wu, Wt[i], s0);
                fr_add(wu, Wt[i + h], s1);
                mont_mul(wu, Wt[i], pq0);
                mont_mul(wu, Wt[i + h], pq1);
                mont_mul(X[i], s0, t);
                fr_add(l0, t, l0);
                mont_mul(Y[i], pq0, t);
                fr_add(l0, t, l0);
                mont_mul(X[i + h], s1, t);
                fr_add(l1, t, l1);
                mont_mul(Y[i + h], pq1, t);
                fr_add(l1, t, l1);
                fr_sub(X[i + h], X[i], da);
                fr_sub(Y[i + h], Y[i], dm);
                fr_sub(s1, s0, ds);
                fr_sub(pq1, pq0, dpq);
                mont_mul(da, ds, t);
                fr_add(l2, t, l2);
                mont_mul(dm, dpq, t);
                fr_add(l2, t, l2);
            }
#pragma omp critical(ogkr_lin_c)
            {
                fr_add(c0, l0, c0);
                fr_add(g1, l1, g1);
                fr_add(c2, l2, c2);
            }
        }
        publish_round(k + j, dep[j], c2, g1, c0, out_coeffs, out_len, out_r);
        u64 rm[4];
        to_mont(out_r[k + j].l, rm);
        fold_table(X, h, rm, nt);
        fold_table(Y, h, rm, nt);
        fold_table(Wt, h, rm, nt);
    }
    free(e);
    free(eu);
    free(Wm);
    free(Wt);
    free(X);
    free(Y);
    return 0;
}

void ogkr_layer_eval(size_t gates, const uint8_t *gate_type, const uint32_t *left,
                     const uint32_t *right, const ogkr_fr *prev, ogkr_fr *out) {
#pragma omp parallel for schedule(static)
    for (size_t g = 0; g < gates; ++g) {
        if (gate_type[g])
            fr_mul(prev[left[g]].l, prev[right[g]].l, out[g].l);
        else
            fr_add(prev[left[g]].l, prev[right[g]].l, out[g].l);
    }
}

void ogkr_mobius(ogkr_fr *vals, int k) {
    size_t n = (size_t)1 << k;
    for (int b = 0; b < k; ++b) {
        size_t bit = (size_t)1 << (k - 1 - b);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i)
            if (i & bit) fr_sub(vals[i].l, vals[i ^ bit].l, vals[i].l);   /* (reads only entries without the bit) */
    }
}

int ogkr_line_restriction(int k, const ogkr_fr *b, const ogkr_fr *c, const ogkr_fr *W,
                          ogkr_fr *out, uint32_t *out_len) {
    if (k < 0 || k > 24) return -1;
    size_t n = (size_t)1 << k;
    ogkr_fr *co = malloc(n * 32);
    u64(*res)[4] = calloc((size_t)(k + 1), 32);
    u64(*grad)[4] = malloc((size_t)(k + 1) * 32), (*cst)[4] = malloc((size_t)(k + 1) * 32);
    if (!co || !res || !grad || !cst) return -2;
    memcpy(co, W, n * 32);
    ogkr_mobius(co, k);
    for (int j = 0; j < k; ++j) {
        u64 g[4];
        fr_sub(c[j].l, b[j].l, g);
        to_mont(g, grad[j]);
        to_mont(b[j].l, cst[j]);
    }
    /* res[d] = coefficient of t^d (ascending while accumulating); every monomial expanded along the line, as the
     * reference does (poly.rs:476-497); the monomials are independent, so they are dealt over the threads */
    int maxdeg = 0;
#pragma omp parallel
    {
        u64 poly[32][4], lres[32][4];
        int lmax = 0;
        memset(lres, 0, sizeof lres);
#pragma omp for schedule(dynamic, 1024) nowait
        for (size_t mono = 0; mono < n; ++mono) {
            const u64 *cf = co[mono].l;
            if (!(cf[0] | cf[1] | cf[2] | cf[3])) continue;
            int deg = 0;
            memcpy(poly[0], cf, 32);
            for (int j = 0; j < k; ++j) {
                if (!((mono >> (k - 1 - j)) & 1)) continue;
                /* poly *= (grad_j t + cst_j) */
                memset(poly[deg + 1], 0, 32);
                for (int d = deg + 1; d >= 1; --d) {
                    u64 x[4], y[4];
                    mont_mul(poly[d - 1], grad[j], x);
                    mont_mul(poly[d], cst[j], y);
                    fr_add(x, y, poly[d]);
                }
                u64 y0[4];
                mont_mul(poly[0], cst[j], y0);
                memcpy(poly[0], y0, 32);
                ++deg;
            }
            if (deg > lmax) lmax = deg;
            for (int d = 0; d <= deg; ++d) fr_add(lres[d], poly[d], lres[d]);
        }
#pragma omp critical(ogkr_line)
        {
            if (lmax > maxdeg) maxdeg = lmax;
            for (int d = 0; d <= k; ++d) fr_add(res[d], lres[d], res[d]);
        }
    }
    *out_len = (uint32_t)(maxdeg + 1);
    for (int d = 0; d <= k; ++d) memcpy(out[k - d].l, res[d], 32);   /* highest first, right-aligned */
    free(co);
    free(res);
    free(grad);
    free(cst);
    return 0;
}
