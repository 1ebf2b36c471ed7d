/* Plain-C dense oracle for the GKR sumcheck hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Never linked into, loaded by or called from the product library
 * (gkr_amd/csrc).  Users: tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py.
 *
 * Restates, on dense evaluation tables, the reference's
 *   prove_sumcheck_opt   rust/src/gkr/sumcheck.rs:36-156
 *   prove_sumcheck       rust/src/gkr/sumcheck.rs:158-214
 *   reduce_multiple_polynomial / l_function   rust/src/gkr/poly.rs:469-500,538-551
 *   calculate_input (forward step)            rust/src/convert.rs:812-831
 *   MiMC7 multi_hash (third-party mimc-rs; call sites sumcheck.rs:45,84)
 * It is the C twin of oracle/dense.py, which tests/test_oracle_equivalence.py
 * shows equal to the term-list restatement oracle/termlist.py and
 * tests/test_oracle_golden.py pins to fixtures made by the reference's own
 * Python prover.  Pinning status of each piece: oracle/__init__.py.
 *
 * All field elements cross this API as 4 little-endian 64-bit limbs of the
 * canonical value (32-byte LE repr, sumcheck.rs:10-22).
 */
#ifndef OGKR_H
#define OGKR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } ogkr_fr;

void ogkr_fr_add(const ogkr_fr *a, const ogkr_fr *b, ogkr_fr *out);
void ogkr_fr_sub(const ogkr_fr *a, const ogkr_fr *b, ogkr_fr *out);
void ogkr_fr_mul(const ogkr_fr *a, const ogkr_fr *b, ogkr_fr *out);
int  ogkr_fr_is_canonical(const ogkr_fr *a);

void ogkr_keccak256(const uint8_t *data, size_t len, uint8_t out[32]);
void ogkr_mimc7_constant(int i, ogkr_fr *out);                 /* i in 0..90 */
void ogkr_mimc7_hash(const ogkr_fr *x, const ogkr_fr *k, ogkr_fr *out);
void ogkr_multi_hash(const ogkr_fr *arr, size_t n, const ogkr_fr *key, ogkr_fr *out);

/* Synthetic workload generator shared (by definition, not by code) with the
 * product: element i of stream `seed` is four splitmix64 outputs of the state
 * seed + 4 i + {1,2,3,4} golden-ratio steps, top limb masked to 61 bits, so the
 * value is < 2^253 < r and needs no rejection. */
void ogkr_fill_table(ogkr_fr *table, size_t count, uint64_t seed);

/* prove_sumcheck on the 2^n evaluations of a multilinear g.  out_coeffs holds
 * n rows of 2 slots, right-aligned (slot 1 = constant term); out_len[j] is the
 * reference's vector length for round j.  threads <= 0: all cores. */
int ogkr_sumcheck_mle(const ogkr_fr *table, int n, ogkr_fr *out_coeffs, uint32_t *out_len,
                      ogkr_fr *out_r, int threads);

/* the same, overwriting the caller's table (no copy: for tables of tens of GiB) */
int ogkr_sumcheck_mle_inplace(ogkr_fr *table, int n, ogkr_fr *out_coeffs, uint32_t *out_len,
                              ogkr_fr *out_r, int threads);

/* prove_sumcheck_opt for one layer: gates g = 0..2^k_i-1 of type gate_type[g]
 * (0 add, 1 mult) with operands left[g], right[g] in [0, 2^k_next); z has k_i
 * entries; W has 2^k_next evaluations.  out_coeffs: 2*k_next rows of 3 slots,
 * right-aligned, highest degree first. */
int ogkr_sumcheck_layer(int k_i, int k_next, const uint8_t *gate_type, const uint32_t *left,
                        const uint32_t *right, const ogkr_fr *z, const ogkr_fr *W,
                        ogkr_fr *out_coeffs, uint32_t *out_len, ogkr_fr *out_r, int threads);

/* The same transcript in time linear in the gates (no 2^{2 k_next}-entry tables; k_next up to 28): U, V and the
 * c-phase row summed over the gate list -- the C twin of oracle/gatesum.py, the checker for wide layers. */
int ogkr_sumcheck_layer_lin(int k_i, int k_next, const uint8_t *gate_type, const uint32_t *left,
                            const uint32_t *right, const ogkr_fr *z, const ogkr_fr *W,
                            ogkr_fr *out_coeffs, uint32_t *out_len, ogkr_fr *out_r, int threads);

/* eq(z, .) weights scattered into the dense predicate tables A, M (2^{2 k_next} each). */
int ogkr_predicate_tables(int k_i, int k_next, const uint8_t *gate_type, const uint32_t *left,
                          const uint32_t *right, const ogkr_fr *z, ogkr_fr *A, ogkr_fr *M);

/* forward gate evaluation of one layer */
void ogkr_layer_eval(size_t gates, const uint8_t *gate_type, const uint32_t *left,
                     const uint32_t *right, const ogkr_fr *prev, ogkr_fr *out);

/* q(t) = W(b + t (c - b)); out has k+1 slots right-aligned, *out_len = 1 + max
 * total degree of a non-zero monomial of W. */
int ogkr_line_restriction(int k, const ogkr_fr *b, const ogkr_fr *c, const ogkr_fr *W,
                          ogkr_fr *out, uint32_t *out_len);

/* evaluation table -> monomial coefficients (MSB-first Moebius transform), in place */
void ogkr_mobius(ogkr_fr *vals, int k);

/* default thread count of every parallel loop below (OpenMP's own default is one per visible CPU) */
void ogkr_set_threads(int threads);
int ogkr_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
