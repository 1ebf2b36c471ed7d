// Host-only neighbours of the boundary (include/gkr_amd.h, "the reference's own argument types" and gkr_verify):
//
//   * prover::prove(&GKRCircuit, &Input) (rust/src/gkr/prover.rs:6-9) receives, per layer, the wiring as lists of 0/1 wire
//     vectors `gate || left || right` (Layer.wire, gkr.rs:35-51, built at convert.rs:715-767) and every layer's values as a
//     multilinear polynomial in term-list form [coeff, e_1 .. e_k] (Input.w, gkr.rs:21-33, get_multi_ext poly.rs:502-536).
//     The library's ABI takes gate index arrays and evaluation tables: gkr_layer_from_wires and gkr_values_from_terms turn
//     the one into the other, gkr_terms_from_coeffs turns the proof's monomial-coefficient tables back into term lists
//     (Proof.d, Proof.input_func) -- a binding needs nothing else to sit exactly at `prove`.
//   * gkr_verify: the relations of the reference's verifier (python/gkr.py:202-231, python/sumcheck.py:55-70; the circom
//     verifier checks the same inside a circuit, verifier.circom:39-71) on a gkr_proof_buf, in C++ on the host's threads,
//     so that proofs of any size the prover takes can be checked without the CPU checker under tests/.
//
// Plain C++ (g++): no device code, no HIP.
#include <stdint.h>
#include <string.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/gkr_amd.h"
#include "fr64.h"
#include "host_cpus.h"

namespace {

using gkr::h64::F;
using gkr::h64::Wide;

inline F load(const gkr_fr& x) {
    F f;
    memcpy(f.l, x.l, 32);
    return f;
}
inline bool canonical(const gkr_fr& x) { return !gkr::h64::geq_mod(load(x)); }
inline bool is_one(const gkr_fr& x) { return x.l[0] == 1 && !(x.l[1] | x.l[2] | x.l[3]); }
inline bool is_zero(const gkr_fr& x) { return !(x.l[0] | x.l[1] | x.l[2] | x.l[3]); }
inline bool same(const F& a, const F& b) { return memcmp(a.l, b.l, 32) == 0; }

const F kOneCanonical = {{1, 0, 0, 0}};
inline F mont_one() { return gkr::h64::to_mont(kOneCanonical); }

// bits of one wire vector, most significant first -> index; false if an entry is neither 0 nor 1
inline bool decode_bits(const gkr_fr* v, int n, uint64_t* out) {
    uint64_t x = 0;
    for (int i = 0; i < n; ++i) {
        if (is_one(v[i]))
            x = (x << 1) | 1;
        else if (is_zero(v[i]))
            x <<= 1;
        else
            return false;
    }
    *out = x;
    return true;
}

// run fn(t, begin, end) over [0, n) on up to `threads` threads (the caller's included)
template <class Fn>
void parallel_for(size_t n, int threads, size_t min_chunk, Fn fn) {
    size_t parts = (size_t)(threads < 1 ? 1 : threads);
    if (parts > (n + min_chunk - 1) / min_chunk) parts = (n + min_chunk - 1) / min_chunk;
    if (parts <= 1) {
        fn(0, (size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = (n + parts - 1) / parts;
    for (size_t t = 1; t < parts; ++t) th.emplace_back([=] { fn((int)t, std::min(n, t * per), std::min(n, (t + 1) * per)); });
    fn(0, (size_t)0, std::min(n, per));
    for (auto& x : th) x.join();
}

int default_threads() {
    const int n = gkr::usable_cpus();
    return n > 64 ? 64 : n;
}

// eq(point, .) over 2^k indices, variable 1 = most significant index bit; point and table in Montgomery form.  The table of
// the first t variables is built by doublings on the calling thread; every thread then expands its own slice of those 2^t
// prefixes by the remaining k - t variables (entry j of a level becomes entries 2j and 2j + 1 of the next, so a prefix's
// descendants are a contiguous block of every later level).
// a table whose storage is kept between layers and never zero-filled (every entry is written before it is read)
struct Table {
    std::unique_ptr<F[]> p;
    size_t cap = 0;
    F* get(size_t n) {
        if (n > cap) {
            p.reset(new F[n]);
            cap = n;
        }
        return p.get();
    }
};

void eq_table(const std::vector<F>& point_m, Table& table, int threads) {
    const int k = (int)point_m.size();
    F* out = table.get((size_t)1 << k);
    out[0] = mont_one();
    auto expand = [&](size_t base, size_t have, int from, int to) {   // the block out[base .. base + have) of level `from` -> level `to`, in place at base << (to - from)
        for (int i = from; i < to; ++i) {
            const size_t b0 = base << (i - from), b1 = base << (i + 1 - from);
            for (size_t j = have; j-- > 0;) {
                const F v = out[b0 + j];
                const F hi = gkr::h64::mont_mul(v, point_m[i]);
                out[b1 + 2 * j] = gkr::h64::sub(v, hi);
                out[b1 + 2 * j + 1] = hi;
            }
            have <<= 1;
        }
    };
    int t = 0;
    while (t < k && ((size_t)1 << t) < (size_t)threads * 4) ++t;
    if (k < 14 || threads <= 1) t = k;
    expand(0, 1, 0, t);
    if (t == k) return;
    // level t sits in out[0 .. 2^t); prefix p's block of level k is out[p << (k - t) .. (p + 1) << (k - t)).  Highest prefix first
    // within a thread would overwrite nothing it still needs only if blocks never overlap their sources: copy level t aside.
    const std::vector<F> prefixes(out, out + ((size_t)1 << t));
    parallel_for((size_t)1 << t, threads, 1, [&](int, size_t a, size_t b) {
        for (size_t p = a; p < b; ++p) {
            const size_t base = p << (k - t);
            out[base] = prefixes[p];
            // (expand works on "block at base of the current level": re-based so that the prefix is entry 0 of its own sub-table)
            size_t have = 1;
            for (int i = t; i < k; ++i) {
                for (size_t j = have; j-- > 0;) {
                    const F v = out[base + j];
                    const F hi = gkr::h64::mont_mul(v, point_m[i]);
                    out[base + 2 * j] = gkr::h64::sub(v, hi);
                    out[base + 2 * j + 1] = hi;
                }
                have <<= 1;
            }
        }
    });
}

// Horner, highest degree first (poly.rs:260-267); coefficients canonical, x Montgomery -> canonical
F horner(const gkr_fr* c, int n, const F& x_m) {
    F acc = {{0, 0, 0, 0}};
    for (int i = 0; i < n; ++i) acc = gkr::h64::add(gkr::h64::mont_mul(acc, x_m), load(c[i]));
    return acc;
}

// sum_mask coeff[mask] prod_{i in mask} x_i for a table of 2^k monomial coefficients (mask bit k-1-j <-> variable j+1):
// the variables are bound one by one, last variable first (c[rest,0] + x c[rest,1]); canonical in, x Montgomery, canonical out
F eval_monomial_table(const gkr_fr* coeffs, int k, const std::vector<F>& x_m, int threads) {
    if (k == 0) return load(coeffs[0]);
    // every thread folds its own contiguous slice of the table down to one value per slice-prefix (the leading variables index
    // the slices), the calling thread folds those
    int t = 0;
    while (t < k - 10 && ((size_t)1 << t) < (size_t)threads * 2) ++t;
    if (threads <= 1) t = 0;
    std::vector<F> top((size_t)1 << t);
    parallel_for((size_t)1 << t, threads, 1, [&](int, size_t a, size_t b) {
        std::vector<F> cur;
        for (size_t p = a; p < b; ++p) {
            const int kk = k - t;                                 // the slice's own variables: t + 1 .. k
            const gkr_fr* c = coeffs + (p << kk);
            if (kk == 0) {
                top[p] = load(c[0]);
                continue;
            }
            cur.resize((size_t)1 << (kk - 1));
            const F& last = x_m[k - 1];
            for (size_t i = 0; i < cur.size(); ++i) cur[i] = gkr::h64::add(load(c[2 * i]), gkr::h64::mont_mul(load(c[2 * i + 1]), last));
            for (int v = kk - 2; v >= 0; --v) {
                const size_t half = (size_t)1 << v;
                const F& x = x_m[t + v];
                for (size_t i = 0; i < half; ++i) cur[i] = gkr::h64::add(cur[2 * i], gkr::h64::mont_mul(cur[2 * i + 1], x));
            }
            top[p] = cur[0];
        }
    });
    for (int v = t - 1; v >= 0; --v) {
        const size_t half = (size_t)1 << v;
        for (size_t i = 0; i < half; ++i) top[i] = gkr::h64::add(top[2 * i], gkr::h64::mont_mul(top[2 * i + 1], x_m[v]));
    }
    return top[0];
}

// every element of a table below r?  (on the threads: a 2^20-entry table is 32 MiB)
bool all_canonical_par(const gkr_fr* v, size_t n, int threads) {
    std::atomic<int> bad{0};
    parallel_for(n, threads, 1 << 15, [&](int, size_t a, size_t b) {
        for (size_t i = a; i < b; ++i)
            if (!canonical(v[i])) {
                bad.store(1);
                return;
            }
    });
    return !bad.load();
}

}  // namespace

extern "C" {

int gkr_layer_from_wires(int k_i, int k_next, const gkr_fr* add_wire, size_t n_add, const gkr_fr* mult_wire, size_t n_mult,
                         uint8_t* gate_type, uint32_t* left, uint32_t* right) {
    if (k_i < 0 || k_i > GKR_MAX_K_I || k_next < 0 || k_next > GKR_MAX_K_NEXT || !gate_type || !left || !right) return GKR_ERR_INVALID;
    if ((n_add && !add_wire) || (n_mult && !mult_wire)) return GKR_ERR_INVALID;
    const size_t gates = (size_t)1 << k_i;
    if (n_add + n_mult != gates) return GKR_ERR_INVALID;     // every slot of a layer is a gate (convert.rs:209-214, 307-342)
    const int width = k_i + 2 * k_next;
    std::vector<uint8_t> seen(gates, 0);
    for (int pass = 0; pass < 2; ++pass) {
        const gkr_fr* wires = pass ? mult_wire : add_wire;
        const size_t n = pass ? n_mult : n_add;
        for (size_t t = 0; t < n; ++t) {
            const gkr_fr* v = wires + t * (size_t)width;
            uint64_t g, l, r;
            if (!decode_bits(v, k_i, &g) || !decode_bits(v + k_i, k_next, &l) || !decode_bits(v + k_i + k_next, k_next, &r)) return GKR_ERR_INVALID;
            if (seen[g]) return GKR_ERR_INVALID;             // a gate named twice
            seen[g] = 1;
            gate_type[g] = (uint8_t)pass;
            left[g] = (uint32_t)l;
            right[g] = (uint32_t)r;
        }
    }
    return GKR_OK;
}

int gkr_values_from_terms(int k, const gkr_fr* terms, size_t n_terms, gkr_fr* out_values) {
    if (k < 0 || k > GKR_MAX_K_NEXT || !out_values || (n_terms && !terms)) return GKR_ERR_INVALID;
    const size_t n = (size_t)1 << k;
    std::vector<F> c(n, F{{0, 0, 0, 0}});
    for (size_t t = 0; t < n_terms; ++t) {
        const gkr_fr* row = terms + t * (size_t)(k + 1);
        if (!canonical(row[0])) return GKR_ERR_NON_CANONICAL;
        uint64_t mask;
        if (!decode_bits(row + 1, k, &mask)) return GKR_ERR_INVALID;    // an exponent above 1: not a multilinear extension
        c[mask] = gkr::h64::add(c[mask], load(row[0]));                 // equal monomials add up (add_poly, poly.rs:293-334)
    }
    // value(x) = sum of the coefficients of the monomials contained in x: one pass per variable
    for (size_t bit = 1; bit < n; bit <<= 1)
        for (size_t i = 0; i < n; ++i)
            if (i & bit) c[i] = gkr::h64::add(c[i], c[i ^ bit]);
    for (size_t i = 0; i < n; ++i) memcpy(out_values[i].l, c[i].l, 32);
    return GKR_OK;
}

int gkr_terms_from_coeffs(int k, const gkr_fr* coeffs, gkr_fr* out_terms, size_t capacity_terms, size_t* n_terms) {
    if (k < 0 || k > GKR_MAX_K_NEXT || !coeffs || !n_terms) return GKR_ERR_INVALID;
    const size_t n = (size_t)1 << k;
    size_t count = 0;
    for (size_t m = 0; m < n; ++m) {
        if (is_zero(coeffs[m])) continue;                   // get_multi_ext drops zero coefficients (poly.rs:522-525)
        if (out_terms && count < capacity_terms) {
            gkr_fr* row = out_terms + count * (size_t)(k + 1);
            row[0] = coeffs[m];
            for (int j = 0; j < k; ++j) row[1 + j] = gkr_fr{{(m >> (k - 1 - j)) & 1, 0, 0, 0}};
        }
        ++count;
    }
    *n_terms = count;
    return out_terms && count > capacity_terms ? GKR_ERR_NOMEM : GKR_OK;
}

int gkr_prove_wires(gkr_ctx* ctx, const gkr_wire_circuit* wc, const gkr_fr* input_terms, size_t n_input_terms, int require_zero_output,
                    gkr_proof_buf* out) {
    if (!ctx || !wc || !wc->k || !out || wc->depth < 1 || wc->depth > 4096 || !wc->n_add || !wc->n_mult || !wc->add_wire || !wc->mult_wire)
        return GKR_ERR_INVALID;
    const uint32_t L = wc->depth;
    for (uint32_t i = 0; i <= L; ++i)
        if (wc->k[i] > (i == 0 ? (uint32_t)GKR_MAX_K_I : (uint32_t)GKR_MAX_K_NEXT)) return GKR_ERR_INVALID;
    std::vector<std::vector<uint8_t>> types(L);
    std::vector<std::vector<uint32_t>> lefts(L), rights(L);
    std::vector<const uint8_t*> p_types(L);
    std::vector<const uint32_t*> p_lefts(L), p_rights(L);
    for (uint32_t i = 0; i < L; ++i) {
        const size_t gates = (size_t)1 << wc->k[i];
        types[i].resize(gates);
        lefts[i].resize(gates);
        rights[i].resize(gates);
        const int rc = gkr_layer_from_wires((int)wc->k[i], (int)wc->k[i + 1], wc->add_wire[i], wc->n_add[i], wc->mult_wire[i], wc->n_mult[i],
                                            types[i].data(), lefts[i].data(), rights[i].data());
        if (rc) return rc;
        p_types[i] = types[i].data();
        p_lefts[i] = lefts[i].data();
        p_rights[i] = rights[i].data();
    }
    std::vector<gkr_fr> values((size_t)1 << wc->k[L]);
    const int rv = gkr_values_from_terms((int)wc->k[L], input_terms, n_input_terms, values.data());
    if (rv) return rv;
    const gkr_circuit_desc desc = {L, wc->k, p_types.data(), p_lefts.data(), p_rights.data()};
    return gkr_prove(ctx, &desc, values.data(), require_zero_output, out);
}

int gkr_verify(const gkr_circuit_desc* circuit, const gkr_proof_buf* proof, int threads, int* accept, uint32_t* failed_layer,
               uint32_t* failed_check) {
    if (!circuit || !proof || !accept || !circuit->k || circuit->depth < 1 || circuit->depth > 4096) return GKR_ERR_INVALID;
    if (!proof->sumcheck_coeffs || !proof->sumcheck_len || !proof->sumcheck_r || !proof->q || !proof->q_len || !proof->z || !proof->r ||
        !proof->d_coeffs || !proof->input_coeffs || !circuit->gate_type || !circuit->left || !circuit->right)
        return GKR_ERR_INVALID;
    const uint32_t L = circuit->depth;
    for (uint32_t i = 0; i <= L; ++i)
        if (circuit->k[i] > (i == 0 ? (uint32_t)GKR_MAX_K_I : (uint32_t)GKR_MAX_K_NEXT)) return GKR_ERR_INVALID;
    if (threads <= 0) threads = default_threads();
    *accept = 0;
    uint32_t layer_out = 0, check_out = 0;
    auto reject = [&](uint32_t layer, uint32_t check) {
        layer_out = layer;
        check_out = check;
        if (failed_layer) *failed_layer = layer;
        if (failed_check) *failed_check = check;
        return GKR_OK;
    };
    const F zero = {{0, 0, 0, 0}};
    const F one_m = mont_one();
    // z[0] = 0 (prover.rs:16-21) and m_0 = D(z[0])
    const gkr_fr* z = proof->z;
    std::vector<F> zi_m(circuit->k[0]);
    for (uint32_t j = 0; j < circuit->k[0]; ++j) {
        if (!is_zero(z[j])) return reject(0, GKR_VERIFY_Z0);
        zi_m[j] = zero;
    }
    if (!all_canonical_par(proof->d_coeffs, (size_t)1 << circuit->k[0], threads)) return reject(0, GKR_VERIFY_NON_CANONICAL);
    F m = eval_monomial_table(proof->d_coeffs, (int)circuit->k[0], zi_m, threads);
    size_t row = 0, qo = 0, zo = circuit->k[0];
    Table tab_z, tab_b, tab_c;
    for (uint32_t i = 0; i < L; ++i) {
        const int k_i = (int)circuit->k[i], k = (int)circuit->k[i + 1];
        const size_t gates = (size_t)1 << k_i;
        if (!circuit->gate_type[i] || !circuit->left[i] || !circuit->right[i]) return GKR_ERR_INVALID;
        // the sumcheck's rounds (python/sumcheck.py:55-70)
        F expected = m;
        std::vector<F> rs_m(2 * (size_t)k);
        for (int j = 0; j < 2 * k; ++j, ++row) {
            const uint32_t len = proof->sumcheck_len[row];
            if (len < 1 || len > 3) return reject(i, GKR_VERIFY_SHAPE);
            const gkr_fr* g = proof->sumcheck_coeffs + row * 3 + (3 - len);
            for (uint32_t t = 0; t < len; ++t)
                if (!canonical(g[t])) return reject(i, GKR_VERIFY_NON_CANONICAL);
            if (!canonical(proof->sumcheck_r[row])) return reject(i, GKR_VERIFY_NON_CANONICAL);
            F at1 = zero;                                    // g(1) = sum of the coefficients, g(0) = the constant term
            for (uint32_t t = 0; t < len; ++t) at1 = gkr::h64::add(at1, load(g[t]));
            if (!same(gkr::h64::add(at1, load(g[len - 1])), expected)) return reject(i, GKR_VERIFY_ROUND_SUM);
            gkr_fr key = {{0, 0, 0, 0}}, h;
            if (gkr_mimc7_multi_hash(g, len, &key, &h) != GKR_OK) return GKR_ERR_INVALID;
            if (memcmp(h.l, proof->sumcheck_r[row].l, 32) != 0) return reject(i, GKR_VERIFY_CHALLENGE);
            rs_m[j] = gkr::h64::to_mont(load(proof->sumcheck_r[row]));
            expected = horner(g, (int)len, rs_m[j]);
        }
        // q(0), q(1), and the last claim against add(z,b*,c*) (q0 + q1) + mult(z,b*,c*) q0 q1 (python/gkr.py:213-219)
        const uint32_t qlen = proof->q_len[i];
        if (qlen < 1 || qlen > (uint32_t)k + 1) return reject(i, GKR_VERIFY_SHAPE);
        const gkr_fr* q = proof->q + qo + ((size_t)k + 1 - qlen);
        for (uint32_t t = 0; t < qlen; ++t)
            if (!canonical(q[t])) return reject(i, GKR_VERIFY_NON_CANONICAL);
        const F q0 = load(q[qlen - 1]);
        F q1 = zero;
        for (uint32_t t = 0; t < qlen; ++t) q1 = gkr::h64::add(q1, load(q[t]));
        std::vector<F> b_m(rs_m.begin(), rs_m.begin() + k), c_m(rs_m.begin() + k, rs_m.end());
        eq_table(zi_m, tab_z, threads);
        eq_table(b_m, tab_b, threads);
        eq_table(c_m, tab_c, threads);
        const F *eq_z = tab_z.p.get(), *eq_b = tab_b.p.get(), *eq_c = tab_c.p.get();
        const uint8_t* gt = circuit->gate_type[i];
        const uint32_t *lf = circuit->left[i], *rt = circuit->right[i];
        const uint32_t limit = (uint32_t)1 << k;
        std::vector<F> part_add((size_t)threads, zero), part_mult((size_t)threads, zero);
        std::atomic<int> bad_gate{0};
        parallel_for(gates, threads, 2048, [&](int t, size_t a, size_t b) {
            F sa = zero, sm = zero;
            for (size_t base = a; base < b; base += 4096) {          // one reduction per 4096 products
                Wide wa = gkr::h64::wide_zero(), wm = gkr::h64::wide_zero();
                const size_t end = std::min(b, base + 4096);
                for (size_t g = base; g < end; ++g) {
                    if (lf[g] >= limit || rt[g] >= limit || gt[g] > 1) {
                        bad_gate.store(1);
                        return;
                    }
                    if (g + 12 < end && lf[g + 12] < limit && rt[g + 12] < limit) {   // (the operands' eq entries are random reads of 32 MiB tables)
                        __builtin_prefetch(&eq_b[lf[g + 12]]);
                        __builtin_prefetch(&eq_c[rt[g + 12]]);
                    }
                    const F bc = gkr::h64::mont_mul(eq_b[lf[g]], eq_c[rt[g]]);
                    gkr::h64::wide_mac(gt[g] ? wm : wa, eq_z[g], bc);
                }
                sa = gkr::h64::add(sa, gkr::h64::wide_reduce(wa));
                sm = gkr::h64::add(sm, gkr::h64::wide_reduce(wm));
            }
            part_add[t] = sa;
            part_mult[t] = sm;
        });
        if (bad_gate.load()) return GKR_ERR_INVALID;
        F add_m = zero, mult_m = zero;                        // Montgomery forms of add_i, mult_i at (z, b*, c*)
        for (int t = 0; t < threads; ++t) {
            add_m = gkr::h64::add(add_m, part_add[t]);
            mult_m = gkr::h64::add(mult_m, part_mult[t]);
        }
        const F q01 = gkr::h64::mont_mul(gkr::h64::to_mont(q0), q1);             // canonical q0 q1
        const F want = gkr::h64::add(gkr::h64::mont_mul(add_m, gkr::h64::add(q0, q1)), gkr::h64::mont_mul(mult_m, q01));
        if (!same(want, expected)) return reject(i, GKR_VERIFY_FINAL_CLAIM);
        // r* = hash of the last round vector (prover.rs:74-78), z[i+1] = l(r*) (poly.rs:538-551), m = q(r*)
        {
            const size_t last = row - 1;
            const uint32_t len = proof->sumcheck_len[last];
            gkr_fr key = {{0, 0, 0, 0}}, h;
            if (gkr_mimc7_multi_hash(proof->sumcheck_coeffs + last * 3 + (3 - len), len, &key, &h) != GKR_OK) return GKR_ERR_INVALID;
            if (memcmp(h.l, proof->r[i].l, 32) != 0) return reject(i, GKR_VERIFY_R_STAR);
        }
        const F rstar_m = gkr::h64::to_mont(load(proof->r[i]));
        zi_m.assign((size_t)k, zero);
        for (int j = 0; j < k; ++j) {
            const F bj = load(proof->sumcheck_r[row - 2 * (size_t)k + j]), cj = load(proof->sumcheck_r[row - (size_t)k + j]);
            const F zj = gkr::h64::add(bj, gkr::h64::mont_mul(rstar_m, gkr::h64::sub(cj, bj)));
            if (!canonical(z[zo + j]) || !same(zj, load(z[zo + j]))) return reject(i, GKR_VERIFY_NEXT_Z);
            zi_m[j] = gkr::h64::to_mont(zj);
        }
        m = horner(q, (int)qlen, rstar_m);
        qo += (size_t)k + 1;
        zo += (size_t)k;
    }
    if (!all_canonical_par(proof->input_coeffs, (size_t)1 << circuit->k[L], threads)) return reject(L, GKR_VERIFY_NON_CANONICAL);
    if (!same(m, eval_monomial_table(proof->input_coeffs, (int)circuit->k[L], zi_m, threads))) return reject(L, GKR_VERIFY_INPUT);
    (void)one_m;
    (void)layer_out;
    (void)check_out;
    *accept = 1;
    if (failed_layer) *failed_layer = 0;
    if (failed_check) *failed_check = GKR_VERIFY_OK;
    return GKR_OK;
}

}  // extern "C"
