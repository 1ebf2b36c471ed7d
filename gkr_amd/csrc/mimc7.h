// MiMC7 (91 rounds) Fiat-Shamir hash on fr32, host and device.
//
// Replaces the third-party mimc-rs calls of the reference:
//   Mimc7::new(91)                       rust/src/gkr/sumcheck.rs:45, prover.rs:10
//   multi_hash(vec, &Fr::from(0))        rust/src/gkr/sumcheck.rs:84,129,152, prover.rs:78
// Algorithm (circomlib mimc7 / upstream mimc-rs):
//   c_0 = 0, c_i = keccak256^{i+1}("mimc") mod r
//   hash(x, k): t = x + k; 91 times { h = t^7; t = h + k + c_{i+1} }; return h + k
//   multi_hash(arr, key): r = key; for a in arr: r = r + a + hash(a, r)
// The 91 constants are produced on the host (keccak.h) once per context and
// handed to the kernels in Montgomery form.
#pragma once
#include "fr32.h"

namespace gkr {

constexpr int kMimcRounds = 91;

// x, k Montgomery; cts = 91 Montgomery constants; result Montgomery
GKR_HD Fr mimc7_hash_mont(const Fr& x, const Fr& k, const Fr* cts) {
    Fr h = fr_zero();
    for (int i = 0; i < kMimcRounds; ++i) {
        Fr t = (i == 0) ? fr_add(x, k) : fr_add(fr_add(h, k), cts[i]);
        Fr t2 = mont_mul(t, t);
        Fr t4 = mont_mul(t2, t2);
        Fr t6 = mont_mul(t4, t2);
        h = mont_mul(t6, t);
    }
    return fr_add(h, k);
}

// arr canonical, n elements; key = 0 as everywhere in the reference; result canonical
GKR_HD Fr mimc7_multi_hash(const Fr* arr, int n, const Fr* cts) {
    Fr r = fr_zero();
    for (int i = 0; i < n; ++i) {
        Fr a = to_mont(arr[i]);
        Fr h = mimc7_hash_mont(a, r, cts);
        r = fr_add(fr_add(r, a), h);
    }
    return from_mont(r);
}

}  // namespace gkr
