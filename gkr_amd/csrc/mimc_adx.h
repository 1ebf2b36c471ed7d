// One transcript's MiMC7 on the scalar integer pipes with mulx / adcx / adox (mimc_adx.cpp); used by the host
// transcript for a single hash (batch 1, the tail of a chunk) when the CPU has BMI2 and ADX.
#pragma once
#include <stdint.h>

namespace gkr {
bool gkr_adx_available();
// Mimc7::new(91).multi_hash(arr, &Fr::from(0))  (rust/src/gkr/sumcheck.rs:84,129,152, prover.rs:78):
// arr: n canonical elements, 4 x 64-bit little-endian limbs; cts_mont: the 91 round constants in Montgomery form
// (R = 2^256, the form the context keeps); out: canonical
void gkr_adx_multi_hash(const uint64_t (*arr)[4], int n, const uint64_t (*cts_mont)[4], uint64_t* out);
}  // namespace gkr
