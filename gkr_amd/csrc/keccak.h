// Host-side keccak-256 (original Keccak padding) used once per context to derive
// the MiMC7 round constants c_i = keccak256^{i+1}("mimc") mod r, the schedule of
// circomlib / mimc-rs that the reference's Mimc7::new(91) computes
// (rust/src/gkr/sumcheck.rs:45).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "fr32.h"

namespace gkr {

void keccak256(const uint8_t* data, size_t len, uint8_t out[32]);

// fills cts[0..90] with the constants in Montgomery form
void mimc7_make_constants(Fr* cts_mont);

}  // namespace gkr
