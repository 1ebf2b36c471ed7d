// Eight-lane AVX-512 IFMA MiMC7 (mimc_ifma.cpp); used by the host transcript when the CPU has it.
#pragma once
#include <stdint.h>

namespace gkr {
bool gkr_ifma_available();
// cts_canonical: the 91 MiMC7 round constants, canonical, 4 x 64-bit little-endian limbs each
void gkr_ifma_init(const uint64_t (*cts_canonical)[4]);
// vec[k][s]: slot s of lane k (right-aligned round vector, `slots` <= 3 slots); len[k] trailing slots are hashed
void gkr_ifma_multi_hash8(const uint64_t (*vec)[3][4], const uint32_t* len, int slots, uint64_t (*out)[4]);
void gkr_ifma_multi_hash16(const uint64_t (*vec)[3][4], const uint32_t* len, int slots, uint64_t (*out)[4]);
}  // namespace gkr
