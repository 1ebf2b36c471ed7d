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
// the host's whole share of one multi-round pass, `count` <= 16 sumchecks side by side (see mimc_ifma.cpp)
void gkr_ifma_pass(const uint64_t* sums, size_t sums_row_words, int count, int J, const uint32_t* final_len, uint64_t (*c0)[16][4],
                   uint64_t (*c1)[16][4], uint64_t (*r)[16][4], uint32_t (*len)[16], uint64_t* weights, size_t w_row_words);
// the host's share of one product pass of the layer sumcheck, `count` <= 16 sumchecks side by side (see mimc_ifma.cpp)
void gkr_ifma_prod_pass(const uint64_t* recs, size_t rec_row_words, int count, int J, const uint32_t (*vec_len)[16], uint64_t (*c2)[16][4],
                        uint64_t (*lin)[16][4], uint64_t (*c0)[16][4], uint64_t (*r)[16][4], uint64_t* weights, size_t w_row_words);
// the host tail of a phase's product passes on eight lanes (see mimc_ifma.cpp; capi_layer.hip, host_tail_pass, is its scalar twin)
void gkr_ifma_tail_pass(uint64_t* tables, size_t stride, uint32_t m, uint32_t jp, const uint64_t* weights, uint32_t J, uint64_t* rec);
}  // namespace gkr
