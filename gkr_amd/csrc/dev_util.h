// Device helpers shared by the kernel translation units (kernels.hip, kernels_wide.hip): wave64 / block reductions of
// wide accumulators, 32-byte element loads and stores, grid sizing.  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fr32.h"

namespace gkr {

// ---------------------------------------------------------------------------
// wave64 / block reductions of wide accumulators
// ---------------------------------------------------------------------------

template <int NL>
__device__ __forceinline__ Acc<NL> wave_sum(Acc<NL> a) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Acc<NL> o;
#pragma unroll
        for (int i = 0; i < NL; ++i) o.l[i] = __shfl_down(a.l[i], off, 64);
        acc_add_acc(a, o);
    }
    return a;  // lane 0 holds the wave total
}

// Sum NA accumulators over the block; thread 0 gets the totals.  smem must hold
// (blockDim.x / 64) * NA accumulators.
template <int NL, int NA>
__device__ __forceinline__ void block_sum(Acc<NL> (&acc)[NA], Acc<NL>* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        acc[k] = wave_sum(acc[k]);
        if (lane == 0) smem[wave * NA + k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nwaves; ++w)
#pragma unroll
            for (int k = 0; k < NA; ++k) acc_add_acc(acc[k], smem[w * NA + k]);
    }
}

__device__ __forceinline__ Fr load_fr(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    Fr f;
    f.l[0] = a.x; f.l[1] = a.y; f.l[2] = a.z; f.l[3] = a.w;
    f.l[4] = b.x; f.l[5] = b.y; f.l[6] = b.z; f.l[7] = b.w;
    return f;
}

__device__ __forceinline__ void store_fr(Fr* p, const Fr& f) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(f.l[0], f.l[1], f.l[2], f.l[3]);
    q[1] = make_uint4(f.l[4], f.l[5], f.l[6], f.l[7]);
}

// device-side build of the round's fixed-multiplier table (device transcript only):
// R_i = r * 2^(32 i) * 2^64 mod p, canonical
__device__ __forceinline__ void store_fixed_mul(FixedMul* out, const Fr& r_canonical) {
    Fr two32 = fr_zero(), two64 = fr_zero();
    two32.l[1] = 1;
    two64.l[2] = 1;
    Fr cur = mont_mul(to_mont(r_canonical), two64);
    const Fr two32_m = to_mont(two32);
    for (int i = 0; i < 8; ++i) {
        for (int c = 0; c < 8; ++c) out->w[i][c] = cur.l[c];
        cur = mont_mul(cur, two32_m);
    }
}

static inline uint32_t blocks_for(uint64_t items, uint32_t cap) {
    uint64_t b = (items + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (uint32_t)b;
}

}  // namespace gkr
