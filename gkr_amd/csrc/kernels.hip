// CDNA4 (gfx950) kernels of the GKR sumcheck hot path.
//
// All kernels work on dense evaluation tables whose index is the bit string of
// the variables with variable 1 as the MOST significant bit (the reference's
// convention, rust/src/gkr/poly.rs:507,117-131).  Binding the leading variable
// therefore pairs entry i with entry i + h (h = half the table): two contiguous,
// perfectly coalesced streams.  One lane owns one 32-byte element
// (2 x global_load_dwordx4).
//
// 254-bit modular integer arithmetic on v_mad_u64_u32 carry chains (fr32.h); the one exception is the multi-round
// fold pass, whose fixed-weight products run as int8 digit products on the matrix cores (mfma_fold.h) because the
// VALU form was instruction-issue bound.  The streaming passes are priced against HBM bandwidth; the gate passes of the
// layer sumcheck are bound by the 254-bit arithmetic, the small kernels of a proof's rounds by latency.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <atomic>

#include "kernels.h"
#include "options.h"
#include "dev_util.h"
#include "gate_seg.h"
#include "mfma_fold.h"
#include "mfma_cross.h"
#include "mimc7.h"

namespace gkr {

// ---------------------------------------------------------------------------
// synthetic workload (bench / parity at full size)
// ---------------------------------------------------------------------------

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ void k_fill_table(Fr* table, size_t count, uint64_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        Fr f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint64_t w = mix64(seed + (4 * (uint64_t)i + (uint64_t)j + 1) * 0x9E3779B97F4A7C15ULL);
            if (j == 3) w &= 0x1FFFFFFFFFFFFFFFULL;
            f.l[2 * j] = (uint32_t)w;
            f.l[2 * j + 1] = (uint32_t)(w >> 32);
        }
        store_fr(table + i, f);
    }
}

// the same stream of values, the entries one rank of gkr_sumcheck_mle_sharded_dev holds: local entry (h, x) of shard p of
// 2^lp is entry h * 2P + 2p + x of the table
__global__ void k_fill_shard(Fr* shard_table, size_t count, uint32_t lp, uint32_t shard, uint64_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t g = ((uint64_t)(i >> 1) << (lp + 1u)) | ((uint64_t)shard << 1) | (i & 1u);
        Fr f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint64_t w = mix64(seed + (4 * g + (uint64_t)j + 1) * 0x9E3779B97F4A7C15ULL);
            if (j == 3) w &= 0x1FFFFFFFFFFFFFFFULL;
            f.l[2 * j] = (uint32_t)w;
            f.l[2 * j + 1] = (uint32_t)(w >> 32);
        }
        store_fr(shard_table + i, f);
    }
}

// ---------------------------------------------------------------------------
// plain multilinear sumcheck (reference: prove_sumcheck, sumcheck.rs:158-214)
// ---------------------------------------------------------------------------

// Round 1: S_lo = sum T[0..h), S_hi = sum T[h..2h) per table, plus "does the
// table depend on its last variable" (needed by the last round's length rule).
// grid = (blocks_per_table, batch)
__global__ void __launch_bounds__(256) k_mle_sum_first(const Fr* __restrict__ tables, size_t table_stride,
                                                       uint32_t h, MlePartial* __restrict__ partials) {
    __shared__ Acc<9> smem[4 * 2];
    __shared__ uint32_t s_dep;
    const Fr* t = tables + (size_t)blockIdx.y * table_stride;
    Acc<9> acc[2] = {acc_zero<9>(), acc_zero<9>()};
    uint32_t dep = 0;
    if (threadIdx.x == 0) s_dep = 0;
    __syncthreads();
    // Blocked distribution: block b owns one contiguous chunk of each stream, and blocks are dispatched
    // in order, so the chip sweeps every stream as a compact moving window (measured with
    // tools/ubench_copy.hip: +5..10 % over a grid-stride loop on this access pattern).
    // h is a power of two >= 2; chunk starts are multiples of 256, so lanes i and i^1 stay neighbours.
    const uint32_t chunk = ((h + gridDim.x - 1) / gridDim.x + 255u) & ~255u;
    const uint32_t begin = blockIdx.x * chunk;
    const uint32_t end = begin + chunk < h ? begin + chunk : h;
    for (uint32_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
        Fr lo = load_fr(t + i), hi = load_fr(t + i + h);
        acc_add_fr(acc[0], lo);
        acc_add_fr(acc[1], hi);
        // element 2m vs 2m+1: the neighbour lane holds the partner (i and i^1 are in the same
        // wave and both active); DPP quad_perm [1,0,3,2] swaps neighbours in the VALU, no LDS trip
        uint32_t dl = 0, dh = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            dl |= lo.l[k] ^ (uint32_t)__builtin_amdgcn_mov_dpp((int)lo.l[k], 0xB1, 0xF, 0xF, true);
            dh |= hi.l[k] ^ (uint32_t)__builtin_amdgcn_mov_dpp((int)hi.l[k], 0xB1, 0xF, 0xF, true);
        }
        dep |= dl | dh;
    }
    if (dep) atomicOr(&s_dep, 1u);
    block_sum<9, 2>(acc, smem);
    if (threadIdx.x == 0) {
        MlePartial* p = partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        p->lo = acc[0];
        p->hi = acc[1];
        p->dep = s_dep;   // valid: block_sum's __syncthreads ordered the atomicOr before this read
    }
}

// Rounds 2..n: fold T_{j-1} with r_{j-1} into T_j and accumulate the two half
// sums of T_j in the same pass.  src has 4q elements, dst 2q.  In-place
// (dst == src) is safe: a thread writes only slots it alone has read.
// grid = (blocks_per_table, batch)
__global__ void __launch_bounds__(256) k_mle_fold_sum(const Fr* __restrict__ src, size_t src_stride,
                                                      Fr* __restrict__ dst, size_t dst_stride, uint32_t q,
                                                      const FixedMul* __restrict__ rtab, uint32_t r_stride,
                                                      MlePartial* __restrict__ partials) {
    __shared__ Acc<9> smem[4 * 2];
    const Fr* s = src + (size_t)blockIdx.y * src_stride;
    Fr* d = dst + (size_t)blockIdx.y * dst_stride;
    const FixedMul T = rtab[(size_t)blockIdx.y * r_stride];   // wave-uniform -> scalar loads, lives in SGPRs
    Acc<9> acc[2] = {acc_zero<9>(), acc_zero<9>()};
    const uint32_t chunk = ((q + gridDim.x - 1) / gridDim.x + 255u) & ~255u;   // blocked distribution, see k_mle_sum_first
    const uint32_t begin = blockIdx.x * chunk;
    const uint32_t end = begin + chunk < q ? begin + chunk : q;
    for (uint32_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
        Fr x0 = load_fr(s + i), x1 = load_fr(s + i + 2 * (size_t)q);
        Fr x2 = load_fr(s + i + q), x3 = load_fr(s + i + 3 * (size_t)q);
        Fr y0, y1;
        fr_fold_fixed2(x0, x1, x2, x3, T, y0, y1);
        store_fr(d + i, y0);
        store_fr(d + i + q, y1);
        acc_add_fr(acc[0], y0);
        acc_add_fr(acc[1], y1);
    }
    block_sum<9, 2>(acc, smem);
    if (threadIdx.x == 0) {
        MlePartial* p = partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        p->lo = acc[0];
        p->hi = acc[1];
        p->dep = 0;
    }
}

// One wave per table: add up the per-block partials, apply the reference's
// length rule, hash the round vector with MiMC7, publish r (canonical + Montgomery).
//   rounds 1..n-1: [c1, c0] unless c1 == 0 (add_poly drops the zero term,
//                  poly.rs:324-327) -> [c0]
//   round n:       [c1, c0] iff the table depends on x_n (no merge, sumcheck.rs:206-207)
// grid = (batch), block = 64
__global__ void __launch_bounds__(64) k_mle_round_hash(const MlePartial* __restrict__ partials, uint32_t nblk,
                                                       uint32_t round, uint32_t n, const Fr* __restrict__ cts,
                                                       Fr* __restrict__ out_coeffs, uint32_t* __restrict__ out_len,
                                                       Fr* __restrict__ out_r, FixedMul* __restrict__ rtab,
                                                       uint32_t* __restrict__ dep_last) {
    const uint32_t b = blockIdx.x;
    const MlePartial* p = partials + (size_t)b * nblk;
    Acc<10> lo = acc_zero<10>(), hi = acc_zero<10>();
    uint32_t dep = 0;
    for (uint32_t i = threadIdx.x; i < nblk; i += 64) {
        acc_add_acc(lo, p[i].lo);
        acc_add_acc(hi, p[i].hi);
        dep |= p[i].dep;
    }
    lo = wave_sum(lo);
    hi = wave_sum(hi);
    dep = __any(dep) ? 1u : 0u;
    if (threadIdx.x == 0) {
        if (round == 0) dep_last[b] = dep;
        Fr c0 = acc_reduce(lo);
        Fr c1 = fr_sub(acc_reduce(hi), c0);
        uint32_t len;
        if (round + 1 < n)
            len = fr_is_zero(c1) ? 1u : 2u;
        else
            len = dep_last[b] ? 2u : 1u;
        Fr vec[2];
        vec[0] = (len == 2) ? c1 : c0;
        vec[1] = c0;
        Fr r = mimc7_multi_hash(vec, (int)len, cts);
        Fr* oc = out_coeffs + ((size_t)b * n + round) * 2;
        oc[0] = (len == 2) ? c1 : fr_zero();
        oc[1] = c0;
        out_len[(size_t)b * n + round] = len;
        out_r[(size_t)b * n + round] = r;
        store_fixed_mul(rtab + (size_t)b * n + round, r);
    }
}

// ---------------------------------------------------------------------------
// Multi-round passes (host transcript, default).  The sums a round needs are linear in the
// table, so the sums of J consecutive rounds follow from the 2^J sub-block sums of the current
// table (sub-block = the entries sharing their J leading index bits): the host derives
// r_j .. r_{j+J-1} from them without touching the table, and ONE pass then binds all J variables,
//     T'[i] = sum_b  w_b * T[b * S + i],     w_b = eq((r_j..r_{j+J-1}), b),   S = |T| / 2^J,
// while accumulating the 2^J' sub-block sums of T' for the next J' rounds.  With J = 5 (matrix-core
// fold, mfma_fold.h; J <= 3 for the v_mad_u64_u32 fold below) a 2^n sumcheck reads T_1 twice and then
// only a tail 32x smaller (~2.07 N elements moved instead of 4 N) and needs ~n/5 host round trips
// instead of n.  Results are the same field elements: bit-exact.
// ---------------------------------------------------------------------------

// sum_b w_b * src[b * S + i] for b < 2^JIN: unreduced products in independent accumulators that
// advance together (four at a time where there are that many), one reduction
template <int JIN>
__device__ __forceinline__ Fr multifold_entry(const Fr* __restrict__ s, uint32_t S, uint32_t i, const Fr* __restrict__ w) {
    if constexpr (JIN == 1) {
        Lazy17 a0 = lazy_zero(), a1 = lazy_zero();
        lazy_mac2_s(a0, load_fr(s + i), w[0], a1, load_fr(s + (size_t)S + i), w[1]);
        lazy_add(a0, a1);
        return lazy_reduce_k8(a0);
    } else if constexpr (JIN <= 3) {
        Fr x[1 << JIN];
#pragma unroll
        for (int b = 0; b < (1 << JIN); ++b) x[b] = load_fr(s + (size_t)b * S + i);
        Lazy17 t;
        weighted_sum_s<(1 << JIN)>(x, w, t);
        return lazy_reduce_k8(t);
    } else {
        // 16 or 32 inputs (small tables only): eight at a time, reduced sums added
        Fr y = fr_zero();
        for (int q = 0; q < (1 << (JIN - 3)); ++q) {
            Fr x[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) x[b] = load_fr(s + (size_t)(q * 8 + b) * S + i);
            Lazy17 t;
            weighted_sum_s<8>(x, w + q * 8, t);
            y = fr_add(y, lazy_reduce_k8(t));
        }
        return y;
    }
}

// The nblk partials of ONE sumcheck (2^jout sub-blocks of bps = nblk / 2^jout consecutive partials) -> the canonical
// sub-block sums in the host record.  Every partial is loaded by its own thread at once (nblk > blockDim: r consecutive
// ones per thread) and the sub-blocks are totalled by shuffles over min(bps / r, 64) lanes -- one round of loads and
// log2 steps on a lone sumcheck's round path, where a wave walking its sub-blocks one after the other paid a load
// latency per sub-block (11 us for 32 sub-blocks).  Called by all threads of a block of <= 1024.
__device__ __forceinline__ void mle_total_and_publish(const MleSubPartial* __restrict__ p, uint32_t nblk, uint32_t jout,
                                                      MleHostRecSub* __restrict__ r, uint32_t ticket) {
    __shared__ uint32_t s_pdep;
    __shared__ Acc<10> s_tot[kMleMaxSub], s_wave[16];
    const uint32_t T = blockDim.x, t = threadIdx.x, lane = t & 63u;
    const uint32_t nsub = 1u << jout, per = nblk > T ? nblk / T : 1u, eff = nblk / per, bps = (nblk >> jout) / per;
    if (t == 0) s_pdep = 0;
    __syncthreads();
    Acc<10> tot = acc_zero<10>();
    uint32_t dep = 0;
    if (t < eff) {
        for (uint32_t i = 0; i < per; ++i) {
            acc_add_acc(tot, p[(size_t)t * per + i].sum);
            dep |= p[(size_t)t * per + i].dep;
        }
    }
    if (__any(dep) && lane == 0) atomicOr(&s_pdep, 1u);
    const uint32_t seg = bps < 64u ? bps : 64u;
    for (uint32_t off = seg >> 1; off >= 1u; off >>= 1) {
        Acc<10> o;
#pragma unroll
        for (int l = 0; l < 10; ++l) o.l[l] = __shfl_down(tot.l[l], off, 64);
        acc_add_acc(tot, o);
    }
    if (bps <= 64u) {
        if (t < eff && (t & (seg - 1u)) == 0) s_tot[t / seg] = tot;
        __syncthreads();
    } else {
        // a sub-block spans bps / 64 waves (<= 16 wave totals in all)
        if (t < eff && lane == 0) s_wave[t >> 6] = tot;
        __syncthreads();
        if (t < nsub) {
            Acc<10> a = acc_zero<10>();
            for (uint32_t w = 0; w < (bps >> 6); ++w) acc_add_acc(a, s_wave[t * (bps >> 6) + w]);
            s_tot[t] = a;
        }
    }
    if (t < nsub) r->sums[t] = acc_reduce(s_tot[t]);
    __syncthreads();   // every record store is issued and waited for before the release below
    if (t == 0) {
        r->dep = s_pdep;
        __hip_atomic_store(&r->seq, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// A latency-bound pass publishes from its LAST block (pub.arrivals != nullptr): every block leaves its partial, the
// last one to arrive at the sumcheck's counter totals them -- k_mle_sub_reduce's arithmetic on this block's waves -- and
// writes the host record: no second launch on the round path of a lone sumcheck (k_mle_sub_reduce there: 6 - 11 us
// plus the ~6 us to a dependent launch, on three passes of a 2^20-point sumcheck).  The partials cross the XCDs' L2s:
// release before the counter, acquire after it.  Streaming passes and passes of many blocks do NOT take this path: the
// fences cost the fold pass 15 % of its bandwidth (measured in round 2), three times what the launch saves.
__device__ __forceinline__ void mle_publish_from_last_block(const MleSubPartial* __restrict__ partials, const MlePublish& pub) {
    __shared__ uint32_t s_last;
    const uint32_t b = blockIdx.y, nblk = gridDim.x;
    if (threadIdx.x == 0) {
        // (thread 0 wrote the block's partial: ITS release is the one that counts.  A release is a write-back of the XCD's
        // L2 and costs ~30 ns, one after the other across the grid: 4096 of them -- every wave of a 1024-block pass --
        // turned an 11 us pass into a 149 us one, which is why only passes of few blocks come here at all)
        __threadfence();
        s_last = atomicAdd(pub.arrivals + b, 1u) == nblk - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x == 0) pub.arrivals[b] = 0u;   // (the next pass's blocks start after this kernel)
    mle_total_and_publish(partials + (size_t)b * nblk, nblk, pub.jout, pub.rec + b, pub.ticket);
}

// pass 0: sub-block sums of the input tables.  grid = (nblk, batch), nblk = 2^J * blocks-per-sub-block,
// every block a contiguous chunk of len / nblk entries.  dep: does the table depend on x_n
// (entry 2m vs 2m+1; the neighbour is read through the cache the partner lane just filled).
__global__ void __launch_bounds__(256) k_mle_sub_sums(const Fr* __restrict__ tables, size_t table_stride, uint32_t len,
                                                      MleSubPartial* __restrict__ partials, MlePublish pub) {
    __shared__ Acc<9> smem[4];
    __shared__ uint32_t s_dep;
    const Fr* t = tables + (size_t)blockIdx.y * table_stride;
    const uint32_t chunk = len / gridDim.x;
    const uint32_t begin = blockIdx.x * chunk, end = begin + chunk;
    Acc<9> acc[1] = {acc_zero<9>()};
    uint32_t dep = 0;
    if (threadIdx.x == 0) s_dep = 0;
    __syncthreads();
    for (uint32_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
        const Fr x = load_fr(t + i);
        acc_add_fr(acc[0], x);
        // entry 2m against 2m+1: the partner sits in the neighbour lane (chunks are multiples of 256 entries, so lanes
        // i and i^1 are in one wave and both active); DPP quad_perm [1,0,3,2] swaps neighbours, no second load
        uint32_t d = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) d |= x.l[k] ^ (uint32_t)__builtin_amdgcn_mov_dpp((int)x.l[k], 0xB1, 0xF, 0xF, true);
        dep |= d;
    }
    if (dep) atomicOr(&s_dep, 1u);
    block_sum<9, 1>(acc, smem);
    if (threadIdx.x == 0) {
        MleSubPartial* p = partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        p->sum = acc[0];
        p->dep = s_dep;
    }
    if (pub.arrivals) mle_publish_from_last_block(partials, pub);
}

// one pass: bind JIN variables with the weights w[0 .. 2^JIN) (Montgomery, wave-uniform), write the
// folded table of S entries, accumulate its sub-block sums.  grid = (nblk, batch), chunk = S / nblk.
template <int JIN>
__global__ void __launch_bounds__(256) k_mle_multifold(const Fr* __restrict__ src, size_t src_stride, Fr* __restrict__ dst,
                                                       size_t dst_stride, uint32_t S, const Fr* __restrict__ weights,
                                                       MleSubPartial* __restrict__ partials, MlePublish pub) {
    __shared__ Acc<9> smem[4];
    const Fr* s = src + (size_t)blockIdx.y * src_stride;
    Fr* d = dst + (size_t)blockIdx.y * dst_stride;
    const Fr* w = weights + (size_t)blockIdx.y * kMleMaxSub;
    const uint32_t chunk = S / gridDim.x;
    const uint32_t begin = blockIdx.x * chunk, end = begin + chunk;
    Acc<9> acc[1] = {acc_zero<9>()};
    for (uint32_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
        const Fr y = multifold_entry<JIN>(s, S, i, w);
        store_fr(d + i, y);
        acc_add_fr(acc[0], y);
    }
    block_sum<9, 1>(acc, smem);
    if (threadIdx.x == 0) {
        MleSubPartial* p = partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        p->sum = acc[0];
        p->dep = 0;
    }
    if (pub.arrivals) mle_publish_from_last_block(partials, pub);
}

// the same pass with the products on the matrix cores (mfma_fold.h): k_mle_fold_plan turns each sumcheck's
// weights into its digit matrix (grid = batch), k_mle_multifold_mfma streams the tables through it
// (chunk = S / nblk a multiple of 64)
template <int JIN>
__global__ void __launch_bounds__(1024) k_mle_fold_plan(const Fr* __restrict__ weights, MfmaFoldPlan* __restrict__ plans) {
    __shared__ __attribute__((aligned(16))) unsigned char digits[32 * 32 * (1 << JIN)];
    mfma_plan_block<JIN>(weights + (size_t)blockIdx.x * kMleMaxSub, plans + blockIdx.x, digits);
}

template <int JIN>
__global__ void __launch_bounds__(256) k_mle_multifold_mfma(const Fr* __restrict__ src, size_t src_stride, Fr* __restrict__ dst,
                                                            size_t dst_stride, uint32_t S, const MfmaFoldPlan* __restrict__ plans,
                                                            MleSubPartial* __restrict__ partials, MlePublish pub) {
    __shared__ Acc<9> smem[4];
    __shared__ __attribute__((aligned(16))) unsigned char digits[JIN > 2 ? 32 * 32 * (1 << JIN) : 16];
    const Fr* s = src + (size_t)blockIdx.y * src_stride;
    Fr* d = dst + (size_t)blockIdx.y * dst_stride;
    const uint32_t chunk = S / gridDim.x;
    const uint32_t begin = blockIdx.x * chunk;
    Acc<9> acc[1] = {acc_zero<9>()};
    mfma_multifold_block<JIN>(s, d, S, plans + blockIdx.y, begin, begin + chunk, blockIdx.x * 5u + blockIdx.y * 3u, acc[0], digits);
    block_sum<9, 1>(acc, smem);
    if (threadIdx.x == 0) {
        MleSubPartial* p = partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        p->sum = acc[0];
        p->dep = 0;
    }
    if (pub.arrivals) mle_publish_from_last_block(partials, pub);
}

// partials of a pass -> 2^jout canonical sub-block sums per table -> pinned host record (mle_total_and_publish).  Streaming
// passes and passes of many blocks publish through this second launch; fusing it into them (the last block to arrive
// reduces: mle_publish_from_last_block) costs a streaming pass 15 % of its bandwidth, three times what the launch saves.
// grid = (batch), block = 512
__global__ void __launch_bounds__(512) k_mle_sub_reduce(const MleSubPartial* __restrict__ partials, uint32_t nblk,
                                                        uint32_t jout, MleHostRecSub* __restrict__ host_rec, uint32_t ticket) {
    mle_total_and_publish(partials + (size_t)blockIdx.x * nblk, nblk, jout, host_rec + blockIdx.x, ticket);
}

// small tables: one block per sumcheck does the whole pass (multifold, sub-block sums, publish).  JIN == 0: no fold,
// only the sub-block sums of the table as it stands (tiny first pass).  grid = (batch), block = 256
__global__ void __launch_bounds__(256) k_mle_sums_small(const Fr* __restrict__ src, size_t src_stride, uint32_t S, uint32_t jout,
                                                        MleHostRecSub* __restrict__ host_rec, uint32_t ticket) {
    const uint32_t b = blockIdx.x;
    const Fr* tbl = src + (size_t)b * src_stride;
    const uint32_t sub = S >> jout, nsub = 1u << jout;
    MleHostRecSub* r = host_rec + b;
    uint32_t dep = 0;
    // 256 / nsub threads per sub-block (nsub <= 32): partial sums, totals through LDS, then one thread per
    // sub-block reduces its total mod p
    __shared__ Acc<9> s_part[256];
    const uint32_t tps = 256u >> jout, sb = threadIdx.x / tps, rr = threadIdx.x % tps;
    Acc<9> acc = acc_zero<9>();
    for (uint32_t i = rr; i < sub; i += tps) {
        const Fr x = load_fr(tbl + (size_t)sb * sub + i);
        acc_add_fr(acc, x);
        dep |= fr_eq(x, load_fr(tbl + (((size_t)sb * sub + i) ^ 1u))) ? 0u : 1u;
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    // tree over the tps partials of each sub-block (tps is a power of two)
    for (uint32_t step = tps >> 1; step >= 1u; step >>= 1) {
        if (rr < step) {
            Acc<9> mine = s_part[threadIdx.x];
            acc_add_acc(mine, s_part[threadIdx.x + step]);
            s_part[threadIdx.x] = mine;
        }
        __syncthreads();
    }
    if (threadIdx.x < nsub) r->sums[threadIdx.x] = acc_reduce(s_part[threadIdx.x * tps]);
    dep = __syncthreads_or(dep);
    if (threadIdx.x == 0) {
        r->dep = dep ? 1u : 0u;
        __hip_atomic_store(&r->seq, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The fold passes of small tables: S <= 512 outputs, each a sum of 2^jin products.  These are the latency-bound tail of
// every sumcheck (2^12 -> 2^7 -> 2^2 at n = 20), so the (output, term) pairs are spread over ALL 1024 threads of the
// block -- thread (i, q) sums the terms b = q * per .. q * per + per - 1 of output i, per = max(1, S 2^jin / 1024) --
// instead of one thread per output walking its 32 products (128 and then 4 busy threads of 256: 31 and 29 us per pass
// for a lone sumcheck; the pass's weights, which sit in pinned host memory, are fetched once per block, not once per
// product).  The partial sums are reduced where they are (any order of exact sums gives the same canonical element),
// totalled per output, and the outputs' sub-block sums are taken from LDS.  grid = (batch), block = 1024
__global__ void __launch_bounds__(1024) k_mle_multifold_small(const Fr* __restrict__ src, size_t src_stride, Fr* __restrict__ dst,
                                                              size_t dst_stride, uint32_t S, uint32_t jin, uint32_t jout,
                                                              const Fr* __restrict__ weights, MleHostRecSub* __restrict__ host_rec,
                                                              uint32_t ticket, Fr* __restrict__ tail, uint32_t tail_stride) {
    __shared__ Fr s_w[kMleMaxSub];
    __shared__ Fr s_y[1024];       // partial sums [q][i], then (the first S) the outputs
    __shared__ Acc<9> s_part[kMleMaxSub];
    const uint32_t b = blockIdx.x;
    const Fr* s = src + (size_t)b * src_stride;
    Fr* d = dst + (size_t)b * dst_stride;
    const uint32_t nterm = 1u << jin;
    if (threadIdx.x < nterm * 8u)
        s_w[threadIdx.x >> 3].l[threadIdx.x & 7u] = reinterpret_cast<const uint32_t*>(weights + (size_t)b * kMleMaxSub)[threadIdx.x];
    __syncthreads();
    const uint32_t items = S << jin, per = items > 1024u ? items >> 10 : 1u, groups = nterm / per;
    const uint32_t i = threadIdx.x & (S - 1u), q = threadIdx.x / S;   // S is a power of two
    if (q < groups) {
        Lazy17 acc = lazy_zero();
        for (uint32_t t = 0; t < per; ++t) {
            const uint32_t term = q * per + t;
            lazy_mac_v(acc, load_fr(s + (size_t)term * S + i), s_w[term]);
        }
        s_y[q * S + i] = lazy_reduce(acc);
    }
    __syncthreads();
    Fr y;
    if (threadIdx.x < S) {
        Acc<9> tot = acc_zero<9>();
        for (uint32_t g = 0; g < groups; ++g) acc_add_fr(tot, s_y[g * S + threadIdx.x]);
        y = acc_reduce(tot);
        store_fr(d + threadIdx.x, y);
        // (the host finishes the sumcheck from here: the folded table into pinned memory, released with the record below)
        if (tail) store_fr(tail + (size_t)b * tail_stride + threadIdx.x, y);
    }
    if (tail) __threadfence_system();
    __syncthreads();   // every partial has been read
    if (threadIdx.x < S) s_y[threadIdx.x] = y;
    __syncthreads();
    // sub-block sums of the outputs: min(sub, 1024 / nsub) threads per sub-block, totals through LDS
    const uint32_t nsub = 1u << jout, sub = S >> jout;
    uint32_t tps = 1024u >> jout;
    if (tps > sub) tps = sub;
    const uint32_t sb = threadIdx.x / tps, rr = threadIdx.x % tps;
    MleHostRecSub* r = host_rec + b;
    const bool mine = sb < nsub;
    Acc<9> acc = acc_zero<9>();
    if (mine)
        for (uint32_t k = rr; k < sub; k += tps) acc_add_fr(acc, s_y[sb * sub + k]);
    // tree over the tps partials of a sub-block: within a wave by shuffles, across waves (tps > 64) through LDS
    const uint32_t within = tps < 64u ? tps : 64u;
    for (uint32_t off = within >> 1; off >= 1u; off >>= 1) {
        Acc<9> o;
#pragma unroll
        for (int l = 0; l < 9; ++l) o.l[l] = __shfl_down(acc.l[l], off, 64);
        acc_add_acc(acc, o);
    }
    if (tps > 64u) {
        // nsub * (tps / 64) <= 16 wave totals
        __shared__ Acc<9> s_wave[16];
        if (mine && (threadIdx.x & 63u) == 0) s_wave[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x < nsub) {
            Acc<9> tot = acc_zero<9>();
            for (uint32_t w = 0; w < (tps >> 6); ++w) acc_add_acc(tot, s_wave[threadIdx.x * (tps >> 6) + w]);
            s_part[threadIdx.x] = tot;
        }
    } else if (mine && rr == 0) {
        s_part[sb] = acc;
    }
    __syncthreads();
    if (threadIdx.x < nsub) r->sums[threadIdx.x] = acc_reduce(s_part[threadIdx.x]);
    __syncthreads();   // every record store is issued and waited for before the release below
    if (threadIdx.x == 0) {
        r->dep = 0u;
        __hip_atomic_store(&r->seq, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Late rounds (host transcript): the table is small, so ONE block per sumcheck folds
// it, totals the two half sums and publishes the host record itself -- one launch
// per round instead of fold + reduce, and no partials round trip.
// grid = (batch), block = 256
__global__ void __launch_bounds__(256) k_mle_fold_sum_small(const Fr* __restrict__ src, size_t src_stride,
                                                            Fr* __restrict__ dst, size_t dst_stride, uint32_t q,
                                                            const FixedMul* __restrict__ rtab,
                                                            MleHostRec* __restrict__ host_rec, uint32_t ticket) {
    __shared__ Acc<9> smem[4 * 2];
    const uint32_t b = blockIdx.x;
    const Fr* s = src + (size_t)b * src_stride;
    Fr* d = dst + (size_t)b * dst_stride;
    const FixedMul T = rtab[b];
    Acc<9> acc[2] = {acc_zero<9>(), acc_zero<9>()};
    for (uint32_t i = threadIdx.x; i < q; i += blockDim.x) {
        Fr x0 = load_fr(s + i), x1 = load_fr(s + i + 2 * (size_t)q);
        Fr x2 = load_fr(s + i + q), x3 = load_fr(s + i + 3 * (size_t)q);
        Fr y0 = fr_fold_fixed(x0, x1, T);
        Fr y1 = fr_fold_fixed(x2, x3, T);
        store_fr(d + i, y0);
        store_fr(d + i + q, y1);
        acc_add_fr(acc[0], y0);
        acc_add_fr(acc[1], y1);
    }
    block_sum<9, 2>(acc, smem);
    if (threadIdx.x == 0) {
        Fr c0 = acc_reduce(acc[0]);
        Fr c1 = fr_sub(acc_reduce(acc[1]), c0);
        MleHostRec* r = host_rec + b;
        r->c0 = c0;
        r->c1 = c1;
        r->dep = 0;
        __hip_atomic_store(&r->seq, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Host-transcript variant of the round tail: one wave per table totals the
// partials and hands the canonical sums to the host through pinned, fine-grained
// memory; the host applies the length rule and hashes (fr64.h).  The record's
// seq word is stored last with a system-scope release; the host spins on it.
// grid = (batch), block = 64
__global__ void __launch_bounds__(64) k_mle_round_reduce(const MlePartial* __restrict__ partials, uint32_t nblk,
                                                         MleHostRec* __restrict__ host_rec, uint32_t ticket) {
    const uint32_t b = blockIdx.x;
    const MlePartial* p = partials + (size_t)b * nblk;
    Acc<10> lo = acc_zero<10>(), hi = acc_zero<10>();
    uint32_t dep = 0;
    for (uint32_t i = threadIdx.x; i < nblk; i += 64) {
        acc_add_acc(lo, p[i].lo);
        acc_add_acc(hi, p[i].hi);
        dep |= p[i].dep;
    }
    lo = wave_sum(lo);
    hi = wave_sum(hi);
    dep = __any(dep) ? 1u : 0u;
    if (threadIdx.x == 0) {
        Fr c0 = acc_reduce(lo);
        Fr c1 = fr_sub(acc_reduce(hi), c0);
        MleHostRec* r = host_rec + b;
        r->c0 = c0;
        r->c1 = c1;
        r->dep = dep;
        __hip_atomic_store(&r->seq, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------
// forward evaluation of one circuit layer (reference: calculate_input,
// rust/src/convert.rs:812-831): out[g] = prev[l] (+|*) prev[r]
// ---------------------------------------------------------------------------

__global__ void k_layer_eval(uint32_t gates, const uint8_t* __restrict__ gate_type,
                             const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                             const Fr* __restrict__ prev, Fr* __restrict__ out, uint32_t prev_stride, const GateSet* __restrict__ sets) {
    if (sets) {   // proofs of different circuits in one launch: this proof's gate arrays
        const GateSet gs = sets[blockIdx.y];
        gate_type = gs.gate_type;
        left = gs.left;
        right = gs.right;
    }
    prev += (size_t)blockIdx.y * prev_stride;   // grid.y = proof of a batch (its own values)
    out += (size_t)blockIdx.y * gates;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < gates; g += gridDim.x * blockDim.x) {
        Fr a = load_fr(prev + left[g]), b = load_fr(prev + right[g]);
        store_fr(out + g, gate_type[g] ? fr_mul(a, b) : fr_add(a, b));
    }
}

// ---------------------------------------------------------------------------
// Product passes.  Both phases of the linear-time layer sumcheck are a sumcheck of  h(t) = W(t) X(t) + Y(t)  over
// three tables of 2^k entries (b-phase: X = U, Y = V; c-phase: X = a_u + W(u) m_u, Y = W(u) a_u).  The sums a round
// needs are bilinear in (W, X) and linear in Y, so -- as for the plain sumcheck's multi-round passes -- the sums of J
// consecutive rounds follow from the sub-block CROSS sums
//     m[a][b] = sum_i W[a S + i] X[b S + i],   sy[a] = sum_i Y[a S + i],   a, b < 2^J,  S = 2^(m - J):
// round 1 reads the entries whose trailing bits agree, binding a variable folds the matrix along both indices.  One
// device round trip per J <= 3 rounds instead of one per round; the host folds a 64-entry matrix between hashes.
// Same round polynomials as the per-round kernels: bit-exact.
// ---------------------------------------------------------------------------

// grid = (blocks, batch), block = 256: this block's tile of i, all 2^J sub-blocks.  TILE = kProdTile entries of every
// sub-block per block for the tables of a circom-sized layer (latency: many small blocks), kProdTileWide for wide layers
// (sub-blocks of >= 1024 entries: fewer partials to total on the round path).
template <uint32_t TILE>
__global__ void __launch_bounds__(256) k_prod_cross(Fr* __restrict__ Wt, Fr* __restrict__ Xt, Fr* __restrict__ Yt, uint32_t m_in, uint32_t jp,
                                                    const Fr* __restrict__ weights, uint32_t J, Fr* __restrict__ partials, uint32_t wstride,
                                                    ProdPassRec* __restrict__ rec, uint32_t ticket, uint32_t* __restrict__ arrivals,
                                                    Fr* __restrict__ tail, uint32_t tail_stride) {
    __shared__ Fr s_w[8];
    __shared__ uint32_t s_last;
    __shared__ Fr s_t[3][8][TILE];               // folded tile: table, sub-block, i
    __shared__ Fr s_red[4][kProdRecValues];
    const uint32_t tid = threadIdx.x, proof = blockIdx.y;
    Fr* T[3] = {Wt + (size_t)proof * wstride, Xt + (size_t)proof * wstride, Yt + (size_t)proof * wstride};
    const uint32_t m = m_in - jp, nsub = 1u << J, S = 1u << (m - J);
    const uint32_t ti = S < TILE ? S : TILE, i0 = blockIdx.x * TILE;
    if (tid < (1u << jp)) s_w[tid] = load_fr(weights + (size_t)proof * 8 + tid);
    __syncthreads();
    // the tile of the tables with the previous pass's variables bound (written back: later passes read it)
    for (uint32_t e = tid; e < 3u * nsub * ti; e += blockDim.x) {
        const uint32_t i = e % ti, a = (e / ti) % nsub, t = e / (ti * nsub);
        const uint32_t idx = a * S + i0 + i;
        Fr v;
        if (jp) {
            // at most eight products: added unreduced, one short reduction (a pass is a latency chain on the round path:
            // 8 x 64 + 64 multiply-adds deep instead of 8 x 128)
            // (all operands requested before the first product: eight dependent-looking loads in a row cost the round path
            // ~1 us each)
            Fr in[8];
#pragma unroll
            for (uint32_t b = 0; b < 8u; ++b)
                if (b < (1u << jp)) in[b] = load_fr(T[t] + ((size_t)b << m) + idx);
            Lazy17 acc = lazy_zero();
#pragma unroll
            for (uint32_t b = 0; b < 8u; ++b)
                if (b < (1u << jp)) lazy_mac_v(acc, in[b], s_w[b]);
            v = lazy_reduce_k8(acc);
            store_fr(T[t] + idx, v);
        } else {
            v = load_fr(T[t] + idx);
        }
        s_t[t][a][i] = v;
        // (the host takes the rest of the phase over from here: the tables as this pass's rounds find them, into pinned memory)
        if (tail) store_fr(tail + ((size_t)proof * 3u + t) * tail_stride + idx, v);
    }
    if (tail) __threadfence_system();   // (before the barrier the record's release stands behind)
    __syncthreads();
    const uint32_t p = tid & 63u, sub = tid >> 6, a = p >> 3, b = p & 7u;
    Fr acc = fr_zero(), accy = fr_zero();
    if (a < nsub && b < nsub) {
        Lazy17 la = lazy_zero();   // TILE / 4 products per thread
        for (uint32_t i = sub; i < ti; i += 4) lazy_mac_v(la, s_t[1][b][i], s_t[0][a][i]);
        acc = lazy_reduce_k8(la);
    }
    if (p < nsub)
        for (uint32_t i = sub; i < ti; i += 4) accy = fr_add(accy, s_t[2][p][i]);
    s_red[sub][p] = acc;
    if (p < 8) s_red[sub][64 + p] = accy;
    __syncthreads();
    // one block per proof (tables of <= 16 entries per sub-block): its sums are the pass's sums, published from here
    const bool alone = gridDim.x == 1;
    if (tid < (uint32_t)kProdRecValues) {
        Fr v = fr_add(fr_add(s_red[0][tid], s_red[1][tid]), fr_add(s_red[2][tid], s_red[3][tid]));
        store_fr(alone ? &rec[proof].v[tid] : partials + ((size_t)proof * gridDim.x + blockIdx.x) * kProdRecValues + tid, v);
    }
    if (alone) {
        __syncthreads();   // every record store is issued and waited for before the release below
        if (tid == 0) __hip_atomic_store(&rec[proof].seq, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (arrivals) {
        // A few blocks per proof (<= kProdFuseBlocks): the LAST block to arrive totals the partials and publishes -- no second
        // launch (a publish kernel cost its ~10 us plus the ~6 us to the next dependent launch on every such pass of every
        // layer).  The partials cross the XCDs' L2s: release (write-back) before the arrival counter, acquire (invalidate)
        // after it; the streaming passes of the plain sumcheck do NOT do this (there the fences cost 15 % of the bandwidth).
        // (ONE release per block, by one thread, after the barrier that orders the other threads' partial stores before it: a
        // release is a write-back of the XCD's L2, ~30 ns each and one after the other across the grid -- mle_publish_from_last_block)
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            s_last = atomicAdd(arrivals + proof, 1u) == gridDim.x - 1u ? 1u : 0u;
        }
        __syncthreads();
        if (s_last) {
            __threadfence();
            const uint32_t q = tid / (uint32_t)kProdRecValues, val = tid % (uint32_t)kProdRecValues;   // three threads per value
            if (q < 3u) {
                // (eight partials requested before the first is added: they sit in other XCDs' memory, ~1 us each when one
                // load waits for the addition before it)
                const Fr* pp = partials + (size_t)proof * gridDim.x * kProdRecValues + val;
                Fr v = fr_zero();
                for (uint32_t b0 = q; b0 < gridDim.x; b0 += 24u) {
                    Fr in[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8u; ++j)
                        if (b0 + 3u * j < gridDim.x) in[j] = load_fr(pp + (size_t)(b0 + 3u * j) * kProdRecValues);
#pragma unroll
                    for (uint32_t j = 0; j < 8u; ++j)
                        if (b0 + 3u * j < gridDim.x) v = fr_add(v, in[j]);
                }
                s_red[q][val] = v;
            }
            __syncthreads();
            if (tid < (uint32_t)kProdRecValues) store_fr(&rec[proof].v[tid], fr_add(fr_add(s_red[0][tid], s_red[1][tid]), s_red[2][tid]));
            __syncthreads();   // every record store is issued and waited for before the release below
            if (tid == 0) {
                arrivals[proof] = 0u;   // (the next pass's blocks start after this kernel)
                __hip_atomic_store(&rec[proof].seq, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// many blocks (a wide layer's first passes): first level of the totals over a grid -- block r of kProdReduceBlocks totals
// every kProdReduceBlocks-th partial (four threads per value), k_prod_publish then totals the kProdReduceBlocks results.
// grid = (kProdReduceBlocks, batch), block = 288
constexpr uint32_t kProdReduceBlocks = 64;
__global__ void __launch_bounds__(320) k_prod_reduce(const Fr* __restrict__ partials, uint32_t blocks, Fr* __restrict__ out) {
    __shared__ Fr s_q[4][kProdRecValues];
    const uint32_t tid = threadIdx.x, q = tid / (uint32_t)kProdRecValues, val = tid % (uint32_t)kProdRecValues;
    const Fr* p = partials + (size_t)blockIdx.y * blocks * kProdRecValues + val;
    Fr v = fr_zero();
    for (uint32_t k = blockIdx.x + q * kProdReduceBlocks; k < blocks; k += 4u * kProdReduceBlocks) v = fr_add(v, load_fr(p + (size_t)k * kProdRecValues));
    s_q[q][val] = v;
    __syncthreads();
    if (tid < (uint32_t)kProdRecValues)
        store_fr(out + ((size_t)blockIdx.y * kProdReduceBlocks + blockIdx.x) * kProdRecValues + tid,
                 fr_add(fr_add(s_q[0][tid], s_q[1][tid]), fr_add(s_q[2][tid], s_q[3][tid])));
}

// grid = (batch), block = Q * 72 (Q = 4, or 14 for the many blocks of a wide layer): totals of the blocks' partials -> the
// pinned record.  Q threads per value, each a Q-th of the blocks (the kernel sits between two hashes of the round path: 32
// dependent additions took 12 us)
__global__ void __launch_bounds__(1024) k_prod_publish(const Fr* __restrict__ partials, uint32_t blocks, ProdPassRec* __restrict__ rec,
                                                       uint32_t ticket) {
    __shared__ Fr s_q[14][kProdRecValues];
    const uint32_t tid = threadIdx.x, Q = blockDim.x / (uint32_t)kProdRecValues, q = tid / (uint32_t)kProdRecValues,
                   val = tid % (uint32_t)kProdRecValues;
    ProdPassRec* r = rec + blockIdx.x;
    if (q < Q) {
        // (four partials requested before the first is added: a load per addition, each waiting for the one before, was ~0.5 us a
        // term; eight at a time do not fit the 64 registers a block of 1024 threads leaves a lane)
        const Fr* p = partials + (size_t)blockIdx.x * blocks * kProdRecValues + val;
        Fr v = fr_zero();
        uint32_t k = q;
        for (; k + 3u * Q < blocks; k += 4u * Q) {
            const Fr a = load_fr(p + (size_t)k * kProdRecValues), b = load_fr(p + (size_t)(k + Q) * kProdRecValues);
            const Fr c = load_fr(p + (size_t)(k + 2u * Q) * kProdRecValues), d = load_fr(p + (size_t)(k + 3u * Q) * kProdRecValues);
            v = fr_add(fr_add(v, a), fr_add(fr_add(b, c), d));
        }
        for (; k < blocks; k += Q) v = fr_add(v, load_fr(p + (size_t)k * kProdRecValues));
        s_q[q][val] = v;
    }
    __syncthreads();
    if (tid < (uint32_t)kProdRecValues) {
        Fr v = s_q[0][tid];
        for (uint32_t j = 1; j < Q; ++j) v = fr_add(v, s_q[j][tid]);
        store_fr(&r->v[tid], v);
    }
    __syncthreads();   // every record store is issued and waited for before the release below
    if (tid == 0) __hip_atomic_store(&r->seq, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// grid = (blocks over 2^k, batch), block = 256
__global__ void __launch_bounds__(256) k_prod_c_setup(const Fr* __restrict__ Wb, uint32_t jp, const Fr* __restrict__ weights,
                                                      const Fr* __restrict__ A, const Fr* __restrict__ M, Fr* __restrict__ X,
                                                      Fr* __restrict__ Y, uint32_t k, uint32_t wstride) {
    __shared__ Fr s_wu;
    const size_t base = (size_t)blockIdx.y * wstride;
    if (threadIdx.x < 64) {
        // W(u): the last b pass's variables bound in what is left of Wb (Montgomery in, Montgomery out) -- one product per
        // lane (at most eight), summed across lanes: one product deep instead of eight on every layer's set-up path
        Fr wu = fr_zero();
        if (threadIdx.x < (1u << jp)) wu = mont_mul(load_fr(Wb + base + threadIdx.x), load_fr(weights + (size_t)blockIdx.y * 8 + threadIdx.x));
#pragma unroll
        for (int off = 4; off >= 1; off >>= 1) {
            Fr o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.l[j] = __shfl_down(wu.l[j], off, 64);
            wu = fr_add(wu, o);
        }
        if (threadIdx.x == 0) s_wu = wu;
    }
    __syncthreads();
    const Fr wu = s_wu;
    const uint32_t n = 1u << k;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        const Fr a = load_fr(A + base + c), mm = load_fr(M + base + c);
        store_fr(X + base + c, fr_add(a, mont_mul(mm, wu)));
        store_fr(Y + base + c, mont_mul(a, wu));
    }
}

// Small transfers as kernels: `words` 32-bit words from src to dst, either of which may be pinned host memory.  The
// round path of a proof makes no transfer call of the runtime (measured: with twelve contexts proving side by side, an
// asynchronous 256 KB host-to-device copy per layer now and then held every context's launches up for 6 - 8 ms).
__global__ void k_copy_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t words) {
    const size_t quads = words / 4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (size_t i = quads * 4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// `rows` rows of `words` 32-bit words each, row r from src + r * src_stride_words to dst + r * dst_stride_words (the first
// value of every proof's table in ONE launch: a launch per proof cost a lockstep group seven launches where a lone proof has one)
__global__ void k_copy_rows(const uint32_t* __restrict__ src, size_t src_stride_words, uint32_t* __restrict__ dst, size_t dst_stride_words,
                            uint32_t words, uint32_t rows) {
    const size_t total = (size_t)words * rows;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / words, w = i % words;
        dst[r * dst_stride_words + w] = src[r * src_stride_words + w];
    }
}

// out[proof][g] = eq(point, g) = prod_i (bit_i(g) ? x_i : 1 - x_i) over the `nvars` coordinates x_i = point[first + i] of
// the proof's point (variable `first` = most significant index bit), canonical or in Montgomery form.  The points are
// read where the host wrote them (pinned memory): no copy engine, no transfer call on the round path.  Every entry is
// its own product of nvars factors (the host's doubling construction gives the same field elements).
// grid = (blocks over 2^nvars, batch), block = 256
__device__ __forceinline__ void eq_table_part(const Fr* __restrict__ points, uint32_t stride, uint32_t first, uint32_t nvars,
                                              Fr* __restrict__ out, uint32_t montgomery, uint32_t bx, uint32_t nbx, uint32_t proof) {
    __shared__ Fr s_f[2][32];   // Montgomery forms of 1 - x_i and x_i
    const Fr* pt = points + (size_t)proof * stride + first;
    if (threadIdx.x < nvars) {
        const Fr x = load_fr(pt + threadIdx.x);
        Fr one = fr_zero();
        one.l[0] = 1u;
        s_f[1][threadIdx.x] = to_mont(x);
        s_f[0][threadIdx.x] = to_mont(fr_sub(one, x));
    }
    __syncthreads();
    // Four lanes per entry: lane q of a quad multiplies the factors i = q, q + 4, ... (a 254-bit product is a ~1.2 us
    // dependent chain on one lane, and these kernels sit on every layer's set-up path: 16 variables = 4 + 2 products deep
    // instead of 15), then the quad's four partial products are multiplied together across lanes.
    const uint32_t n = 1u << nvars, q = threadIdx.x & 3u;
    for (uint32_t g = (bx * blockDim.x + threadIdx.x) >> 2; g < n; g += (nbx * blockDim.x) >> 2) {
        Fr p = fr_mont_one();
        bool any = false;
        for (uint32_t i = q; i < nvars; i += 4) {
            const Fr f = s_f[(g >> (nvars - 1u - i)) & 1u][i];
            p = any ? mont_mul(p, f) : f;
            any = true;
        }
#pragma unroll
        for (int step = 1; step <= 2; step <<= 1) {
            Fr o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.l[j] = __shfl_xor(p.l[j], step, 64);
            p = mont_mul(p, o);
        }
        if (q == 0) {
            if (!montgomery) p = from_mont(p);
            store_fr(out + ((size_t)proof << nvars) + g, p);
        }
    }
}

// W(u) = sum_{i < 2^jp} Wb[i] * weights[i] of one proof (Montgomery in, Montgomery out), by the calling wave's first lanes:
// what k_prod_c_setup computes per block, once per proof (the eq-table launch of the c-phase leaves it for the wide row pass)
__device__ __forceinline__ void c_phase_wu(const Fr* Wb, const Fr* weights, uint32_t jp, uint32_t wstride, uint32_t proof, Fr* wu_out) {
    const uint32_t lane = threadIdx.x & 63u;
    Fr wu = fr_zero();
    if (lane < (1u << jp)) wu = mont_mul(load_fr(Wb + (size_t)proof * wstride + lane), load_fr(weights + (size_t)proof * 8 + lane));
#pragma unroll
    for (int off = 4; off >= 1; off >>= 1) {
        Fr o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.l[j] = __shfl_down(wu.l[j], off, 64);
        wu = fr_add(wu, o);
    }
    if (lane == 0) store_fr(wu_out + proof, wu);
}

// The same table for MANY variables (eq(u, .) of a wide layer: 2^20 entries x 20 factors were 237 us): a block owns 4096
// consecutive entries, i.e. one value of the leading nvars - 12 index bits.  Its product over those bits once (four lanes,
// strided factors, as above), the sixteen products over the next four bits, the 256 over the last eight from two
// sixteen-entry tables -- then TWO products per entry: (hi * mid[..]) * lo[..].  Same field elements as eq_table_part.
// nvars >= 12.  grid = (2^(nvars - 12), batch), block = 1024 (four entries = eight products per thread; with 256 threads a
// thread's sixteen entries were a chain of 32 products -- 40 us for the eight blocks of a 2^15-entry table, on a device that
// fourteen proving threads keep short of queues, not of lanes)
__global__ void __launch_bounds__(1024) k_eq_table_split(const Fr* __restrict__ points, uint32_t stride, uint32_t first, uint32_t nvars,
                                                        Fr* __restrict__ out, uint32_t montgomery, const Fr* __restrict__ wu_Wb,
                                                        const Fr* __restrict__ wu_weights, uint32_t wu_jp, uint32_t wu_wstride, Fr* __restrict__ wu_out) {
    __shared__ Fr s_f[2][32];
    __shared__ Fr s_a[16], s_b[16], s_hm[16], s_lo[256];
    __shared__ Fr s_hi;
    const uint32_t proof = blockIdx.y, tid = threadIdx.x, nh = nvars - 12u, hi_idx = blockIdx.x;
    const Fr* pt = points + (size_t)proof * stride + first;
    if (tid < nvars) {
        const Fr x = load_fr(pt + tid);
        Fr one = fr_zero();
        one.l[0] = 1u;
        s_f[1][tid] = to_mont(x);
        s_f[0][tid] = to_mont(fr_sub(one, x));
    }
    __syncthreads();
    if (tid < 4u) {   // the leading bits' product: lane q takes factors q, q + 4, ...; then the four partial products together
        Fr p = fr_mont_one();
        for (uint32_t i = tid; i < nh; i += 4u) p = mont_mul(p, s_f[(hi_idx >> (nh - 1u - i)) & 1u][i]);
#pragma unroll
        for (int step = 1; step <= 2; step <<= 1) {
            Fr o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.l[j] = __shfl_xor(p.l[j], step, 64);
            p = mont_mul(p, o);
        }
        if (tid == 0) s_hi = montgomery ? p : from_mont(p);   // canonical here makes every entry canonical without a product per entry
    } else if (tid >= 64u && tid < 64u + 48u) {   // three sixteen-entry tables over four variables each: mid, lo-a, lo-b
        const uint32_t which = (tid - 64u) >> 4, e = (tid - 64u) & 15u, v0 = nh + 4u * which;
        Fr p = s_f[(e >> 3) & 1u][v0];
        p = mont_mul(p, s_f[(e >> 2) & 1u][v0 + 1u]);
        p = mont_mul(p, s_f[(e >> 1) & 1u][v0 + 2u]);
        p = mont_mul(p, s_f[e & 1u][v0 + 3u]);
        (which == 0 ? s_hm : (which == 1 ? s_a : s_b))[e] = p;
    }
    __syncthreads();
    if (tid < 256u) s_lo[tid] = mont_mul(s_a[tid >> 4], s_b[tid & 15u]);
    else if (tid < 256u + 16u) s_hm[tid - 256u] = mont_mul(s_hi, s_hm[tid - 256u]);
    __syncthreads();
    Fr* dst = out + ((size_t)proof << nvars) + ((size_t)hi_idx << 12);
    const Fr lo = s_lo[tid & 255u];
    const uint32_t m0 = (tid >> 8) * 4u;   // a quarter of the sixteen mid entries per 256 threads
#pragma unroll
    for (uint32_t m = m0; m < m0 + 4u; ++m) store_fr(dst + (m << 8) + (tid & 255u), mont_mul(s_hm[m], lo));
    if (wu_out && hi_idx == 0u && tid < 64u) c_phase_wu(wu_Wb, wu_weights, wu_jp, wu_wstride, proof, wu_out);
}

__global__ void __launch_bounds__(256) k_eq_table(const Fr* __restrict__ points, uint32_t stride, uint32_t first, uint32_t nvars,
                                                  Fr* __restrict__ out, uint32_t montgomery, const Fr* __restrict__ wu_Wb,
                                                  const Fr* __restrict__ wu_weights, uint32_t wu_jp, uint32_t wu_wstride, Fr* __restrict__ wu_out) {
    eq_table_part(points, stride, first, nvars, out, montgomery, blockIdx.x, gridDim.x, blockIdx.y);
    if (wu_out && blockIdx.x == 0u && threadIdx.x < 64u) c_phase_wu(wu_Wb, wu_weights, wu_jp, wu_wstride, blockIdx.y, wu_out);
}

// Everything a layer's sumcheck needs before its first gate pass, in ONE launch (seven launches of 5 - 20 us each sat
// on every layer's set-up path: the two eq tables, two memsets, two Montgomery copies of W, the dependence flags):
//   blocks [0, nb_hi):   E_hi = eq(z[0 .. kh), .)   canonical
//   next nb_lo blocks:   E_lo = eq(z[kh .. k_i), .) Montgomery
//   next nb_mont blocks: Wb = Wc = Montgomery form of W (skipped when Wb is null)
//   last block:          dep[b] = 1 iff W differs somewhere across bit (k-1-b)  (k_depends; no memset, no global atomics),
//                        also left in pinned host memory (host_dep, may be null) for the host transcript's length rule
// grid = (nb_hi + nb_lo + nb_mont + 1, batch), block = 256
__global__ void __launch_bounds__(256) k_layer_prologue(const Fr* __restrict__ points, uint32_t k_i, uint32_t kh, uint32_t kl,
                                                        Fr* __restrict__ e_hi, Fr* __restrict__ e_lo, const Fr* __restrict__ W,
                                                        Fr* __restrict__ Wb, Fr* __restrict__ Wc, uint32_t k, uint32_t* __restrict__ dep,
                                                        uint32_t* __restrict__ host_dep, uint32_t nb_hi, uint32_t nb_lo, uint32_t nb_mont,
                                                        uint32_t* __restrict__ wide_bits) {
    const uint32_t bx = blockIdx.x, proof = blockIdx.y, n = 1u << k;
    if (bx < nb_hi) {
        eq_table_part(points, k_i, 0u, kh, e_hi, 0u, bx, nb_hi, proof);
    } else if (bx < nb_hi + nb_lo) {
        eq_table_part(points, k_i, kh, kl, e_lo, 1u, bx - nb_hi, nb_lo, proof);
    } else if (bx < nb_hi + nb_lo + nb_mont) {
        if (!Wb) return;
        const size_t base = (size_t)proof << k;
        for (uint32_t i = (bx - nb_hi - nb_lo) * blockDim.x + threadIdx.x; i < n; i += nb_mont * blockDim.x) {
            const Fr m = to_mont(load_fr(W + base + i));
            store_fr(Wb + base + i, m);
            store_fr(Wc + base + i, m);
        }
    } else {
        if (!dep) {
            // A wide table's flags are found over a grid (launch_depends_wide) -- but only a table that does NOT depend on some
            // variable needs the whole scan: that it does depend is shown by any one pair.  This block looks at the pairs of
            // the table's first 256 entries and STORES what it finds in the grid kernel's word (no memset before it): for a
            // generic table that is every bit, and the grid kernel's blocks leave at once (23.7 -> ~4 us at 2^20 values).
            if (!wide_bits) return;
            __shared__ uint32_t s_found;
            if (threadIdx.x == 0) s_found = 0;
            __syncthreads();
            const Fr* w = W + ((size_t)proof << k);
            uint32_t bits = 0;
            if (threadIdx.x < n) {
                const Fr a = load_fr(w + threadIdx.x);
                for (uint32_t b = 0; b < k; ++b) {
                    const uint32_t bit = 1u << (k - 1 - b);
                    if (!(threadIdx.x & bit) && !fr_eq(a, load_fr(w + (threadIdx.x ^ bit)))) bits |= 1u << b;
                }
            }
            if (bits) atomicOr(&s_found, bits);
            __syncthreads();
            if (threadIdx.x == 0) wide_bits[proof] = s_found;
            return;
        }
        __shared__ uint32_t s_dep;
        if (threadIdx.x == 0) s_dep = 0;
        __syncthreads();
        const Fr* w = W + ((size_t)proof << k);
        uint32_t bits = 0;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const Fr a = load_fr(w + i);
            for (uint32_t b = 0; b < k; ++b) {
                const uint32_t bit = 1u << (k - 1 - b);
                if (!(bits >> b & 1u) && !(i & bit) && !fr_eq(a, load_fr(w + (i ^ bit)))) bits |= 1u << b;
            }
        }
        if (bits) atomicOr(&s_dep, bits);
        __syncthreads();
        if (threadIdx.x < 32) {
            const uint32_t f = (s_dep >> threadIdx.x) & 1u;
            dep[(size_t)proof * 32 + threadIdx.x] = f;
            if (host_dep) host_dep[(size_t)proof * 32 + threadIdx.x] = f;   // pinned: the host reads it once a later kernel has released round 0's record
        }
    }
}

// q(t) = W(b + t (c - b)): the layer's values restricted to the line through the two points the sumcheck ended at
// (reduce_multiple_polynomial, poly.rs:469-500) -- the variables bound one after the other on the evaluation table
// with l_j(t) = b_j + t (c_j - b_j) in place of a challenge,
//     P'[i](t) = P[i](t) + l_j(t) (P[i + h](t) - P[i](t)),
// entries being coefficient vectors in t (lowest degree first) that grow by one coefficient per variable.  One block
// per proof; the table ping-pongs between two halves of `scratch` (entry i of the table with j variables bound at
// i * (j + 1); never more than 2^k elements); a third 2^k elements hold W's monomial coefficients (Moebius
// transform), of which only the support counts: *q_len = 1 + the largest total degree of a non-zero monomial
// (:484-497).  out: k + 1 slots, highest degree first.  bc: b_1..b_k, c_1..c_k per proof (canonical; may be pinned
// host memory).  grid = (batch), block = 256.
__global__ void __launch_bounds__(256) k_line_restriction(const Fr* __restrict__ W, uint32_t k, const Fr* __restrict__ bc,
                                                          Fr* __restrict__ scratch, Fr* __restrict__ out,
                                                          uint32_t* __restrict__ out_len) {
    // (the tables live in LDS -- 3 * 2^k elements, 48 KB at k = 9 -- and the line's coefficients are converted once, up
    // front: through global scratch, with b_j, c_j fetched from pinned memory and converted inside every step, the
    // 2k dependent steps took ~60 us of every layer's path to the next layer's z)
    extern __shared__ uint4 s_line_raw[];
    __shared__ uint32_t s_maxdeg;
    __shared__ Fr s_cst[16], s_grad[16];
    const uint32_t n = 1u << k, tid = threadIdx.x;
    const Fr* w = W + ((size_t)blockIdx.x << k);
    const Fr* line = bc + (size_t)blockIdx.x * 2u * k;
    Fr* const base = reinterpret_cast<Fr*>(s_line_raw);
    Fr* buf[2] = {base, base + n};
    Fr* mono = base + 2u * n;
    (void)scratch;
    if (tid == 0) s_maxdeg = 0;
    if (tid < k) {
        const Fr bj = load_fr(line + tid), cj = load_fr(line + k + tid);
        s_cst[tid] = to_mont(bj);
        s_grad[tid] = to_mont(fr_sub(cj, bj));
    }
    for (uint32_t i = tid; i < n; i += blockDim.x) {
        const Fr v = load_fr(w + i);
        store_fr(buf[0] + i, v);
        store_fr(mono + i, v);
    }
    __syncthreads();
    // monomial coefficients, variable 1 = most significant index bit (get_multi_ext, poly.rs:502-536)
    for (uint32_t bit = n >> 1; bit; bit >>= 1) {
        for (uint32_t i = tid; i < n; i += blockDim.x)
            if (i & bit) store_fr(mono + i, fr_sub(load_fr(mono + i), load_fr(mono + (i ^ bit))));
        __syncthreads();
    }
    uint32_t deg = 0;
    for (uint32_t i = tid; i < n; i += blockDim.x)
        if (!fr_is_zero(load_fr(mono + i))) deg = max(deg, (uint32_t)__popc(i));
    if (deg) atomicMax(&s_maxdeg, deg);
    // the k bindings
    uint32_t h = n >> 1;
    for (uint32_t j = 0; j < k; ++j, h >>= 1) {
        const Fr* src = buf[j & 1];
        Fr* dst = buf[(j & 1) ^ 1];
        const Fr cst = s_cst[j], grad = s_grad[j];
        const uint32_t in_len = j + 1, out_len_j = j + 2;
        for (uint32_t item = tid; item < h * out_len_j; item += blockDim.x) {
            const uint32_t i = item / out_len_j, m = item - i * out_len_j;
            const Fr* lo = src + (size_t)i * in_len;
            const Fr* hi = src + (size_t)(i + h) * in_len;
            Fr v = fr_zero();
            if (m < in_len) {
                const Fr l = load_fr(lo + m);
                v = fr_add(l, mont_mul(fr_sub(load_fr(hi + m), l), cst));
            }
            if (m > 0) v = fr_add(v, mont_mul(fr_sub(load_fr(hi + m - 1), load_fr(lo + m - 1)), grad));
            store_fr(dst + (size_t)i * out_len_j + m, v);
        }
        __syncthreads();
    }
    const Fr* fin = buf[k & 1];
    for (uint32_t d = tid; d <= k; d += blockDim.x) store_fr(out + (size_t)blockIdx.x * (k + 1u) + (k - d), load_fr(fin + d));
    if (tid == 0) out_len[blockIdx.x] = s_maxdeg + 1u;
}

// The line's coefficients for the step kernels below, once per restriction: l_j(t) = b_j + t (c_j - b_j) as (cst_j,
// grad_j) in Montgomery form, in device memory.  Every step used to fetch b_j, c_j from pinned host memory and convert
// them itself: a PCIe round trip and two products at the head of each of its k launches -- 4.5 of a step's 11 us, and these
// launches are a seventh of a many-circuit proving step's kernel time.  grid = (batch), block = 64
__global__ void __launch_bounds__(64) k_line_coeffs(uint32_t k, const Fr* __restrict__ bc, Fr* __restrict__ bcm) {
    const uint32_t j = threadIdx.x;
    if (j >= k) return;
    const Fr* line = bc + (size_t)blockIdx.x * 2u * k;
    const Fr bj = load_fr(line + j), cj = load_fr(line + k + j);
    store_fr(bcm + (size_t)blockIdx.x * 2u * k + j, to_mont(bj));
    store_fr(bcm + (size_t)blockIdx.x * 2u * k + k + j, to_mont(fr_sub(cj, bj)));
}

// The same for wide layers (k > 9), where one block per proof would run the 2^k (k + 1) products of a step on 256
// threads: the Moebius part and the set-up in one block (subtractions only), then one launch per variable over
// (blocks, batch), then the read-out.  Same buffers, same results.
__global__ void __launch_bounds__(256) k_line_init(const Fr* __restrict__ W, uint32_t k, Fr* __restrict__ scratch, uint32_t* __restrict__ out_len) {
    __shared__ uint32_t s_maxdeg;
    const uint32_t n = 1u << k, tid = threadIdx.x;
    const Fr* w = W + ((size_t)blockIdx.x << k);
    Fr* buf0 = scratch + (size_t)blockIdx.x * 3u * n;
    Fr* mono = buf0 + 2u * n;
    if (tid == 0) s_maxdeg = 0;
    for (uint32_t i = tid; i < n; i += blockDim.x) {
        const Fr v = load_fr(w + i);
        store_fr(buf0 + i, v);
        store_fr(mono + i, v);
    }
    __syncthreads();
    for (uint32_t bit = n >> 1; bit; bit >>= 1) {
        for (uint32_t i = tid; i < n; i += blockDim.x)
            if (i & bit) store_fr(mono + i, fr_sub(load_fr(mono + i), load_fr(mono + (i ^ bit))));
        __syncthreads();
    }
    uint32_t deg = 0;
    for (uint32_t i = tid; i < n; i += blockDim.x)
        if (!fr_is_zero(load_fr(mono + i))) deg = max(deg, (uint32_t)__popc(i));
    if (deg) atomicMax(&s_maxdeg, deg);
    __syncthreads();
    if (tid == 0) out_len[blockIdx.x] = s_maxdeg + 1u;
}

// variable j of every proof: grid = (blocks, batch)
__global__ void __launch_bounds__(256) k_line_step(uint32_t k, uint32_t j, const Fr* __restrict__ bc, Fr* __restrict__ scratch) {
    const uint32_t n = 1u << k, h = n >> (j + 1u);
    const Fr* line = bc + (size_t)blockIdx.y * 2u * k;
    Fr* base = scratch + (size_t)blockIdx.y * 3u * n;
    const Fr* src = base + ((j & 1u) ? n : 0u);
    Fr* dst = base + ((j & 1u) ? 0u : n);
    const Fr cst = load_fr(line + j), grad = load_fr(line + k + j);   // (k_line_coeffs: Montgomery form, device memory)
    const uint32_t in_len = j + 1u, out_len_j = j + 2u;
    for (uint32_t item = blockIdx.x * blockDim.x + threadIdx.x; item < h * out_len_j; item += gridDim.x * blockDim.x) {
        const uint32_t i = item / out_len_j, m = item - i * out_len_j;
        const Fr* lo = src + (size_t)i * in_len;
        const Fr* hi = src + (size_t)(i + h) * in_len;
        Fr v = fr_zero();
        if (m < in_len) {
            const Fr l = load_fr(lo + m);
            v = fr_add(l, mont_mul(fr_sub(load_fr(hi + m), l), cst));
        }
        if (m > 0) v = fr_add(v, mont_mul(fr_sub(load_fr(hi + m - 1), load_fr(lo + m - 1)), grad));
        store_fr(dst + (size_t)i * out_len_j + m, v);
    }
}

// The last steps of a wide layer's line restriction in ONE launch (from variable j0 on the table has 2^(k - j0) <= 64 entries):
// one block per proof binds the remaining variables one after the other (the same arithmetic as k_line_step, a barrier
// between the steps), then writes q -- highest degree first -- and its length (1 + the largest degree of a non-zero monomial
// of W, found by the set-up kernels).  grid = (batch), block = 256
__global__ void __launch_bounds__(256) k_line_tail(uint32_t k, uint32_t j0, const Fr* __restrict__ bc, Fr* __restrict__ scratch,
                                                   Fr* __restrict__ out, const uint32_t* __restrict__ maxdeg, uint32_t* __restrict__ out_len) {
    const uint32_t n = 1u << k;
    const Fr* line = bc + (size_t)blockIdx.x * 2u * k;
    Fr* base = scratch + (size_t)blockIdx.x * 3u * n;
    for (uint32_t j = j0; j < k; ++j) {
        const uint32_t h = n >> (j + 1u);
        const Fr* src = base + ((j & 1u) ? n : 0u);
        Fr* dst = base + ((j & 1u) ? 0u : n);
        const Fr cst = load_fr(line + j), grad = load_fr(line + k + j);
        const uint32_t in_len = j + 1u, out_len_j = j + 2u;
        for (uint32_t item = threadIdx.x; item < h * out_len_j; item += blockDim.x) {
            const uint32_t i = item / out_len_j, m = item - i * out_len_j;
            const Fr* lo = src + (size_t)i * in_len;
            const Fr* hi = src + (size_t)(i + h) * in_len;
            Fr v = fr_zero();
            if (m < in_len) {
                const Fr l = load_fr(lo + m);
                v = fr_add(l, mont_mul(fr_sub(load_fr(hi + m), l), cst));
            }
            if (m > 0) v = fr_add(v, mont_mul(fr_sub(load_fr(hi + m - 1), load_fr(lo + m - 1)), grad));
            store_fr(dst + (size_t)i * out_len_j + m, v);
        }
        __threadfence_block();
        __syncthreads();
    }
    const Fr* fin = base + ((k & 1u) ? n : 0u);
    for (uint32_t d = threadIdx.x; d <= k; d += blockDim.x) store_fr(out + (size_t)blockIdx.x * (k + 1u) + (k - d), load_fr(fin + d));
    if (threadIdx.x == 0) out_len[blockIdx.x] = maxdeg[blockIdx.x] + 1u;   // (out_len may be pinned host memory: a plain store)
}

__global__ void k_line_out(uint32_t k, const Fr* __restrict__ scratch, Fr* __restrict__ out) {
    const uint32_t n = 1u << k;
    const Fr* fin = scratch + (size_t)blockIdx.x * 3u * n + ((k & 1u) ? n : 0u);
    for (uint32_t d = threadIdx.x; d <= k; d += blockDim.x) store_fr(out + (size_t)blockIdx.x * (k + 1u) + (k - d), load_fr(fin + d));
}

// canonical -> Montgomery copy of a small table (W for the layer kernel)
__global__ void k_to_mont(const Fr* __restrict__ in, Fr* __restrict__ out, uint32_t count) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
        store_fr(out + i, to_mont(load_fr(in + i)));
}

// out[i] = Montgomery form of in[i * stride + offset]: the W copy a shard binds its c' variables on
__global__ void k_to_mont_strided(const Fr* __restrict__ in, Fr* __restrict__ out, uint32_t count, uint32_t stride,
                                  uint32_t offset) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x)
        store_fr(out + i, to_mont(load_fr(in + (size_t)i * stride + offset)));
}

// flag |= 1 iff a[i] != b[i] for some i (does a table depend on a variable that is a rank bit?)
__global__ void k_tables_differ(const Fr* __restrict__ a, const Fr* __restrict__ b, size_t count, uint32_t* __restrict__ flag) {
    uint32_t d = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        d |= fr_eq(load_fr(a + i), load_fr(b + i)) ? 0u : 1u;
    if (d) atomicOr(flag, 1u);
}

// the last fold of a shard: two entries -> one
__global__ void k_fold_pair(const Fr* __restrict__ src, Fr* __restrict__ dst, const FixedMul* __restrict__ rtab) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const FixedMul T = *rtab;
        store_fr(dst, fr_fold_fixed(load_fr(src), load_fr(src + 1), T));
    }
}

// dep[b] = 1 iff W differs somewhere across bit (k-1-b), i.e. iff a stored
// monomial of the MLE carries variable b+1 (length rule, poly.rs:388-420)
__global__ void k_depends(const Fr* __restrict__ W, uint32_t k, uint32_t* __restrict__ dep) {
    const uint32_t n = 1u << k;
    W += (size_t)blockIdx.y << k;   // grid.y = proof of a batch: its own W, its own 32 flags
    dep += (size_t)blockIdx.y * 32;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        Fr a = load_fr(W + i);
        for (uint32_t b = 0; b < k; ++b) {
            const uint32_t bit = 1u << (k - 1 - b);
            if (!(i & bit) && !fr_eq(a, load_fr(W + (i ^ bit)))) atomicOr(dep + b, 1u);
        }
    }
}

// ---------------------------------------------------------------------------
// wiring predicates (reference: chi_w_for_binary + add_poly at build time,
// convert.rs:715-767, then partial_eval_binary_form at z, prover.rs:24-37)
//   E[g] = prod_i (bit_i(g) ? z_i : 1 - z_i);  A[(l << k) | r] += E[g] (add gates),
//   M[...] += E[g] (mult gates).
// No modular atomic exists: scatter the eight 32-bit limbs with 64-bit integer
// atomics into a widened cell (8 x u64), then normalise.  Integer adds commute,
// so the result is exact and order independent.
// ---------------------------------------------------------------------------

// E[g] = E_hi[g >> kl] * E_lo[g & (2^kl - 1)]: the eq(z, .) weight of gate g as ONE product of two
// small host-built tables (E_hi canonical over the leading k_i - kl bits of g, E_lo in Montgomery
// form over the trailing kl bits; both L2-resident), instead of k_i products per gate.
// Shard (log_p, p): keep only gates whose right operand has low bits p (the trailing-variable
// partition of the hypercube, one shard per GPU); cell = (l << (k_next - log_p)) | (r >> log_p).
__global__ void k_predicate_scatter(uint32_t k_i, uint32_t k_next, const uint8_t* __restrict__ gate_type,
                                    const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                    const Fr* __restrict__ e_hi, const Fr* __restrict__ e_lo_mont, uint32_t kl,
                                    unsigned long long* __restrict__ wideA, unsigned long long* __restrict__ wideM,
                                    uint32_t* __restrict__ bad, uint32_t log_p, uint32_t shard) {
    const uint64_t gates = 1ull << k_i;
    const uint32_t lmask = (1u << kl) - 1u;
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < gates; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t l = left[g], r = right[g], ty = gate_type[g];
        if ((l >> k_next) | (r >> k_next) | (ty > 1u)) {
            atomicOr(bad, 1u);
            continue;
        }
        if ((r & ((1u << log_p) - 1u)) != shard) continue;
        const Fr e = mont_mul(load_fr(e_hi + (g >> kl)), load_fr(e_lo_mont + ((uint32_t)g & lmask)));
        unsigned long long* cell = (ty ? wideM : wideA) + (((size_t)l << (k_next - log_p)) | (r >> log_p)) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(cell + j, (unsigned long long)e.l[j]);
    }
}

// --- predicate build by counting sort (default): 2 u32 atomics per gate instead of 8 u64, no
// widened tables.  cells = [A table | M table] (2 * N entries):
//   count:  counts[cell(g)] += 1
//   scan:   offsets = exclusive prefix sum of counts; cursor = offsets
//   fill:   list[cursor[cell(g)]++] = g
//   sum:    out[cell] = sum over its list segment of E[g]   (modular adds commute: any order)
__device__ __forceinline__ bool pred_cell(uint32_t l, uint32_t r, uint32_t ty, uint32_t k_next, uint32_t log_p,
                                          uint32_t shard, size_t ncells, size_t& cell, uint32_t* bad) {
    if ((l >> k_next) | (r >> k_next) | (ty > 1u)) {
        atomicOr(bad, 1u);
        return false;
    }
    if ((r & ((1u << log_p) - 1u)) != shard) return false;
    cell = (ty ? ncells : 0) + (((size_t)l << (k_next - log_p)) | (r >> log_p));
    return true;
}

__global__ void k_pred_count(uint32_t k_i, uint32_t k_next, const uint8_t* __restrict__ gate_type,
                             const uint32_t* __restrict__ left, const uint32_t* __restrict__ right, uint32_t log_p,
                             uint32_t shard, size_t ncells, uint32_t* __restrict__ counts, uint32_t* __restrict__ bad) {
    const uint64_t gates = 1ull << k_i;
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < gates; g += (uint64_t)gridDim.x * blockDim.x) {
        size_t cell;
        if (pred_cell(left[g], right[g], gate_type[g], k_next, log_p, shard, ncells, cell, bad)) atomicAdd(counts + cell, 1u);
    }
}

__global__ void k_pred_fill(uint32_t k_i, uint32_t k_next, const uint8_t* __restrict__ gate_type,
                            const uint32_t* __restrict__ left, const uint32_t* __restrict__ right, uint32_t log_p,
                            uint32_t shard, size_t ncells, uint32_t* __restrict__ cursor, uint32_t* __restrict__ list,
                            uint32_t* __restrict__ bad) {
    const uint64_t gates = 1ull << k_i;
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < gates; g += (uint64_t)gridDim.x * blockDim.x) {
        size_t cell;
        if (pred_cell(left[g], right[g], gate_type[g], k_next, log_p, shard, ncells, cell, bad))
            list[atomicAdd(cursor + cell, 1u)] = (uint32_t)g;
    }
}

// exclusive scan, three passes: 2048 entries per block
constexpr uint32_t kScanPerBlock = 2048;
__global__ void __launch_bounds__(256) k_scan_blocks(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                     uint32_t* __restrict__ block_sums, size_t n) {
    __shared__ uint32_t wave_tot[4];
    const size_t base = (size_t)blockIdx.x * kScanPerBlock + (size_t)threadIdx.x * 8;
    uint32_t v[8], run = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0u;
        run += v[i];
    }
    // inclusive scan of the per-thread totals across the block
    uint32_t x = run;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wave] = x;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
    uint32_t excl = wbase + x - run;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (base + i < n) out[base + i] = excl;
        excl += v[i];
    }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = wbase + x;
}

__global__ void __launch_bounds__(1024) k_scan_sums(uint32_t* __restrict__ block_sums, uint32_t nblocks) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t val = i < nblocks ? block_sums[i] : 0u;
        uint32_t x = val;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) wave_tot[wave] = x;
        __syncthreads();
        uint32_t wbase = carry;
        for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
        if (i < nblocks) block_sums[i] = wbase + x - val;   // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) carry = wbase + x;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) k_scan_add(uint32_t* __restrict__ offsets, uint32_t* __restrict__ cursor,
                                                  const uint32_t* __restrict__ block_sums, size_t n) {
    const uint32_t add = block_sums[blockIdx.x];
    const size_t base = (size_t)blockIdx.x * kScanPerBlock;
    for (uint32_t t = threadIdx.x; t < kScanPerBlock; t += 256) {
        const size_t i = base + t;
        if (i < n) {
            const uint32_t v = offsets[i] + add;
            offsets[i] = v;
            if (cursor) cursor[i] = v;
        }
    }
}

// one thread per cell: out = sum of E[g] over the cell's segment [offsets, cursor)
__global__ void __launch_bounds__(256) k_pred_sum(size_t ncells, const uint32_t* __restrict__ offsets,
                                                  const uint32_t* __restrict__ cursor, const uint32_t* __restrict__ list,
                                                  const Fr* __restrict__ e_hi, const Fr* __restrict__ e_lo_mont, uint32_t kl,
                                                  uint32_t kh, Fr* __restrict__ out_A, Fr* __restrict__ out_M) {
    const uint32_t lmask = (1u << kl) - 1u;
    // grid.y = proof of a batch: same gates (same cell lists), its own eq tables and output tables
    e_hi += (size_t)blockIdx.y << kh;
    e_lo_mont += (size_t)blockIdx.y << kl;
    out_A += (size_t)blockIdx.y * ncells;
    out_M += (size_t)blockIdx.y * ncells;
    for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < 2 * ncells; c += (size_t)gridDim.x * blockDim.x) {
        const uint32_t b = offsets[c], e = cursor[c];
        Fr acc = fr_zero();
        for (uint32_t i = b; i < e; ++i) {
            const uint32_t g = list[i];
            acc = fr_add(acc, mont_mul(load_fr(e_hi + (g >> kl)), load_fr(e_lo_mont + (g & lmask))));
        }
        store_fr(c < ncells ? out_A + c : out_M + (c - ncells), acc);
    }
}

// ---------------------------------------------------------------------------
// Gate lists for the linear-time layer sumcheck (no 2^{2k}-entry predicate tables at all): the gates grouped by
// their LEFT operand (what U, V of the b-phase sum over) and by their RIGHT operand (what the rows a_u, m_u of the
// c-phase sum over).  One counting sort over 2 * 2^k buckets -- bucket b = left operand b, bucket 2^k + c = right
// operand c -- into one list of 2 G entries; shared by all proofs of a batch (same gates).
// ---------------------------------------------------------------------------
// One atomic for all the lanes of a wave that hit the same counter as the wave's first (then second) pending lane: the gates
// of a compiled layer come in runs -- 64 consecutive relay gates all read the zero slot (convert.rs:307-342), every other mult
// gate one hot wire -- and 2^19 single atomics on ONE address took 2.8 ms where 2^20 over random addresses take 0.13
// (profiles/r05/b_*).  The lanes that went together get consecutive positions in lane order: a run of gates stays a run in
// its bucket's list (what the item passes' gathers like).  Called by every lane of the wave; -> the lane's position (its old
// counter value), meaningless where !active.
__device__ __forceinline__ uint32_t wave_atomic_inc(uint32_t* __restrict__ counters, uint32_t idx, bool active) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t pos = 0;
    bool pending = active;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const unsigned long long todo = __ballot(pending);
        if (!todo) break;
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lead_idx = (uint32_t)__shfl((int)idx, leader, 64);
        const bool mine = pending && idx == lead_idx;
        const unsigned long long same = __ballot(mine);
        const uint32_t cnt = (uint32_t)__popcll(same);
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(counters + lead_idx, cnt);
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (mine) {
            pos = base + (uint32_t)__popcll(same & below);
            pending = false;
        }
    }
    if (pending) pos = atomicAdd(counters + idx, 1u);
    return pos;
}

__global__ void k_gate_count(uint64_t gates, uint32_t k, const uint8_t* __restrict__ gate_type, const uint32_t* __restrict__ left,
                             const uint32_t* __restrict__ right, uint32_t* __restrict__ counts, uint32_t* __restrict__ bad) {
    const uint32_t n = 1u << k;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g0 = blockIdx.x * (uint64_t)blockDim.x; g0 < gates; g0 += stride) {   // (every lane of a wave stays in the loop)
        const uint64_t g = g0 + threadIdx.x;
        bool ok = g < gates;
        uint32_t l = 0, r = 0;
        if (ok) {
            l = left[g];
            r = right[g];
            if (l >= n || r >= n || gate_type[g] > 1) {
                atomicOr(bad, 1u);
                ok = false;
            }
        }
        (void)wave_atomic_inc(counts, l, ok);
        (void)wave_atomic_inc(counts + n, r, ok);
    }
}

__global__ void k_gate_fill(uint64_t gates, uint32_t k, const uint8_t* __restrict__ gate_type, const uint32_t* __restrict__ left,
                            const uint32_t* __restrict__ right, uint32_t* __restrict__ cursor, uint32_t* __restrict__ list,
                            uint32_t* __restrict__ meta) {
    const uint32_t n = 1u << k;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g0 = blockIdx.x * (uint64_t)blockDim.x; g0 < gates; g0 += stride) {
        const uint64_t g = g0 + threadIdx.x;
        bool ok = g < gates;
        uint32_t l = 0, r = 0, t = 0;
        if (ok) {
            l = left[g];
            r = right[g];
            ok = !(l >= n || r >= n || gate_type[g] > 1);
            t = ok ? (uint32_t)gate_type[g] << 31 : 0u;
        }
        const uint32_t pl = wave_atomic_inc(cursor, l, ok), pr = wave_atomic_inc(cursor + n, r, ok);
        if (ok) {
            list[pl] = (uint32_t)g;
            meta[pl] = r | t;   // what the sums over this bucket need of the gate besides its index: the other operand, the type
            list[pr] = (uint32_t)g;
            meta[pr] = l | t;
        }
    }
}

// The same lists for LARGE layers (2^16 gates and more, k <= 12) without a global atomic per gate.  With 2^24 gates
// on 2 * 2^12 buckets the global counters are hit ~2000 times each and the two passes above take 1.1 ms apiece;
// here every block owns a contiguous run of gates and keeps its histogram -- then its write cursors -- in LDS
// (2 * 2^k counters <= 32 KB):
//   k_gate_count_lds   block histogram in LDS  ->  hist[bucket][block]   (bucket-major: one scan gives every block
//                                                   its start inside every bucket)
//   exclusive scan over 2 * 2^k * blocks entries (the kernels below), k_gate_bucket_bounds picks the buckets' bounds
//   k_gate_fill_lds    cursors = the block's starts, in LDS; list[cursor++] = gate
// The order of the gates inside a bucket differs from the global-atomic version; every consumer sums a bucket.
__global__ void __launch_bounds__(256) k_gate_count_lds(uint64_t gates, uint32_t k, uint32_t per_block,
                                                        const uint8_t* __restrict__ gate_type, const uint32_t* __restrict__ left,
                                                        const uint32_t* __restrict__ right, uint32_t* __restrict__ hist,
                                                        uint32_t* __restrict__ bad) {
    extern __shared__ uint32_t s_hist[];
    const uint32_t n = 1u << k, nb2 = 2u * n;
    for (uint32_t i = threadIdx.x; i < nb2; i += blockDim.x) s_hist[i] = 0u;
    __syncthreads();
    const uint64_t begin = (uint64_t)blockIdx.x * per_block, end = begin + per_block < gates ? begin + per_block : gates;
    for (uint64_t g = begin + threadIdx.x; g < end; g += blockDim.x) {
        const uint32_t l = left[g], r = right[g];
        if (l >= n || r >= n || gate_type[g] > 1) {
            atomicOr(bad, 1u);
            continue;
        }
        atomicAdd(&s_hist[l], 1u);
        atomicAdd(&s_hist[n + r], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb2; i += blockDim.x) hist[(size_t)i * gridDim.x + blockIdx.x] = s_hist[i];
}

__global__ void __launch_bounds__(256) k_gate_bucket_bounds(const uint32_t* __restrict__ starts, uint32_t nb2, uint32_t blocks,
                                                            const uint32_t* __restrict__ hist, uint32_t* __restrict__ offsets,
                                                            uint32_t* __restrict__ cursor) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb2) return;
    offsets[b] = starts[(size_t)b * blocks];
    const size_t last = (size_t)b * blocks + blocks - 1;
    cursor[b] = starts[last] + hist[last];   // one past the bucket's last gate
}

__global__ void __launch_bounds__(256) k_gate_fill_lds(uint64_t gates, uint32_t k, uint32_t per_block,
                                                       const uint8_t* __restrict__ gate_type, const uint32_t* __restrict__ left,
                                                       const uint32_t* __restrict__ right, const uint32_t* __restrict__ starts,
                                                       uint32_t* __restrict__ list, uint32_t* __restrict__ meta) {
    extern __shared__ uint32_t s_cur[];
    const uint32_t n = 1u << k, nb2 = 2u * n;
    for (uint32_t i = threadIdx.x; i < nb2; i += blockDim.x) s_cur[i] = starts[(size_t)i * gridDim.x + blockIdx.x];
    __syncthreads();
    const uint64_t begin = (uint64_t)blockIdx.x * per_block, end = begin + per_block < gates ? begin + per_block : gates;
    for (uint64_t g = begin + threadIdx.x; g < end; g += blockDim.x) {
        const uint32_t l = left[g], r = right[g];
        if (l >= n || r >= n || gate_type[g] > 1) continue;
        const uint32_t t = (uint32_t)gate_type[g] << 31, pl = atomicAdd(&s_cur[l], 1u), pr = atomicAdd(&s_cur[n + r], 1u);
        list[pl] = (uint32_t)g;
        meta[pl] = r | t;
        list[pr] = (uint32_t)g;
        meta[pr] = l | t;
    }
}

// U[b] = sum over the gates with left operand b of E[g] * (add ? 1 : W[right]),  V[b] = sum over its add gates of
// E[g] * W[right];  E[g] = e_hi[g >> kl] * e_lo[g & mask] (e_lo and W in Montgomery form, so products are canonical).
// grid = (2^k buckets, batch); any block size that is a multiple of 64.
// The gate arrays may hold a contiguous SHARD of the layer (one rank's gates when a layer is split across GPUs):
// list entries index the shard's arrays, and gate_base + entry is the gate's index in the layer, which is what E
// depends on.
__global__ void __launch_bounds__(256) k_gate_uv(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ cursor,
                                                 const uint32_t* __restrict__ list, const uint32_t* __restrict__ meta,
                                                 const Fr* __restrict__ e_hi,
                                                 const Fr* __restrict__ e_lo_mont, uint32_t kl, uint32_t kh,
                                                 const Fr* __restrict__ W, Fr* __restrict__ U, Fr* __restrict__ V, uint32_t wstride,
                                                 uint32_t gate_base, const GateSet* __restrict__ sets) {
    __shared__ Acc<9> smem[4 * 2];
    const uint32_t b = blockIdx.x, lmask = (1u << kl) - 1u;
    if (sets) {   // this proof's circuit (block-uniform: scalar loads)
        const GateSet gs = sets[blockIdx.y];
        const ptrdiff_t moff = meta - list;
        offsets = gs.offsets;
        cursor = gs.cursor;
        list = gs.list;
        meta = gs.list + moff;
    }
    e_hi += (size_t)blockIdx.y << kh;
    e_lo_mont += (size_t)blockIdx.y << kl;
    W += (size_t)blockIdx.y * wstride;
    // E[g] is a reduced product (it is an operand); E[g] W[right] only ever enters a sum, so those products are added
    // as full 512-bit integers and reduced once per thread: 128 + 64 multiply-adds per gate instead of 2 x 128
    Acc<9> acc[2] = {acc_zero<9>(), acc_zero<9>()};
    Lazy17 lu = lazy_zero(), lv = lazy_zero();
    for (uint32_t i = offsets[b] + threadIdx.x; i < cursor[b]; i += blockDim.x) {
        // (the list entry carries the other operand and the type: gathering right[g] and gate_type[g] by gate index
        // cost two random DRAM transactions per gate, which -- not the arithmetic -- bounded this kernel)
        const uint32_t gg = list[i] + gate_base, mt = meta[i];
        const Fr e = mont_mul(load_fr(e_hi + (gg >> kl)), load_fr(e_lo_mont + (gg & lmask)));
        const Fr w = load_fr(W + (mt & 0x7fffffffu));
        // (the accumulator is chosen per lane without a branch: a divergent one ran the 64 multiply-adds twice)
        const bool mult = (mt >> 31) != 0u;
        lazy_mac_sel(lu, lv, mult, e, w);
        if (!mult) acc_add_fr(acc[0], e);
    }
    acc_add_fr(acc[0], lazy_reduce(lu));
    acc_add_fr(acc[1], lazy_reduce(lv));
    block_sum<9, 2>(acc, smem);
    if (threadIdx.x == 0) {
        store_fr(U + (size_t)blockIdx.y * wstride + b, acc_reduce(acc[0]));
        store_fr(V + (size_t)blockIdx.y * wstride + b, acc_reduce(acc[1]));
    }
}

// a_u[c] = sum over the add gates with right operand c of E[g] * eq(u, left[g]),  m_u[c] the same over its mult gates
// (eq in Montgomery form).  grid = (2^k buckets, batch).
__global__ void __launch_bounds__(256) k_gate_rows(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ cursor,
                                                   const uint32_t* __restrict__ list, const uint32_t* __restrict__ meta,
                                                   const Fr* __restrict__ e_hi,
                                                   const Fr* __restrict__ e_lo_mont, uint32_t kl, uint32_t kh,
                                                   const Fr* __restrict__ eq_mont, Fr* __restrict__ A_row, Fr* __restrict__ M_row,
                                                   uint32_t k, uint32_t wstride, uint32_t gate_base, const GateSet* __restrict__ sets) {
    __shared__ Acc<9> smem[4 * 2];
    const uint32_t c = blockIdx.x, lmask = (1u << kl) - 1u, bucket = (1u << k) + c;
    if (sets) {
        const GateSet gs = sets[blockIdx.y];
        const ptrdiff_t moff = meta - list;
        offsets = gs.offsets;
        cursor = gs.cursor;
        list = gs.list;
        meta = gs.list + moff;
    }
    e_hi += (size_t)blockIdx.y << kh;
    e_lo_mont += (size_t)blockIdx.y << kl;
    eq_mont += (size_t)blockIdx.y * wstride;
    Acc<9> acc[2] = {acc_zero<9>(), acc_zero<9>()};
    Lazy17 la = lazy_zero(), lm = lazy_zero();   // unreduced sums of E[g] eq(u, left[g]), as in k_gate_uv
    for (uint32_t i = offsets[bucket] + threadIdx.x; i < cursor[bucket]; i += blockDim.x) {
        const uint32_t gg = list[i] + gate_base, mt = meta[i];
        const Fr e = mont_mul(load_fr(e_hi + (gg >> kl)), load_fr(e_lo_mont + (gg & lmask)));
        const Fr q = load_fr(eq_mont + (mt & 0x7fffffffu));
        lazy_mac_sel(lm, la, (mt >> 31) != 0u, e, q);
    }
    acc_add_fr(acc[0], lazy_reduce(la));
    acc_add_fr(acc[1], lazy_reduce(lm));
    block_sum<9, 2>(acc, smem);
    if (threadIdx.x == 0) {
        store_fr(A_row + (size_t)blockIdx.y * wstride + c, acc_reduce(acc[0]));
        store_fr(M_row + (size_t)blockIdx.y * wstride + c, acc_reduce(acc[1]));
    }
}

// ---------------------------------------------------------------------------
// Segments of the sorted gate lists (gate_seg.h).
// The block-private sort leaves bucket b's gates as blocks-many runs in ascending block order, run j = the gates of
// sort block j, and `starts[b * blocks + j]` is where it begins: `m` consecutive sort blocks make one segment.
// ---------------------------------------------------------------------------
// items per segment; segment s = bucket2 * runs + run over both halves (bucket2 < 2 * 2^k); cnt has nseg + 1 entries
__global__ void __launch_bounds__(256) k_seg_count(const uint32_t* __restrict__ starts, const uint32_t* __restrict__ hist, uint32_t blocks,
                                                   uint32_t m, uint32_t runs, uint32_t nseg, uint32_t* __restrict__ cnt) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > nseg) return;
    if (s == nseg) {
        cnt[s] = 0;
        return;
    }
    const uint32_t b2 = s / runs, run = s - b2 * runs;
    const uint32_t first = run * m, last = (first + m < blocks ? first + m : blocks) - 1;
    const size_t row = (size_t)b2 * blocks;
    const uint32_t len = starts[row + last] + hist[row + last] - starts[row + first];
    cnt[s] = (len + kSegCap - 1) / kSegCap;
}
__global__ void __launch_bounds__(256) k_seg_fill(const uint32_t* __restrict__ starts, const uint32_t* __restrict__ hist, uint32_t blocks,
                                                  uint32_t m, uint32_t runs, uint32_t nseg, const uint32_t* __restrict__ off,
                                                  uint2* __restrict__ items, uint32_t* __restrict__ bucket_begin) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > nseg) return;
    const uint32_t b2 = s / runs, run = s - b2 * runs;
    if (run == 0) bucket_begin[b2] = off[s];   // (s == nseg: b2 = 2 * 2^k, the total)
    if (s == nseg) return;
    const uint32_t first = run * m, last = (first + m < blocks ? first + m : blocks) - 1;
    const size_t row = (size_t)b2 * blocks;
    const uint32_t begin = starts[row + first], len = starts[row + last] + hist[row + last] - begin;
    uint32_t o = off[s];
    for (uint32_t done = 0; done < len; done += kSegCap, ++o)
        items[o] = make_uint2(begin + done, (len - done < kSegCap ? len - done : kSegCap) | (run << 8));
}
// counting sort of the items by (half, decreasing length): block-private histograms in LDS, bin-major global layout,
// one scan, block-private cursors (the scheme of k_gate_count_lds / k_gate_fill_lds with 2 * kSegCap bins)
constexpr uint32_t kSegBins = 2 * kSegCap;
__device__ __forceinline__ uint32_t seg_bin(uint32_t i, uint32_t half1_begin, uint32_t len) {
    return (i >= half1_begin ? kSegCap : 0u) + (kSegCap - len);   // len in 1 .. kSegCap
}
__global__ void __launch_bounds__(256) k_seg_len_hist(const uint2* __restrict__ items, const uint32_t* __restrict__ bucket_begin, uint32_t nb,
                                                      uint32_t per_block, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_h[kSegBins];
    if (threadIdx.x < kSegBins) s_h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t half1 = bucket_begin[nb], total = bucket_begin[2 * nb];
    const uint32_t begin = blockIdx.x * per_block, end = begin + per_block < total ? begin + per_block : total;
    for (uint32_t i = begin + threadIdx.x; i < end; i += blockDim.x) atomicAdd(&s_h[seg_bin(i, half1, items[i].y & 0xffu)], 1u);
    __syncthreads();
    if (threadIdx.x < kSegBins) hist[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s_h[threadIdx.x];
}
__global__ void __launch_bounds__(256) k_seg_len_fill(const uint2* __restrict__ items, const uint32_t* __restrict__ bucket_begin, uint32_t nb,
                                                      uint32_t per_block, const uint32_t* __restrict__ starts, uint32_t* __restrict__ order) {
    __shared__ uint32_t s_c[kSegBins];
    if (threadIdx.x < kSegBins) s_c[threadIdx.x] = starts[(size_t)threadIdx.x * gridDim.x + blockIdx.x];
    __syncthreads();
    const uint32_t half1 = bucket_begin[nb], total = bucket_begin[2 * nb];
    const uint32_t begin = blockIdx.x * per_block, end = begin + per_block < total ? begin + per_block : total;
    for (uint32_t i = begin + threadIdx.x; i < end; i += blockDim.x) order[atomicAdd(&s_c[seg_bin(i, half1, items[i].y & 0xffu)], 1u)] = i;
}

// Groups: 64 consecutive items of a half's length-sorted order = one wave's work.  group_len = the gates of its first
// (longest) item; slot = half * groups + w.
__global__ void __launch_bounds__(256) k_seg_group_len(const uint2* __restrict__ items, const uint32_t* __restrict__ order,
                                                       const uint32_t* __restrict__ bucket_begin, uint32_t nb, uint32_t groups,
                                                       uint32_t* __restrict__ group_len) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot > 2 * groups) return;
    if (slot == 2 * groups) {
        group_len[slot] = 0;
        return;
    }
    const uint32_t half = slot >= groups ? 1u : 0u, w = slot - half * groups;
    const uint32_t begin = bucket_begin[half ? nb : 0], end = bucket_begin[half ? 2 * nb : nb];
    const uint32_t pos = begin + 64u * w;
    group_len[slot] = pos < end ? (items[order[pos]].y & 0xffu) : 0u;
}
// one wave per group: the entries of its items, step-major (see GateSegs::packed).  An item's entries are contiguous in
// list / meta, so the wave reads item after item with its lanes (one or two cache lines per load instead of 64 per
// step), transposes in LDS and writes whole 256-byte steps.  grid = ceil(2 * groups / 4), block = 256
__global__ void __launch_bounds__(256) k_seg_pack(const uint2* __restrict__ items, const uint32_t* __restrict__ order,
                                                  const uint32_t* __restrict__ bucket_begin, uint32_t nb, uint32_t groups,
                                                  const uint32_t* __restrict__ group_len, const uint32_t* __restrict__ group_off,
                                                  const uint32_t* __restrict__ list, const uint32_t* __restrict__ meta, uint32_t gate_base,
                                                  uint32_t shift, uint32_t* __restrict__ packed) {
    __shared__ uint32_t tile[4][kSegCap][65];   // [wave][step][lane], rows padded: the column writes hit 32 banks
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t slot = blockIdx.x * 4u + wave;
    if (slot >= 2 * groups) return;
    const uint32_t half = slot >= groups ? 1u : 0u, w = slot - half * groups;
    const uint32_t n = group_len[slot];
    if (!n) return;
    const uint32_t begin = bucket_begin[half ? nb : 0], end = bucket_begin[half ? 2 * nb : nb];
    const uint32_t pos = begin + 64u * w + lane;
    uint32_t first = 0, len = 0;
    if (pos < end) {
        const uint2 it = items[order[pos]];
        first = it.x;
        len = it.y & 0xffu;
    }
    const uint32_t lmask = (1u << shift) - 1u;
    for (uint32_t j = 0; j < n; ++j) tile[wave][j][lane] = 0u;
    // inside an item the add gates come first (any order gives the same sums): the pass then changes the accumulator
    // its products go to once per item instead of selecting it per gate
    const uint64_t below = ((uint64_t)1 << lane) - 1u;
    for (uint32_t q = 0; q < 64u; ++q) {
        const uint32_t fq = __shfl(first, (int)q, 64), lq = __shfl(len, (int)q, 64);
        const bool have = lane < lq;
        const uint32_t mt = have ? meta[fq + lane] : 0u;
        const bool is_add = have && !(mt >> 31);
        const uint64_t valid = __ballot(have), addm = __ballot(is_add);
        if (have) {
            const uint32_t at = is_add ? (uint32_t)__popcll(addm & below) : (uint32_t)(__popcll(addm) + __popcll(valid & ~addm & below));
            tile[wave][at][q] = ((list[fq + lane] + gate_base) & lmask) | ((mt & 0x7fffffffu) << shift) | (mt & 0x80000000u);
        }
    }
    uint32_t* dst = packed + (size_t)group_off[slot] * 64u + lane;
    for (uint32_t j = 0; j < n; ++j) dst[(size_t)j * 64u] = tile[wave][j][lane];
}

// One item per lane, longest items first (the lanes of a wave run the same number of steps): the unreduced sums of
// E_lo[g & mask] * T[other operand] over the item's gates (gate_seg.h, seg_gate), left as 256-bit representatives for
// k_seg_combine.  ROWS = false: the left-operand buckets with T = W (U, V); true: the right-operand buckets with
// T = eq(u, .) (the row).  Per gate a wave streams 8 bytes (one coalesced load per step), gathers E_lo from L2 and
// T from LDS (LDS_T: the 2^k-entry table copied into the block's LDS once; the blocks are resident and take groups
// until none is left) -- the passes' first form gathered three operands per gate from L2 and was bound by those
// cache-line transfers, not by its arithmetic.  The operands of step j + 1 are loaded before step j's products.
// grid = (resident blocks, batch), block = 1024 (LDS_T) / 256
template <bool ROWS, bool LDS_T>
__global__ void __launch_bounds__(LDS_T ? 1024 : 256) k_seg_pass(const uint2* __restrict__ items, const uint32_t* __restrict__ order,
                                                                  const uint32_t* __restrict__ bucket_begin, uint32_t nb, uint32_t groups,
                                                                  const uint32_t* __restrict__ group_len, const uint32_t* __restrict__ group_off,
                                                                  const uint32_t* __restrict__ packed, const Fr* __restrict__ e_lo_mont, uint32_t shift,
                                                                  const Fr* __restrict__ T, uint32_t tlen, uint32_t tstride, Fr* __restrict__ X,
                                                                  Fr* __restrict__ Y, uint32_t pstride) {
    extern __shared__ uint4 s_raw[];
    e_lo_mont += (size_t)blockIdx.y << shift;
    T += (size_t)blockIdx.y * tstride;
    const Fr* Tt = T;
    if (LDS_T) {
        const uint4* src = reinterpret_cast<const uint4*>(T);
        for (uint32_t i = threadIdx.x; i < 2 * tlen; i += 4 * blockDim.x) {   // four loads in flight per thread
            uint4 v0 = src[i], v1, v2, v3;
            const bool h1 = i + blockDim.x < 2 * tlen, h2 = i + 2 * blockDim.x < 2 * tlen, h3 = i + 3 * blockDim.x < 2 * tlen;
            if (h1) v1 = src[i + blockDim.x];
            if (h2) v2 = src[i + 2 * blockDim.x];
            if (h3) v3 = src[i + 3 * blockDim.x];
            s_raw[i] = v0;
            if (h1) s_raw[i + blockDim.x] = v1;
            if (h2) s_raw[i + 2 * blockDim.x] = v2;
            if (h3) s_raw[i + 3 * blockDim.x] = v3;
        }
        __syncthreads();
        Tt = reinterpret_cast<const Fr*>(s_raw);
    }
    const uint32_t half = ROWS ? 1u : 0u;
    const uint32_t begin = bucket_begin[half ? nb : 0], end = bucket_begin[half ? 2 * nb : nb];
    const uint32_t lane = threadIdx.x & 63u, waves = blockDim.x >> 6, lmask = (1u << shift) - 1u;
    const uint32_t ngroups = (end - begin + 63u) / 64u;
    for (uint32_t w = blockIdx.x * waves + (threadIdx.x >> 6); w < ngroups; w += gridDim.x * waves) {
        const uint32_t slot = half * groups + w, n = group_len[slot];
        const uint32_t pos = begin + 64u * w + lane;
        uint32_t idx = 0, len = 0;
        if (pos < end) {
            idx = order[pos];
            len = items[idx].y & 0xffu;
        }
        const uint32_t* src = packed + (size_t)group_off[slot] * 64u + lane;
        // R0: the accumulator this lane's products go to -- its add gates' sum first, its mult gates' sum after the one
        // exchange at the item's first mult gate (k_seg_pack puts an item's add gates first); R1: the other one
        Lazy17 R0 = lazy_zero(), R1 = lazy_zero();
        bool sw = false;
        // Entries are requested four steps ahead and E_lo one step ahead: a step's two dependent loads (the entry from
        // HBM, then the gather it addresses) would otherwise cost ~2 us per step with nothing to hide behind.
        const uint32_t last = n - 1u, omask = (1u << (31u - shift)) - 1u;
        uint32_t ent = src[0];
        uint32_t p1 = src[(size_t)(1u < last ? 1u : last) * 64u], p2 = src[(size_t)(2u < last ? 2u : last) * 64u],
                 p3 = src[(size_t)(3u < last ? 3u : last) * 64u];
        Fr e = load_fr(e_lo_mont + (ent & lmask)), t;
        if (!LDS_T) t = load_fr(Tt + ((ent >> shift) & omask));
        for (uint32_t j = 0; j < n; ++j) {
            const uint32_t p4 = src[(size_t)(j + 4u < last ? j + 4u : last) * 64u];
            const uint32_t ent_n = p1;
            const Fr e_n = load_fr(e_lo_mont + (ent_n & lmask));   // in flight while this step's products run
            Fr t_n;
            if (LDS_T)
                t = load_fr(Tt + ((ent >> shift) & omask));   // LDS: no need to look ahead
            else
                t_n = load_fr(Tt + ((ent_n >> shift) & omask));
            // seg_gate_ordered (gate_seg.h), with the exchange skipped by the whole wave when no lane is at its first mult gate
            const bool live = j < len, go = live && (ent >> 31) != 0u && !sw;
            if (__any(go)) {
                if (go) {
#pragma unroll
                    for (int c = 0; c < 17; ++c) asm("v_swap_b32 %0, %1" : "+v"(R0.l[c]), "+v"(R1.l[c]));
                }
                sw = sw || go;
            }
            if (live) {
                lazy_mac_v(R0, e, t);
                // (U, V: an add gate's term without a second factor joins the mult gates' sum, which is R1 until the exchange)
                if (!ROWS && !sw) lazy_add_hi(R1, e, true);
            }
            ent = ent_n;
            e = e_n;
            if (!LDS_T) t = t_n;
            p1 = p2;
            p2 = p3;
            p3 = p4;
        }
        if (pos < end) {
            const size_t o = (size_t)blockIdx.y * pstride + (idx - begin);
            // ROWS: add gates -> X, mult gates -> Y;  U, V: mult gates (+ the add gates' plain terms) -> X, add gates -> Y
            Lazy17 L0, L1;
            seg_item_sums<ROWS>(R0, R1, sw, L0, L1);
            store_fr(X + o, lazy_reduce_partial32(L0));
            store_fr(Y + o, lazy_reduce_partial32(L1));
        }
    }
}
// out0[bucket] = sum over the bucket's items of E_hi[run] * X[item], out1 likewise with Y: half a wave (32 lanes) per
// bucket -- the two unreduced sums are added up across the lanes as they are (17-limb adds over DPP shuffles) and reduced
// once, so the fixed cost of a wave (the reductions) is shared by two buckets.  grid = (2^k / 2, batch), block = 64
__global__ void __launch_bounds__(64) k_seg_combine(const uint2* __restrict__ items, const uint32_t* __restrict__ bucket_begin, uint32_t b_first,
                                                    uint32_t nb, const Fr* __restrict__ X, const Fr* __restrict__ Y, uint32_t pstride,
                                                    const Fr* __restrict__ e_hi, uint32_t kh, uint32_t run_base, Fr* __restrict__ out0,
                                                    Fr* __restrict__ out1, uint32_t wstride, CPhaseFuse fuse) {
    const uint32_t lane = threadIdx.x & 31u, bucket = 2u * blockIdx.x + (threadIdx.x >> 5);
    const bool live = bucket < nb;
    const uint32_t half_begin = bucket_begin[b_first], lo = live ? bucket_begin[b_first + bucket] : 0u, hi = live ? bucket_begin[b_first + bucket + 1] : 0u;
    e_hi += ((size_t)blockIdx.y << kh) + run_base;
    X += (size_t)blockIdx.y * pstride;
    Y += (size_t)blockIdx.y * pstride;
    Lazy17 A = lazy_zero(), B = lazy_zero();
    for (uint32_t i = lo + lane; i < hi; i += 32) {
        const Fr eh = load_fr(e_hi + (items[i].y >> 8));
        lazy_mac_v(A, load_fr(X + (i - half_begin)), eh);
        lazy_mac_v(B, load_fr(Y + (i - half_begin)), eh);
    }
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
        Lazy17 oa, ob;
#pragma unroll
        for (int j = 0; j < 17; ++j) {
            oa.l[j] = __shfl_xor(A.l[j], off, 64);
            ob.l[j] = __shfl_xor(B.l[j], off, 64);
        }
        lazy_add(A, oa);
        lazy_add(B, ob);
    }
    const Fr r0 = lazy_reduce(A), r1 = lazy_reduce(B);
    if (live && lane == 0) {
        store_fr(out0 + (size_t)blockIdx.y * wstride + bucket, r0);
        store_fr(out1 + (size_t)blockIdx.y * wstride + bucket, r1);
    }
    if (fuse.Wb) {
        // the rows are a_u, m_u: the c-phase's tables X = a_u + W(u) m_u, Y = W(u) a_u straight from here (k_prod_c_setup's
        // arithmetic; one launch less between the row pass and the c-phase's first product pass)
        const size_t base = (size_t)blockIdx.y * wstride;
        Fr wu = fr_zero();
        if (threadIdx.x < (1u << fuse.jp)) wu = mont_mul(load_fr(fuse.Wb + base + threadIdx.x), load_fr(fuse.weights + (size_t)blockIdx.y * 8 + threadIdx.x));
#pragma unroll
        for (int off = 4; off >= 1; off >>= 1) {
            Fr o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.l[j] = __shfl_down(wu.l[j], off, 64);
            wu = fr_add(wu, o);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) wu.l[j] = __shfl(wu.l[j], 0, 64);
        const Fr x = fr_add(r0, mont_mul(r1, wu)), y = mont_mul(r0, wu);
        if (live && lane == 0) {
            store_fr(fuse.X + base + bucket, x);
            store_fr(fuse.Y + base + bucket, y);
        }
    }
}

// widened cell -> canonical Fr.  limb sums < 2^32 * 2^32; value < 2^32 r.
__global__ void k_predicate_normalise(const unsigned long long* __restrict__ wide, Fr* __restrict__ out, size_t cells) {
    for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < cells; c += (size_t)gridDim.x * blockDim.x) {
        Acc<10> a = acc_zero<10>();
        unsigned long long carry = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned long long w = wide[c * 8 + j];
            unsigned long long lo = (w & 0xffffffffull) + (carry & 0xffffffffull);
            a.l[j] = (uint32_t)lo;
            carry = (w >> 32) + (carry >> 32) + (lo >> 32);
        }
        a.l[8] = (uint32_t)carry;
        a.l[9] = (uint32_t)(carry >> 32);
        store_fr(out + c, acc_reduce(a));
    }
}

// ---------------------------------------------------------------------------
// host-callable launchers
// ---------------------------------------------------------------------------

void launch_fill_table(Fr* table, size_t count, uint64_t seed, hipStream_t s) {
    hipLaunchKernelGGL(k_fill_table, dim3(blocks_for(count, 4096)), dim3(256), 0, s, table, count, seed);
}
void launch_fill_shard(Fr* shard_table, size_t count, uint32_t lp, uint32_t shard, uint64_t seed, hipStream_t s) {
    hipLaunchKernelGGL(k_fill_shard, dim3(blocks_for(count, 4096)), dim3(256), 0, s, shard_table, count, lp, shard, seed);
}

// ---- what the box itself gives (bench.py quotes these beside its roofline fractions) -----------------
// one 16-byte element per thread, no loop: the copy shape that measured fastest on this chip (tools/ubench_copy.hip)
__global__ void __launch_bounds__(256) k_ubench_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_ubench_read(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, uint32_t per_block) {
    const size_t base = (size_t)blockIdx.x * per_block;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t t = threadIdx.x; t < per_block; t += 256) {
        const size_t i = base + t;
        if (i < n) {
            const uint4 v = in[i];
            acc.x ^= v.x, acc.y ^= v.y, acc.z ^= v.z, acc.w ^= v.w;
        }
    }
    if (acc.x == 0x12345u && acc.y == 0x777u && acc.z == 0x1u) out[0] = acc;
}
// a chain of dependent 254-bit Montgomery products per lane (128 v_mad_u64_u32 + 128 v_addc each): with the chip full
// of waves this is the arithmetic ceiling of every kernel whose work is modular products (tools/ubench_f64mont.hip)
__global__ void __launch_bounds__(64) k_ubench_modmul(Fr* io, uint32_t mask, int reps) {
    const uint32_t g = blockIdx.x * 64 + threadIdx.x;
    Fr x = load_fr(io + (g & mask)), y = load_fr(io + ((g + 1u) & mask));
    for (int r = 0; r < reps; ++r) x = mont_mul(x, y);
    if (fr_is_zero(x)) store_fr(io + (g & mask), x);   // (never: keeps the chain alive)
}
void launch_ubench_copy(const void* in, void* out, size_t bytes, hipStream_t s) {
    const size_t n = bytes / 16;
    hipLaunchKernelGGL(k_ubench_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, static_cast<const uint4*>(in), static_cast<uint4*>(out), n);
}
void launch_ubench_read(const void* in, void* out, size_t bytes, hipStream_t s) {
    const size_t n = bytes / 16;
    hipLaunchKernelGGL(k_ubench_read, dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, s, static_cast<const uint4*>(in), static_cast<uint4*>(out), n, 4096u);
}
void launch_ubench_modmul(Fr* io, uint32_t entries_pow2, uint32_t waves, int reps, hipStream_t s) {
    hipLaunchKernelGGL(k_ubench_modmul, dim3(waves), dim3(64), 0, s, io, entries_pow2 - 1u, reps);
}

uint32_t mle_blocks_per_table(uint32_t items, uint32_t batch) {
    // ~1024 items (4 iterations) per block, but enough blocks across the batch to fill the chip;
    // never more than one block per 256 items
    const uint32_t per_block = opt(OPT_items_per_block) >= 256 ? (uint32_t)opt(OPT_items_per_block) : 1024u;
    uint32_t b = (items + per_block - 1) / per_block;
    const uint32_t fill = (2048 + batch - 1) / batch;
    if (b < fill) b = fill;
    const uint32_t most = (items + 255) / 256;
    if (b > most) b = most;
    if (b > kMaxBlocksPerTable) b = kMaxBlocksPerTable;
    if (b < 1) b = 1;
    return b;
}

void launch_mle_sum_first(const Fr* tables, size_t stride, uint32_t h, uint32_t batch, uint32_t nblk,
                          MlePartial* partials, hipStream_t s) {
    hipLaunchKernelGGL(k_mle_sum_first, dim3(nblk, batch), dim3(256), 0, s, tables, stride, h, partials);
}

void launch_mle_fold_sum(const Fr* src, size_t src_stride, Fr* dst, size_t dst_stride, uint32_t q, uint32_t batch,
                         uint32_t nblk, const FixedMul* rtab, uint32_t r_stride, MlePartial* partials, hipStream_t s) {
    hipLaunchKernelGGL(k_mle_fold_sum, dim3(nblk, batch), dim3(256), 0, s, src, src_stride, dst, dst_stride, q, rtab,
                       r_stride, partials);
}

void launch_mle_round_hash(const MlePartial* partials, uint32_t nblk, uint32_t round, uint32_t n, uint32_t batch,
                           const Fr* cts, Fr* out_coeffs, uint32_t* out_len, Fr* out_r, FixedMul* rtab,
                           uint32_t* dep_last, hipStream_t s) {
    hipLaunchKernelGGL(k_mle_round_hash, dim3(batch), dim3(64), 0, s, partials, nblk, round, n, cts, out_coeffs,
                       out_len, out_r, rtab, dep_last);
}

void launch_layer_eval(uint32_t gates, const uint8_t* gate_type, const uint32_t* left, const uint32_t* right,
                       const Fr* prev, Fr* out, uint32_t batch, uint32_t prev_stride, hipStream_t s, const GateSet* sets) {
    hipLaunchKernelGGL(k_layer_eval, dim3(blocks_for(gates, 4096), batch), dim3(256), 0, s, gates, gate_type, left, right,
                       prev, out, prev_stride, sets);
}

// a wide layer's LATER passes: the pending fold (the previous pass's three variables bound, 2^20 and more entries of three
// tables read for it) on the matrix cores as the plain sumcheck's fold passes are (mfma_fold.h) -- the weights are constants
// of the proof -- in place, then the cross sums of the folded tables as a pass with nothing pending.  Three launches for
// one, where the one spends its time on 8 x 3 products of 254-bit numbers per folded entry.
// grid = (batch), block = 256
__global__ void __launch_bounds__(256) k_prod_fold_plan(const Fr* __restrict__ weights, MfmaFoldPlan* __restrict__ plans) {
    __shared__ __attribute__((aligned(16))) unsigned char digits[32 * 32 * 8];
    mfma_plan_block<3>(weights + (size_t)blockIdx.x * 8, plans + blockIdx.x, digits);
}
// grid = (S / 256, 3 * batch), block = 256: a 64-entry tile per wave (the chunk length the fold passes measured best at)
__global__ void __launch_bounds__(256) k_prod_fold_mfma(Fr* Wt, Fr* Xt, Fr* Yt, uint32_t S, const MfmaFoldPlan* __restrict__ plans, uint32_t wstride) {
    __shared__ __attribute__((aligned(16))) unsigned char digits[32 * 32 * 8];
    const uint32_t table = blockIdx.y % 3u, proof = blockIdx.y / 3u;
    Fr* T = (table == 0u ? Wt : table == 1u ? Xt : Yt) + (size_t)proof * wstride;
    const uint32_t chunk = S / gridDim.x, begin = blockIdx.x * chunk;
    Acc<9> unused = acc_zero<9>();
    // (in place: an output entry is stored by the wave that read the eight entries it is made of, after it read them)
    mfma_multifold_block<3>(T, T, S, plans + proof, begin, begin + chunk, blockIdx.x * 5u + blockIdx.y * 3u, unused, digits);
}
size_t prod_fold_plan_bytes() { return sizeof(MfmaFoldPlan); }

void launch_prod_pass(Fr* W, Fr* X, Fr* Y, uint32_t m_in, uint32_t jp, const Fr* weights, uint32_t J, Fr* partials, uint32_t wstride,
                      ProdPassRec* rec, uint32_t ticket, uint32_t batch, hipStream_t s, uint32_t* arrivals, void* fold_plans, Fr* tail,
                      uint32_t tail_stride) {
    const bool mfma = !opt(OPT_no_mfma_cross);
    // (the threshold counts a lockstep group's proofs in: k_prod_cross<8> over 64 blocks x 7 proofs takes 43 us where one proof's
    // takes 26, the fold kernels a few us either way)
    uint32_t log_batch = 0;
    while ((2u << log_batch) <= batch) ++log_batch;
    const uint32_t fold_min = opt(OPT_prod_fold_min_log2) > 0 ? (uint32_t)opt(OPT_prod_fold_min_log2) : kProdFoldMinM;
    if (mfma && fold_plans && jp == 3u && m_in + log_batch >= fold_min && m_in >= 14u) {
        MfmaFoldPlan* plans = static_cast<MfmaFoldPlan*>(fold_plans);
        const uint32_t Sf = 1u << (m_in - 3u);
        hipLaunchKernelGGL(k_prod_fold_plan, dim3(batch), dim3(256), 0, s, weights, plans);
        hipLaunchKernelGGL(k_prod_fold_mfma, dim3(Sf / 256u, 3u * batch), dim3(256), 0, s, W, X, Y, Sf, plans, wstride);
        m_in -= 3u;
        jp = 0u;
    }
    const uint32_t S = 1u << (m_in - jp - J);
    const uint32_t cross_min = opt(OPT_prod_cross_min_log2) > 0 ? (uint32_t)opt(OPT_prod_cross_min_log2) : kCrossMinM;
    if (mfma && jp == 0u && J == 3u && m_in >= cross_min && m_in >= 13u) {
        // a wide layer's first pass of a phase (or a later one, folded above): the 64 cross sums as int8 matrix products (mfma_cross.h)
        const uint32_t kc = cross_pass_kc(S, batch), nblk = S / kc;
        // (publishing from the last block to arrive, as the small passes do, was measured and is slower here: its 37 dependent
        // additions per value over partials in other XCDs' memory cost more than the launch they save)
        if (kc == 2048u)
            hipLaunchKernelGGL(k_prod_cross_mfma<2048>, dim3(nblk, batch), dim3(512), 0, s, W, X, Y, m_in, partials, wstride);
        else if (kc == 1024u)
            hipLaunchKernelGGL(k_prod_cross_mfma<1024>, dim3(nblk, batch), dim3(512), 0, s, W, X, Y, m_in, partials, wstride);
        else if (kc == 512u)
            hipLaunchKernelGGL(k_prod_cross_mfma<512>, dim3(nblk, batch), dim3(512), 0, s, W, X, Y, m_in, partials, wstride);
        else if (kc == 256u)
            hipLaunchKernelGGL(k_prod_cross_mfma<256>, dim3(nblk, batch), dim3(512), 0, s, W, X, Y, m_in, partials, wstride);
        else
            hipLaunchKernelGGL(k_prod_cross_mfma<128>, dim3(nblk, batch), dim3(512), 0, s, W, X, Y, m_in, partials, wstride);
        hipLaunchKernelGGL(k_prod_publish, dim3(batch), dim3((nblk > 32 ? 14 : 4) * kProdRecValues), 0, s, partials, nblk, rec, ticket);
        return;
    }
    const uint32_t blocks = prod_pass_blocks(S);
    uint32_t* fused = (arrivals && blocks > 1 && blocks <= kProdFuseBlocks) ? arrivals : nullptr;   // the last block publishes
    if (prod_pass_tile(S) == kProdTileWide)
        hipLaunchKernelGGL(k_prod_cross<kProdTileWide>, dim3(blocks, batch), dim3(256), 0, s, W, X, Y, m_in, jp, weights, J, partials, wstride, rec, ticket,
                           fused, tail, tail_stride);
    else
        hipLaunchKernelGGL(k_prod_cross<kProdTile>, dim3(blocks, batch), dim3(256), 0, s, W, X, Y, m_in, jp, weights, J, partials, wstride, rec, ticket,
                           fused, tail, tail_stride);
    if (fused) return;
    if (blocks > 1024) {
        // (the second level's input sits behind the partials: launch_prod_pass's callers size the scratch with prod_pass_scratch_values)
        Fr* level2 = partials + (size_t)batch * blocks * kProdRecValues;
        hipLaunchKernelGGL(k_prod_reduce, dim3(kProdReduceBlocks, batch), dim3(4 * kProdRecValues), 0, s, partials, blocks, level2);
        hipLaunchKernelGGL(k_prod_publish, dim3(batch), dim3(4 * kProdRecValues), 0, s, level2, kProdReduceBlocks, rec, ticket);
    } else if (blocks > 1) {
        // (fourteen threads per value from 32 blocks on: with four a thread's chain of additions was up to 64 long)
        hipLaunchKernelGGL(k_prod_publish, dim3(batch), dim3((blocks > 32 ? 14 : 4) * kProdRecValues), 0, s, partials, blocks, rec, ticket);
    }
}

void launch_prod_c_setup(const Fr* Wb, uint32_t jp, const Fr* weights, const Fr* A, const Fr* M, Fr* X, Fr* Y, uint32_t k, uint32_t wstride,
                         uint32_t batch, hipStream_t s) {
    hipLaunchKernelGGL(k_prod_c_setup, dim3(blocks_for(1u << k, 1024), batch), dim3(256), 0, s, Wb, jp, weights, A, M, X, Y, k, wstride);
}

// ---- sum over ranks on the device (gkr_exchange_dev): field elements <-> eight 32-bit limbs in int64, the form an
// integer SUM all-reduce adds exactly (RCCL has no modular sum).  Element 2 * each is the "some rank failed" flag.
__global__ void __launch_bounds__(256) k_exchange_widen(const Fr* __restrict__ a, const Fr* __restrict__ b, uint32_t each,
                                                        const uint32_t* __restrict__ flag, uint32_t local_flag, long long* __restrict__ limbs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * each) {
        const Fr v = load_fr(i < each ? a + i : b + (i - each));
#pragma unroll
        for (int j = 0; j < 8; ++j) limbs[(size_t)i * 8 + j] = (long long)v.l[j];
    } else if (i == 2 * each) {
        limbs[(size_t)i * 8] = (long long)(((flag && *flag) || local_flag) ? 1 : 0);
#pragma unroll
        for (int j = 1; j < 8; ++j) limbs[(size_t)i * 8 + j] = 0;
    }
}
// limb sums of < 2^31 addends -> canonical values mod r; the summed flag goes to a word the host can read (pinned)
__global__ void __launch_bounds__(256) k_exchange_narrow(const long long* __restrict__ limbs, Fr* __restrict__ a, Fr* __restrict__ b, uint32_t each,
                                                         uint32_t* __restrict__ flag_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * each) {
        Acc<10> acc = acc_zero<10>();
        unsigned long long carry = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned long long w = (unsigned long long)limbs[(size_t)i * 8 + j];
            const unsigned long long lo = (w & 0xffffffffull) + (carry & 0xffffffffull);
            acc.l[j] = (uint32_t)lo;
            carry = (w >> 32) + (carry >> 32) + (lo >> 32);
        }
        acc.l[8] = (uint32_t)carry;
        acc.l[9] = (uint32_t)(carry >> 32);
        store_fr(i < each ? a + i : b + (i - each), acc_reduce(acc));
    } else if (i == 2 * each && flag_out) {
        *flag_out = limbs[(size_t)i * 8] ? 1u : 0u;
    }
}
void launch_exchange_widen(const Fr* a, const Fr* b, uint32_t each, const uint32_t* flag, uint32_t local_flag, long long* limbs, hipStream_t s) {
    hipLaunchKernelGGL(k_exchange_widen, dim3((2 * each + 1 + 255) / 256), dim3(256), 0, s, a, b, each, flag, local_flag, limbs);
}
void launch_exchange_narrow(const long long* limbs, Fr* a, Fr* b, uint32_t each, uint32_t* flag_out, hipStream_t s) {
    hipLaunchKernelGGL(k_exchange_narrow, dim3((2 * each + 1 + 255) / 256), dim3(256), 0, s, limbs, a, b, each, flag_out);
}

void launch_copy_words(const void* src, void* dst, size_t words, hipStream_t s) {
    if (!words) return;
    hipLaunchKernelGGL(k_copy_words, dim3(blocks_for(words / 4 + 1, 1024)), dim3(256), 0, s, static_cast<const uint32_t*>(src),
                       static_cast<uint32_t*>(dst), words);
}

void launch_copy_rows(const void* src, size_t src_stride_words, void* dst, size_t dst_stride_words, uint32_t words, uint32_t rows, hipStream_t s) {
    if (!words || !rows) return;
    hipLaunchKernelGGL(k_copy_rows, dim3(blocks_for((size_t)words * rows, 1024)), dim3(256), 0, s, static_cast<const uint32_t*>(src), src_stride_words,
                       static_cast<uint32_t*>(dst), dst_stride_words, words, rows);
}

void launch_eq_table(const Fr* points, uint32_t stride, uint32_t first, uint32_t nvars, Fr* out, bool montgomery, uint32_t batch,
                     hipStream_t s, const CPhaseFuse* wu_job, Fr* wu_out, uint32_t wstride) {
    const Fr* wb = wu_job && wu_out ? wu_job->Wb : nullptr;
    const Fr* ww = wu_job && wu_out ? wu_job->weights : nullptr;
    const uint32_t jp = wu_job && wu_out ? wu_job->jp : 0u;
    Fr* wo = wu_job && wu_out ? wu_out : nullptr;
    if (nvars >= 14) {   // many variables: a few products per block, two per entry (k_eq_table_split)
        hipLaunchKernelGGL(k_eq_table_split, dim3(1u << (nvars - 12u), batch), dim3(1024), 0, s, points, stride, first, nvars, out, montgomery ? 1u : 0u,
                           wb, ww, jp, wstride, wo);
        return;
    }
    hipLaunchKernelGGL(k_eq_table, dim3(blocks_for((size_t)4 << nvars, 4096), batch), dim3(256), 0, s, points, stride, first, nvars, out,
                       montgomery ? 1u : 0u, wb, ww, jp, wstride, wo);
}

void launch_layer_prologue(const Fr* points, uint32_t k_i, uint32_t kh, uint32_t kl, Fr* e_hi, Fr* e_lo, const Fr* W, Fr* Wb, Fr* Wc,
                           uint32_t k, uint32_t* dep, uint32_t* host_dep, uint32_t batch, hipStream_t s, uint32_t* wide_bits) {
    const uint32_t nb_hi = blocks_for((size_t)4 << kh, 4096), nb_lo = blocks_for((size_t)4 << kl, 4096);
    const uint32_t nb_mont = Wb ? blocks_for((size_t)1 << k, 1024) : 0u;
    hipLaunchKernelGGL(k_layer_prologue, dim3(nb_hi + nb_lo + nb_mont + 1, batch), dim3(256), 0, s, points, k_i, kh, kl, e_hi, e_lo, W, Wb,
                       Wc, k, dep, host_dep, nb_hi, nb_lo, nb_mont, wide_bits);
}

void launch_line_restriction(const Fr* W, uint32_t k, const Fr* bc, Fr* scratch, uint32_t* deg_scratch, Fr* bcm, Fr* out, uint32_t* out_len,
                             uint32_t batch, hipStream_t s, LinePart part) {
    const bool stepwise = opt(OPT_line_stepwise) != 0;   // (test hook: the wide-layer form at every width)
    if (k <= 9 && !stepwise) {
        hipLaunchKernelGGL(k_line_restriction, dim3(batch), dim3(256), (size_t)3 * sizeof(Fr) << k, s, W, k, bc, scratch, out, out_len);
        return;
    }
    // (the set-up -- copy, Moebius transform, largest monomial degree -- in one block per proof up to 2^12 values, over a
    // grid beyond: kernels_wide.hip)
    if (k > 12) {
        // wide layers: the set-up over a grid (kernels_wide.hip), one launch per variable while the table is large, the last
        // six variables, q and its length in one block per proof -- 4 + (k - 6) + 1 launches instead of 2 k + 6 (the proving
        // thread issues them between two layers' sumchecks)
        // (the set-up needs W only, the rest the line: a caller that has W before it has the line -- the last layer of a proof,
        // whose restriction nothing else overlaps -- issues the two parts apart)
        if (part != LinePart::finish) launch_line_setup_wide(W, k, scratch, deg_scratch, batch, s);
        if (part == LinePart::prepare) return;
        hipLaunchKernelGGL(k_line_coeffs, dim3(batch), dim3(64), 0, s, k, bc, bcm);
        const uint32_t j0 = k - 6u;
        for (uint32_t j = 0; j < j0; ++j) {
            const uint32_t items = (1u << (k - j - 1u)) * (j + 2u);
            hipLaunchKernelGGL(k_line_step, dim3(blocks_for(items, 4096), batch), dim3(256), 0, s, k, j, bcm, scratch);
        }
        hipLaunchKernelGGL(k_line_tail, dim3(batch), dim3(256), 0, s, k, j0, bcm, scratch, out, deg_scratch, out_len);
        return;
    }
    hipLaunchKernelGGL(k_line_coeffs, dim3(batch), dim3(64), 0, s, k, bc, bcm);
    hipLaunchKernelGGL(k_line_init, dim3(batch), dim3(256), 0, s, W, k, scratch, out_len);
    for (uint32_t j = 0; j < k; ++j) {
        const uint32_t items = (1u << (k - j - 1u)) * (j + 2u);
        hipLaunchKernelGGL(k_line_step, dim3(blocks_for(items, 4096), batch), dim3(256), 0, s, k, j, bcm, scratch);
    }
    hipLaunchKernelGGL(k_line_out, dim3(batch), dim3(64), 0, s, k, scratch, out);
}

void launch_to_mont(const Fr* in, Fr* out, uint32_t count, hipStream_t s) {
    hipLaunchKernelGGL(k_to_mont, dim3(blocks_for(count, 1024)), dim3(256), 0, s, in, out, count);
}

void launch_to_mont_strided(const Fr* in, Fr* out, uint32_t count, uint32_t stride, uint32_t offset, hipStream_t s) {
    hipLaunchKernelGGL(k_to_mont_strided, dim3(blocks_for(count, 1024)), dim3(256), 0, s, in, out, count, stride, offset);
}

void launch_tables_differ(const Fr* a, const Fr* b, size_t count, uint32_t* flag, hipStream_t s) {
    hipLaunchKernelGGL(k_tables_differ, dim3(blocks_for(count, 2048)), dim3(256), 0, s, a, b, count, flag);
}

void launch_fold_pair(const Fr* src, Fr* dst, const FixedMul* rtab, hipStream_t s) {
    hipLaunchKernelGGL(k_fold_pair, dim3(1), dim3(64), 0, s, src, dst, rtab);
}

void launch_depends(const Fr* W, uint32_t k, uint32_t* dep, uint32_t batch, hipStream_t s) {
    hipLaunchKernelGGL(k_depends, dim3(blocks_for(1u << k, 1024), batch), dim3(256), 0, s, W, k, dep);
}

void launch_predicate_scatter(uint32_t k_i, uint32_t k_next, const uint8_t* gate_type, const uint32_t* left,
                              const uint32_t* right, const Fr* e_hi, const Fr* e_lo_mont, uint32_t kl,
                              unsigned long long* wideA, unsigned long long* wideM, uint32_t* bad, uint32_t log_p,
                              uint32_t shard, hipStream_t s) {
    hipLaunchKernelGGL(k_predicate_scatter, dim3(blocks_for(1ull << k_i, 4096)), dim3(256), 0, s, k_i, k_next,
                       gate_type, left, right, e_hi, e_lo_mont, kl, wideA, wideM, bad, log_p, shard);
}

// counting-sort predicate build; counts/offsets/cursor: 2 * ncells u32 each, block_sums: ceil(2 ncells / 2048),
// list: 2^k_i u32.  All scratch is caller-provided.
void launch_predicate_sorted(uint32_t k_i, uint32_t k_next, const uint8_t* gate_type, const uint32_t* left,
                             const uint32_t* right, const Fr* e_hi, const Fr* e_lo_mont, uint32_t kl, uint32_t log_p,
                             uint32_t shard, size_t ncells, uint32_t* counts, uint32_t* offsets, uint32_t* cursor,
                             uint32_t* block_sums, uint32_t* list, uint32_t* bad, Fr* out_A, Fr* out_M, uint32_t batch,
                             hipStream_t s) {
    const size_t n = 2 * ncells;
    const uint32_t gblocks = blocks_for(1ull << k_i, 4096);
    const uint32_t sblocks = (uint32_t)((n + kScanPerBlock - 1) / kScanPerBlock);
    hipLaunchKernelGGL(k_pred_count, dim3(gblocks), dim3(256), 0, s, k_i, k_next, gate_type, left, right, log_p, shard, ncells,
                       counts, bad);
    hipLaunchKernelGGL(k_scan_blocks, dim3(sblocks), dim3(256), 0, s, counts, offsets, block_sums, n);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, block_sums, sblocks);
    hipLaunchKernelGGL(k_scan_add, dim3(sblocks), dim3(256), 0, s, offsets, cursor, block_sums, n);
    hipLaunchKernelGGL(k_pred_fill, dim3(gblocks), dim3(256), 0, s, k_i, k_next, gate_type, left, right, log_p, shard, ncells,
                       cursor, list, bad);
    hipLaunchKernelGGL(k_pred_sum, dim3(blocks_for(n, 8192), batch), dim3(256), 0, s, ncells, offsets, cursor, list, e_hi,
                       e_lo_mont, kl, k_i - kl, out_A, out_M);
}

// gate lists by left and by right operand: counts / offsets / cursor 2 * 2^k u32 each (counts zeroed by the caller),
// block_sums ceil(2 * 2^k / 2048) + 1, list 2 * 2^k_i u32
// blocks of the LDS-privatised sort (0: the layer is small or k too large -- use the global-atomic passes), and the
// scratch it needs: two arrays of 2 * 2^k * blocks counters (histograms, their scan) + the scan's block sums
uint32_t gate_lists_lds_blocks(uint64_t gates, uint32_t k) {
    const bool off = opt(OPT_gate_sort_global) != 0;
    if (off || k > 12 || gates < ((uint64_t)1 << 16)) return 0;
    uint64_t b = (gates + 16383) / 16384;   // >= 16384 gates per block: the 32 KB histogram flush must stay small beside them
    if (b > 1024) b = 1024;
    return (uint32_t)b;
}
size_t gate_lists_lds_scratch_words(uint64_t gates, uint32_t k) {
    const uint32_t blocks = gate_lists_lds_blocks(gates, k);
    if (!blocks) return 0;
    const size_t n = ((size_t)2 << k) * blocks;
    return 2 * n + (n + kScanPerBlock - 1) / kScanPerBlock + 1;
}

// ---- segments: where they apply and what they need ---------------------------------------------------------------------
static uint32_t log2_exact(uint64_t x) {   // log2 of a power of two, else 64
    if (!x || (x & (x - 1))) return 64;
    uint32_t l = 0;
    while ((x >> l) != 1) ++l;
    return l;
}
struct SegPlan {
    uint32_t shift = 0, runs = 0, run_base = 0, m = 0, blocks = 0, nseg = 0, bound = 0, half_bound = 0, len_blocks = 0, groups = 0;
    uint64_t packed_half = 0;
};
static SegPlan seg_plan(GateSpan span, uint32_t k_i, uint32_t k) {
    SegPlan p;
    const bool off = opt(OPT_gate_segments_off) != 0;
    // (measured, MI355X, gate passes of a layer with k = k_i / 2, ms per pass segment / bucket form: 2^16 gates 0.089 / 0.024,
    // 2^20 0.090 / 0.052, 2^22 0.104 / 0.125, 2^24 0.245 / 0.43 -- the segment form has ~85 us of fixed latency (resident
    // blocks, the one-wave-per-bucket combine); profiles/r03/q_segment_threshold.jsonl)
    const uint32_t min_log2 = (uint32_t)opt(OPT_gate_segments_min_log2);
    const uint32_t mean_log2 = opt(OPT_gate_segment_log2) >= 2 && opt(OPT_gate_segment_log2) <= 8 ? (uint32_t)opt(OPT_gate_segment_log2) : kSegMeanLog2;
    const uint32_t blocks = gate_lists_lds_blocks(span.count, k);
    if (off || !blocks || span.count < ((uint64_t)1 << min_log2) || span.count > ((uint64_t)1 << 31)) return p;
    const uint64_t per_block = (span.count + blocks - 1) / blocks;
    const uint32_t pb = log2_exact(per_block);
    if (pb == 64 || span.count % per_block) return p;              // blocks of a power of two of gates, all full
    uint32_t shift = k + mean_log2 > pb ? k + mean_log2 : pb;       // segments of 2^mean_log2 gates on average
    if (shift > k_i) shift = k_i;
    if (shift < pb || shift > 20 || shift + k > 31 || (span.base & (((uint64_t)1 << shift) - 1))) return p;   // E_lo: at most 2^20 entries; a packed entry is 32 bits; aligned span
    p.shift = shift;
    p.m = 1u << (shift - pb);
    p.blocks = blocks;
    p.runs = (blocks + p.m - 1) / p.m;
    p.run_base = (uint32_t)(span.base >> shift);
    p.nseg = (2u << k) * p.runs;
    const uint64_t half = ((uint64_t)p.runs << k) + span.count / kSegCap + 1;
    if (2 * half > ((uint64_t)1 << 31) || p.runs >= (1u << 24)) return SegPlan();
    p.half_bound = (uint32_t)half;
    p.bound = (uint32_t)(2 * half);
    p.len_blocks = (p.bound + 16383) / 16384;
    if (p.len_blocks > 1024) p.len_blocks = 1024;
    p.groups = p.half_bound / 64 + 1;
    // a half's groups, each padded to its longest item: sorted by length, the padding telescopes to < 64 * kSegCap
    // entries, plus a last partial group
    p.packed_half = span.count + 2 * 64 * (uint64_t)kSegCap + 64;
    return p;
}
uint32_t gate_seg_shift(GateSpan span, uint32_t k_i, uint32_t k) {
    const SegPlan p = seg_plan(span, k_i, k);
    return p.shift ? p.shift : k_i / 2;
}
size_t gate_segs_words(GateSpan span, uint32_t k_i, uint32_t k) {
    const SegPlan p = seg_plan(span, k_i, k);
    if (!p.shift) return 0;
    return 3 * (size_t)p.bound + ((size_t)2 << k) + 2 + 2 * (size_t)p.groups + (2 * (size_t)p.groups + 2) + 2 * (size_t)p.packed_half;
}
size_t gate_segs_scratch_words(GateSpan span, uint32_t k_i, uint32_t k) {
    const SegPlan p = seg_plan(span, k_i, k);
    if (!p.shift) return 0;
    const size_t a = (size_t)p.nseg + 1, b = (size_t)kSegBins * p.len_blocks, c = 2 * (size_t)p.groups + 1;
    return 2 * a + (a + kScanPerBlock - 1) / kScanPerBlock + 1 + 2 * b + (b + kScanPerBlock - 1) / kScanPerBlock + 1 +
           (c + kScanPerBlock - 1) / kScanPerBlock + 1;
}
size_t gate_seg_partial_elems(GateSpan span, uint32_t k_i, uint32_t k) {
    const SegPlan p = seg_plan(span, k_i, k);
    return p.shift ? 2 * (size_t)p.half_bound : 0;
}

void launch_gate_lists(GateSpan span, uint32_t k_i, uint32_t k, const uint8_t* gate_type, const uint32_t* left, const uint32_t* right,
                       uint32_t* counts, uint32_t* offsets, uint32_t* cursor, uint32_t* block_sums, uint32_t* list, uint32_t* bad,
                       uint32_t* lds_scratch, GateSegs* segs, uint32_t* seg_scratch, hipStream_t s) {
    const uint64_t gates = span.count;
    const size_t n = (size_t)2 << k;
    const uint32_t lblocks = lds_scratch ? gate_lists_lds_blocks(gates, k) : 0;
    if (segs) {
        segs->shift = 0;
    }
    if (lblocks) {
        const size_t cells = n * lblocks;
        uint32_t *hist = lds_scratch, *starts = lds_scratch + cells, *sums = lds_scratch + 2 * cells;
        const uint32_t per_block = (uint32_t)((gates + lblocks - 1) / lblocks);
        const uint32_t sblocks = (uint32_t)((cells + kScanPerBlock - 1) / kScanPerBlock);
        const size_t lds = n * sizeof(uint32_t);
        hipLaunchKernelGGL(k_gate_count_lds, dim3(lblocks), dim3(256), lds, s, gates, k, per_block, gate_type, left, right, hist, bad);
        hipLaunchKernelGGL(k_scan_blocks, dim3(sblocks), dim3(256), 0, s, hist, starts, sums, cells);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, sums, sblocks);
        hipLaunchKernelGGL(k_scan_add, dim3(sblocks), dim3(256), 0, s, starts, (uint32_t*)nullptr, sums, cells);
        hipLaunchKernelGGL(k_gate_bucket_bounds, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, starts, (uint32_t)n, lblocks, hist, offsets,
                           cursor);
        hipLaunchKernelGGL(k_gate_fill_lds, dim3(lblocks), dim3(256), lds, s, gates, k, per_block, gate_type, left, right, starts, list,
                           list + gate_list_words(gates));
        const SegPlan p = seg_plan(span, k_i, k);
        if (segs && segs->words && seg_scratch && p.shift) {
            // the segments of both halves: count the items, scan, fill; then the items of each half by decreasing length
            const size_t a = (size_t)p.nseg + 1, b = (size_t)kSegBins * p.len_blocks;
            const uint32_t ablocks = (uint32_t)((a + kScanPerBlock - 1) / kScanPerBlock), bblocks = (uint32_t)((b + kScanPerBlock - 1) / kScanPerBlock);
            uint32_t *cnt = seg_scratch, *off = cnt + a, *asums = off + a, *lh = asums + ablocks + 1, *ls = lh + b, *bsums = ls + b,
                     *csums = bsums + bblocks + 1;
            segs->shift = p.shift;
            segs->runs = p.runs;
            segs->run_base = p.run_base;
            segs->bound = p.bound;
            segs->half_bound = p.half_bound;
            segs->groups = p.groups;
            segs->nb = 1u << k;
            segs->packed_half = p.packed_half;
            const dim3 sg((unsigned)((a + 255) / 256));
            hipLaunchKernelGGL(k_seg_count, sg, dim3(256), 0, s, starts, hist, lblocks, p.m, p.runs, p.nseg, cnt);
            hipLaunchKernelGGL(k_scan_blocks, dim3(ablocks), dim3(256), 0, s, cnt, off, asums, a);
            hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, asums, ablocks);
            hipLaunchKernelGGL(k_scan_add, dim3(ablocks), dim3(256), 0, s, off, (uint32_t*)nullptr, asums, a);
            hipLaunchKernelGGL(k_seg_fill, sg, dim3(256), 0, s, starts, hist, lblocks, p.m, p.runs, p.nseg, off, segs->items(), segs->bucket_begin());
            const uint32_t per = (p.bound + p.len_blocks - 1) / p.len_blocks;
            hipLaunchKernelGGL(k_seg_len_hist, dim3(p.len_blocks), dim3(256), 0, s, segs->items(), segs->bucket_begin(), 1u << k, per, lh);
            hipLaunchKernelGGL(k_scan_blocks, dim3(bblocks), dim3(256), 0, s, lh, ls, bsums, b);
            hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, bsums, bblocks);
            hipLaunchKernelGGL(k_scan_add, dim3(bblocks), dim3(256), 0, s, ls, (uint32_t*)nullptr, bsums, b);
            hipLaunchKernelGGL(k_seg_len_fill, dim3(p.len_blocks), dim3(256), 0, s, segs->items(), segs->bucket_begin(), 1u << k, per, ls, segs->order());
            // the groups of 64 items, where each begins in the step-major copy of the entries, and that copy
            const size_t c = 2 * (size_t)p.groups + 1;
            const uint32_t cblocks = (uint32_t)((c + kScanPerBlock - 1) / kScanPerBlock);
            hipLaunchKernelGGL(k_seg_group_len, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, s, segs->items(), segs->order(), segs->bucket_begin(),
                               1u << k, p.groups, segs->group_len());
            hipLaunchKernelGGL(k_scan_blocks, dim3(cblocks), dim3(256), 0, s, segs->group_len(), segs->group_off(), csums, c);
            hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, csums, cblocks);
            hipLaunchKernelGGL(k_scan_add, dim3(cblocks), dim3(256), 0, s, segs->group_off(), (uint32_t*)nullptr, csums, c);
            hipLaunchKernelGGL(k_seg_pack, dim3((2 * p.groups + 3) / 4), dim3(256), 0, s, segs->items(), segs->order(), segs->bucket_begin(), 1u << k, p.groups,
                               segs->group_len(), segs->group_off(), list, list + gate_list_words(gates), (uint32_t)span.base, p.shift,
                               segs->packed());
        }
        return;
    }
    const uint32_t gblocks = blocks_for(gates, 4096);
    const uint32_t sblocks = (uint32_t)((n + kScanPerBlock - 1) / kScanPerBlock);
    hipLaunchKernelGGL(k_gate_count, dim3(gblocks), dim3(256), 0, s, gates, k, gate_type, left, right, counts, bad);
    hipLaunchKernelGGL(k_scan_blocks, dim3(sblocks), dim3(256), 0, s, counts, offsets, block_sums, n);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, block_sums, sblocks);
    hipLaunchKernelGGL(k_scan_add, dim3(sblocks), dim3(256), 0, s, offsets, cursor, block_sums, n);
    hipLaunchKernelGGL(k_gate_fill, dim3(gblocks), dim3(256), 0, s, gates, k, gate_type, left, right, cursor, list, list + gate_list_words(gates));
}

// compute units of the CURRENT device (a process may drive several: gkr_ctx_create_multi), read once per device id
static uint32_t device_cus() {
    static std::atomic<uint32_t> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256u;
    std::atomic<uint32_t>& slot = cus[dev & 63];
    uint32_t n = slot.load(std::memory_order_relaxed);
    if (!n) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 1) ? (uint32_t)prop.multiProcessorCount : 256u;
        slot.store(n, std::memory_order_relaxed);
    }
    return n;
}
template <bool ROWS, bool LDS_T>
static void launch_seg_pass_t(const GateSegs& g, uint32_t k, const Fr* e_lo_mont, const Fr* T, Fr* X, Fr* Y, LayerBatch lb, hipStream_t s) {
    const uint32_t tlen = 1u << k;
    const size_t lds = LDS_T ? (size_t)tlen * sizeof(Fr) : 0;
    // the attribute is per device (a process may drive several: gkr_ctx_create_multi): one bit per device id, set once each
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (LDS_T && !((attr_set.load(std::memory_order_relaxed) >> (dev & 63)) & 1u)) {
        // (a device whose blocks cannot take the table -- not gfx950 -- refuses here: the gather form runs instead)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seg_pass<ROWS, LDS_T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            launch_seg_pass_t<ROWS, false>(g, k, e_lo_mont, T, X, Y, lb, s);
            return;
        }
        attr_set.fetch_or((uint64_t)1 << (dev & 63), std::memory_order_relaxed);
    }
    // resident blocks: one 1024-thread block per CU with the table in LDS; 256-thread blocks, four per CU, without
    const uint32_t blocks = LDS_T ? device_cus() : 4 * device_cus();
    hipLaunchKernelGGL((k_seg_pass<ROWS, LDS_T>), dim3(blocks, lb.batch), dim3(LDS_T ? 1024 : 256), lds, s, g.items(), g.order(), g.bucket_begin(),
                       g.nb, g.groups, g.group_len(), g.group_off(), g.packed(), e_lo_mont, g.shift, T, tlen,
                       (uint32_t)lb.wstride, X, Y, g.half_bound);
}
static void launch_seg_pass(bool rows, const GateSegs& g, uint32_t k, const Fr* e_lo_mont, const Fr* T, Fr* X, Fr* Y, LayerBatch lb, hipStream_t s) {
    const bool no_lds = opt(OPT_gate_segments_no_lds) != 0;
    const bool lds_t = !no_lds && ((size_t)sizeof(Fr) << k) <= 128 * 1024;
    if (rows)
        lds_t ? launch_seg_pass_t<true, true>(g, k, e_lo_mont, T, X, Y, lb, s) : launch_seg_pass_t<true, false>(g, k, e_lo_mont, T, X, Y, lb, s);
    else
        lds_t ? launch_seg_pass_t<false, true>(g, k, e_lo_mont, T, X, Y, lb, s) : launch_seg_pass_t<false, false>(g, k, e_lo_mont, T, X, Y, lb, s);
}

void launch_exclusive_scan(const uint32_t* in, uint32_t* out, uint32_t* block_sums, size_t n, hipStream_t s) {
    const uint32_t sblocks = (uint32_t)((n + kScanPerBlock - 1) / kScanPerBlock);
    hipLaunchKernelGGL(k_scan_blocks, dim3(sblocks), dim3(256), 0, s, in, out, block_sums, n);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, block_sums, sblocks);
    hipLaunchKernelGGL(k_scan_add, dim3(sblocks), dim3(256), 0, s, out, (uint32_t*)nullptr, block_sums, n);
}

static uint32_t bucket_threads(uint64_t gates, uint32_t k) { return (gates >> k) > 64u ? 256u : 64u; }   // gates per bucket on average

void launch_gate_uv(GateSpan span, uint32_t k_i, uint32_t k, const uint32_t* offsets, const uint32_t* cursor, const uint32_t* list,
                    const Fr* e_hi, const Fr* e_lo_mont, uint32_t kl, const Fr* W, Fr* U, Fr* V, LayerBatch lb, const GateSegs* segs,
                    Fr* partials, hipStream_t s, const GateSet* sets) {
    if (segs && segs->shift && partials && !sets) {   // (the segment form takes one circuit per launch: callers keep layers that have segments out of groups)
        Fr *X = partials, *Y = partials + (size_t)segs->half_bound * lb.batch;
        launch_seg_pass(false, *segs, k, e_lo_mont, W, X, Y, lb, s);
        hipLaunchKernelGGL(k_seg_combine, dim3(((1u << k) + 1u) / 2u, lb.batch), dim3(64), 0, s, segs->items(), segs->bucket_begin(), 0u, 1u << k, X, Y,
                           segs->half_bound, e_hi, k_i - segs->shift, segs->run_base, U, V, (uint32_t)lb.wstride, CPhaseFuse{});
        return;
    }
    hipLaunchKernelGGL(k_gate_uv, dim3(1u << k, lb.batch), dim3(bucket_threads(span.count, k)), 0, s, offsets, cursor, list,
                       list + gate_list_words(span.count), e_hi, e_lo_mont, kl, k_i - kl, W, U, V, (uint32_t)lb.wstride, (uint32_t)span.base, sets);
}

bool launch_gate_rows(GateSpan span, uint32_t k_i, uint32_t k, const uint32_t* offsets, const uint32_t* cursor, const uint32_t* list,
                      const Fr* e_hi, const Fr* e_lo_mont, uint32_t kl, const Fr* eq_mont, Fr* A_row, Fr* M_row, LayerBatch lb,
                      const GateSegs* segs, Fr* partials, hipStream_t s, const CPhaseFuse* fuse, const GateSet* sets) {
    if (segs && segs->shift && partials && !sets) {
        Fr *X = partials, *Y = partials + (size_t)segs->half_bound * lb.batch;
        launch_seg_pass(true, *segs, k, e_lo_mont, eq_mont, X, Y, lb, s);
        hipLaunchKernelGGL(k_seg_combine, dim3(((1u << k) + 1u) / 2u, lb.batch), dim3(64), 0, s, segs->items(), segs->bucket_begin(), 1u << k, 1u << k, X, Y,
                           segs->half_bound, e_hi, k_i - segs->shift, segs->run_base, A_row, M_row, (uint32_t)lb.wstride, fuse ? *fuse : CPhaseFuse{});
        return fuse != nullptr;
    }
    hipLaunchKernelGGL(k_gate_rows, dim3(1u << k, lb.batch), dim3(bucket_threads(span.count, k)), 0, s, offsets, cursor, list,
                       list + gate_list_words(span.count), e_hi, e_lo_mont, kl, k_i - kl, eq_mont, A_row, M_row, k, (uint32_t)lb.wstride,
                       (uint32_t)span.base, sets);
    return false;
}

void launch_predicate_normalise(const unsigned long long* wide, Fr* out, size_t cells, hipStream_t s) {
    hipLaunchKernelGGL(k_predicate_normalise, dim3(blocks_for(cells, 4096)), dim3(256), 0, s, wide, out, cells);
}

void launch_mle_fold_sum_small(const Fr* src, size_t src_stride, Fr* dst, size_t dst_stride, uint32_t q, uint32_t batch,
                               const FixedMul* rtab, MleHostRec* host_rec, uint32_t ticket, hipStream_t s) {
    hipLaunchKernelGGL(k_mle_fold_sum_small, dim3(batch), dim3(256), 0, s, src, src_stride, dst, dst_stride, q, rtab,
                       host_rec, ticket);
}

// blocks per table for a pass whose per-table work is `items` entries split into 2^jout sub-blocks:
// a power of two, >= 2^jout, ~1024 items per block, enough blocks across the batch to fill the chip
uint32_t mle_pass_blocks(uint32_t items, uint32_t jout, uint32_t batch) {
    uint32_t want = mle_blocks_per_table(items, batch);
    uint32_t b = 1u << jout;
    while (b < want && b * 2 <= items && b * 2 <= kMaxBlocksPerTable) b <<= 1;
    if (b > items) b = items;   // tiny tables: one entry per block at the very least
    return b;
}

void launch_mle_sub_sums(const Fr* tables, size_t stride, uint32_t len, uint32_t batch, uint32_t nblk, MleSubPartial* partials,
                         hipStream_t s, const MlePublish* publish) {
    hipLaunchKernelGGL(k_mle_sub_sums, dim3(nblk, batch), dim3(256), 0, s, tables, stride, len, partials, publish ? *publish : MlePublish{});
}

void launch_mle_sub_reduce(const MleSubPartial* partials, uint32_t nblk, uint32_t jout, uint32_t batch, MleHostRecSub* host_rec,
                           uint32_t ticket, hipStream_t s) {
    hipLaunchKernelGGL(k_mle_sub_reduce, dim3(batch), dim3(512), 0, s, partials, nblk, jout, host_rec, ticket);
}

// matrix-core form unless the option no_mfma_fold is set or the chunk is not a whole number of 64-entry wave tiles
bool mle_multifold_uses_mfma(uint32_t S, uint32_t nblk) {
    const bool off = opt(OPT_no_mfma_fold) != 0;
    return !off && nblk > 0 && S % nblk == 0 && (S / nblk) % 64u == 0;
}

size_t mle_fold_plan_bytes() { return sizeof(MfmaFoldPlan); }

// the plans of `batch` sumchecks (batch * mle_fold_plan_bytes() of device memory) from their weights; only needed
// when mle_multifold_uses_mfma says so.  One thread per row of the digit matrix (32 * 2^jin rows: the kernel sits on a lone
// sumcheck's round path, where four rows per thread cost 14 us instead of 7)
void launch_mle_fold_plan(int jin, const Fr* weights, void* plans, uint32_t batch, hipStream_t s) {
    MfmaFoldPlan* pl = static_cast<MfmaFoldPlan*>(plans);
    if (jin == 1)
        hipLaunchKernelGGL(k_mle_fold_plan<1>, dim3(batch), dim3(64), 0, s, weights, pl);
    else if (jin == 2)
        hipLaunchKernelGGL(k_mle_fold_plan<2>, dim3(batch), dim3(128), 0, s, weights, pl);
    else if (jin == 3)
        hipLaunchKernelGGL(k_mle_fold_plan<3>, dim3(batch), dim3(256), 0, s, weights, pl);
    else if (jin == 4)
        hipLaunchKernelGGL(k_mle_fold_plan<4>, dim3(batch), dim3(512), 0, s, weights, pl);
    else
        hipLaunchKernelGGL(k_mle_fold_plan<5>, dim3(batch), dim3(1024), 0, s, weights, pl);
}

// plans: what launch_mle_fold_plan built for this pass (matrix-core form), unused otherwise
void launch_mle_multifold(int jin, const Fr* src, size_t src_stride, Fr* dst, size_t dst_stride, uint32_t S, uint32_t batch,
                          uint32_t nblk, const Fr* weights, const void* plans, MleSubPartial* partials, hipStream_t s,
                          const MlePublish* publish) {
    dim3 grid(nblk, batch);
    const MlePublish pub = publish ? *publish : MlePublish{};
    if (mle_multifold_uses_mfma(S, nblk)) {
        const MfmaFoldPlan* pl = static_cast<const MfmaFoldPlan*>(plans);
        if (jin == 1)
            hipLaunchKernelGGL(k_mle_multifold_mfma<1>, grid, dim3(256), 0, s, src, src_stride, dst, dst_stride, S, pl, partials, pub);
        else if (jin == 2)
            hipLaunchKernelGGL(k_mle_multifold_mfma<2>, grid, dim3(256), 0, s, src, src_stride, dst, dst_stride, S, pl, partials, pub);
        else if (jin == 3)
            hipLaunchKernelGGL(k_mle_multifold_mfma<3>, grid, dim3(256), 0, s, src, src_stride, dst, dst_stride, S, pl, partials, pub);
        else if (jin == 4)
            hipLaunchKernelGGL(k_mle_multifold_mfma<4>, grid, dim3(256), 0, s, src, src_stride, dst, dst_stride, S, pl, partials, pub);
        else
            hipLaunchKernelGGL(k_mle_multifold_mfma<5>, grid, dim3(256), 0, s, src, src_stride, dst, dst_stride, S, pl, partials, pub);
        return;
    }
    if (jin == 1)
        hipLaunchKernelGGL(k_mle_multifold<1>, grid, dim3(256), 0, s, src, src_stride, dst, dst_stride, S, weights, partials, pub);
    else if (jin == 2)
        hipLaunchKernelGGL(k_mle_multifold<2>, grid, dim3(256), 0, s, src, src_stride, dst, dst_stride, S, weights, partials, pub);
    else
        hipLaunchKernelGGL(k_mle_multifold<3>, grid, dim3(256), 0, s, src, src_stride, dst, dst_stride, S, weights, partials, pub);
}

// blocks per table for a multifold pass of S output entries split into 2^jout sub-blocks: a power of two >= 2^jout.
// The matrix-core kernel pays a per-block set-up (the sumcheck's digit matrix into LDS), which argues for few, long
// chunks -- but what decides its bandwidth is how many TABLES the resident blocks work on at once: every table is
// read as 2^J streams 32 * S bytes apart, and with 8 long chunks per table the ~512 resident blocks spread over 64
// tables = 2048 concurrent DRAM streams, whose bandwidth then depends on where the allocation's pages happen to sit
// (5.3 / 5.8 / 6.1+ TB/s "modes", fixed per allocation; profiles/r02/c_fold_blocks_sweep.txt).  With 128 short
// chunks per table (256 entries = one 64-entry tile per wave) the same blocks cover 4 tables, and every placement
// runs at the fast rate (6.1 - 6.5 TB/s in 13 of 13 allocations).  Hence: as many blocks as chunks of
// GKR_FOLD_MIN_CHUNK (default 256) entries allow, up to GKR_FOLD_BLOCKS (default 65536) over the batch.
uint32_t mle_multifold_blocks(uint32_t S, uint32_t jout, uint32_t batch) {
    const bool off = opt(OPT_no_mfma_fold) != 0;
    if (off) return mle_pass_blocks(S, jout, batch);
    const uint32_t target = opt(OPT_fold_blocks) > 0 ? (uint32_t)opt(OPT_fold_blocks) : 65536u;
    const uint32_t mc = (uint32_t)opt(OPT_fold_min_chunk);
    const uint32_t min_chunk = mc >= 64u && (mc & (mc - 1u)) == 0 ? mc : 256u;
    uint32_t b = 1u << jout;
    while ((uint64_t)b * batch < target && S / (2u * b) >= min_chunk && 2u * b <= kMaxBlocksPerTable) b <<= 1;
    if (b > S) b = S;
    return b;
}

void launch_mle_multifold_small(int jin, const Fr* src, size_t src_stride, Fr* dst, size_t dst_stride, uint32_t S, uint32_t jout,
                                uint32_t batch, const Fr* weights, MleHostRecSub* host_rec, uint32_t ticket, hipStream_t s, Fr* tail, uint32_t tail_stride) {
    dim3 grid(batch);
    if (jin == 0)
        hipLaunchKernelGGL(k_mle_sums_small, grid, dim3(256), 0, s, src, src_stride, S, jout, host_rec, ticket);
    else
        hipLaunchKernelGGL(k_mle_multifold_small, grid, dim3(1024), 0, s, src, src_stride, dst, dst_stride, S, (uint32_t)jin, jout, weights,
                           host_rec, ticket, tail, tail_stride);
}

void launch_mle_round_reduce(const MlePartial* partials, uint32_t nblk, uint32_t batch, MleHostRec* host_rec,
                             uint32_t ticket, hipStream_t s) {
    hipLaunchKernelGGL(k_mle_round_reduce, dim3(batch), dim3(64), 0, s, partials, nblk, host_rec, ticket);
}


}  // namespace gkr
