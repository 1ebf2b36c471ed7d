// CDNA4 (gfx950) kernels for WIDE layers of the GKR layer sumcheck: next-layer tables of 2^14 .. 2^28 values
// (prove_sumcheck_opt, rust/src/gkr/sumcheck.rs:36-156, accepts any GKRCircuit; the compiler pads a layer to whatever
// 2^k it needs, rust/src/convert.rs:209-214, and WIDTH_LIMIT = 20 merges all constraints of an R1CS into at most 20
// sub-circuits, convert.rs:10-11,171-186 -- a 10^5-constraint circuit has layers far beyond 2^14 values).
//
// The linear-time form of kernels.hip has no inherent width limit: U, V, the c-phase row and W are 2^k-entry tables in
// HBM.  What changes with the width is the SHAPE of the work:
//   * 2^k buckets of the gate lists with only a few gates each (a circom layer has about as many gates as the next
//     layer has values) -- a block per bucket would launch 2^20 blocks for one gate apiece.  Here the buckets are cut into
//     ITEMS of at most sixteen gates, sorted by length and summed a lane per item against a materialised eq(z, .) table
//     (k_items_pass); the few buckets far longer than the rest -- the constant wires every relay gate reads
//     (convert.rs:307-342) -- are many items whose partial sums a combine step adds (k_items_combine);
//   * tables the one-block helpers cannot walk: the dependence flags (k_depends_wide), the Moebius transform
//     (k_mobius_pass) and the set-up of the line restriction (k_line_copy, k_line_maxdeg) run over a grid.
// Same field elements as every other form: bit-exact (tests/test_gpu_wide_layers.py).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "dev_util.h"
#include "kernels.h"
#include "options.h"

namespace gkr {

// ---------------------------------------------------------------------------
// gate passes of wide layers: the buckets of the sorted gate lists cut into ITEMS, a lane per item
// ---------------------------------------------------------------------------
// What a gate pass sums (sumcheck.rs:50-63, 97-124 in the linear-time form): per bucket b of the left (U, V) resp. right
// (a_u, m_u) operand, sum over the bucket's gates of eq(z, g) * T[other operand(g)].  Round 4's form gave a bucket a group of
// lanes and counted 5560 (2^20 gates over 2^20 values) and 1144 (2^24 over 2^18) wave-instructions per 64 gates against the
// 128 of the product itself (profiles/r04/q_*): a REDUCED product per gate just to form eq(z, g) = E_hi * E_lo, ~1000
// instructions of per-lane fixed cost (two reductions, the group's shuffles, two canonical reductions) whether a lane walked
// one gate or sixteen, and lanes idling on the longest bucket of their wave.  This form:
//   * eq(z, .) is MATERIALISED per sumcheck as a 2^k_i-entry table (k_eq_table_split, tens of microseconds) and gathered: per
//     gate ONE unreduced 256 x 256-bit multiply-add;
//   * a bucket is cut into items of <= kItemMax gates, the items are sorted by length (a counting sort, built once per
//     circuit) and handled a lane per item, 64 items of (nearly) equal length per wave: every lane of a wave runs the same
//     number of steps; the items' entries are stored step-major (a wave's step = 512 contiguous bytes);
//   * a wave whose items hold ONE gate each -- most of a circom layer, where a value feeds one gate -- does one reduced
//     product per lane and stores: no accumulators, no reductions;
//   * an item that is its bucket writes the bucket's two outputs itself; buckets of several items (the constant wire half the
//     relay gates read, convert.rs:307-342: ONE bucket with half the layer) leave partial sums that a second, small kernel
//     adds per bucket (a lane per bucket; a wave per bucket beyond kLongItems items);
//   * buckets without gates are items of length zero (their outputs are zero): no memset of the output tables.
// The same field elements as every other form: integer adds commute, every sum is reduced mod r at the end -- bit-exact.
constexpr uint32_t kItemMax = 16;
constexpr uint32_t kLongItems = 64;
constexpr uint32_t kClasses = kItemMax + 1;   // item lengths 0 .. kItemMax
constexpr uint32_t kChunkItems = 1024;        // a long bucket's items are summed a wave per chunk of this many, then a wave per bucket

// the plan of ONE half (left-operand buckets / right-operand buckets) as offsets (in u32 words) from the half's base
struct PlanLayout {
    uint32_t nb;              // buckets per half
    uint32_t cap_items, cap_groups, cap_multi, cap_long, cap_chunks;
    size_t cap_packed;        // u64 entries
    size_t items_per, item_first, item_bucket, sorted, desc, group_len, group_off, scan_sums, multi, longb, long_chunk0, chunk_slot, packed, half_words;
};
// header of a half: [0] items, [1] groups, [2] buckets of 2 .. kLongItems items, [3] buckets of more ("long"), [4] groups of two
// and more steps (sorted by length, they come first), [5] chunks of kChunkItems items the long buckets are summed in,
// [8 + c] items of length c, [32 + c] the sort's cursors
constexpr uint32_t kPlanHdrWords = 64;
static PlanLayout plan_layout(uint64_t gates, uint32_t k) {
    PlanLayout L;
    L.nb = 1u << k;
    const uint64_t nb = L.nb;
    L.cap_items = (uint32_t)(gates / kItemMax + nb + 64);                 // sum of max(1, ceil(len / kItemMax)) over the buckets
    L.cap_groups = L.cap_items / 64 + 2;
    L.cap_multi = (uint32_t)(gates / kItemMax + 1);
    L.cap_long = (uint32_t)(gates / ((uint64_t)kItemMax * kLongItems) + 1);
    L.cap_chunks = L.cap_items / kChunkItems + L.cap_long + 1;
    L.cap_packed = (size_t)gates + 64u * 2u * kItemMax + 64u;              // padding only where the length changes inside a wave
    size_t w = kPlanHdrWords;
    auto take = [&](size_t n) { const size_t at = w; w += (n + 3) & ~(size_t)3; return at; };
    L.items_per = take(nb);
    L.item_first = take(nb + 1);
    L.item_bucket = take(L.cap_items);
    L.sorted = take(L.cap_items);
    L.desc = take(L.cap_items);
    L.group_len = take(L.cap_groups);
    L.group_off = take(L.cap_groups + 1);
    L.scan_sums = take((L.cap_items > nb ? L.cap_items : nb) / 2048 + 4);
    L.multi = take(L.cap_multi);
    L.longb = take(L.cap_long);
    L.long_chunk0 = take(L.cap_long);
    L.chunk_slot = take(L.cap_chunks);
    L.packed = take(2 * L.cap_packed);
    L.half_words = w;
    return L;
}
size_t gate_plan_words(uint64_t gates, uint32_t k) { return 2 * plan_layout(gates, k).half_words; }
// per proof, one half at a time: two sums per item, and behind them two per chunk of a long bucket
size_t gate_plan_partial_elems(uint64_t gates, uint32_t k) {
    const PlanLayout L = plan_layout(gates, k);
    return 2 * (size_t)L.cap_items + 2 * (size_t)L.cap_chunks;
}

// packed entry: gate index (28 bits) | other operand (24 bits) << 28 | gate type << 63; all ones = no gate in this step
constexpr unsigned long long kNoGate = ~0ull;

// one thread per bucket of the half: items of the bucket, and the buckets that need a combine step
__global__ void __launch_bounds__(256) k_plan_count(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ cursor, uint32_t nb,
                                                    uint32_t* __restrict__ hdr, uint32_t* __restrict__ items_per, uint32_t* __restrict__ multi,
                                                    uint32_t* __restrict__ longb, uint32_t* __restrict__ long_chunk0, uint32_t* __restrict__ chunk_slot) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const uint32_t len = cursor[b] - offsets[b];
    const uint32_t n = len ? (len + kItemMax - 1u) / kItemMax : 1u;   // (an empty bucket is one item of length zero)
    items_per[b] = n;
    if (n > kLongItems) {
        const uint32_t slot = atomicAdd(hdr + 3, 1u), nch = (n + kChunkItems - 1u) / kChunkItems, c0 = atomicAdd(hdr + 5, nch);
        longb[slot] = b;
        long_chunk0[slot] = c0;
        for (uint32_t c = 0; c < nch; ++c) chunk_slot[c0 + c] = slot;   // (few: a bucket of 2^24 gates has 1024 chunks)
    } else if (n > 1u) {
        multi[atomicAdd(hdr + 2, 1u)] = b;
    }
}
// one thread per item: its bucket (a search in the buckets' first items), its length class counted
__global__ void __launch_bounds__(256) k_plan_items(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ cursor, uint32_t nb,
                                                    uint32_t* __restrict__ hdr, const uint32_t* __restrict__ items_per,
                                                    const uint32_t* __restrict__ item_first, uint32_t* __restrict__ item_bucket) {
    __shared__ uint32_t s_hist[kClasses];
    if (threadIdx.x < kClasses) s_hist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t n_items = item_first[nb - 1u] + items_per[nb - 1u];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0u) {
        hdr[0] = n_items;
        hdr[1] = (n_items + 63u) / 64u;
    }
    if (i < n_items) {
        uint32_t lo = 0, hi = nb - 1u;   // the last bucket whose first item is <= i
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1u) >> 1;
            if (item_first[mid] <= i) lo = mid; else hi = mid - 1u;
        }
        item_bucket[i] = lo;
        const uint32_t len = cursor[lo] - offsets[lo], j = i - item_first[lo];
        const uint32_t ilen = len - kItemMax * j < kItemMax ? len - kItemMax * j : kItemMax;
        atomicAdd(&s_hist[ilen], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kClasses && s_hist[threadIdx.x]) atomicAdd(hdr + 8 + threadIdx.x, s_hist[threadIdx.x]);
}
// the counting sort by length, longest first: a block reserves a range per class, its items take it in turn
__global__ void __launch_bounds__(256) k_plan_sort(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ cursor,
                                                   uint32_t* __restrict__ hdr, const uint32_t* __restrict__ item_first,
                                                   const uint32_t* __restrict__ item_bucket, uint32_t* __restrict__ sorted) {
    __shared__ uint32_t s_count[kClasses], s_base[kClasses];
    if (threadIdx.x < kClasses) s_count[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t n_items = hdr[0], i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t ilen = 0, rank = 0;
    if (i < n_items) {
        const uint32_t b = item_bucket[i], len = cursor[b] - offsets[b], j = i - item_first[b];
        ilen = len - kItemMax * j < kItemMax ? len - kItemMax * j : kItemMax;
        rank = atomicAdd(&s_count[ilen], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kClasses) {
        uint32_t start = 0;   // classes in descending order of length
        for (uint32_t c = kItemMax; c > threadIdx.x; --c) start += hdr[8 + c];
        s_base[threadIdx.x] = start + (s_count[threadIdx.x] ? atomicAdd(hdr + 32 + threadIdx.x, s_count[threadIdx.x]) : 0u);
    }
    __syncthreads();
    if (i < n_items) sorted[s_base[ilen] + rank] = i;
}
// one thread per group of 64 sorted items: its steps = the length of its first (longest) item
__global__ void __launch_bounds__(256) k_plan_groups(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ cursor,
                                                     const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ item_first,
                                                     const uint32_t* __restrict__ item_bucket, const uint32_t* __restrict__ sorted,
                                                     uint32_t* __restrict__ group_len, uint32_t cap_groups, uint32_t* __restrict__ hdr_out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= cap_groups) return;
    uint32_t ilen = 0;
    if (g < hdr[1]) {
        const uint32_t i = sorted[64u * g], b = item_bucket[i], len = cursor[b] - offsets[b], j = i - item_first[b];
        ilen = len - kItemMax * j < kItemMax ? len - kItemMax * j : kItemMax;
    }
    group_len[g] = ilen;
    if (ilen >= 2u) atomicAdd(hdr_out + 4, 1u);
}
// a wave per group: the items' entries step-major, and what a lane needs to know of its item
//   desc = bucket | (the item is its whole bucket) << 31
__global__ void __launch_bounds__(256) k_plan_pack(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ cursor,
                                                   const uint32_t* __restrict__ list, const uint32_t* __restrict__ meta,
                                                   const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ items_per,
                                                   const uint32_t* __restrict__ item_first, const uint32_t* __restrict__ item_bucket,
                                                   const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ group_len,
                                                   const uint32_t* __restrict__ group_off, uint32_t* __restrict__ desc,
                                                   unsigned long long* __restrict__ packed) {
    const uint32_t lane = threadIdx.x & 63u, g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (g >= hdr[1]) return;
    const uint32_t pos = 64u * g + lane, steps = group_len[g];
    uint32_t ilen = 0, base = 0;
    if (pos < hdr[0]) {
        const uint32_t i = sorted[pos], b = item_bucket[i], len = cursor[b] - offsets[b], j = i - item_first[b];
        ilen = len - kItemMax * j < kItemMax ? len - kItemMax * j : kItemMax;
        base = offsets[b] + kItemMax * j;
        desc[pos] = b | (items_per[b] == 1u ? 0x80000000u : 0u);
    }
    unsigned long long* out = packed + (size_t)group_off[g] * 64u + lane;
    for (uint32_t t = 0; t < steps; ++t) {
        unsigned long long e = kNoGate;
        if (t < ilen) {
            const uint32_t mt = meta[base + t];
            e = (unsigned long long)list[base + t] | ((unsigned long long)(mt & 0x7fffffffu) << 28) | ((unsigned long long)(mt >> 31) << 63);
        }
        out[(size_t)t * 64u] = e;
    }
}

void launch_gate_plan(GateSpan span, uint32_t k, const uint32_t* offsets, const uint32_t* cursor, const uint32_t* list, uint32_t* plan, hipStream_t s) {
    const PlanLayout L = plan_layout(span.count, k);
    const uint32_t* meta = list + gate_list_words(span.count);
    for (uint32_t half = 0; half < 2; ++half) {
        uint32_t* P = plan + (size_t)half * L.half_words;
        const uint32_t *off = offsets + (size_t)half * L.nb, *cur = cursor + (size_t)half * L.nb;
        (void)hipMemsetAsync(P, 0, kPlanHdrWords * sizeof(uint32_t), s);
        hipLaunchKernelGGL(k_plan_count, dim3((L.nb + 255u) / 256u), dim3(256), 0, s, off, cur, L.nb, P, P + L.items_per, P + L.multi, P + L.longb, P + L.long_chunk0,
                           P + L.chunk_slot);
        launch_exclusive_scan(P + L.items_per, P + L.item_first, P + L.scan_sums, L.nb, s);
        hipLaunchKernelGGL(k_plan_items, dim3((L.cap_items + 255u) / 256u), dim3(256), 0, s, off, cur, L.nb, P, P + L.items_per, P + L.item_first, P + L.item_bucket);
        hipLaunchKernelGGL(k_plan_sort, dim3((L.cap_items + 255u) / 256u), dim3(256), 0, s, off, cur, P, P + L.item_first, P + L.item_bucket, P + L.sorted);
        hipLaunchKernelGGL(k_plan_groups, dim3((L.cap_groups + 255u) / 256u), dim3(256), 0, s, off, cur, P, P + L.item_first, P + L.item_bucket, P + L.sorted,
                           P + L.group_len, L.cap_groups, P);
        launch_exclusive_scan(P + L.group_len, P + L.group_off, P + L.scan_sums, L.cap_groups, s);
        hipLaunchKernelGGL(k_plan_pack, dim3((L.cap_groups * 64u + 255u) / 256u), dim3(256), 0, s, off, cur, list, meta, P, P + L.items_per, P + L.item_first,
                           P + L.item_bucket, P + L.sorted, P + L.group_len, P + L.group_off, P + L.desc,
                           reinterpret_cast<unsigned long long*>(P + L.packed));
    }
}

// value < 2^256 (any representative) -> canonical: 2^256 < 6 r
__device__ __forceinline__ Fr fr_canonical(Fr x) {
#pragma unroll
    for (int t = 0; t < 5; ++t) x = fr_reduce_once(x);
    return x;
}

// eq(z, g): gathered from the materialised table, or -- layers whose table would not stay in the 256 MiB Infinity Cache
// (a random 32-byte gather from 512 MiB fetched 1.95 GB per pass over 2^24 gates: profiles/r05/b_*) -- one reduced product of
// the two half tables E_hi (canonical) and E_lo (Montgomery), both a few hundred KiB
struct EqSource {
    const Fr* E;          // batch x 2^k_i, or null: the split form
    const Fr* e_hi;       // batch x 2^(k_i - kl)
    const Fr* e_lo_mont;  // batch x 2^kl
    uint32_t k_i, kl;
};
__device__ __forceinline__ Fr eq_at(const EqSource& q, uint32_t g) {
    if (q.E) return load_fr(q.E + ((size_t)blockIdx.y << q.k_i) + g);
    return mont_mul(load_fr(q.e_hi + ((size_t)blockIdx.y << (q.k_i - q.kl)) + (g >> q.kl)),
                    load_fr(q.e_lo_mont + ((size_t)blockIdx.y << q.kl) + (g & ((1u << q.kl) - 1u))));
}

// E[g] = E_hi[g >> kl] * E_lo[g & (2^kl - 1)], canonical: the table the passes gather, from the two half tables the layer's
// prologue has just built -- one product per entry (building it from the point, k_eq_table_split, was 23 - 34 us on every
// layer's path; this is ~5 at 2^15 entries).  grid = (blocks, batch), block = 256
__global__ void __launch_bounds__(256) k_eq_outer(const Fr* __restrict__ e_hi, const Fr* __restrict__ e_lo_mont, uint32_t k_i, uint32_t kl,
                                                  Fr* __restrict__ E) {
    const size_t n = (size_t)1 << k_i;
    e_hi += (size_t)blockIdx.y << (k_i - kl);
    e_lo_mont += (size_t)blockIdx.y << kl;
    E += (size_t)blockIdx.y << k_i;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < n; g += (size_t)gridDim.x * blockDim.x)
        store_fr(E + g, mont_mul(load_fr(e_hi + (g >> kl)), load_fr(e_lo_mont + (g & (((size_t)1 << kl) - 1)))));
}
void launch_eq_outer(const Fr* e_hi, const Fr* e_lo_mont, uint32_t k_i, uint32_t kl, Fr* E, uint32_t batch, hipStream_t s) {
    hipLaunchKernelGGL(k_eq_outer, dim3(blocks_for((uint64_t)1 << k_i, 16384), batch), dim3(256), 0, s, e_hi, e_lo_mont, k_i, kl, E);
}

// The passes: grid = (groups / 4 rounded up, batch), block = 256 = four waves = four groups.
//   ROWS == false (U, V):      mult gate: P += e t;   add gate: Q += e t and P += e      -> out0 = U, out1 = V
//   ROWS == true  (a_u, m_u):  add gate:  P += e t;   mult gate: Q += e t               -> out0 = a_u, out1 = m_u
// e = eq(z, g), canonical; t = T[other operand], Montgomery form (W resp. eq(u, .)).
// Groups whose items hold one gate or none (sorted by length, such items fill whole waves) take a short way: one reduced
// product per item, no accumulators.  (As a kernel of its own that path was slower: 0.095 / 0.102 ms per pass against
// 0.077 / 0.093 at 2^20 gates over 2^20 values, profiles/r05/b_*.)
constexpr uint32_t kSinglePerLane = 4;   // items of one gate (or none) a lane takes in one go

// where a bucket's two sums go: the pass's output tables, or -- the row pass of a whole layer on one rank (WideCFuse) -- straight
// into the c-phase's tables X = a_u + W(u) m_u, Y = W(u) a_u (a, m canonical, wu Montgomery: canonical products)
template <bool ROWS>
__device__ __forceinline__ void put_bucket(Fr* out0, Fr* out1, const WideCFuse& fuse, const Fr& wu, uint32_t b, const Fr& a, const Fr& m) {
    if (ROWS && fuse.X) {
        store_fr(fuse.X + b, fr_add(a, mont_mul(m, wu)));
        store_fr(fuse.Y + b, mont_mul(a, wu));
    } else {
        store_fr(out0 + b, a);
        store_fr(out1 + b, m);
    }
}

template <bool ROWS>
__global__ void __launch_bounds__(256) k_items_pass(const uint32_t* __restrict__ plan, PlanLayout L, EqSource eq, const Fr* __restrict__ T,
                                                    uint32_t wstride, Fr* __restrict__ out0, Fr* __restrict__ out1, Fr* __restrict__ partials,
                                                    size_t pstride, uint32_t gate_base, const GateSet* __restrict__ sets, WideCFuse fuse) {
    Fr wu = fr_zero();
    if (ROWS && fuse.X) {
        wu = load_fr(fuse.wu + blockIdx.y);
        fuse.X += (size_t)blockIdx.y * wstride;
        fuse.Y += (size_t)blockIdx.y * wstride;
    }
    if (sets) plan = sets[blockIdx.y].plan;
    plan += ROWS ? L.half_words : 0;
    const uint32_t lane = threadIdx.x & 63u, w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_items = plan[0], n_groups = plan[1], n_multi = plan[4];   // sorted by length: the first n_multi groups have two and more steps
    T += (size_t)blockIdx.y * wstride;
    out0 += (size_t)blockIdx.y * wstride;
    out1 += (size_t)blockIdx.y * wstride;
    partials += (size_t)blockIdx.y * pstride;
    const unsigned long long* packed = reinterpret_cast<const unsigned long long*>(plan + L.packed);
    if (w >= n_multi) {
        // Groups whose items hold one gate or none: one reduced product per item, no accumulators; a lane takes kSinglePerLane
        // items of as many groups, all their loads in flight together (these waves are two dependent gathers and a store: a
        // wave per group left the chip waiting for wave launches -- ~900 waves in flight on 1024 SIMDs, profiles/r05/b_*)
        const uint32_t g0 = n_multi + (w - n_multi) * kSinglePerLane;
        if (g0 >= n_groups) return;
        unsigned long long en[kSinglePerLane];
        uint32_t d[kSinglePerLane], slot[kSinglePerLane];
        bool live[kSinglePerLane];
#pragma unroll
        for (uint32_t q = 0; q < kSinglePerLane; ++q) {
            const uint32_t g = g0 + q, pos = 64u * g + lane;
            live[q] = g < n_groups && pos < n_items;
            const uint32_t steps = g < n_groups ? plan[L.group_len + g] : 0u;
            en[q] = live[q] && steps ? packed[(size_t)plan[L.group_off + g] * 64u + lane] : kNoGate;
            d[q] = live[q] ? plan[L.desc + pos] : 0u;
            slot[q] = live[q] && !(d[q] >> 31) ? plan[L.sorted + pos] : 0u;
        }
        Fr e[kSinglePerLane], t[kSinglePerLane];
#pragma unroll
        for (uint32_t q = 0; q < kSinglePerLane; ++q) {
            const bool has = en[q] != kNoGate;
            e[q] = eq_at(eq, has ? (uint32_t)(en[q] & 0xfffffffu) + gate_base : 0u);
            t[q] = load_fr(T + (has ? (uint32_t)((en[q] >> 28) & 0xffffffu) : 0u));
        }
#pragma unroll
        for (uint32_t q = 0; q < kSinglePerLane; ++q) {
            const bool has = en[q] != kNoGate, mult = (en[q] >> 63) != 0ull;
            const Fr et = mont_mul(e[q], t[q]);
            Fr o0, o1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t p = has ? et.l[i] : 0u, ee = has ? e[q].l[i] : 0u;
                o0.l[i] = ROWS ? (mult ? 0u : p) : (mult ? p : ee);
                o1.l[i] = ROWS ? (mult ? p : 0u) : (mult ? 0u : p);
            }
            if (live[q]) {
                if (d[q] >> 31) {
                    put_bucket<ROWS>(out0, out1, fuse, wu, d[q] & 0x7fffffffu, o0, o1);
                } else {
                    store_fr(partials + 2u * (size_t)slot[q], o0);
                    store_fr(partials + 2u * (size_t)slot[q] + 1u, o1);
                }
            }
        }
        return;
    }
    const uint32_t g = w, pos = 64u * g + lane, steps = plan[L.group_len + g];
    const bool live = pos < n_items;
    const uint32_t d = live ? plan[L.desc + pos] : 0u;
    const unsigned long long* ent = packed + (size_t)plan[L.group_off + g] * 64u + lane;
    Lazy17 P = lazy_zero(), Q = lazy_zero();
    unsigned long long en = ent[0];
    for (uint32_t st = 0; st < steps; ++st) {
        const unsigned long long cur = en;
        if (st + 1u < steps) en = ent[(size_t)(st + 1u) * 64u];   // the next step's entry is on its way while this one's products run
        if (cur != kNoGate) {
            const Fr e = eq_at(eq, (uint32_t)(cur & 0xfffffffu) + gate_base);
            const Fr t = load_fr(T + (uint32_t)((cur >> 28) & 0xffffffu));
            const bool mult = (cur >> 63) != 0ull;
            lazy_mac_sel(P, Q, ROWS ? !mult : mult, e, t);
            if (!ROWS) lazy_add_hi(P, e, !mult);
        }
    }
    // <= kItemMax products and <= kItemMax terms e * 2^256 per accumulator: the 32-term partial reduction applies
    Fr rp = lazy_reduce_partial32(P), rq = lazy_reduce_partial32(Q);
    if (live) {
        const bool whole = (d >> 31) != 0u;
        if (whole) {   // (wave-uniform in all but the waves where the buckets' lengths change)
            rp = fr_canonical(rp);
            rq = fr_canonical(rq);
        }
        if (whole) {
            put_bucket<ROWS>(out0, out1, fuse, wu, d & 0x7fffffffu, rp, rq);
        } else {
            store_fr(partials + 2u * (size_t)plan[L.sorted + pos], rp);
            store_fr(partials + 2u * (size_t)plan[L.sorted + pos] + 1u, rq);
        }
    }
}

// The combine step, ONE launch: blocks [0, multi_blocks) take the buckets of 2 .. kLongItems items, a lane per bucket (it adds
// its items' partial sums -- values below 2^256 -- and reduces); the blocks behind them take the buckets of more items (the
// constant wires every relay gate reads: ONE bucket with half the layer) a wave per chunk of kChunkItems items, and the wave
// that completes a bucket's LAST chunk adds the chunks and writes the bucket (arrive: one zeroed counter per proof and long
// bucket, left zero again).  One wave over the 32 768 items of a 2^19-gate bucket took 0.15 ms -- longer than the pass itself;
// three launches (buckets, chunks, long buckets) cost a layer's chain six launches per sumcheck.
template <bool ROWS>
__global__ void __launch_bounds__(256) k_items_combine(const uint32_t* __restrict__ plan, PlanLayout L, Fr* __restrict__ partials, size_t pstride,
                                                       Fr* __restrict__ out0, Fr* __restrict__ out1, uint32_t wstride, uint32_t multi_blocks,
                                                       uint32_t* __restrict__ arrive, const GateSet* __restrict__ sets, WideCFuse fuse) {
    Fr wu = fr_zero();
    if (ROWS && fuse.X) {
        wu = load_fr(fuse.wu + blockIdx.y);
        fuse.X += (size_t)blockIdx.y * wstride;
        fuse.Y += (size_t)blockIdx.y * wstride;
    }
    if (sets) plan = sets[blockIdx.y].plan;
    plan += ROWS ? L.half_words : 0;
    Fr* base = partials + (size_t)blockIdx.y * pstride;
    out0 += (size_t)blockIdx.y * wstride;
    out1 += (size_t)blockIdx.y * wstride;
    if (blockIdx.x < multi_blocks) {
        const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
        if (m >= plan[2]) return;
        const uint32_t b = plan[L.multi + m], first = plan[L.item_first + b], n = plan[L.items_per + b];
        const Fr* part = base + 2u * (size_t)first;
        Acc<9> a0 = acc_zero<9>(), a1 = acc_zero<9>();
        for (uint32_t j = 0; j < n; ++j) {
            acc_add_fr(a0, load_fr(part + 2u * (size_t)j));
            acc_add_fr(a1, load_fr(part + 2u * (size_t)j + 1u));
        }
        put_bucket<ROWS>(out0, out1, fuse, wu, b, acc_reduce(a0), acc_reduce(a1));
        return;
    }
    const uint32_t lane = threadIdx.x & 63u, wave = ((blockIdx.x - multi_blocks) * blockDim.x + threadIdx.x) >> 6,
                   nwaves = ((gridDim.x - multi_blocks) * blockDim.x) >> 6;
    arrive += (size_t)blockIdx.y * L.cap_long;
    Fr* chunk_sums = base + 2u * (size_t)L.cap_items;
    for (uint32_t c = wave; c < plan[5]; c += nwaves) {
        const uint32_t slot = plan[L.chunk_slot + c], b = plan[L.longb + slot], first = plan[L.item_first + b], n = plan[L.items_per + b];
        const uint32_t c0 = plan[L.long_chunk0 + slot], nch = (n + kChunkItems - 1u) / kChunkItems;
        const uint32_t lo = (c - c0) * kChunkItems, hi = lo + kChunkItems < n ? lo + kChunkItems : n;
        const Fr* part = base + 2u * (size_t)first;
        Acc<9> a0 = acc_zero<9>(), a1 = acc_zero<9>();
        for (uint32_t j = lo + lane; j < hi; j += 64u) {
            acc_add_fr(a0, load_fr(part + 2u * (size_t)j));
            acc_add_fr(a1, load_fr(part + 2u * (size_t)j + 1u));
        }
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        uint32_t last = 0;
        if (lane == 0) {
            store_fr(chunk_sums + 2u * (size_t)c, acc_reduce(a0));
            store_fr(chunk_sums + 2u * (size_t)c + 1u, acc_reduce(a1));
            __threadfence();   // (the chunk's sums cross the XCDs' L2s: written back before the arrival is counted)
            last = atomicAdd(arrive + slot, 1u) == nch - 1u ? 1u : 0u;
        }
        last = (uint32_t)__shfl((int)last, 0, 64);
        if (last) {
            __threadfence();
            Acc<9> t0 = acc_zero<9>(), t1 = acc_zero<9>();
            for (uint32_t j = lane; j < nch; j += 64u) {
                acc_add_fr(t0, load_fr(chunk_sums + 2u * (size_t)(c0 + j)));
                acc_add_fr(t1, load_fr(chunk_sums + 2u * (size_t)(c0 + j) + 1u));
            }
            t0 = wave_sum(t0);
            t1 = wave_sum(t1);
            if (lane == 0) {
                put_bucket<ROWS>(out0, out1, fuse, wu, b, acc_reduce(t0), acc_reduce(t1));
                arrive[slot] = 0u;   // (the next pass's waves start after this kernel)
            }
        }
    }
}

size_t gate_plan_arrive_words(uint64_t gates, uint32_t k) { return plan_layout(gates, k).cap_long; }   // per proof
void gate_plan_counts_offsets(uint64_t gates, uint32_t k, size_t* half1_word_offset) { *half1_word_offset = plan_layout(gates, k).half_words; }

template <bool ROWS>
static void launch_items_pass_t(GateSpan span, uint32_t k_i, uint32_t k, const uint32_t* plan, const GateEq& ge, const Fr* T, Fr* out0, Fr* out1, LayerBatch lb,
                                Fr* partials, uint32_t* arrive, hipStream_t s, const GateSet* sets, const GatePlanCounts* counts, const WideCFuse* fuse_in = nullptr) {
    const WideCFuse fuse = fuse_in ? *fuse_in : WideCFuse{};
    const PlanLayout L = plan_layout(span.count, k);
    const size_t pstride = gate_plan_partial_elems(span.count, k);
    const EqSource eq{ge.E, ge.e_hi, ge.e_lo_mont, k_i, ge.kl};
    const uint32_t half = ROWS ? 1u : 0u;
    // Grids from the plan's counts where the host knows them (read back when the plan was built), else from its capacities: the
    // kernels read the counts the build left on the device, waves beyond them leave at once.
    const bool known = counts && counts->known;
    const uint32_t groups = known ? counts->hdr[half][1] : L.cap_groups;
    const uint32_t n_multi = known ? counts->hdr[half][2] : L.cap_multi, n_long = known ? counts->hdr[half][3] : L.cap_long,
                   n_chunks = known ? counts->hdr[half][5] : L.cap_chunks;
    // (waves: one per group of two and more steps, one per kSinglePerLane groups of the rest -- at most `groups`)
    hipLaunchKernelGGL((k_items_pass<ROWS>), dim3((groups + 3u) / 4u, lb.batch), dim3(256), 0, s, plan, L, eq, T, (uint32_t)lb.wstride, out0, out1,
                       partials, pstride, (uint32_t)span.base, sets, fuse);
    if (known && n_multi == 0u && n_long == 0u) return;   // every bucket is one item: nothing to combine
    const uint32_t multi_blocks = (n_multi + 255u) / 256u;
    const uint32_t chunk_blocks = n_long ? (n_chunks < 1024u ? (n_chunks + 3u) / 4u : 256u) : 0u;
    hipLaunchKernelGGL((k_items_combine<ROWS>), dim3(multi_blocks + chunk_blocks, lb.batch), dim3(256), 0, s, plan, L, partials, pstride, out0, out1,
                       (uint32_t)lb.wstride, multi_blocks, arrive, sets, fuse);
}
void launch_gate_uv_wide(GateSpan span, uint32_t k_i, uint32_t k, const uint32_t* plan, const GateEq& eq, const Fr* W, Fr* U, Fr* V, LayerBatch lb, Fr* partials,
                         uint32_t* arrive, hipStream_t s, const GateSet* sets, const GatePlanCounts* counts) {
    launch_items_pass_t<false>(span, k_i, k, plan, eq, W, U, V, lb, partials, arrive, s, sets, counts);
}
void launch_gate_rows_wide(GateSpan span, uint32_t k_i, uint32_t k, const uint32_t* plan, const GateEq& eq, const Fr* eq_mont, Fr* A_row, Fr* M_row, LayerBatch lb,
                           Fr* partials, uint32_t* arrive, hipStream_t s, const GateSet* sets, const GatePlanCounts* counts, const WideCFuse* fuse) {
    launch_items_pass_t<true>(span, k_i, k, plan, eq, eq_mont, A_row, M_row, lb, partials, arrive, s, sets, counts, fuse);
}

// every entry < r?  (a large W handed over in host memory is validated where it lands: the host loop over 2^20 entries
// cost more than the layer's gate passes)  *flag |= 1 on an entry >= r.  grid = blocks, block = 256
__global__ void __launch_bounds__(256) k_check_canonical(const Fr* __restrict__ t, size_t n, uint32_t* __restrict__ flag) {
    bool bad = false;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) bad |= !fr_is_canonical(load_fr(t + i));
    if (__any(bad) && (threadIdx.x & 63u) == 0) atomicOr(flag, 1u);
}
void launch_check_canonical(const Fr* t, size_t n, uint32_t* flag, hipStream_t s) {
    hipLaunchKernelGGL(k_check_canonical, dim3(blocks_for(n, 2048)), dim3(256), 0, s, t, n, flag);
}

// ---------------------------------------------------------------------------
// dependence flags of a wide W (length rule, get_univariate_coeff, poly.rs:388-420): bit b of *bits is set iff W
// differs somewhere across index bit (k - 1 - b).  grid = (blocks, batch); bits: one zeroed word per proof.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_depends_wide(const Fr* __restrict__ W, uint32_t k, uint32_t* __restrict__ bits) {
    __shared__ uint32_t s_bits;
    const uint32_t n = 1u << k;
    const Fr* w = W + ((size_t)blockIdx.y << k);
    if (threadIdx.x == 0) s_bits = 0u;
    __syncthreads();
    const uint32_t all = k >= 32u ? 0xffffffffu : (1u << k) - 1u;
    const uint32_t known = __hip_atomic_load(bits + blockIdx.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // what other blocks (or the layer's prologue) found already
    if (known == all) return;   // (uniform over the block)
    uint32_t mine = known;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n && mine != all; i += gridDim.x * blockDim.x) {
        const Fr a = load_fr(w + i);
        for (uint32_t b = 0; b < k; ++b) {
            const uint32_t bit = 1u << (k - 1u - b);
            if (!((mine >> b) & 1u) && !(i & bit) && !fr_eq(a, load_fr(w + (i ^ bit)))) mine |= 1u << b;
        }
    }
    if (mine & ~known) atomicOr(&s_bits, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_bits) atomicOr(bits + blockIdx.y, s_bits);
}
// the 32 flags per proof from the word, on the device and (when not null) in pinned host memory
__global__ void k_dep_publish(const uint32_t* __restrict__ bits, uint32_t* __restrict__ dep, uint32_t* __restrict__ host_dep) {
    const uint32_t f = (bits[blockIdx.x] >> threadIdx.x) & 1u;
    dep[(size_t)blockIdx.x * 32 + threadIdx.x] = f;
    if (host_dep) host_dep[(size_t)blockIdx.x * 32 + threadIdx.x] = f;
}
void launch_depends_wide(const Fr* W, uint32_t k, uint32_t* bits, uint32_t* dep, uint32_t* host_dep, uint32_t batch, hipStream_t s, bool bits_preset) {
    if (!bits_preset) (void)hipMemsetAsync(bits, 0, sizeof(uint32_t) * batch, s);
    hipLaunchKernelGGL(k_depends_wide, dim3(blocks_for((uint64_t)1 << k, 1024), batch), dim3(256), 0, s, W, k, bits);
    hipLaunchKernelGGL(k_dep_publish, dim3(batch), dim3(32), 0, s, bits, dep, host_dep);
}

// ---------------------------------------------------------------------------
// Moebius transform over a grid: evaluation table -> monomial coefficients, variable 1 = most significant index bit
// (get_multi_ext, poly.rs:502-536).  One launch per variable: entries with the bit set subtract their partner without
// it (which this pass does not write).  grid = (blocks, batch), tables `stride` elements apart.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mobius_pass(Fr* __restrict__ t, uint32_t k, uint32_t bit, size_t stride) {
    Fr* p = t + (size_t)blockIdx.y * stride;
    const uint32_t half = 1u << (k - 1u);
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < half; j += gridDim.x * blockDim.x) {
        const uint32_t lo = j & (bit - 1u), i = ((j & ~(bit - 1u)) << 1) | bit | lo;   // the j-th index with `bit` set
        store_fr(p + i, fr_sub(load_fr(p + i), load_fr(p + (i ^ bit))));
    }
}
// Up to ten variables per launch (the transform over different variables commutes): a block takes the 2^m entries that
// differ only in index bits s .. s + m - 1 into LDS, runs the m butterfly stages there and writes them back -- a table of
// 2^20 entries in two launches instead of twenty (a proof's line restrictions and coefficient tables issued 35 launches per
// layer from the proving thread, as many as the layer's sumcheck).  grid = (2^(k - m), batch), block = 256
constexpr uint32_t kMobiusBitsPerLaunch = 10;
__global__ void __launch_bounds__(256) k_mobius_lds(Fr* __restrict__ t, uint32_t k, uint32_t s, uint32_t m, size_t stride) {
    __shared__ Fr sh[1u << kMobiusBitsPerLaunch];
    Fr* p = t + (size_t)blockIdx.y * stride;
    const uint32_t n = 1u << m, lo_mask = (1u << s) - 1u;
    const uint32_t lo = blockIdx.x & lo_mask, hi = blockIdx.x >> s;
    const size_t base = ((size_t)hi << (s + m)) | lo;
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) sh[e] = load_fr(p + base + ((size_t)e << s));
    __syncthreads();
    for (uint32_t bit = 1u; bit < n; bit <<= 1) {
        for (uint32_t j = threadIdx.x; j < (n >> 1); j += blockDim.x) {
            const uint32_t l = j & (bit - 1u), i = ((j & ~(bit - 1u)) << 1) | bit | l;
            sh[i] = fr_sub(sh[i], sh[i ^ bit]);
        }
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) store_fr(p + base + ((size_t)e << s), sh[e]);
}
void launch_mobius(Fr* tables, uint32_t k, size_t stride, uint32_t batch, hipStream_t s) {
    if (k == 0) return;
    for (uint32_t b = 0; b < k; b += kMobiusBitsPerLaunch) {
        const uint32_t m = k - b < kMobiusBitsPerLaunch ? k - b : kMobiusBitsPerLaunch;
        hipLaunchKernelGGL(k_mobius_lds, dim3(1u << (k - m), batch), dim3(256), 0, s, tables, k, b, m, stride);
    }
}

// ---------------------------------------------------------------------------
// set-up of the stepwise line restriction (kernels.hip, k_line_step) for wide layers: scratch = 3 * 2^k elements per
// proof -- the table (first 2^k), its ping-pong half, and W's monomial coefficients, of which only the support counts:
// q's length = 1 + the largest total degree of a non-zero monomial (reduce_multiple_polynomial, poly.rs:484-497).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_line_copy(const Fr* __restrict__ W, uint32_t k, Fr* __restrict__ scratch, uint32_t* __restrict__ maxdeg) {
    const uint32_t n = 1u << k;
    if (blockIdx.x == 0 && threadIdx.x == 0) maxdeg[blockIdx.y] = 0u;   // (k_line_maxdeg, two launches later, raises it)
    const Fr* w = W + ((size_t)blockIdx.y << k);
    Fr* base = scratch + (size_t)blockIdx.y * 3u * n;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Fr v = load_fr(w + i);
        store_fr(base + i, v);
        store_fr(base + 2u * (size_t)n + i, v);
    }
}
__global__ void __launch_bounds__(256) k_line_maxdeg(uint32_t k, const Fr* __restrict__ scratch, uint32_t* __restrict__ maxdeg) {
    __shared__ uint32_t s_deg;
    const uint32_t n = 1u << k;
    const Fr* mono = scratch + (size_t)blockIdx.y * 3u * n + 2u * (size_t)n;
    if (threadIdx.x == 0) s_deg = 0u;
    __syncthreads();
    uint32_t deg = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if ((uint32_t)__popc(i) > deg && !fr_is_zero(load_fr(mono + i))) deg = (uint32_t)__popc(i);
    if (deg) atomicMax(&s_deg, deg);
    __syncthreads();
    if (threadIdx.x == 0 && s_deg) atomicMax(maxdeg + blockIdx.y, s_deg);
}
// (the length itself, maxdeg + 1, is written with q by k_line_tail)
void launch_line_setup_wide(const Fr* W, uint32_t k, Fr* scratch, uint32_t* maxdeg_scratch, uint32_t batch, hipStream_t s) {
    const size_t n = (size_t)1 << k;
    hipLaunchKernelGGL(k_line_copy, dim3(blocks_for(n, 2048), batch), dim3(256), 0, s, W, k, scratch, maxdeg_scratch);
    launch_mobius(scratch + 2 * n, k, 3 * n, batch, s);
    hipLaunchKernelGGL(k_line_maxdeg, dim3(blocks_for(n, 1024), batch), dim3(256), 0, s, k, scratch, maxdeg_scratch);
}

// ---------------------------------------------------------------------------
// One plain sumcheck split over ranks (gkr_sumcheck_mle_sharded_dev): what crosses the ranks is, per pass, the 2^J
// sub-block sums of every table -- they are linear in the table, so the sums of the whole table are the sums over ranks
// of the shards' sums (the multi-GPU twin of the rayon reduce, sumcheck.rs:62 applied to prove_sumcheck :158-214) --
// and once, at the end, the few entries every shard has left.  Field elements travel as eight 32-bit limbs in int64 (an
// integer SUM all-reduce of those is exact; RCCL has no modular sum).
//   per table: 2^J sums | dep ("the shard depends on the last variable") | fail   ->  (2^J + 2) x 8 int64
// ---------------------------------------------------------------------------
// grid = (batch), block = 64: records in DEVICE memory (written by k_mle_sub_reduce / k_mle_multifold_small) -> limbs
__global__ void __launch_bounds__(64) k_mle_xwiden(const MleHostRecSub* __restrict__ rec, uint32_t J, uint32_t local_fail, long long* __restrict__ limbs) {
    const uint32_t b = blockIdx.x, n = (1u << J) + 2u, e = threadIdx.x;
    if (e >= n) return;
    long long* out = limbs + ((size_t)b * n + e) * 8;
    if (e < (1u << J)) {
        const Fr v = load_fr(&rec[b].sums[e]);
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = local_fail ? 0 : (long long)v.l[j];
    } else {
        const uint32_t f = e == (1u << J) ? (local_fail ? 0u : rec[b].dep) : local_fail;
        out[0] = (long long)(f ? 1 : 0);
#pragma unroll
        for (int j = 1; j < 8; ++j) out[j] = 0;
    }
}
// limb sums -> the pinned host record of every table (sums mod r, dep = some rank's shard depends on the last variable),
// seq = ticket last; *fail_out (pinned, may be null) = some rank failed
__global__ void __launch_bounds__(64) k_mle_xnarrow(const long long* __restrict__ limbs, uint32_t J, MleHostRecSub* __restrict__ host_rec,
                                                    uint32_t ticket, uint32_t* __restrict__ fail_out) {
    const uint32_t b = blockIdx.x, n = (1u << J) + 2u, e = threadIdx.x;
    const long long* in = limbs + ((size_t)b * n + e) * 8;
    MleHostRecSub* r = host_rec + b;
    if (e < (1u << J)) {
        Acc<10> acc = acc_zero<10>();
        unsigned long long carry = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned long long w = (unsigned long long)in[j];
            const unsigned long long lo = (w & 0xffffffffull) + (carry & 0xffffffffull);
            acc.l[j] = (uint32_t)lo;
            carry = (w >> 32) + (carry >> 32) + (lo >> 32);
        }
        acc.l[8] = (uint32_t)carry;
        acc.l[9] = (uint32_t)(carry >> 32);
        store_fr(&r->sums[e], acc_reduce(acc));
    } else if (e == (1u << J)) {
        r->dep = in[0] ? 1u : 0u;
    } else if (e == (1u << J) + 1u) {
        if (in[0] && fail_out) __hip_atomic_store(fail_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();   // every record store is issued and waited for before the release below
    if (e == 0) __hip_atomic_store(&r->seq, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_mle_xwiden(const MleHostRecSub* rec, uint32_t J, uint32_t batch, uint32_t local_fail, long long* limbs, hipStream_t s) {
    hipLaunchKernelGGL(k_mle_xwiden, dim3(batch), dim3(64), 0, s, rec, J, local_fail, limbs);
}
void launch_mle_xnarrow(const long long* limbs, uint32_t J, uint32_t batch, MleHostRecSub* host_rec, uint32_t ticket, uint32_t* fail_out,
                        hipStream_t s) {
    hipLaunchKernelGGL(k_mle_xnarrow, dim3(batch), dim3(64), 0, s, limbs, J, host_rec, ticket, fail_out);
}

// The shards' last entries gathered into the tail table every rank finishes on.  Shard p of P = 2^lp holds, of the table
// with the leading variables bound, the entries whose index bits lp .. 1 are p (the last variable -- bit 0 -- stays
// inside every shard: "does T depend on x_n" is then a local neighbour compare, exact without comparing shards across
// ranks): local entry (h, x_n) is tail entry h * 2P + 2p + x_n.  An all-gather as a SUM all-reduce of zero-padded
// buffers: per table 2^(t + lp) elements x 8 int64, then one more element: the fail flag.
// grid = (blocks, batch), block = 256
__global__ void __launch_bounds__(256) k_mle_gather_widen(const Fr* __restrict__ src, size_t stride, uint32_t t, uint32_t lp, uint32_t shard,
                                                          uint32_t local_fail, uint32_t batch, long long* __restrict__ limbs) {
    const uint32_t total = 1u << (t + lp), b = blockIdx.y;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += gridDim.x * blockDim.x) {
        long long* out = limbs + ((size_t)b * total + g) * 8;
        const bool mine = ((g >> 1) & ((1u << lp) - 1u)) == shard && !local_fail;
        Fr v = fr_zero();
        if (mine) v = load_fr(src + (size_t)b * stride + (((g >> (lp + 1u)) << 1) | (g & 1u)));
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = (long long)v.l[j];
    }
    if (b == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        long long* f = limbs + (size_t)batch * total * 8;
        f[0] = local_fail ? 1 : 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) f[j] = 0;
    }
}
__global__ void __launch_bounds__(256) k_mle_gather_narrow(const long long* __restrict__ limbs, uint32_t tn, uint32_t batch, Fr* __restrict__ tail,
                                                           uint32_t* __restrict__ fail_out) {
    const uint32_t total = 1u << tn, b = blockIdx.y;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += gridDim.x * blockDim.x) {
        const long long* in = limbs + ((size_t)b * total + g) * 8;
        Acc<10> acc = acc_zero<10>();
        unsigned long long carry = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned long long w = (unsigned long long)in[j];
            const unsigned long long lo = (w & 0xffffffffull) + (carry & 0xffffffffull);
            acc.l[j] = (uint32_t)lo;
            carry = (w >> 32) + (carry >> 32) + (lo >> 32);
        }
        acc.l[8] = (uint32_t)carry;
        acc.l[9] = (uint32_t)(carry >> 32);
        store_fr(tail + (size_t)b * total + g, acc_reduce(acc));
    }
    if (b == 0 && blockIdx.x == 0 && threadIdx.x == 0 && fail_out && limbs[(size_t)batch * total * 8])
        __hip_atomic_store(fail_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_mle_gather_widen(const Fr* src, size_t stride, uint32_t t, uint32_t lp, uint32_t shard, uint32_t local_fail, uint32_t batch,
                             long long* limbs, hipStream_t s) {
    hipLaunchKernelGGL(k_mle_gather_widen, dim3(blocks_for((uint64_t)1 << (t + lp), 64), batch), dim3(256), 0, s, src, stride, t, lp, shard, local_fail,
                       batch, limbs);
}
void launch_mle_gather_narrow(const long long* limbs, uint32_t tn, uint32_t batch, Fr* tail, uint32_t* fail_out, hipStream_t s) {
    hipLaunchKernelGGL(k_mle_gather_narrow, dim3(blocks_for((uint64_t)1 << tn, 64), batch), dim3(256), 0, s, limbs, tn, batch, tail, fail_out);
}

}  // namespace gkr
