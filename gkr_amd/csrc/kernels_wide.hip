// CDNA4 (gfx950) kernels for WIDE layers of the GKR layer sumcheck: next-layer tables of 2^14 .. 2^28 values
// (prove_sumcheck_opt, rust/src/gkr/sumcheck.rs:36-156, accepts any GKRCircuit; the compiler pads a layer to whatever
// 2^k it needs, rust/src/convert.rs:209-214, and WIDTH_LIMIT = 20 merges all constraints of an R1CS into at most 20
// sub-circuits, convert.rs:10-11,171-186 -- a 10^5-constraint circuit has layers far beyond 2^14 values).
//
// The linear-time form of kernels.hip has no inherent width limit: U, V, the c-phase row and W are 2^k-entry tables in
// HBM.  What changes with the width is the SHAPE of the work:
//   * 2^k buckets of the gate lists with only a few gates each (a circom layer has about as many gates as the next
//     layer has values) -- a block per bucket would launch 2^20 blocks for one gate apiece.  Here a GROUP of L = 1..64
//     lanes sums a bucket, about eight gates per lane (k_gate_group), and the few buckets far longer than the rest -- the constant wires every
//     relay gate reads (convert.rs:307-342) -- are cut into units of 256 / 1024 gates summed a wave per unit (k_gate_heavy,
//     k_heavy_combine), so that no lane ever walks more than 64 gates;
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
// gate passes: a group of lanes per bucket, heavy buckets in units
// ---------------------------------------------------------------------------

// Lanes of a wave that sum one bucket: about EIGHT gates per lane.  A lane's fixed cost -- two reductions of its unreduced
// sums, the group's shuffles, two more reductions -- is ~1000 instructions whether it walked one gate or sixteen; measured on
// MI355X, ms per gate pass (profiles/r04/h_gate_group_lanes_sweep.txt): 2^24 gates over 2^18 buckets (64 per bucket) 1.74 with
// a whole wave per bucket, 0.52 with four lanes; 2^20 over 2^15 (32 per bucket) 0.216 with 32 lanes, 0.065 with four; 2^20
// over 2^20 (one per bucket) 0.40 with four lanes, 0.17 with one.
uint32_t gate_group_lanes_log2(uint64_t gates, uint32_t k) {
    const int forced = (int)opt(OPT_gate_group_lanes_log2);   // (measurement knob)
    if (forced >= 0 && forced <= 6) return (uint32_t)forced;
    const uint64_t mean = gates >> k;
    uint32_t lg = 0;
    while (lg < 6 && ((uint64_t)8 << lg) < mean) ++lg;
    return lg;
}
uint32_t gate_heavy_threshold(uint64_t gates, uint32_t k) { return kHeavyPerLane << gate_group_lanes_log2(gates, k); }   // (at least 64: a lane group's first lane)
// capacities of one half (left-operand buckets / right-operand buckets)
static inline size_t heavy_cap_buckets(uint64_t gates, uint32_t k) { return (size_t)(gates / gate_heavy_threshold(gates, k)) + 1; }
static inline size_t heavy_cap_units(uint64_t gates, uint32_t k) { return (size_t)(gates / gate_heavy_unit(gates)) + heavy_cap_buckets(gates, k); }
size_t gate_heavy_words(uint64_t gates, uint32_t k) {
    // header (8 words) | per half: heavy buckets {bucket, first unit, units} (3 words each) | units {bucket, chunk} (2 words each)
    return 8 + 2 * (3 * heavy_cap_buckets(gates, k) + 2 * heavy_cap_units(gates, k));
}
size_t gate_heavy_partial_elems(uint64_t gates, uint32_t k) { return 2 * heavy_cap_units(gates, k); }

struct HeavyView {
    uint32_t* hdr;        // [2 * half] heavy buckets, [2 * half + 1] units
    uint32_t* buckets;    // 3 words per heavy bucket of this half
    uint32_t* units;      // 2 words per unit of this half
};
static inline HeavyView heavy_view(uint32_t* words, uint64_t gates, uint32_t k, uint32_t half) {
    const size_t cb = heavy_cap_buckets(gates, k), cu = heavy_cap_units(gates, k);
    uint32_t* base = words + 8 + (size_t)half * (3 * cb + 2 * cu);
    return HeavyView{words, base, base + 3 * cb};
}

// one thread per bucket of both halves: buckets longer than `threshold` go to the half's work lists
__global__ void __launch_bounds__(256) k_heavy_list(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ cursor, uint32_t nb,
                                                    uint32_t threshold, uint32_t unit, uint32_t* __restrict__ hdr, uint32_t* __restrict__ bucketsL,
                                                    uint32_t* __restrict__ unitsL, uint32_t* __restrict__ bucketsR, uint32_t* __restrict__ unitsR) {
    const uint32_t b2 = blockIdx.x * blockDim.x + threadIdx.x;
    if (b2 >= 2u * nb) return;
    const uint32_t len = cursor[b2] - offsets[b2];
    if (len <= threshold) return;
    const uint32_t half = b2 >= nb ? 1u : 0u, nunits = (len + unit - 1u) / unit;
    const uint32_t slot = atomicAdd(hdr + 2u * half, 1u), first = atomicAdd(hdr + 2u * half + 1u, nunits);
    uint32_t* hb = (half ? bucketsR : bucketsL) + 3u * (size_t)slot;
    hb[0] = b2;
    hb[1] = first;
    hb[2] = nunits;
    uint32_t* un = (half ? unitsR : unitsL) + 2u * (size_t)first;
    for (uint32_t c = 0; c < nunits; ++c) {
        un[2u * c] = b2;
        un[2u * c + 1u] = c;
    }
}

// one gate's terms.  e = eq(z, g) (canonical, a reduced product: it is an operand), t = W[right] resp. eq(u, left) in
// Montgomery form.  The products e t only ever enter a sum: added unreduced, reduced once per lane.
//   ROWS == false (U, V):      mult gate: P += e t;   add gate: Q += e t and S += e
//   ROWS == true  (a_u, m_u):  add gate:  P += e t;   mult gate: Q += e t
template <bool ROWS>
__device__ __forceinline__ void gate_term(Lazy17& P, Lazy17& Q, Acc<9>& S, uint32_t gg, uint32_t mt, const Fr* __restrict__ e_hi,
                                          const Fr* __restrict__ e_lo_mont, uint32_t kl, uint32_t lmask, const Fr* __restrict__ T) {
    const Fr e = mont_mul(load_fr(e_hi + (gg >> kl)), load_fr(e_lo_mont + (gg & lmask)));
    const Fr t = load_fr(T + (mt & 0x7fffffffu));
    const bool mult = (mt >> 31) != 0u;
    lazy_mac_sel(P, Q, ROWS ? !mult : mult, e, t);
    if (!ROWS && !mult) acc_add_fr(S, e);
}

// sum of an accumulator over the 2^lg lanes of a group (lanes of one wave); every lane of the group gets the total
template <int NL>
__device__ __forceinline__ Acc<NL> group_sum(Acc<NL> a, uint32_t lg) {
    for (uint32_t off = 1u; off < (1u << lg); off <<= 1) {
        Acc<NL> o;
#pragma unroll
        for (int i = 0; i < NL; ++i) o.l[i] = __shfl_xor(a.l[i], (int)off, 64);
        acc_add_acc(a, o);
    }
    return a;
}

// grid = (2^k * L / 256, batch), block = 256: group `tid >> lg` of the grid sums bucket (first_bucket + group).
// out0 / out1: U, V resp. a_u, m_u (stride wstride per proof).  Buckets longer than `threshold` are left to the units.
template <bool ROWS>
__global__ void __launch_bounds__(256) k_gate_group(const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ cursor,
                                                    const uint32_t* __restrict__ list, const uint32_t* __restrict__ meta,
                                                    const Fr* __restrict__ e_hi, const Fr* __restrict__ e_lo_mont, uint32_t kl, uint32_t kh,
                                                    const Fr* __restrict__ T, Fr* __restrict__ out0, Fr* __restrict__ out1, uint32_t k,
                                                    uint32_t wstride, uint32_t gate_base, uint32_t lg, uint32_t threshold,
                                                    const GateSet* __restrict__ sets) {
    if (sets) {   // proofs of different circuits in one launch: this proof's lists (block-uniform)
        const GateSet gs = sets[blockIdx.y];
        const ptrdiff_t moff = meta - list;
        offsets = gs.offsets;
        cursor = gs.cursor;
        list = gs.list;
        meta = gs.list + moff;
    }
    const uint32_t L = 1u << lg, sub = threadIdx.x & (L - 1u), nb = 1u << k;
    const uint32_t bl = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> lg);   // bucket within the half
    const uint32_t lmask = (1u << kl) - 1u;
    e_hi += (size_t)blockIdx.y << kh;
    e_lo_mont += (size_t)blockIdx.y << kl;
    T += (size_t)blockIdx.y * wstride;
    Lazy17 P = lazy_zero(), Q = lazy_zero();
    Acc<9> S = acc_zero<9>();
    bool live = bl < nb;
    uint32_t i = 0, end = 0;
    if (live) {
        const uint32_t bucket = ROWS ? nb + bl : bl;
        i = offsets[bucket];
        end = cursor[bucket];
        if (end - i > threshold) {
            live = false;   // a heavy bucket: its units write it (k_heavy_combine)
            end = i;
        }
        i += sub;
    }
    for (; i < end; i += L) gate_term<ROWS>(P, Q, S, list[i] + gate_base, meta[i], e_hi, e_lo_mont, kl, lmask, T);
    Acc<9> a0 = S, a1 = acc_zero<9>();
    acc_add_fr(a0, lazy_reduce(P));
    acc_add_fr(a1, lazy_reduce(Q));
    a0 = group_sum(a0, lg);
    a1 = group_sum(a1, lg);
    if (live && sub == 0) {
        store_fr(out0 + (size_t)blockIdx.y * wstride + bl, acc_reduce(a0));
        store_fr(out1 + (size_t)blockIdx.y * wstride + bl, acc_reduce(a1));
    }
}

// a wave per unit (gate_heavy_unit() gates of a heavy bucket); grid = (blocks, batch), any number of blocks
template <bool ROWS>
__global__ void __launch_bounds__(256) k_gate_heavy(const uint32_t* __restrict__ hdr, uint32_t half, const uint32_t* __restrict__ units,
                                                    const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ cursor,
                                                    const uint32_t* __restrict__ list, const uint32_t* __restrict__ meta,
                                                    const Fr* __restrict__ e_hi, const Fr* __restrict__ e_lo_mont, uint32_t kl, uint32_t kh,
                                                    const Fr* __restrict__ T, Fr* __restrict__ partials, size_t pstride, uint32_t wstride,
                                                    uint32_t gate_base, uint32_t unit, const GateSet* __restrict__ sets) {
    if (sets) {
        const GateSet gs = sets[blockIdx.y];
        const ptrdiff_t moff = meta - list, uoff = units - hdr;
        offsets = gs.offsets;
        cursor = gs.cursor;
        list = gs.list;
        meta = gs.list + moff;
        hdr = gs.heavy;
        units = gs.heavy + uoff;
    }
    const uint32_t nunits = hdr[2u * half + 1u];
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t lmask = (1u << kl) - 1u;
    e_hi += (size_t)blockIdx.y << kh;
    e_lo_mont += (size_t)blockIdx.y << kl;
    T += (size_t)blockIdx.y * wstride;
    partials += (size_t)blockIdx.y * pstride;
    for (uint32_t u = wave; u < nunits; u += nwaves) {
        const uint32_t bucket = units[2u * u], chunk = units[2u * u + 1u];
        const uint32_t begin = offsets[bucket] + chunk * unit, stop = cursor[bucket];
        const uint32_t end = begin + unit < stop ? begin + unit : stop;
        Lazy17 P = lazy_zero(), Q = lazy_zero();
        Acc<9> S = acc_zero<9>();
        for (uint32_t i = begin + lane; i < end; i += 64u) gate_term<ROWS>(P, Q, S, list[i] + gate_base, meta[i], e_hi, e_lo_mont, kl, lmask, T);
        Acc<9> a0 = S, a1 = acc_zero<9>();
        acc_add_fr(a0, lazy_reduce(P));
        acc_add_fr(a1, lazy_reduce(Q));
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        if (lane == 0) {
            store_fr(partials + 2u * (size_t)u, acc_reduce(a0));
            store_fr(partials + 2u * (size_t)u + 1u, acc_reduce(a1));
        }
    }
}

// a wave per heavy bucket: the totals of its units -> the bucket's two outputs
__global__ void __launch_bounds__(256) k_heavy_combine(const uint32_t* __restrict__ hdr, uint32_t half, const uint32_t* __restrict__ buckets,
                                                       uint32_t nb, const Fr* __restrict__ partials, size_t pstride, Fr* __restrict__ out0,
                                                       Fr* __restrict__ out1, uint32_t wstride, const GateSet* __restrict__ sets) {
    if (sets) {
        const ptrdiff_t boff = buckets - hdr;
        hdr = sets[blockIdx.y].heavy;
        buckets = hdr + boff;
    }
    const uint32_t nheavy = hdr[2u * half];
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    partials += (size_t)blockIdx.y * pstride;
    for (uint32_t h = wave; h < nheavy; h += nwaves) {
        const uint32_t bucket = buckets[3u * h], first = buckets[3u * h + 1u], n = buckets[3u * h + 2u];
        Acc<9> a0 = acc_zero<9>(), a1 = acc_zero<9>();
        for (uint32_t c = lane; c < n; c += 64u) {
            acc_add_fr(a0, load_fr(partials + 2u * (size_t)(first + c)));
            acc_add_fr(a1, load_fr(partials + 2u * (size_t)(first + c) + 1u));
        }
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        if (lane == 0) {
            const uint32_t bl = half ? bucket - nb : bucket;
            store_fr(out0 + (size_t)blockIdx.y * wstride + bl, acc_reduce(a0));
            store_fr(out1 + (size_t)blockIdx.y * wstride + bl, acc_reduce(a1));
        }
    }
}

void launch_gate_heavy_lists(GateSpan span, uint32_t k, const uint32_t* offsets, const uint32_t* cursor, uint32_t* heavy_words, hipStream_t s) {
    const uint32_t nb = 1u << k;
    (void)hipMemsetAsync(heavy_words, 0, 8 * sizeof(uint32_t), s);
    const HeavyView L = heavy_view(heavy_words, span.count, k, 0), R = heavy_view(heavy_words, span.count, k, 1);
    hipLaunchKernelGGL(k_heavy_list, dim3((2u * nb + 255u) / 256u), dim3(256), 0, s, offsets, cursor, nb, gate_heavy_threshold(span.count, k),
                       gate_heavy_unit(span.count), heavy_words,
                       L.buckets, L.units, R.buckets, R.units);
}

template <bool ROWS>
static void launch_gate_group_t(GateSpan span, uint32_t k_i, uint32_t k, const uint32_t* offsets, const uint32_t* cursor, const uint32_t* list,
                                const Fr* e_hi, const Fr* e_lo_mont, uint32_t kl, const Fr* T, Fr* out0, Fr* out1, LayerBatch lb,
                                uint32_t* heavy_words, Fr* heavy_partials, hipStream_t s, const uint32_t* host_hdr, const GateSet* sets) {
    const uint32_t lg = gate_group_lanes_log2(span.count, k), threshold = gate_heavy_threshold(span.count, k);
    const uint32_t* meta = list + gate_list_words(span.count);
    const uint64_t threads = (uint64_t)1 << (k + lg);
    hipLaunchKernelGGL((k_gate_group<ROWS>), dim3((unsigned)((threads + 255) / 256), lb.batch), dim3(256), 0, s, offsets, cursor, list, meta, e_hi,
                       e_lo_mont, kl, k_i - kl, T, out0, out1, k, (uint32_t)lb.wstride, (uint32_t)span.base, lg, threshold, sets);
    const uint32_t half = ROWS ? 1u : 0u;
    // (the work lists' header as the host read it back when the lists were built: a half without a heavy bucket -- most
    // layers -- skips two launches per pass)
    if (host_hdr && host_hdr[2u * half] == 0u) return;
    const HeavyView hv = heavy_view(heavy_words, span.count, k, half);
    const size_t pstride = gate_heavy_partial_elems(span.count, k);
    hipLaunchKernelGGL((k_gate_heavy<ROWS>), dim3(512, lb.batch), dim3(256), 0, s, hv.hdr, half, hv.units, offsets, cursor, list, meta, e_hi, e_lo_mont,
                       kl, k_i - kl, T, heavy_partials, pstride, (uint32_t)lb.wstride, (uint32_t)span.base, gate_heavy_unit(span.count), sets);
    hipLaunchKernelGGL(k_heavy_combine, dim3(64, lb.batch), dim3(256), 0, s, hv.hdr, half, hv.buckets, 1u << k, heavy_partials, pstride, out0, out1,
                       (uint32_t)lb.wstride, sets);
}

void launch_gate_uv_wide(GateSpan span, uint32_t k_i, uint32_t k, const uint32_t* offsets, const uint32_t* cursor, const uint32_t* list,
                         const Fr* e_hi, const Fr* e_lo_mont, uint32_t kl, const Fr* W, Fr* U, Fr* V, LayerBatch lb, uint32_t* heavy_words,
                         Fr* heavy_partials, hipStream_t s, const uint32_t* host_hdr, const GateSet* sets) {
    launch_gate_group_t<false>(span, k_i, k, offsets, cursor, list, e_hi, e_lo_mont, kl, W, U, V, lb, heavy_words, heavy_partials, s, host_hdr, sets);
}
void launch_gate_rows_wide(GateSpan span, uint32_t k_i, uint32_t k, const uint32_t* offsets, const uint32_t* cursor, const uint32_t* list,
                           const Fr* e_hi, const Fr* e_lo_mont, uint32_t kl, const Fr* eq_mont, Fr* A_row, Fr* M_row, LayerBatch lb,
                           uint32_t* heavy_words, Fr* heavy_partials, hipStream_t s, const uint32_t* host_hdr, const GateSet* sets) {
    launch_gate_group_t<true>(span, k_i, k, offsets, cursor, list, e_hi, e_lo_mont, kl, eq_mont, A_row, M_row, lb, heavy_words, heavy_partials, s, host_hdr, sets);
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
    uint32_t mine = __hip_atomic_load(bits + blockIdx.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // what other blocks found already
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n && mine != all; i += gridDim.x * blockDim.x) {
        const Fr a = load_fr(w + i);
        for (uint32_t b = 0; b < k; ++b) {
            const uint32_t bit = 1u << (k - 1u - b);
            if (!((mine >> b) & 1u) && !(i & bit) && !fr_eq(a, load_fr(w + (i ^ bit)))) mine |= 1u << b;
        }
    }
    if (mine) atomicOr(&s_bits, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_bits) atomicOr(bits + blockIdx.y, s_bits);
}
// the 32 flags per proof from the word, on the device and (when not null) in pinned host memory
__global__ void k_dep_publish(const uint32_t* __restrict__ bits, uint32_t* __restrict__ dep, uint32_t* __restrict__ host_dep) {
    const uint32_t f = (bits[blockIdx.x] >> threadIdx.x) & 1u;
    dep[(size_t)blockIdx.x * 32 + threadIdx.x] = f;
    if (host_dep) host_dep[(size_t)blockIdx.x * 32 + threadIdx.x] = f;
}
void launch_depends_wide(const Fr* W, uint32_t k, uint32_t* bits, uint32_t* dep, uint32_t* host_dep, uint32_t batch, hipStream_t s) {
    (void)hipMemsetAsync(bits, 0, sizeof(uint32_t) * batch, s);
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
