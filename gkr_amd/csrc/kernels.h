// Launchers of the CDNA4 kernels (kernels.hip).  Internal to the library; the
// public surface is include/gkr_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "fr32.h"

namespace gkr {

constexpr uint32_t kMaxBlocksPerTable = 2048;
constexpr uint32_t kMaxLayerBlocks = 2048;

struct MlePartial {
    Acc<9> lo, hi;
    uint32_t dep, pad;
};

// a batch of layer sumchecks that share their gates (proofs of one circuit): per-proof strides of the
// predicate tables, the W copies and the partials; batch = 1 with zero strides is the single sumcheck
struct LayerBatch {
    uint32_t batch;
    uint32_t pstride;
    size_t tstride, wstride;
};
inline LayerBatch single_layer() { return LayerBatch{1u, 0u, 0, 0}; }

struct LayerPartial {
    Acc<9> c0, g1, c2;
    uint32_t pad;
};

// pinned-host hand-off records of the host transcript (128 B each, seq written last)
struct MleHostRec {
    Fr c0, c1;
    uint32_t dep, seq;
    uint32_t pad[14];
};
struct LayerHostRec {
    Fr c0, g1, c2;
    uint32_t seq;
    uint32_t pad[7];
};
static_assert(sizeof(MleHostRec) == 128 && sizeof(LayerHostRec) == 128, "hand-off records are one 128-byte line");

// Multi-round passes of the layer sumcheck's product form (kernels.hip, "product passes"): a pass hands the host the
// 4^J cross sums of the sub-blocks of W and X and the 2^J sub-block sums of Y, J <= kProdMaxJ rounds' worth.
constexpr int kProdMaxJ = 3;
constexpr int kProdRecValues = 72;   // m[a * 8 + b], a, b < 2^J; then sy[a] at 64 + a
struct ProdPassRec {
    Fr v[kProdRecValues];
    uint32_t seq;
    uint32_t pad[7];
};
static_assert(sizeof(ProdPassRec) == kProdRecValues * 32 + 32, "product-pass record layout");
// entries per block of a pass (one tile) and blocks per proof for tables whose sub-blocks have S entries
constexpr uint32_t kProdTile = 8, kProdTileWide = 32;
inline uint32_t prod_pass_tile(uint32_t S) { return S >= 1024u ? kProdTileWide : kProdTile; }
inline uint32_t prod_pass_blocks(uint32_t S) { return S <= kProdTile ? 1u : S / prod_pass_tile(S); }
// most blocks per proof any pass over tables of 2^k entries takes (sub-blocks of 1 .. 2^(k-1) entries)
inline uint32_t prod_pass_max_blocks(uint32_t k) {
    uint32_t m = 1;
    for (uint32_t l = 0; l < k; ++l) m = prod_pass_blocks(1u << l) > m ? prod_pass_blocks(1u << l) : m;
    return m;
}
// Fr values of pass scratch per proof for tables of 2^k entries (the blocks' partials, and behind them the first-level
// totals of passes with more than 1024 blocks)
inline size_t prod_pass_scratch_values(uint32_t k) { return ((size_t)prod_pass_max_blocks(k) + 64u) * kProdRecValues; }
// One pass on the tables W (Montgomery), X, Y (canonical) of 2^m_in entries per proof (stride wstride): bind the jp
// variables of the previous pass with the 2^jp Montgomery weights at weights + proof * 8 (in place; jp = 0: none), then
// the cross sums for the next J rounds -> the pinned records (seq = ticket, system-scope release); partials: scratch of
// batch x prod_pass_blocks(2^(m_in - jp - 1)) x 72 values for passes that span several blocks per proof.
// arrivals (may be null): `batch` zeroed words; passes of 2 .. kProdFuseBlocks blocks per proof then publish from their last
// block instead of a second launch (the words are zero again when the kernel ends).
// tail (may be null; pinned host memory, batch x 3 tables x tail_stride entries): the pass also leaves there the tables its
// rounds work on (2^(m_in - jp) <= tail_stride entries each, the pending fold applied) -- the host runs the rest of the
// phase's passes itself (capi_layer.hip, host tail).
// fold_plans (may be null): batch * prod_fold_plan_bytes() of device memory; passes with three variables pending over tables
// of 2^kProdFoldMinM entries and more then fold on the matrix cores first (unless the option no_mfma_cross is set).
constexpr uint32_t kProdFuseBlocks = 64;
constexpr uint32_t kProdFoldMinM = 17;
size_t prod_fold_plan_bytes();
void launch_prod_pass(Fr* W, Fr* X, Fr* Y, uint32_t m_in, uint32_t jp, const Fr* weights, uint32_t J, Fr* partials, uint32_t wstride,
                      ProdPassRec* rec, uint32_t ticket, uint32_t batch, hipStream_t s, uint32_t* arrivals = nullptr, void* fold_plans = nullptr,
                      Fr* tail = nullptr, uint32_t tail_stride = 0);
// start of the c-phase: W(u) = sum_b w_b Wb[b] over the 2^jp entries left of Wb, then X = A + W(u) M, Y = W(u) A over
// the 2^k entries of the rows A, M (gate_rows), per proof
void launch_prod_c_setup(const Fr* Wb, uint32_t jp, const Fr* weights, const Fr* A, const Fr* M, Fr* X, Fr* Y, uint32_t k, uint32_t wstride,
                         uint32_t batch, hipStream_t s);

constexpr uint32_t kSmallFoldQuarter = 2048;   // rounds whose output half has <= this many entries use one block per sumcheck
void launch_mle_fold_sum_small(const Fr* src, size_t src_stride, Fr* dst, size_t dst_stride, uint32_t q, uint32_t batch,
                               const FixedMul* rtab, MleHostRec* host_rec, uint32_t ticket, hipStream_t s);
void launch_mle_round_reduce(const MlePartial* partials, uint32_t nblk, uint32_t batch, MleHostRec* host_rec,
                             uint32_t ticket, hipStream_t s);

// multi-round passes (see kernels.hip)
struct MleSubPartial {
    Acc<9> sum;
    uint32_t dep, pad[2];
};
constexpr int kMlePassMaxRounds = 5;                   // rounds one pass can cover (matrix-core fold: mfma_fold.h)
constexpr int kMleMaxSub = 1 << kMlePassMaxRounds;     // sub-block sums / fold weights per sumcheck and pass
struct MleHostRecSub {   // pinned hand-off record: up to 32 sub-block sums, seq written last
    Fr sums[kMleMaxSub];
    uint32_t dep, seq;
    uint32_t pad[14];
};
static_assert(sizeof(MleHostRecSub) == 1088, "hand-off record layout");
// a pass that publishes its own sums (latency-bound passes: the last block to arrive totals the partials; arrivals == nullptr:
// launch_mle_sub_reduce does it)
struct MlePublish {
    MleHostRecSub* rec = nullptr;   // the first sumcheck of the launch
    uint32_t* arrivals = nullptr;   // one zeroed counter per sumcheck of the launch; left zero
    uint32_t ticket = 0, jout = 0;
};
constexpr uint32_t kSmallPassEntries = 512;   // passes whose output has <= this many entries run as one block per sumcheck
uint32_t mle_pass_blocks(uint32_t items, uint32_t jout, uint32_t batch);
uint32_t mle_multifold_blocks(uint32_t S, uint32_t jout, uint32_t batch);
bool mle_multifold_uses_mfma(uint32_t S, uint32_t nblk);
// Every pass: per-block partial sums, then a reduce kernel that writes the pinned host record.
void launch_mle_sub_sums(const Fr* tables, size_t stride, uint32_t len, uint32_t batch, uint32_t nblk, MleSubPartial* partials,
                         hipStream_t s, const MlePublish* publish = nullptr);
void launch_mle_sub_reduce(const MleSubPartial* partials, uint32_t nblk, uint32_t jout, uint32_t batch, MleHostRecSub* host_rec,
                           uint32_t ticket, hipStream_t s);
size_t mle_fold_plan_bytes();
void launch_mle_fold_plan(int jin, const Fr* weights, void* plans, uint32_t batch, hipStream_t s);
void launch_mle_multifold(int jin, const Fr* src, size_t src_stride, Fr* dst, size_t dst_stride, uint32_t S, uint32_t batch,
                          uint32_t nblk, const Fr* weights, const void* plans, MleSubPartial* partials, hipStream_t s,
                          const MlePublish* publish = nullptr);
// the host's share of a pass ON THE DEVICE (kernels_transcript.hip): J rounds of `count` sumchecks from their records
void launch_mle_pass_hash_lanes(const MleHostRecSub* rec, uint32_t count, uint32_t J, uint32_t round0, uint32_t n_out, bool final_pass,
                                bool first_pass, const Fr* cts, uint32_t* dep_last, Fr* weights, Fr* out_coeffs, uint32_t* out_len, Fr* out_r,
                                hipStream_t s);
// tail (may be null; pinned host memory, tail_stride entries per sumcheck): the folded table (S <= tail_stride entries) is left there
// too -- the host binds the remaining variables itself (capi_mle.hip, the host tail of a lone sumcheck).
void launch_mle_multifold_small(int jin, const Fr* src, size_t src_stride, Fr* dst, size_t dst_stride, uint32_t S, uint32_t jout,
                                uint32_t batch, const Fr* weights, MleHostRecSub* host_rec, uint32_t ticket, hipStream_t s, Fr* tail = nullptr, uint32_t tail_stride = 0);
void launch_layer_round_reduce(const LayerPartial* partials, uint32_t nblk, LayerHostRec* host_rec, uint32_t ticket,
                               LayerBatch lb, hipStream_t s);
void launch_fold_small(Fr* W, uint32_t hw, const FixedMul* rtab, LayerBatch lb, hipStream_t s);
// gate lists by left / right operand and the 2^k-entry tables summed straight from them (no dense predicate tables).
// GateSpan: the gate arrays hold gates base .. base + count - 1 of the layer (the whole layer, or one rank's shard).
struct GateSpan {
    uint64_t base, count;
};
// lds_scratch: gate_lists_lds_scratch_words(gates, k) words (or null): large layers sort with block-private LDS
// histograms instead of one global atomic per gate and operand
uint32_t gate_lists_lds_blocks(uint64_t gates, uint32_t k);
size_t gate_lists_lds_scratch_words(uint64_t gates, uint32_t k);
// `list`: 2 * gate_list_words(gates) words -- the gate indices of every bucket (left-operand buckets, then right-operand
// buckets), and after them, entry for entry, what the sums over a bucket need of the gate besides its index: the OTHER
// operand in the low bits, the gate type in bit 31 (read in list order instead of gathered by gate index)
inline size_t gate_list_words(uint64_t gates) { return 2 * (size_t)gates + 1; }
// Segments of the sorted lists (gate_seg.h): the bucket lists cut where gate >> shift changes, so that E_hi is a common
// factor of a segment; segments cut into items of <= kSegCap gates, both halves (left-operand buckets, right-operand
// buckets) in one item array, `order` = the items of each half by decreasing length.  Built once per circuit by
// launch_gate_lists right after the block-private sort (whose per-block starts ARE the segment bounds).
struct GateSegs {
    uint32_t shift = 0;      // log2 gates per run; 0: no segments (small layer, unaligned shard): the bucket kernels run
    uint32_t runs = 0;       // runs the span covers
    uint32_t run_base = 0;   // the span's first run in the layer (index into E_hi)
    uint32_t bound = 0;      // capacity of items / order (both halves together)
    uint32_t half_bound = 0; // most items one half can have
    uint32_t groups = 0;     // most groups (64 items, one wave's work) one half can have
    uint32_t nb = 0;         // 2^k buckets per half
    uint64_t packed_half = 0;   // capacity of one half of `packed`, in entries
    uint32_t* words = nullptr;  // one allocation of gate_segs_words()
    // items | order | bucket_begin | group_len | group_off | packed
    uint2* items() const { return reinterpret_cast<uint2*>(words); }      // {first list entry, len | run << 8}
    uint32_t* order() const { return words + 2 * (size_t)bound; }          // the items of each half by decreasing length
    uint32_t* bucket_begin() const { return order() + bound; }             // 2 nb + 1 item offsets (+ 1 pad)
    uint32_t* group_len() const { return bucket_begin() + 2 * (size_t)nb + 2; }   // 2 * groups: gates of a group's longest item
    uint32_t* group_off() const { return group_len() + 2 * (size_t)groups; }      // 2 * groups + 1: first entry / 64 of the group in `packed`
    // The list entries once more, in the order the pass reads them: group after group, inside a group step-major --
    // entry (step j, lane l) = gate j of the group's item l -- so that a wave's load of one step is 256 contiguous bytes
    // (per-lane walks through list / meta cost one cache line per gate and array).
    // entry = (gate & mask) | other operand << shift | type << 31   (shift + k <= 31)
    uint32_t* packed() const { return group_off() + 2 * (size_t)groups + 2; }
};
// the split point of eq(z, g) = E_hi[g >> shift] * E_lo[g & mask] the gate passes of this span use: k_i / 2 for the
// bucket kernels, the segment shift where segments apply (a function of the span and the widths only)
uint32_t gate_seg_shift(GateSpan span, uint32_t k_i, uint32_t k);
size_t gate_segs_words(GateSpan span, uint32_t k_i, uint32_t k);           // u32 words of GateSegs::words (0: no segments)
size_t gate_segs_scratch_words(GateSpan span, uint32_t k_i, uint32_t k);   // u32 words of build scratch
size_t gate_seg_partial_elems(GateSpan span, uint32_t k_i, uint32_t k);    // Fr elements of pass scratch per proof
void launch_gate_lists(GateSpan span, uint32_t k_i, uint32_t k, const uint8_t* gate_type, const uint32_t* left, const uint32_t* right,
                       uint32_t* counts, uint32_t* offsets, uint32_t* cursor, uint32_t* block_sums, uint32_t* list, uint32_t* bad,
                       uint32_t* lds_scratch, GateSegs* segs, uint32_t* seg_scratch, hipStream_t s);
// The sorted gate lists (and gate arrays) of ONE circuit layer on the device, as the passes over the gates read them.  A pass
// over a batch of proofs of one circuit takes the pointers as launch arguments; a pass over proofs of DIFFERENT circuits whose
// layers have the same shape (gkr_prove_many's lockstep groups: the reference's par_iter over the (circuit, input) pairs of a
// step, aggregator.rs:411-416, as ONE launch per pass) takes `sets`, a device table with one entry per proof of the launch,
// and ignores the pointer arguments.
struct GateSet {
    const uint32_t* offsets;      // 2 * 2^k bucket starts
    const uint32_t* cursor;       // 2 * 2^k bucket ends
    const uint32_t* list;         // gate indices by bucket; the entries' meta words follow at list + gate_list_words(gates)
    const uint32_t* plan;         // the wide layers' item plan (gate_plan_words; may be null)
    const uint8_t* gate_type;     // the layer's gate arrays (layer evaluation)
    const uint32_t* left;
    const uint32_t* right;
    uint64_t pad;
};
static_assert(sizeof(GateSet) == 64, "one cache line per proof");
// e_hi / e_lo_mont split at `kl` = gate_seg_shift() when segs->shift != 0 (partials: gate_seg_partial_elems() * batch
// elements of scratch), else at any kl
void launch_gate_uv(GateSpan span, uint32_t k_i, uint32_t k, const uint32_t* offsets, const uint32_t* cursor, const uint32_t* list,
                    const Fr* e_hi, const Fr* e_lo_mont, uint32_t kl, const Fr* W, Fr* U, Fr* V, LayerBatch lb, const GateSegs* segs,
                    Fr* partials, hipStream_t s, const GateSet* sets = nullptr);
// the c-phase's set-up done by the row pass itself (launch_prod_c_setup's arguments; Wb null: not fused)
struct CPhaseFuse {
    const Fr* Wb = nullptr;
    const Fr* weights = nullptr;
    Fr* X = nullptr;
    Fr* Y = nullptr;
    uint32_t jp = 0;
};
// fuse (may be null): also write X = a_u + W(u) m_u, Y = W(u) a_u; returns true when it did (the segment form), false when
// the caller still has to launch_prod_c_setup (the bucket form)
bool launch_gate_rows(GateSpan span, uint32_t k_i, uint32_t k, const uint32_t* offsets, const uint32_t* cursor, const uint32_t* list,
                      const Fr* e_hi, const Fr* e_lo_mont, uint32_t kl, const Fr* eq_mont, Fr* A_row, Fr* M_row, LayerBatch lb,
                      const GateSegs* segs, Fr* partials, hipStream_t s, const CPhaseFuse* fuse = nullptr, const GateSet* sets = nullptr);
void launch_fill_table(Fr* table, size_t count, uint64_t seed, hipStream_t s);
void launch_fill_shard(Fr* shard_table, size_t count, uint32_t lp, uint32_t shard, uint64_t seed, hipStream_t s);
// sum over ranks without leaving the device: a, b (each `each` elements) + one flag element <-> (2 each + 1) x 8 int64 limbs
void launch_exchange_widen(const Fr* a, const Fr* b, uint32_t each, const uint32_t* flag, uint32_t local_flag, long long* limbs, hipStream_t s);
void launch_exchange_narrow(const long long* limbs, Fr* a, Fr* b, uint32_t each, uint32_t* flag_out, hipStream_t s);
// ceilings of the box (gkr_ubench_ceilings): a plain copy, a read-only stream, chains of Montgomery products
void launch_ubench_copy(const void* in, void* out, size_t bytes, hipStream_t s);
void launch_ubench_read(const void* in, void* out, size_t bytes, hipStream_t s);
void launch_ubench_modmul(Fr* io, uint32_t entries_pow2, uint32_t waves, int reps, hipStream_t s);

uint32_t mle_blocks_per_table(uint32_t items, uint32_t batch);
void launch_mle_sum_first(const Fr* tables, size_t stride, uint32_t h, uint32_t batch, uint32_t nblk,
                          MlePartial* partials, hipStream_t s);
void launch_mle_fold_sum(const Fr* src, size_t src_stride, Fr* dst, size_t dst_stride, uint32_t q, uint32_t batch,
                         uint32_t nblk, const FixedMul* rtab, uint32_t r_stride, MlePartial* partials, hipStream_t s);
void launch_mle_round_hash(const MlePartial* partials, uint32_t nblk, uint32_t round, uint32_t n, uint32_t batch,
                           const Fr* cts, Fr* out_coeffs, uint32_t* out_len, Fr* out_r, FixedMul* rtab,
                           uint32_t* dep_last, hipStream_t s);

void launch_layer_eval(uint32_t gates, const uint8_t* gate_type, const uint32_t* left, const uint32_t* right,
                       const Fr* prev, Fr* out, uint32_t batch, uint32_t prev_stride, hipStream_t s, const GateSet* sets = nullptr);
// words 32-bit words src -> dst (16-byte aligned when words >= 4); either side may be pinned host memory
void launch_copy_words(const void* src, void* dst, size_t words, hipStream_t s);
// rows x words: row r from src + r * src_stride_words to dst + r * dst_stride_words (32-bit words)
void launch_copy_rows(const void* src, size_t src_stride_words, void* dst, size_t dst_stride_words, uint32_t words, uint32_t rows, hipStream_t s);
// out[proof][g] = eq(points[proof * stride + first ..+nvars), g), nvars <= 28; points may be pinned host memory
// wu_job / wu_out (both or neither): the launch also leaves W(u) = sum_{i < 2^jp} Wb[i] * weights[i] of every proof (Montgomery;
// k_prod_c_setup's scalar) in wu_out[proof] -- the wide layers' row pass writes the c-phase's tables with it (WideCFuse)
struct CPhaseFuse;
void launch_eq_table(const Fr* points, uint32_t stride, uint32_t first, uint32_t nvars, Fr* out, bool montgomery, uint32_t batch,
                     hipStream_t s, const CPhaseFuse* wu_job = nullptr, Fr* wu_out = nullptr, uint32_t wstride = 0);
// one launch for a layer's set-up: E_hi (canonical, kh leading coordinates of the proof's point), E_lo (Montgomery, kl
// trailing ones), the two Montgomery copies of W (Wb null: none) and the 32 dependence flags per proof, in dep and -- when not null -- in
// pinned host memory host_dep (k_layer_prologue)
void launch_layer_prologue(const Fr* points, uint32_t k_i, uint32_t kh, uint32_t kl, Fr* e_hi, Fr* e_lo, const Fr* W, Fr* Wb, Fr* Wc,
                           uint32_t k, uint32_t* dep, uint32_t* host_dep, uint32_t batch, hipStream_t s, uint32_t* wide_bits = nullptr);
// q(t) = W(b + t (c - b)) per proof: W batch x 2^k, bc batch x 2k (b then c), scratch batch x 3 * 2^k, deg_scratch batch
// words (device), out batch x (k + 1) highest degree first, out_len batch
// bcm: batch * 2k elements of device scratch (the line's coefficients in Montgomery form; unused for k <= 9)
// part (layers of more than 2^12 values only; smaller ones ignore it and must be called with `all`): `prepare` issues what needs
// W alone (the copy, the Moebius transform, the largest degree), `finish` the rest (the line's coefficients, the steps, q).
enum class LinePart { all, prepare, finish };
void launch_line_restriction(const Fr* W, uint32_t k, const Fr* bc, Fr* scratch, uint32_t* deg_scratch, Fr* bcm, Fr* out, uint32_t* out_len,
                             uint32_t batch, hipStream_t s, LinePart part = LinePart::all);

// ---- wide layers (kernels_wide.hip): next-layer tables of 2^14 values and more -------------------------------------------
// The gate passes over ITEMS: the buckets of the sorted lists cut into pieces of at most sixteen gates, sorted by length, a
// lane per item; eq(z, .) gathered from a materialised 2^k_i-entry table E (canonical).  The plan (items, their order, the
// step-major entries) is built once per circuit layer by launch_gate_plan, right after the sort.
constexpr uint32_t kWideMinK = 13;        // layers with k_next >= this take the item passes
size_t gate_plan_words(uint64_t gates, uint32_t k);             // u32 words of the plan (both halves)
size_t gate_plan_partial_elems(uint64_t gates, uint32_t k);     // Fr elements of pass scratch per proof
void launch_gate_plan(GateSpan span, uint32_t k, const uint32_t* offsets, const uint32_t* cursor, const uint32_t* list, uint32_t* plan, hipStream_t s);
// eq(z, .) for the passes: E (batch x 2^k_i entries, canonical) when materialised, else null and the two half tables of the
// layer's prologue (e_hi canonical, batch x 2^(k_i - kl); e_lo_mont Montgomery, batch x 2^kl): one reduced product per gate.
// W / eq_mont: batch tables of 2^k entries in Montgomery form (stride lb.wstride).
struct GateEq {
    const Fr* E;
    const Fr* e_hi;
    const Fr* e_lo_mont;
    uint32_t kl;
};
// eq(z, g) per gate: gathered from the materialised 2^k_i-entry table up to 2^20 gates (32 MiB per proof), formed from the two
// half tables (one reduced product per gate, both halves cache-resident) beyond.  Same-box A/B, ms of gate passes + table
// build per sumcheck, table / halves (profiles/r06/e_wide_eq_source_*): (20,15) 0.113 / 0.134, (20,20) 0.174 / 0.155, circom-
// shaped (20,20) 0.195 / 0.204, (21,16) 0.184 / 0.153, (21,21) 0.327 / 0.288, (22,16) 0.306 / 0.238, (22,22) 0.697 / 0.575 -- the
// table's random 32-byte reads move 128-byte lines, and from 64 MiB on that costs more than the product (round 5 kept the table
// up to k_i = 22, while it stays in the 256 MiB Infinity Cache).
constexpr uint32_t kGateEqTableMaxKi = 20;
// The plan's counts as the host read them back when the plan was built (hdr[half]: [1] groups, [2] buckets of 2 .. 64 items, [3]
// buckets of more, [5] their chunks): exact grids, and no combine launch for a half whose buckets are all one item.  In a lockstep
// group: the largest count over the members.
struct GatePlanCounts {
    uint32_t hdr[2][8];
    bool known;
};
size_t gate_plan_arrive_words(uint64_t gates, uint32_t k);     // zeroed u32 counters per proof (the combine step's long buckets)
void gate_plan_counts_offsets(uint64_t gates, uint32_t k, size_t* half1_word_offset);   // where the second half's header starts in the plan
void launch_gate_uv_wide(GateSpan span, uint32_t k_i, uint32_t k, const uint32_t* plan, const GateEq& eq, const Fr* W, Fr* U, Fr* V, LayerBatch lb, Fr* partials,
                         uint32_t* arrive, hipStream_t s, const GateSet* sets = nullptr, const GatePlanCounts* counts = nullptr);
// fuse (may be null): the rows are not written; every bucket's (a_u, m_u) goes straight into the c-phase's tables
// X = a_u + W(u) m_u, Y = W(u) a_u (k_prod_c_setup's arithmetic: one launch and 64 B of traffic per bucket less per sumcheck)
struct WideCFuse {
    const Fr* wu = nullptr;   // W(u) per proof, Montgomery, device memory (launch_eq_table's wu_out)
    Fr* X = nullptr;
    Fr* Y = nullptr;
};
void launch_gate_rows_wide(GateSpan span, uint32_t k_i, uint32_t k, const uint32_t* plan, const GateEq& eq, const Fr* eq_mont, Fr* A_row, Fr* M_row, LayerBatch lb,
                           Fr* partials, uint32_t* arrive, hipStream_t s, const GateSet* sets = nullptr, const GatePlanCounts* counts = nullptr,
                           const WideCFuse* fuse = nullptr);
// E (batch x 2^k_i, canonical) = the outer product of the prologue's half tables (e_hi canonical, e_lo_mont Montgomery, split at kl)
void launch_eq_outer(const Fr* e_hi, const Fr* e_lo_mont, uint32_t k_i, uint32_t kl, Fr* E, uint32_t batch, hipStream_t s);
// exclusive scan of n 32-bit counts (block_sums: n / 2048 + 1 words of scratch)
void launch_exclusive_scan(const uint32_t* in, uint32_t* out, uint32_t* block_sums, size_t n, hipStream_t s);
// *flag |= 1 if some entry of t[0 .. n) is >= r (flag zeroed by the caller)
void launch_check_canonical(const Fr* t, size_t n, uint32_t* flag, hipStream_t s);
// dependence flags over a grid: bits = one word of scratch per proof, zeroed by the launcher unless `bits_preset` (the layer's
// prologue stored what a first look found: launch_layer_prologue's wide_bits); dep / host_dep as launch_layer_prologue
void launch_depends_wide(const Fr* W, uint32_t k, uint32_t* bits, uint32_t* dep, uint32_t* host_dep, uint32_t batch, hipStream_t s, bool bits_preset = false);
// in-place Moebius transform (evaluations -> monomial coefficients, MSB-first) of `batch` tables of 2^k, `stride` apart
void launch_mobius(Fr* tables, uint32_t k, size_t stride, uint32_t batch, hipStream_t s);
void launch_line_setup_wide(const Fr* W, uint32_t k, Fr* scratch, uint32_t* maxdeg_scratch, uint32_t batch, hipStream_t s);

// ---- one plain sumcheck split over ranks (kernels_wide.hip): the per-pass exchange of the sub-block sums and the final gather
// rec: the pass's records in DEVICE memory; limbs: batch x (2^J + 2) x 8 int64 (sums | dep | fail per table)
void launch_mle_xwiden(const MleHostRecSub* rec, uint32_t J, uint32_t batch, uint32_t local_fail, long long* limbs, hipStream_t s);
void launch_mle_xnarrow(const long long* limbs, uint32_t J, uint32_t batch, MleHostRecSub* host_rec, uint32_t ticket, uint32_t* fail_out,
                        hipStream_t s);
// the 2^t entries every shard has left -> the tail table of 2^(t + lp) entries (limbs: batch x 2^(t+lp) x 8 + 8 int64)
void launch_mle_gather_widen(const Fr* src, size_t stride, uint32_t t, uint32_t lp, uint32_t shard, uint32_t local_fail, uint32_t batch,
                             long long* limbs, hipStream_t s);
void launch_mle_gather_narrow(const long long* limbs, uint32_t tn, uint32_t batch, Fr* tail, uint32_t* fail_out, hipStream_t s);
void launch_to_mont(const Fr* in, Fr* out, uint32_t count, hipStream_t s);
void launch_depends(const Fr* W, uint32_t k, uint32_t* dep, uint32_t batch, hipStream_t s);
void launch_predicate_scatter(uint32_t k_i, uint32_t k_next, const uint8_t* gate_type, const uint32_t* left,
                              const uint32_t* right, const Fr* e_hi, const Fr* e_lo_mont, uint32_t kl,
                              unsigned long long* wideA, unsigned long long* wideM, uint32_t* bad, uint32_t log_p,
                              uint32_t shard, hipStream_t s);
void launch_to_mont_strided(const Fr* in, Fr* out, uint32_t count, uint32_t stride, uint32_t offset, hipStream_t s);
void launch_tables_differ(const Fr* a, const Fr* b, size_t count, uint32_t* flag, hipStream_t s);
void launch_fold_pair(const Fr* src, Fr* dst, const FixedMul* rtab, hipStream_t s);
void launch_predicate_sorted(uint32_t k_i, uint32_t k_next, const uint8_t* gate_type, const uint32_t* left,
                             const uint32_t* right, const Fr* e_hi, const Fr* e_lo_mont, uint32_t kl, uint32_t log_p,
                             uint32_t shard, size_t ncells, uint32_t* counts, uint32_t* offsets, uint32_t* cursor,
                             uint32_t* block_sums, uint32_t* list, uint32_t* bad, Fr* out_A, Fr* out_M, uint32_t batch,
                             hipStream_t s);
void launch_predicate_normalise(const unsigned long long* wide, Fr* out, size_t cells, hipStream_t s);

uint32_t layer_blocks(uint32_t h);
void launch_layer_round(const Fr* A, const Fr* M, uint32_t h, uint32_t k, uint32_t phase, uint32_t hb, const Fr* Wb,
                        const Fr* Wc, uint32_t nblk, LayerPartial* partials, LayerBatch lb, hipStream_t s);
uint32_t launch_layer_round_b(bool fold, const Fr* A_src, const Fr* M_src, Fr* A_dst, Fr* M_dst, uint32_t hb, uint32_t kc,
                              const FixedMul* rtab, const Fr* Wb, const Fr* Wc, LayerPartial* partials, LayerBatch lb,
                              hipStream_t s);
void launch_layer_fold(Fr* A, Fr* M, uint32_t h, const FixedMul* rtab, LayerBatch lb, hipStream_t s);
void launch_layer_round_hash(const LayerPartial* partials, uint32_t nblk, uint32_t round, uint32_t k,
                             const uint32_t* dep, const Fr* cts, Fr* out_coeffs, uint32_t* out_len, Fr* out_r,
                             FixedMul* rtab, Fr* Wb, Fr* Wc, hipStream_t s);

}  // namespace gkr
