// The GKR layer sumcheck (prove_sumcheck_opt, rust/src/gkr/sumcheck.rs:36-156): predicate tables, the linear-time form over gate
// lists with product passes, the gate-sharded form, resident layers, the dense step-wise sessions.  C ABI: include/gkr_amd.h.
#include "capi_internal.h"

namespace gkr_host {

void host_prod_pass_scalar(const uint64_t* recs, size_t rec_row_words, int count, int J, const uint32_t (*vec_len)[16],
                                  uint64_t (*c2)[16][4], uint64_t (*lin)[16][4], uint64_t (*c0)[16][4], uint64_t (*r)[16][4],
                                  uint64_t* weights, size_t w_row_words) {
    using namespace gkr::h64;
    const F* cts = host_mimc_constants64();
    const F one_m = to_mont(F{{1, 0, 0, 0}});
    for (int k = 0; k < count; ++k) {
        F M[64], SY[8], rm[gkr::kProdMaxJ];
        const F* rec = reinterpret_cast<const F*>(recs + (size_t)k * rec_row_words);
        const int n = 1 << J;
        for (int a = 0; a < n; ++a) {
            for (int b = 0; b < n; ++b) M[a * 8 + b] = rec[a * 8 + b];
            SY[a] = rec[64 + a];
        }
        for (int t = 0; t < J; ++t) {
            const int half = 1 << (J - t - 1);
            F p00 = M[0], p01 = M[half], p10 = M[half * 8], p11 = M[half * 8 + half], s0 = SY[0], s1 = SY[half];
            for (int x = 1; x < half; ++x) {
                p00 = add(p00, M[x * 8 + x]);
                p01 = add(p01, M[x * 8 + half + x]);
                p10 = add(p10, M[(half + x) * 8 + x]);
                p11 = add(p11, M[(half + x) * 8 + half + x]);
                s0 = add(s0, SY[x]);
                s1 = add(s1, SY[half + x]);
            }
            const F vc0 = add(p00, s0), g1 = add(p11, s1);
            const F vc2 = sub(add(p11, p00), add(p10, p01));
            const F vlin = sub(sub(g1, vc0), vc2);
            const uint32_t ln = vec_len[t][k];
            const F vec[3] = {vc2, vlin, vc0};
            const F rc = host_multi_hash(vec + (3 - ln), (int)ln, cts);
            memcpy(c2[t][k], &vc2, 32);
            memcpy(lin[t][k], &vlin, 32);
            memcpy(c0[t][k], &vc0, 32);
            memcpy(r[t][k], &rc, 32);
            rm[t] = to_mont(rc);
            for (int ra = 0; ra < half; ++ra)
                for (int cb = 0; cb < 2 * half; ++cb) M[ra * 8 + cb] = add(M[ra * 8 + cb], mont_mul(sub(M[(half + ra) * 8 + cb], M[ra * 8 + cb]), rm[t]));
            for (int ra = 0; ra < half; ++ra)
                for (int cb = 0; cb < half; ++cb) M[ra * 8 + cb] = add(M[ra * 8 + cb], mont_mul(sub(M[ra * 8 + half + cb], M[ra * 8 + cb]), rm[t]));
            for (int ra = 0; ra < half; ++ra) SY[ra] = add(SY[ra], mont_mul(sub(SY[half + ra], SY[ra]), rm[t]));
        }
        if (!weights) continue;
        F tmp[8];
        tmp[0] = one_m;
        int cur = 1;
        for (int t = 0; t < J; ++t) {
            const F nr = sub(one_m, rm[t]);
            for (int b = cur; b-- > 0;) {
                tmp[2 * b + 1] = mont_mul(tmp[b], rm[t]);
                tmp[2 * b] = mont_mul(tmp[b], nr);
            }
            cur <<= 1;
        }
        memcpy(weights + (size_t)k * w_row_words, tmp, sizeof(F) << J);
    }
}

// The HOST TAIL of a phase's product passes.  Once the tables are small (2^host_tail_log2 entries and fewer) a device pass is a
// latency chain -- launch, ~15 us of kernel for a few hundred products, the record's way back -- and costs more than the
// products do on one host core.  The last device pass of the phase therefore also leaves the three tables in pinned memory
// (k_prod_cross's `tail`), and the host does what the later passes' kernels would: binds the previous pass's variables
// (T'[i] = sum_b w_b T[b S + i], the weights in Montgomery form) and forms the next rounds' record (m[a][b] and the sub-block
// sums of Y, as k_prod_cross defines them) -- exact field arithmetic, the same canonical values.  W is in Montgomery form and
// stays so; X and Y are canonical.
// tables: [3][stride] (W, X, Y) of 2^m entries each, folded in place to 2^(m - jp); rec: the record's 72 values
void host_tail_pass_scalar(gkr::h64::F* tables, size_t stride, uint32_t m, uint32_t jp, const gkr::h64::F* weights, uint32_t J, gkr::h64::F* rec) {
    using namespace gkr::h64;
    const uint32_t mf = m - jp, len = 1u << mf;
    if (jp) {   // (sums of 2^jp products with one reduction each: wide_mac / wide_reduce, fr64.h)
        for (int t = 0; t < 3; ++t) {
            F* T = tables + (size_t)t * stride;
            for (uint32_t i = 0; i < len; ++i) {
                Wide acc = wide_zero();
                for (uint32_t b = 0; b < (1u << jp); ++b) wide_mac(acc, T[((size_t)b << mf) + i], weights[b]);
                T[i] = wide_reduce(acc);
            }
        }
    }
    const uint32_t nsub = 1u << J, S = len >> J;
    const F *W = tables, *X = tables + stride, *Y = tables + 2 * stride;
    for (uint32_t a = 0; a < nsub; ++a) {
        for (uint32_t b = 0; b < nsub; ++b) {
            Wide acc = wide_zero();
            for (uint32_t i = 0; i < S; ++i) wide_mac(acc, W[a * S + i], X[b * S + i]);
            rec[a * 8 + b] = wide_reduce(acc);
        }
        F y = Y[a * S];
        for (uint32_t i = 1; i < S; ++i) y = add(y, Y[a * S + i]);
        rec[64 + a] = y;
    }
}

// (eight products per instruction group where the CPU has AVX-512 IFMA: mimc_ifma.cpp, gkr_ifma_tail_pass -- the same values)
void host_tail_pass(gkr::h64::F* tables, size_t stride, uint32_t m, uint32_t jp, const gkr::h64::F* weights, uint32_t J, gkr::h64::F* rec) {
    if (host_ifma_ready())
        gkr::gkr_ifma_tail_pass(&tables[0].l[0], stride, m, jp, weights ? &weights[0].l[0] : nullptr, J, &rec[0].l[0]);
    else
        host_tail_pass_scalar(tables, stride, m, jp, weights, J, rec);
}

// Host transcript, default schedule (kernels.hip "Multi-round passes"): a pass hands the host the
// 2^J sub-block sums of the current table; the host runs J rounds on them (J <= 5 hashes in a row,
// eight or sixteen sumchecks per IFMA call), derives the 2^J fold weights, and the next pass binds all J
// variables at once.  Length rules as in run_mle_batch.
// ------------------------------------------------------------- predicate tables
// builds canonical A, M (2^{2k} each) in device memory from device gate arrays

// shard (log_p, p) keeps the gates whose right operand has low bits p; tables then have 2^{2k - log_p} entries.
// batch > 1: `batch` proofs of one circuit -- same gates (the cell lists are built once), z is batch x k_i,
// d_A / d_M hold batch tables of N entries each.
// E[g] = eq(z, g) = E_hi[g >> kl] * E_lo[g & mask]: two small tables per proof, built on the device from the points
// the host left in pinned memory (k_eq_table), E_lo in Montgomery form so that the product of the two is canonical.
static int upload_eq_tables(gkr_ctx* ctx, int k_i, const gkr_fr* z, int batch, Fr** e_hi_out, Fr** e_lo_out, int kl = -1) {
    if (kl < 0) kl = k_i / 2;
    const int kh = k_i - kl;
    Fr *e_hi = nullptr, *e_lo = nullptr;
    WS(ctx, "pred.ehi", Fr, (size_t)batch << kh, e_hi);
    WS(ctx, "pred.elo", Fr, (size_t)batch << kl, e_lo);
    // the points go to pinned memory, the tables are built on the device from there (k_eq_table): no transfer call
    gkr_fr* hz = nullptr;
    HIP_TRY(ctx, ctx->pinned_host("pred.z", sizeof(gkr_fr) * (size_t)batch * (k_i ? k_i : 1), reinterpret_cast<void**>(&hz)));
    memcpy(hz, z, sizeof(gkr_fr) * (size_t)batch * k_i);
    gkr::launch_eq_table(reinterpret_cast<const Fr*>(hz), (uint32_t)k_i, 0u, (uint32_t)kh, e_hi, false, (uint32_t)batch, ctx->stream);
    gkr::launch_eq_table(reinterpret_cast<const Fr*>(hz), (uint32_t)k_i, (uint32_t)kh, (uint32_t)kl, e_lo, true, (uint32_t)batch, ctx->stream);
    *e_hi_out = e_hi;
    *e_lo_out = e_lo;
    return GKR_OK;
}

int build_predicates(gkr_ctx* ctx, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r,
                     const gkr_fr* z, Fr* d_A, Fr* d_M, uint32_t log_p = 0, uint32_t shard = 0, int batch = 1) {
    const size_t N = (size_t)1 << (2 * k - log_p);
    hipStream_t s = ctx->stream;
    Fr *e_hi = nullptr, *e_lo = nullptr;
    uint32_t* bad = nullptr;
    const int kl = k_i / 2;
    WS(ctx, "pred.bad", uint32_t, 1, bad);
    {
        const int rc_eq = upload_eq_tables(ctx, k_i, z, batch, &e_hi, &e_lo);
        if (rc_eq) return rc_eq;
    }
    HIP_TRY(ctx, hipMemsetAsync(bad, 0, 4, s));
    const bool use_atomics = gkr::opt(gkr::OPT_predicate_atomics) != 0;
    if (!use_atomics || batch > 1) {
        // counting sort by cell, then one modular sum per cell (per proof)
        uint32_t *counts = nullptr, *offsets = nullptr, *cursor = nullptr, *bsums = nullptr, *list = nullptr;
        WS(ctx, "pred.counts", uint32_t, 2 * N, counts);
        WS(ctx, "pred.offsets", uint32_t, 2 * N, offsets);
        WS(ctx, "pred.cursor", uint32_t, 2 * N, cursor);
        WS(ctx, "pred.bsums", uint32_t, (2 * N + 2047) / 2048 + 1, bsums);
        WS(ctx, "pred.list", uint32_t, (size_t)1 << k_i, list);
        HIP_TRY(ctx, hipMemsetAsync(counts, 0, 2 * N * sizeof(uint32_t), s));
        Timed t(ctx, "predicate_sorted", (double)((size_t)1 << k_i) * (2 * 9.0 + 8.0) + (double)N * 2.0 * (3 * 4.0 + 32.0) * batch);
        gkr::launch_predicate_sorted(k_i, k, d_gt, d_l, d_r, e_hi, e_lo, (uint32_t)kl, log_p, shard, N, counts, offsets, cursor,
                                     bsums, list, bad, d_A, d_M, (uint32_t)batch, s);
    } else {
        // widened-atomic scatter (kept for comparison): 8 u64 limb atomics per gate into 64-byte cells
        unsigned long long *wideA = nullptr, *wideM = nullptr;
        WS(ctx, "pred.wideA", unsigned long long, N * 8, wideA);
        WS(ctx, "pred.wideM", unsigned long long, N * 8, wideM);
        HIP_TRY(ctx, hipMemsetAsync(wideA, 0, N * 64, s));
        HIP_TRY(ctx, hipMemsetAsync(wideM, 0, N * 64, s));
        {
            Timed t(ctx, "predicate_scatter", (double)((size_t)1 << k_i) * (9.0 + 64.0));
            gkr::launch_predicate_scatter(k_i, k, d_gt, d_l, d_r, e_hi, e_lo, (uint32_t)kl, wideA, wideM, bad, log_p, shard, s);
        }
        {
            Timed t(ctx, "predicate_normalise", (double)N * 2.0 * (64.0 + 32.0));
            gkr::launch_predicate_normalise(wideA, d_A, N, s);
            gkr::launch_predicate_normalise(wideM, d_M, N, s);
        }
    }
    uint32_t hbad = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));   // also keeps the host tables alive until their upload is done
    if (hbad) return ctx->fail(GKR_ERR_INVALID, "gate type or operand index out of range");
    return GKR_OK;
}

// ------------------------------------------------------------- layer sumcheck
// `batch` layer sumchecks that share their gates (the same layer of `batch` proofs of one circuit), or
// one (batch = 1).  z: batch x k_i challenges (host); d_W: batch tables of 2^k canonical values;
// outputs: per proof 2k rows (out_coeffs 3 slots per row), laid out [proof][round] with the given strides.
// One rank's share of a layer split across GPUs by GATES (gkr_sumcheck_layer_sharded): the device gate arrays hold
// gates gate_base .. gate_base + gate_count - 1, and the two tables that are sums over gates -- (U, V) before the
// b-rounds, the row (a_u, m_u) before the c-rounds -- are completed by the caller's sum-over-ranks hook.
int run_layer_batch_impl(gkr_ctx* ctx, int batch, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r,
                         const gkr_fr* z, const Fr* d_W, gkr_fr* const* out_coeffs, uint32_t* const* out_len, gkr_fr* const* out_r,
                         const LayerShardArgs* shard, GateLists* cached, const LayerGroup* group);

// smallest k_next whose layers take the lane-group gate passes of wide layers (kernels_wide.hip)
static bool layer_is_wide(gkr::GateSpan span, int k_i, int k) {
    const int wide_min_k = gkr::opt(gkr::OPT_gate_groups_min_k) >= 0 ? (int)gkr::opt(gkr::OPT_gate_groups_min_k) : (int)gkr::kWideMinK;
    return k >= wide_min_k && gkr::gate_segs_words(span, (uint32_t)k_i, (uint32_t)k) == 0;
}

// the counts a plan's build left in its two half headers -> host (queued; valid after the stream has been waited for)
static hipError_t queue_plan_counts_readback(const uint32_t* plan, uint64_t gates, int k, gkr::GatePlanCounts* out, hipStream_t s) {
    size_t half1 = 0;
    gkr::gate_plan_counts_offsets(gates, (uint32_t)k, &half1);
    hipError_t e = hipMemcpyAsync(out->hdr[0], plan, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(out->hdr[1], plan + half1, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    return e;
}

int build_cached_gate_lists(gkr_ctx* ctx, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r, GateLists* cached) {
    if (!cached) return ctx->fail(GKR_ERR_INVALID, "no list cache");
    if (cached->ready) return GKR_OK;
    const gkr::GateSpan span{0, (uint64_t)1 << k_i};
    if (gkr::gate_segs_words(span, (uint32_t)k_i, (uint32_t)k) != 0) return ctx->fail(GKR_ERR_INVALID, "a layer of the segment form's size in a lockstep group");
    hipStream_t s = ctx->stream;
    const bool wide = layer_is_wide(span, k_i, k);
    const size_t nb2 = (size_t)2 << k;
    uint32_t *g_counts = nullptr, *g_bsums = nullptr, *bad = nullptr, *lds_scratch = nullptr;
    WS(ctx, "pred.bad", uint32_t, 1, bad);
    WS(ctx, "gates.counts", uint32_t, nb2, g_counts);
    WS(ctx, "gates.bsums", uint32_t, (nb2 + 2047) / 2048 + 1, g_bsums);
    if (!cached->offsets) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&cached->offsets), nb2 * sizeof(uint32_t)));
    if (!cached->cursor) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&cached->cursor), nb2 * sizeof(uint32_t)));
    if (!cached->list) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&cached->list), 2 * gkr::gate_list_words(span.count) * sizeof(uint32_t)));
    if (wide && !cached->plan) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&cached->plan), gkr::gate_plan_words(span.count, (uint32_t)k) * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMemsetAsync(bad, 0, 4, s));
    HIP_TRY(ctx, hipMemsetAsync(g_counts, 0, nb2 * sizeof(uint32_t), s));
    if (const size_t words = gkr::gate_lists_lds_scratch_words(span.count, (uint32_t)k)) WS(ctx, "gates.lds", uint32_t, words, lds_scratch);
    {
        Timed t(ctx, "gate_lists", (double)span.count * (9.0 + 4 * 4.0));
        gkr::launch_gate_lists(span, (uint32_t)k_i, (uint32_t)k, d_gt, d_l, d_r, g_counts, cached->offsets, cached->cursor, g_bsums, cached->list, bad, lds_scratch,
                               &cached->segs, nullptr, s);
        if (wide) gkr::launch_gate_plan(span, (uint32_t)k, cached->offsets, cached->cursor, cached->list, cached->plan, s);
    }
    uint32_t hbad = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, s));
    if (wide) HIP_TRY(ctx, queue_plan_counts_readback(cached->plan, span.count, k, &cached->plan_counts, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (hbad) return ctx->fail(GKR_ERR_INVALID, "gate type or operand index out of range");
    cached->plan_counts.known = wide;
    cached->ready = true;
    return GKR_OK;
}

// Gate lists that this call built (cached->ready false on entry) count as ready only if the whole call succeeded: a bad
// gate, a HIP error or a timeout after the sort was queued must not leave half-validated lists marked usable.
int run_layer_batch(gkr_ctx* ctx, int batch, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r,
                    const gkr_fr* z, const Fr* d_W, gkr_fr* const* out_coeffs, uint32_t* const* out_len, gkr_fr* const* out_r,
                    const LayerShardArgs* shard, GateLists* cached, const LayerGroup* group) {
    const bool was_ready = cached && cached->ready;
    if (group && (ctx->transcript != GKR_TRANSCRIPT_HOST || shard || !was_ready)) return ctx->fail(GKR_ERR_INVALID, "a lockstep group needs the host transcript and prepared gate lists");
    int rc = GKR_OK;
    if (ctx->transcript != GKR_TRANSCRIPT_HOST && batch > 1 && !shard) {
        // The device transcript hashes on one lane per sumcheck and its round kernels take one proof: the proofs of a
        // batch go through one after the other (complete and host-free, not fast: ~1 ms per round and proof).
        const size_t wlen = (size_t)1 << k;
        for (int b = 0; b < batch && rc == GKR_OK; ++b)
            rc = run_layer_batch_impl(ctx, 1, k_i, k, d_gt, d_l, d_r, z + (size_t)b * k_i, d_W + (size_t)b * wlen, out_coeffs + b, out_len + b,
                                      out_r + b, nullptr, cached, nullptr);
    } else {
        rc = run_layer_batch_impl(ctx, batch, k_i, k, d_gt, d_l, d_r, z, d_W, out_coeffs, out_len, out_r, shard, cached, group);
    }
    if (rc && cached && !was_ready) cached->ready = false;
    return rc;
}

int run_layer_batch_impl(gkr_ctx* ctx, int batch, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r,
                         const gkr_fr* z, const Fr* d_W, gkr_fr* const* out_coeffs, uint32_t* const* out_len, gkr_fr* const* out_r,
                         const LayerShardArgs* shard, GateLists* cached, const LayerGroup* group) {
    const size_t N = (size_t)1 << (2 * k);
    const size_t wlen = (size_t)1 << k;
    const uint32_t v = 2 * k;
    const bool host_tx = ctx->transcript == GKR_TRANSCRIPT_HOST;
    if (!host_tx && batch != 1) return ctx->fail(GKR_ERR_INVALID, "batched proving needs the host transcript");
    if (k < 1) return ctx->fail(GKR_ERR_DEGENERATE, "k_next == 0: v = 0 underflows in the reference (sumcheck.rs:49)");
    if (shard && (!host_tx || batch != 1)) return ctx->fail(GKR_ERR_INVALID, "a gate-sharded layer needs the host transcript and one proof");
    // (argument checks come before anything is queued: a rank that returns here has not left its peers inside a collective
    // -- the same arguments fail on every rank)
    if (shard && shard->dev && (shard->dev->capacity < gkr_exchange_limbs(k) || !shard->dev->d_limbs || !shard->dev->fn))
        return ctx->fail(GKR_ERR_INVALID, "the exchange buffer is smaller than gkr_exchange_limbs(k_next) int64");
    if (k > kMaxLayerK || k_i > kMaxLayerKi) return ctx->fail(GKR_ERR_INVALID, "layer wider than the library's limits (gkr_amd.h: GKR_MAX_K_NEXT, GKR_MAX_K_I)");
    if (!host_tx && k > kMaxDenseK)
        return ctx->fail(GKR_ERR_INVALID, "the device transcript works on dense 2^(2 k_next)-entry predicate tables: k_next <= 14 (GKR_MAX_K_NEXT_DEVICE_TRANSCRIPT)");
    const gkr::GateSpan span{shard ? shard->gate_base : 0, shard ? shard->gate_count : (uint64_t)1 << k_i};
    hipStream_t s = ctx->stream;
    Fr *A = nullptr, *M = nullptr, *Wb = nullptr, *Wc = nullptr, *d_coeffs = nullptr, *d_r_out = nullptr;
    gkr::FixedMul* d_rtab = nullptr;
    uint32_t *d_len = nullptr, *dep = nullptr;
    gkr::LayerPartial* partials = nullptr;
    // Two forms (the same transcript): with the host transcript -- the default -- the layer polynomial is summed in time
    // linear in the gates: no 2^{2k}-entry tables at all, U, V and the c-phase row come straight from the gates grouped by
    // left / right operand (kernels.hip, k_gate_* / k_seg_*; kernels_wide.hip for wide layers), and the rounds run as
    // product passes.  The device transcript works on the dense predicate tables (kernels_layer_dense.hip).
    const bool sparse = host_tx;
    // Wide layers (2^13 buckets and more per half, each with a few gates): the gate passes run with a group of lanes per
    // bucket and the rare long buckets in units (kernels_wide.hip) -- a block per bucket would be 2^20 blocks for a gate apiece.
    // The option gate_groups_min_k moves the switch (tests run the form on small layers too).
    const bool wide = sparse && layer_is_wide(span, k_i, k);
    const gkr::GateSet* sets = group ? group->d_sets : nullptr;   // lockstep group: per-proof gate lists
    const size_t tlen = sparse ? wlen : N;   // entries of A / M per proof: the single c-phase row, or the whole table
    WS(ctx, sparse ? "layer.Arow" : "layer.A", Fr, tlen * batch, A);
    WS(ctx, sparse ? "layer.Mrow" : "layer.M", Fr, tlen * batch, M);
    WS(ctx, "layer.Wb", Fr, wlen * batch, Wb);
    WS(ctx, "layer.Wc", Fr, wlen * batch, Wc);
    WS(ctx, "layer.coeffs", Fr, (size_t)v * 3, d_coeffs);
    WS(ctx, "layer.r", Fr, v, d_r_out);
    WS(ctx, "layer.rtab", gkr::FixedMul, v, d_rtab);
    WS(ctx, "layer.len", uint32_t, v, d_len);
    WS(ctx, "layer.dep", uint32_t, 32 * (size_t)batch, dep);
    WS(ctx, "layer.partials", gkr::LayerPartial, (size_t)gkr::kMaxLayerBlocks * batch, partials);
    Fr *U = nullptr, *V = nullptr, *d_eq = nullptr;
    gkr_fr* h_u = nullptr;   // pinned: u = (r_1 .. r_k) of every proof, from which the device builds eq(u, .)
    Fr *e_hi = nullptr, *e_lo = nullptr;
    uint32_t *g_offsets = nullptr, *g_cursor = nullptr, *g_list = nullptr, *g_plan = nullptr;
    Fr *item_partials = nullptr, *E = nullptr;   // wide layers: the item passes' scratch, eq(z, .) as a table
    uint32_t* gate_arrive = nullptr;             // ... and the combine step's arrival counters (zero between passes)
    const gkr::GatePlanCounts* plan_counts = group ? &group->plan_counts : (cached && cached->ready ? &cached->plan_counts : nullptr);
    // where eq(z, g) is split into E_hi, E_lo: in the middle, or -- large layers, whose gate passes run over segments
    // of the sorted lists (gate_seg.h) -- where the segments are cut
    const uint32_t kl = gkr::gate_seg_shift(span, (uint32_t)k_i, (uint32_t)k);
    gkr::GateSegs local_segs;
    gkr::GateSegs* segs = cached ? &cached->segs : &local_segs;
    Fr* seg_partials = nullptr;
    if (sparse) {
        WS(ctx, "layer.U", Fr, wlen * batch, U);
        WS(ctx, "layer.V", Fr, wlen * batch, V);
        WS(ctx, "layer.eq", Fr, wlen * batch, d_eq);
        HIP_TRY(ctx, ctx->pinned_host("layer.u", sizeof(gkr_fr) * (size_t)k * batch, reinterpret_cast<void**>(&h_u)));
    }
    int rc = GKR_OK;
    uint32_t* bad = nullptr;
    uint32_t* h_dep = nullptr;   // pinned: which variables W depends on, per proof; the device leaves it there before round 0
    if (host_tx) HIP_TRY(ctx, ctx->pinned_host("layer.hdep", sizeof(uint32_t) * 32 * batch, reinterpret_cast<void**>(&h_dep)));
    bool lists_fresh = true;   // the gate lists are built (and the gates validated) in this call
    if (sparse) {
        uint32_t *g_counts = nullptr, *g_bsums = nullptr;
        const size_t nb2 = (size_t)2 << k;
        WS(ctx, "pred.bad", uint32_t, 1, bad);
        {
            // the eq tables of z (built on the device from the points in pinned memory), the Montgomery copies of W and the
            // dependence flags: one launch (k_layer_prologue)
            const int kh = k_i - (int)kl;
            WS(ctx, "pred.ehi", Fr, (size_t)batch << kh, e_hi);
            WS(ctx, "pred.elo", Fr, (size_t)batch << kl, e_lo);
            gkr_fr* hz = nullptr;
            HIP_TRY(ctx, ctx->pinned_host("pred.z", sizeof(gkr_fr) * (size_t)batch * (k_i ? k_i : 1), reinterpret_cast<void**>(&hz)));
            memcpy(hz, z, sizeof(gkr_fr) * (size_t)batch * k_i);
            // (the dependence flags of a table beyond 2^13 values are found over a grid, not by the prologue's one block)
            const bool dep_wide = k > 13;
            uint32_t* dep_bits = nullptr;
            if (dep_wide) WS(ctx, "layer.depbits", uint32_t, (size_t)batch, dep_bits);
            gkr::launch_layer_prologue(reinterpret_cast<const Fr*>(hz), (uint32_t)k_i, (uint32_t)kh, kl, e_hi, e_lo, d_W, Wb, Wc, (uint32_t)k, dep_wide ? nullptr : dep, h_dep, (uint32_t)batch, s,
                                       dep_bits);
            // (the prologue's last block has stored what the table's first 256 entries show: a generic table's grid scan finds
            // every bit set and leaves at once)
            if (dep_wide) gkr::launch_depends_wide(d_W, (uint32_t)k, dep_bits, dep, h_dep, (uint32_t)batch, s, true);
        }
        if (wide) {
            // eq(z, g) for every gate index of the layer (of the whole layer also when this rank holds a share of the gates: the
            // lists carry indices relative to the share's first gate, the passes add it back), canonical
            WS(ctx, "gates.itempart", Fr, gkr::gate_plan_partial_elems(span.count, (uint32_t)k) * batch, item_partials);
            {
                const size_t words = gkr::gate_plan_arrive_words(span.count, (uint32_t)k) * (size_t)batch;
                const size_t alloc = words < 4096 ? 4096 : words;   // (one size for the small cases: zeroed once)
                WS(ctx, "gates.arrive", uint32_t, alloc, gate_arrive);
                if (ctx->gate_arrive_zeroed != gate_arrive || words > 4096) {
                    HIP_TRY(ctx, hipMemsetAsync(gate_arrive, 0, alloc * sizeof(uint32_t), s));
                    ctx->gate_arrive_zeroed = gate_arrive;
                }
            }
            if ((uint32_t)k_i <= gkr::kGateEqTableMaxKi) {
                WS(ctx, "pred.E", Fr, (size_t)batch << k_i, E);
                Timed t(ctx, "eq_table_z", ((double)batch * 32.0) * (double)((size_t)1 << k_i));
                gkr::launch_eq_outer(e_hi, e_lo, (uint32_t)k_i, kl, E, (uint32_t)batch, s);   // (the prologue above built the halves)
            }
        }
        if (!(cached && cached->ready)) HIP_TRY(ctx, hipMemsetAsync(bad, 0, 4, s));   // (only the list build writes it)
        if (const size_t pe = gkr::gate_seg_partial_elems(span, (uint32_t)k_i, (uint32_t)k)) WS(ctx, "gates.segpart", Fr, pe * batch, seg_partials);
        if (cached && cached->ready) {
            lists_fresh = false;
            g_offsets = cached->offsets;   // the circuit's lists from an earlier call (validated then)
            g_cursor = cached->cursor;
            g_list = cached->list;
            g_plan = cached->plan;
            if (wide && !g_plan) return ctx->fail(GKR_ERR_INVALID, "cached gate lists were built without the wide layer's item plan");
        } else {
            WS(ctx, "gates.counts", uint32_t, nb2, g_counts);
            WS(ctx, "gates.bsums", uint32_t, (nb2 + 2047) / 2048 + 1, g_bsums);
            if (cached) {
                if (!cached->offsets) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&cached->offsets), nb2 * sizeof(uint32_t)));
                if (!cached->cursor) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&cached->cursor), nb2 * sizeof(uint32_t)));
                if (!cached->list) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&cached->list), 2 * gkr::gate_list_words(span.count) * sizeof(uint32_t)));
                if (wide && !cached->plan)
                    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&cached->plan), gkr::gate_plan_words(span.count, (uint32_t)k) * sizeof(uint32_t)));
                g_offsets = cached->offsets;
                g_cursor = cached->cursor;
                g_list = cached->list;
                g_plan = cached->plan;
            } else {
                WS(ctx, "gates.offsets", uint32_t, nb2, g_offsets);
                WS(ctx, "gates.cursor", uint32_t, nb2, g_cursor);
                WS(ctx, "gates.list", uint32_t, 2 * gkr::gate_list_words(span.count), g_list);
                if (wide) WS(ctx, "gates.plan", uint32_t, gkr::gate_plan_words(span.count, (uint32_t)k), g_plan);
            }
            HIP_TRY(ctx, hipMemsetAsync(g_counts, 0, nb2 * sizeof(uint32_t), s));
            uint32_t *lds_scratch = nullptr, *seg_scratch = nullptr;
            if (const size_t words = gkr::gate_lists_lds_scratch_words(span.count, (uint32_t)k)) WS(ctx, "gates.lds", uint32_t, words, lds_scratch);
            if (const size_t words = gkr::gate_segs_words(span, (uint32_t)k_i, (uint32_t)k)) {
                if (cached) {
                    if (!cached->segs.words) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&cached->segs.words), words * sizeof(uint32_t)));
                } else {
                    WS(ctx, "gates.segs", uint32_t, words, local_segs.words);
                }
                WS(ctx, "gates.segscratch", uint32_t, gkr::gate_segs_scratch_words(span, (uint32_t)k_i, (uint32_t)k), seg_scratch);
            }
            Timed t(ctx, "gate_lists", (double)span.count * (9.0 + 4 * 4.0));
            gkr::launch_gate_lists(span, (uint32_t)k_i, (uint32_t)k, d_gt, d_l, d_r, g_counts, g_offsets, g_cursor, g_bsums, g_list, bad, lds_scratch,
                                   segs, seg_scratch, s);
            if (wide) gkr::launch_gate_plan(span, (uint32_t)k, g_offsets, g_cursor, g_list, g_plan, s);
            if (cached) cached->ready = true;   // a bad gate fails the call below and the prepared circuit is dropped
        }
    } else {
        rc = build_predicates(ctx, k_i, k, d_gt, d_l, d_r, z, A, M, 0, 0, batch);
        if (rc) return rc;
    }
    gkr::LayerBatch lb{(uint32_t)batch, gkr::kMaxLayerBlocks, tlen, wlen};
    if (!sparse) {   // (the gate-list form did all of this in its prologue launch above)
        HIP_TRY(ctx, hipMemsetAsync(dep, 0, sizeof(uint32_t) * 32 * batch, s));
        gkr::launch_to_mont(d_W, Wb, (uint32_t)(wlen * batch), s);
        gkr::launch_to_mont(d_W, Wc, (uint32_t)(wlen * batch), s);
        gkr::launch_depends(d_W, k, dep, (uint32_t)batch, s);
    }
    if (sparse) {
        Timed t(ctx, "gate_uv", (double)span.count * 8.0 * batch);   // HBM: the 8-byte list entry per gate (operands are L2 gathers)
        if (wide)
            gkr::launch_gate_uv_wide(span, (uint32_t)k_i, (uint32_t)k, g_plan, gkr::GateEq{E, e_hi, e_lo, kl}, Wc, U, V, lb, item_partials, gate_arrive, s, sets, plan_counts);
        else
            gkr::launch_gate_uv(span, (uint32_t)k_i, (uint32_t)k, g_offsets, g_cursor, g_list, e_hi, e_lo, kl, Wc, U, V, lb, segs, seg_partials, s, sets);
    }

    // The two tables d_a, d_b (`each` elements) := their sums over all ranks.  One more element travels along: "some rank
    // failed" (a bad gate seen on the device, or `local_fail`: this rank's own error status), so that every rank enters
    // every collective and all of them leave with an error together instead of one leaving the others inside it.
    // Device exchange: widen -> the caller's all-reduce on this stream -> narrow, no host copy and no synchronisation;
    // the summed flag lands in pinned memory and is looked at when the next record has landed (xflag_check).
    uint32_t* h_xflag = nullptr;
    if (shard && shard->dev) {
        HIP_TRY(ctx, ctx->pinned_host("layer.xflag", 64, reinterpret_cast<void**>(&h_xflag)));
        *h_xflag = 0;
    }
    auto sum_over_ranks = [&](Fr* d_a, Fr* d_b, size_t each, const uint32_t* d_flag, int local_fail) -> int {
        if (shard->dev) {
            Timed t(ctx, "exchange", 0.0);
            long long* limbs = reinterpret_cast<long long*>(shard->dev->d_limbs);
            gkr::launch_exchange_widen(d_a, d_b, (uint32_t)each, d_flag, local_fail ? 1u : 0u, limbs, s);
            const int arc = shard->dev->fn(shard->dev->user, (2 * each + 1) * 8, static_cast<void*>(s));
            gkr::launch_exchange_narrow(limbs, d_a, d_b, (uint32_t)each, h_xflag, s);
            if (arc) return ctx->fail(GKR_ERR_INVALID, "the device sum-over-ranks hook failed (status " + std::to_string(arc) + ")");
            HIP_TRY(ctx, hipGetLastError());
            return local_fail;
        }
        const auto t0 = std::chrono::steady_clock::now();
        gkr_fr* buf = nullptr;   // pinned, kept by the context: no pageable staging vector per exchange
        HIP_TRY(ctx, ctx->pinned_host("layer.xbuf", sizeof(gkr_fr) * (2 * each + 1), reinterpret_cast<void**>(&buf)));
        uint32_t hflag = local_fail ? 1u : 0u;
        gkr::launch_copy_words(d_a, buf, each * 8, s);
        gkr::launch_copy_words(d_b, buf + each, each * 8, s);
        // (no early return between here and the hook: the peers are on their way into the collective)
        if (d_flag && !local_fail && hipMemcpyAsync(&hflag, d_flag, 4, hipMemcpyDeviceToHost, s) != hipSuccess) hflag = 1u;
        const hipError_t se = hipStreamSynchronize(s);
        if (se != hipSuccess) hflag = 1u;   // still enter the collective: the peers are on their way into it
        buf[2 * each] = gkr_fr{{(uint64_t)(hflag != 0), 0, 0, 0}};
        const int arc = shard->allreduce(shard->user, buf, 2 * each + 1);
        if (se != hipSuccess) return ctx->hip_fail(se, "hipStreamSynchronize before the sum over ranks");
        if (arc) return ctx->fail(GKR_ERR_INVALID, "the sum-over-ranks hook failed (status " + std::to_string(arc) + ")");
        if (!all_canonical(buf, 2 * each + 1)) return ctx->fail(GKR_ERR_NON_CANONICAL, "the sum-over-ranks hook returned a value >= r");
        const bool some_failed = (buf[2 * each].l[0] | buf[2 * each].l[1] | buf[2 * each].l[2] | buf[2 * each].l[3]) != 0;
        gkr::launch_copy_words(buf, d_a, each * 8, s);
        gkr::launch_copy_words(buf + each, d_b, each * 8, s);   // (the next exchange waits for the stream before it rewrites buf)
        if (ctx->profile == 1)
            ctx->add_host_sample("exchange", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (local_fail) return local_fail;
        if (some_failed) return ctx->fail(GKR_ERR_INVALID, "gate type or operand index out of range, or another failure, on some rank");
        return GKR_OK;
    };
    // device exchange: has the flag that travelled with the exchanges come back set?  Valid once a kernel queued after
    // the narrow step has published something the host waited for.
    auto xflag_check = [&]() -> int {
        if (h_xflag && __atomic_load_n(h_xflag, __ATOMIC_ACQUIRE))
            return ctx->fail(GKR_ERR_INVALID, "gate type or operand index out of range, or another failure, on some rank");
        return GKR_OK;
    };
    if (shard) {
        rc = sum_over_ranks(U, V, wlen, lists_fresh ? bad : nullptr, GKR_OK);
        if (rc) return rc;
    }

    const bool ifma = host_ifma_ready();
    gkr::SpinPool* pool = nullptr;
    if (host_tx) {
        // (h_dep is read when round 0 is hashed, i.e. after a LATER kernel of this stream has released that round's record:
        // the prologue launch wrote it)
        // (gate-sharded with the device exchange: the flag travels with the first exchange and is looked at after the
        // first round's record, on every rank alike -- a rank that left here would leave its peers inside a collective)
        if (sparse && lists_fresh && !(shard && shard->dev)) {   // lists found in the circuit cache were validated when they were built
            uint32_t hbad = 0;
            HIP_TRY(ctx, hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, s));
            if (wide && cached) HIP_TRY(ctx, queue_plan_counts_readback(g_plan, span.count, k, &cached->plan_counts, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
            if (hbad) return ctx->fail(GKR_ERR_INVALID, "gate type or operand index out of range");
            if (wide && cached) cached->plan_counts.known = true;
        }
        if (batch >= 16) pool = ctx->host_pool();
    }
    // Product passes (kernels.hip): both phases as sumchecks of W X + Y over three small tables, up to three rounds per
    // device round trip.
    if (sparse) {
        gkr::ProdPassRec* prec = nullptr;
        Fr *h_pw = nullptr, *d_ppart = nullptr, *Xc = nullptr, *Yc = nullptr;
        HIP_TRY(ctx, ctx->pinned_host("layer.prec", sizeof(gkr::ProdPassRec) * batch, reinterpret_cast<void**>(&prec)));
        HIP_TRY(ctx, ctx->pinned_host("layer.pw", sizeof(Fr) * 8 * batch, reinterpret_cast<void**>(&h_pw)));
        WS(ctx, "layer.ppart", Fr, (size_t)batch * gkr::prod_pass_scratch_values((uint32_t)k), d_ppart);
        WS(ctx, "layer.X", Fr, wlen * batch, Xc);
        WS(ctx, "layer.Y", Fr, wlen * batch, Yc);
        unsigned char* d_fold_plans = nullptr;   // (wide layers: the later passes' pending folds on the matrix cores)
        if (k >= 14) WS(ctx, "layer.foldplans", unsigned char, (size_t)batch * gkr::prod_fold_plan_bytes(), d_fold_plans);
        // (passes of a few blocks per proof publish from their last block: one arrival counter per proof, zero between passes)
        const bool no_fused_publish = gkr::opt(gkr::OPT_no_fused_publish) != 0;
        uint32_t* d_arrivals = nullptr;
        if (!no_fused_publish) {
            WS(ctx, "layer.arrivals", uint32_t, (size_t)(batch < 4096 ? 4096 : batch), d_arrivals);   // (one size for every batch: zeroed once)
            if (ctx->arrivals_zeroed != d_arrivals) {
                HIP_TRY(ctx, hipMemsetAsync(d_arrivals, 0, sizeof(uint32_t) * (size_t)(batch < 4096 ? 4096 : batch), s));
                ctx->arrivals_zeroed = d_arrivals;
            }
        }
        // the host tail (host_tail_pass above): for batches of a few proofs -- where a step waits for its chain of hand-offs, not
        // for its hashing throughput -- the passes over tables of 2^tail_log2 entries and fewer run on the host
        const long long tail_opt = gkr::opt(gkr::OPT_host_tail_log2), tail_batch_opt = gkr::opt(gkr::OPT_host_tail_max_batch);
        const uint32_t tail_log2 = tail_opt < 0 ? 0u : tail_opt == 0 ? 6u : (uint32_t)(tail_opt > 12 ? 12 : tail_opt);
        const bool tail_on = tail_log2 >= 3u && batch <= (tail_batch_opt > 0 ? tail_batch_opt : 8);
        Fr* h_tail = nullptr;
        if (tail_on) HIP_TRY(ctx, ctx->pinned_host("layer.tail", sizeof(Fr) * 3 * ((size_t)batch << tail_log2), reinterpret_cast<void**>(&h_tail)));
        const bool dbg_sections = gkr::debug_timing();
        double us_launch = 0, us_wait = 0, us_pieces = 0, us_phase1 = 0;
        const double t_passes0 = dbg_sections ? now_us_dbg() : 0.0;
        gkr::SpinPool::Session session(pool, nullptr);
        uint32_t round0 = 0, jp = 0;
        bool second_exchange_done = false, tail_active = false;
        const size_t tail_stride = (size_t)1 << tail_log2;
        uint32_t tail_m = 0;   // log2 of the host tables' length (before the pending fold), while the tail is active
        for (int phase = 0; phase < 2 && rc == GKR_OK; ++phase) {
            Fr *Tw = Wb, *Tx = U, *Ty = V;
            // what is left of Wb -- the 2^jp entries the last pass's weights bind into W(u) -- is on the host when the b-phase ended in
            // the host tail.  The wide layers' fused set-up reads it where it is (pinned memory; one wave per proof); the other forms
            // get it back on the device by a copy KERNEL (a hipMemcpyAsync here cost ~25 us of the chain: the round path makes no
            // transfer call of the runtime).
            const bool wu_from_host_tail = phase == 1 && tail_active && wide && !shard;
            if (phase == 1 && tail_active && !wu_from_host_tail)
                gkr::launch_copy_rows(h_tail, 3 * tail_stride * 8, Wb, wlen * 8, 8u << jp, (uint32_t)batch, s);
            tail_active = false;
            const double t_ph1 = dbg_sections ? now_us_dbg() : 0.0;
            if (phase == 1) {
                // all of b is bound: the rows of a, m at u = (r_1 .. r_k), then the c-phase's tables X = a_u + W(u) m_u,
                // Y = W(u) a_u (W(u): the last b pass's fold of what is left of Wb)
                for (int b = 0; b < batch; ++b) memcpy(h_u + (size_t)b * k, out_r[b], sizeof(gkr_fr) * k);
                // (a wide layer whose gates are all on this rank: the eq-table launch also leaves W(u), and the row pass writes the
                // c-phase's tables X, Y itself -- no k_prod_c_setup launch, no pass over the rows)
                const gkr::CPhaseFuse fuse{wu_from_host_tail ? h_tail : Wb, h_pw, Xc, Yc, jp};
                Fr* d_wu = nullptr;
                if (wide && !shard) WS(ctx, "layer.wu", Fr, (size_t)batch, d_wu);
                gkr::launch_eq_table(reinterpret_cast<const Fr*>(h_u), (uint32_t)k, 0u, (uint32_t)k, d_eq, true, (uint32_t)batch, s, d_wu ? &fuse : nullptr, d_wu,
                                     (uint32_t)(wu_from_host_tail ? 3 * tail_stride : wlen));
                bool c_tables_done = false;   // (one rank holds all gates: the row pass writes X, Y too)
                {
                    Timed t(ctx, "gate_rows", (double)span.count * 8.0 * batch);
                    if (wide) {
                        const gkr::WideCFuse wfuse{d_wu, Xc, Yc};
                        gkr::launch_gate_rows_wide(span, (uint32_t)k_i, (uint32_t)k, g_plan, gkr::GateEq{E, e_hi, e_lo, kl}, d_eq, A, M, lb, item_partials, gate_arrive, s, sets, plan_counts,
                                                   d_wu ? &wfuse : nullptr);
                        c_tables_done = d_wu != nullptr;
                    } else
                        c_tables_done = gkr::launch_gate_rows(span, (uint32_t)k_i, (uint32_t)k, g_offsets, g_cursor, g_list, e_hi, e_lo, kl, d_eq, A, M, lb,
                                                              segs, seg_partials, s, shard ? nullptr : &fuse, sets);
                }
                if (shard) {   // every rank summed its own gates: the rows are complete after one exchange
                    second_exchange_done = true;
                    rc = sum_over_ranks(A, M, wlen, nullptr, GKR_OK);
                    if (rc) break;
                }
                if (!c_tables_done) gkr::launch_prod_c_setup(Wb, jp, h_pw, A, M, Xc, Yc, (uint32_t)k, (uint32_t)wlen, (uint32_t)batch, s);
                Tw = Wc;
                Tx = Xc;
                Ty = Yc;
                jp = 0;
            }
            if (dbg_sections && phase == 1) us_phase1 += now_us_dbg() - t_ph1;
            uint32_t m = (uint32_t)k;   // log2 of the tables' length before the pending fold
            for (uint32_t rem = (uint32_t)k; rem > 0 && rc == GKR_OK;) {
                // (the rounds that do not fill a pass of three come LAST.  First -- so that the pass over the whole table forms
                // 4^J = 4 or 16 cross sums per index instead of 64 -- was measured on wide layers and is slower: k = 20 0.61 ->
                // 0.80 ms of product passes per sumcheck, k = 22 1.44 -> 3.33: the second pass then folds into a table four or
                // two times larger and crosses THAT with J = 3.)
                const uint32_t J = rem < (uint32_t)gkr::kProdMaxJ ? rem : (uint32_t)gkr::kProdMaxJ;
                const bool on_host = tail_active;
                // (this pass exports the tables if they are small enough and a later pass of the phase is there to be saved)
                const bool exports = tail_on && !on_host && m - jp <= tail_log2 && rem > J;
                if (!on_host) {
                    const uint32_t ticket = ++ctx->ticket;
                    const double tl0 = dbg_sections ? now_us_dbg() : 0.0;
                    {
                        Timed t(ctx, "layer_prod_pass", 0.0);
                        gkr::launch_prod_pass(Tw, Tx, Ty, m, jp, h_pw, J, d_ppart, (uint32_t)wlen, prec, ticket, (uint32_t)batch, s, d_arrivals, d_fold_plans,
                                              exports ? h_tail : nullptr, (uint32_t)tail_stride);
                    }
                    if (hipError_t le = hipGetLastError(); le != hipSuccess) {
                        rc = ctx->hip_fail(le, "launch of a layer pass");
                        break;
                    }
                    const double tl1 = dbg_sections ? now_us_dbg() : 0.0;
                    rc = wait_records(ctx, prec, batch, ticket);
                    if (dbg_sections) {
                        us_launch += tl1 - tl0;
                        us_wait += now_us_dbg() - tl1;
                    }
                    if (!rc) rc = xflag_check();
                    if (rc) break;
                }
                const uint32_t m_before = on_host ? tail_m : 0u, jp_before = jp;
                m -= jp;
                const int chunk = hash_chunk_size(batch, pool ? pool->workers() + 1 : 1, ctx->crew_member ? ctx->help_share : 0);
                std::atomic<int> next{0};
                const std::function<bool()> work = [&]() -> bool {
                    const int first = next.fetch_add(chunk, std::memory_order_relaxed);
                    if (first >= batch) return false;
                    const int cnt = batch - first < chunk ? batch - first : chunk;
                    uint64_t c2[gkr::kProdMaxJ][16][4], lin[gkr::kProdMaxJ][16][4], c0[gkr::kProdMaxJ][16][4], rr[gkr::kProdMaxJ][16][4];
                    uint32_t vl[gkr::kProdMaxJ][16];
                    const bool acct = accounting_on();
                    const double tp0 = acct ? now_us_dbg() : 0.0;
                    if (on_host)   // (the weights of the previous pass are still in h_pw: the pass function below replaces them)
                        for (int i = 0; i < cnt; ++i)
                            host_tail_pass(reinterpret_cast<gkr::h64::F*>(h_tail + (size_t)(first + i) * 3 * tail_stride), tail_stride, m_before, jp_before,
                                           reinterpret_cast<const gkr::h64::F*>(h_pw + (size_t)(first + i) * 8), J,
                                           reinterpret_cast<gkr::h64::F*>(&prec[first + i].v[0]));
                    for (uint32_t t = 0; t < J; ++t)
                        for (int i = 0; i < cnt; ++i) vl[t][i] = 2u + (h_dep[(size_t)(first + i) * 32 + (round0 + t) % k] ? 1u : 0u);
                    (ifma && cnt >= 3 ? gkr::gkr_ifma_prod_pass : host_prod_pass_scalar)(
                        reinterpret_cast<const uint64_t*>(prec + first), sizeof(gkr::ProdPassRec) / 8, cnt, (int)J, vl, c2, lin, c0, rr,
                        reinterpret_cast<uint64_t*>(h_pw + (size_t)first * 8), 32);
                    const double tp1 = acct ? now_us_dbg() : 0.0;
                    for (int i = 0; i < cnt; ++i) {
                        const int b = first + i;
                        for (uint32_t t = 0; t < J; ++t) {
                            const uint32_t round = round0 + t;
                            gkr_fr* oc = out_coeffs[b] + (size_t)round * 3;
                            memset(&oc[0], 0, 32);
                            if (vl[t][i] == 3) memcpy(&oc[0], c2[t][i], 32);
                            memcpy(&oc[1], lin[t][i], 32);
                            memcpy(&oc[2], c0[t][i], 32);
                            out_len[b][round] = vl[t][i];
                            memcpy(&out_r[b][round], rr[t][i], 32);
                        }
                    }
                    if (acct) account_piece(cnt, tp1 - tp0, now_us_dbg() - tp0);
                    return true;
                };
                const double tw0 = dbg_sections ? now_us_dbg() : 0.0;
                run_pieces(pool, &work, batch > chunk, ctx->rounds_ahead + (int)(v - round0));
                if (dbg_sections) us_pieces += now_us_dbg() - tw0;
                if (exports) tail_active = true;
                if (exports || on_host) tail_m = m;   // (the host's tables: 2^m entries, this pass's J variables pending)
                jp = J;
                round0 += J;
                rem -= J;
            }
        }
        session.close();
        if (dbg_sections)
            fprintf(stderr, "[gkr timing] layer k_i=%d k=%d batch=%d passes: %.0f us = launch calls %.0f + waiting for records %.0f + hashing pieces %.0f + c-phase set-up calls %.0f + other %.0f\n",
                    k_i, k, batch, now_us_dbg() - t_passes0, us_launch, us_wait, us_pieces, us_phase1,
                    now_us_dbg() - t_passes0 - us_launch - us_wait - us_pieces - us_phase1);
        // a rank that failed between the exchanges still enters the second one (flag set): its peers are waiting in it.
        // (Not when the failure is the travelling flag itself: then every rank is leaving at this very point.)
        if (rc && shard && !second_exchange_done && !(h_xflag && __atomic_load_n(h_xflag, __ATOMIC_ACQUIRE)))
            (void)sum_over_ranks(A, M, wlen, nullptr, rc);
        if (rc) {
            (void)hipStreamSynchronize(s);
            ctx->arrivals_zeroed = nullptr;   // (a pass that was given up may have left its counters half way)
            return rc;
        }
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(s));
        rc = xflag_check();
        if (rc) return rc;
        ctx->drain_events();
        return GKR_OK;
    }
    // ---- device transcript: rounds over the dense predicate tables (kernels_layer_dense.hip), hashed by one lane per
    // sumcheck (k_layer_round_hash); one uninterrupted stream of launches, one copy-back.  The fold with r_j is deferred
    // into the pass that computes round j+1's sums (b-phase: the fused kernel; c-phase: a separate fold of the remaining row).
    const gkr::FixedMul* pending = nullptr;   // challenge tables not yet applied to A, M
    const bool no_fused = gkr::opt(gkr::OPT_layer_no_fused) != 0;
    for (uint32_t round = 0; round < v; ++round) {
        const uint32_t h = (uint32_t)(N >> (round + 1));   // half of the table this round sums over
        const uint32_t phase = round < (uint32_t)k ? 0u : 1u;
        const uint32_t hb = phase == 0 ? (h >> k) : 0u;
        uint32_t nblk = 0;
        if (phase == 0 && !no_fused) {
            Timed t(ctx, "layer_round_fused", (pending ? (double)h * 2.0 * 6.0 : (double)h * 2.0 * 2.0) * 32.0 * batch);
            nblk = gkr::launch_layer_round_b(pending != nullptr, A, M, A, M, hb, (uint32_t)k, pending, Wb, Wc, partials, lb, s);
            pending = nullptr;
        } else {
            if (pending) {
                Timed t(ctx, "layer_fold", (double)h * 2.0 * 6.0 * 32.0 * batch);
                gkr::launch_layer_fold(A, M, 2 * h, pending, lb, s);
                pending = nullptr;
            }
            nblk = gkr::layer_blocks(h);
            if (nblk * (uint32_t)batch > 4096u) nblk = 4096u / batch ? 4096u / batch : 1u;
            Timed t(ctx, "layer_round", (double)h * 4.0 * 32.0 * batch);
            gkr::launch_layer_round(A, M, h, k, phase, hb, Wb, Wc, nblk, partials, lb, s);
        }
        Timed t(ctx, "layer_round_hash", 0.0);
        gkr::launch_layer_round_hash(partials, nblk, round, k, dep, ctx->d_cts, d_coeffs, d_len, d_r_out, d_rtab, Wb, Wc, s);
        pending = d_rtab + round;
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_coeffs[0], d_coeffs, (size_t)v * 3 * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(out_len[0], d_len, v * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(out_r[0], d_r_out, v * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    ctx->drain_events();
    return GKR_OK;
}

int run_layer(gkr_ctx* ctx, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r, const gkr_fr* z,
              const Fr* d_W, gkr_fr* out_coeffs, uint32_t* out_len, gkr_fr* out_r) {
    return run_layer_batch(ctx, 1, k_i, k, d_gt, d_l, d_r, z, d_W, &out_coeffs, &out_len, &out_r);
}

// evaluation table -> monomial coefficients, variable 1 = most significant bit

}  // namespace gkr_host

// =========================================================================== C ABI

extern "C" {

// ---- layer sumcheck / predicates / layer eval -------------------------------------

static int upload_gates(gkr_ctx* ctx, size_t gates, const uint8_t* gt, const uint32_t* l, const uint32_t* r,
                        DevBuf<uint8_t>& dgt, DevBuf<uint32_t>& dl, DevBuf<uint32_t>& dr) {
    HIP_TRY(ctx, dgt.alloc(gates));
    HIP_TRY(ctx, dl.alloc(gates));
    HIP_TRY(ctx, dr.alloc(gates));
    HIP_TRY(ctx, hipMemcpyAsync(dgt.p, gt, gates, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dl.p, l, gates * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(dr.p, r, gates * 4, hipMemcpyHostToDevice, ctx->stream));
    return GKR_OK;
}

static int check_layer_args(gkr_ctx* ctx, int k_i, int k_next, const uint8_t* gt, const uint32_t* l, const uint32_t* r,
                            const gkr_fr* z) {
    if (!gt || !l || !r || (k_i > 0 && !z)) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    if (k_i < 0 || k_i > 28) return ctx->fail(GKR_ERR_INVALID, "k_i must be in [0, 28]");
    if (k_next == 0) return ctx->fail(GKR_ERR_DEGENERATE, "k_next == 0: v = 0 underflows in the reference (sumcheck.rs:49)");
    if (k_next < 0 || k_next > kMaxLayerK || k_i > kMaxLayerKi) return ctx->fail(GKR_ERR_INVALID, "k_next must be in [1, GKR_MAX_K_NEXT], k_i in [0, GKR_MAX_K_I]");
    if (k_i > 0 && !all_canonical(z, k_i)) return ctx->fail(GKR_ERR_NON_CANONICAL, "z entry >= r");
    // every gate is validated on the device by the scatter kernel; small layers are also checked here
    // so that the error names the cause
    const size_t gates = (size_t)1 << k_i;
    for (size_t g = 0; g < gates && gates <= ((size_t)1 << 16); ++g) {
        if (gt[g] > 1) return ctx->fail(GKR_ERR_INVALID, "gate_type must be 0 (add) or 1 (mult)");
        if ((l[g] >> k_next) || (r[g] >> k_next)) return ctx->fail(GKR_ERR_INVALID, "gate operand index out of range");
    }
    return GKR_OK;
}

int gkr_sumcheck_layer(gkr_ctx* ctx, int k_i, int k_next, const uint8_t* gate_type, const uint32_t* left,
                       const uint32_t* right, const gkr_fr* z, const gkr_fr* W, gkr_fr* out_coeffs, uint32_t* out_len,
                       gkr_fr* out_r) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!W || !out_coeffs || !out_len || !out_r) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    int rc = check_layer_args(ctx, k_i, k_next, gate_type, left, right, z);
    if (rc) return rc;
    if (!all_canonical(W, (size_t)1 << k_next)) return ctx->fail(GKR_ERR_NON_CANONICAL, "W entry >= r");
    GKR_ENTER(ctx);
    DevBuf<uint8_t> dgt;
    DevBuf<uint32_t> dl, dr;
    DevBuf<Fr> dW;
    rc = upload_gates(ctx, (size_t)1 << k_i, gate_type, left, right, dgt, dl, dr);
    if (rc) return rc;
    HIP_TRY(ctx, dW.alloc((size_t)1 << k_next));
    HIP_TRY(ctx, hipMemcpyAsync(dW.p, W, sizeof(Fr) << k_next, hipMemcpyHostToDevice, ctx->stream));
    return run_layer(ctx, k_i, k_next, dgt.p, dl.p, dr.p, z, dW.p, out_coeffs, out_len, out_r);
}

int gkr_sumcheck_layer_sharded(gkr_ctx* ctx, int k_i, int k_next, uint64_t gate_first, uint64_t gate_count,
                               const uint8_t* gate_type, const uint32_t* left, const uint32_t* right, const gkr_fr* z,
                               const gkr_fr* W, gkr_allreduce_fn allreduce, void* user, gkr_fr* out_coeffs, uint32_t* out_len,
                               gkr_fr* out_r) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!W || !out_coeffs || !out_len || !out_r || !allreduce || (k_i > 0 && !z)) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    if (gate_count && (!gate_type || !left || !right)) return ctx->fail(GKR_ERR_INVALID, "null gate array");
    if (k_i < 0 || k_i > 28) return ctx->fail(GKR_ERR_INVALID, "k_i must be in [0, 28]");
    if (k_next == 0) return ctx->fail(GKR_ERR_DEGENERATE, "k_next == 0: v = 0 underflows in the reference (sumcheck.rs:49)");
    if (k_next < 0 || k_next > kMaxLayerK) return ctx->fail(GKR_ERR_INVALID, "k_next must be in [1, GKR_MAX_K_NEXT]");
    if (gate_first + gate_count > ((uint64_t)1 << k_i)) return ctx->fail(GKR_ERR_INVALID, "gate range exceeds the layer's 2^k_i gates");
    if (k_i > 0 && !all_canonical(z, k_i)) return ctx->fail(GKR_ERR_NON_CANONICAL, "z entry >= r");
    if (!all_canonical(W, (size_t)1 << k_next)) return ctx->fail(GKR_ERR_NON_CANONICAL, "W entry >= r");
    // gates are validated on the device (k_gate_count); a bad one fails every rank through the first exchange
    GKR_ENTER(ctx);
    DevBuf<uint8_t> dgt;
    DevBuf<uint32_t> dl, dr;
    DevBuf<Fr> dW;
    const size_t n_alloc = gate_count ? (size_t)gate_count : 1;
    HIP_TRY(ctx, dgt.alloc(n_alloc));
    HIP_TRY(ctx, dl.alloc(n_alloc));
    HIP_TRY(ctx, dr.alloc(n_alloc));
    if (gate_count) {
        HIP_TRY(ctx, hipMemcpyAsync(dgt.p, gate_type, gate_count, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(dl.p, left, gate_count * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(dr.p, right, gate_count * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, dW.alloc((size_t)1 << k_next));
    HIP_TRY(ctx, hipMemcpyAsync(dW.p, W, sizeof(Fr) << k_next, hipMemcpyHostToDevice, ctx->stream));
    LayerShardArgs sh;
    sh.gate_base = gate_first;
    sh.gate_count = gate_count;
    sh.allreduce = allreduce;
    sh.user = user;
    return run_layer_batch(ctx, 1, k_i, k_next, dgt.p, dl.p, dr.p, z, dW.p, &out_coeffs, &out_len, &out_r, &sh);
}

int gkr_sumcheck_layer_device(gkr_ctx* ctx, int k_i, int k_next, uint64_t gate_first, uint64_t gate_count, const void* d_gate_type,
                              const void* d_left, const void* d_right, const gkr_fr* z, const gkr_fr* W, gkr_allreduce_fn allreduce,
                              void* user, gkr_fr* out_coeffs, uint32_t* out_len, gkr_fr* out_r) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!W || !out_coeffs || !out_len || !out_r || (k_i > 0 && !z)) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    if (!d_gate_type || !d_left || !d_right) return ctx->fail(GKR_ERR_INVALID, "null device gate array");
    if (k_i < 0 || k_i > 28) return ctx->fail(GKR_ERR_INVALID, "k_i must be in [0, 28]");
    if (k_next == 0) return ctx->fail(GKR_ERR_DEGENERATE, "k_next == 0: v = 0 underflows in the reference (sumcheck.rs:49)");
    if (k_next < 0 || k_next > (allreduce ? 13 : 14)) return ctx->fail(GKR_ERR_INVALID, "k_next out of range");
    if (gate_first + gate_count > ((uint64_t)1 << k_i)) return ctx->fail(GKR_ERR_INVALID, "gate range exceeds the layer's 2^k_i gates");
    if (!allreduce && (gate_first != 0 || gate_count != ((uint64_t)1 << k_i)))
        return ctx->fail(GKR_ERR_INVALID, "without an exchange hook the arrays must hold the whole layer");
    if (k_i > 0 && !all_canonical(z, k_i)) return ctx->fail(GKR_ERR_NON_CANONICAL, "z entry >= r");
    if (!all_canonical(W, (size_t)1 << k_next)) return ctx->fail(GKR_ERR_NON_CANONICAL, "W entry >= r");
    GKR_ENTER(ctx);
    Fr* dW = nullptr;
    HIP_TRY(ctx, ctx->workspace("layer.Win", sizeof(Fr) << k_next, reinterpret_cast<void**>(&dW)));
    HIP_TRY(ctx, hipMemcpyAsync(dW, W, sizeof(Fr) << k_next, hipMemcpyHostToDevice, ctx->stream));
    const uint8_t* gt = static_cast<const uint8_t*>(d_gate_type);
    const uint32_t* dl = static_cast<const uint32_t*>(d_left);
    const uint32_t* dr = static_cast<const uint32_t*>(d_right);
    if (!allreduce) return run_layer(ctx, k_i, k_next, gt, dl, dr, z, dW, out_coeffs, out_len, out_r);
    LayerShardArgs sh;
    sh.gate_base = gate_first;
    sh.gate_count = gate_count;
    sh.allreduce = allreduce;
    sh.user = user;
    return run_layer_batch(ctx, 1, k_i, k_next, gt, dl, dr, z, dW, &out_coeffs, &out_len, &out_r, &sh);
}

struct gkr_resident_layer {
    int k_i = 0, k = 0;
    uint64_t first = 0, count = 0;
    uint8_t* gt = nullptr;
    uint32_t *l = nullptr, *r = nullptr;
    GateLists lists;
};

void gkr_resident_layer_free(gkr_ctx* ctx, gkr_resident_layer* layer) {
    if (!layer) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (layer->gt) (void)hipFree(layer->gt);
    if (layer->l) (void)hipFree(layer->l);
    if (layer->r) (void)hipFree(layer->r);
    layer->lists.release();
    delete layer;
}

int gkr_resident_layer_create(gkr_ctx* ctx, int k_i, int k_next, uint64_t gate_first, uint64_t gate_count, const uint8_t* gate_type,
                              const uint32_t* left, const uint32_t* right, gkr_resident_layer** out) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!out) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    *out = nullptr;
    if ((!gate_type || !left || !right) && gate_count) return ctx->fail(GKR_ERR_INVALID, "null gate array");
    if (k_i < 0 || k_i > 28) return ctx->fail(GKR_ERR_INVALID, "k_i must be in [0, 28]");
    if (k_next == 0) return ctx->fail(GKR_ERR_DEGENERATE, "k_next == 0: v = 0 underflows in the reference (sumcheck.rs:49)");
    if (k_next < 0 || k_next > kMaxLayerK || k_i > kMaxLayerKi) return ctx->fail(GKR_ERR_INVALID, "k_next must be in [1, GKR_MAX_K_NEXT], k_i in [0, GKR_MAX_K_I]");
    if (gate_first + gate_count > ((uint64_t)1 << k_i)) return ctx->fail(GKR_ERR_INVALID, "gate range exceeds the layer's 2^k_i gates");
    GKR_ENTER(ctx);
    std::unique_ptr<gkr_resident_layer, void (*)(gkr_resident_layer*)> L(new gkr_resident_layer(), [](gkr_resident_layer* p) {
        gkr_resident_layer_free(nullptr, p);
    });
    L->k_i = k_i;
    L->k = k_next;
    L->first = gate_first;
    L->count = gate_count;
    const size_t n = gate_count ? (size_t)gate_count : 1;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&L->gt), n));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&L->l), n * 4));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&L->r), n * 4));
    if (gate_count) {
        HIP_TRY(ctx, hipMemcpy(L->gt, gate_type, gate_count, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(L->l, left, gate_count * 4, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(L->r, right, gate_count * 4, hipMemcpyHostToDevice));
    }
    *out = L.release();
    return GKR_OK;
}

// A layer's W from host memory into the context's workspace, validated (every entry < r, sumcheck.rs:16,21 unwrap()s): small
// tables through pinned memory and a copy kernel (no transfer call on a small proof's path, see k_copy_words), tables of 2^16
// entries and more by the copy engine with the check on the device -- the host loop and the staging copy of a 2^20-entry W
// took longer than the layer's gate passes.
static int upload_W(gkr_ctx* ctx, const gkr_fr* W, int k, Fr** out) {
    const size_t n = (size_t)1 << k;
    Fr* dW = nullptr;
    HIP_TRY(ctx, ctx->workspace("layer.Win", sizeof(Fr) << k, reinterpret_cast<void**>(&dW)));
    if (k < 16) {
        if (!all_canonical(W, n)) return ctx->fail(GKR_ERR_NON_CANONICAL, "W entry >= r");
        gkr_fr* hW = nullptr;
        HIP_TRY(ctx, ctx->pinned_host("layer.hWin", sizeof(gkr_fr) << k, reinterpret_cast<void**>(&hW)));
        memcpy(hW, W, sizeof(gkr_fr) << k);
        gkr::launch_copy_words(hW, dW, ((size_t)8) << k, ctx->stream);
    } else {
        uint32_t* d_flag = nullptr;
        WS(ctx, "layer.Wflag", uint32_t, 1, d_flag);
        uint32_t hflag = 0;
        HIP_TRY(ctx, hipMemsetAsync(d_flag, 0, 4, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(dW, W, sizeof(Fr) << k, hipMemcpyHostToDevice, ctx->stream));
        gkr::launch_check_canonical(dW, n, d_flag, ctx->stream);
        HIP_TRY(ctx, hipMemcpyAsync(&hflag, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (hflag) return ctx->fail(GKR_ERR_NON_CANONICAL, "W entry >= r");
    }
    *out = dW;
    return GKR_OK;
}

int gkr_resident_layer_sumcheck(gkr_ctx* ctx, gkr_resident_layer* layer, const gkr_fr* z, const gkr_fr* W, gkr_allreduce_fn allreduce,
                                void* user, gkr_fr* out_coeffs, uint32_t* out_len, gkr_fr* out_r) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!layer || !W || !out_coeffs || !out_len || !out_r || (layer->k_i > 0 && !z)) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    const int k_i = layer->k_i, k = layer->k;
    if (!allreduce && (layer->first != 0 || layer->count != ((uint64_t)1 << k_i)))
        return ctx->fail(GKR_ERR_INVALID, "without an exchange hook the layer must be whole");
    if (k_i > 0 && !all_canonical(z, k_i)) return ctx->fail(GKR_ERR_NON_CANONICAL, "z entry >= r");
    GKR_ENTER(ctx);
    Fr* dW = nullptr;
    if (const int urc = upload_W(ctx, W, k, &dW)) return urc;
    LayerShardArgs sh;
    sh.gate_base = layer->first;
    sh.gate_count = layer->count;
    sh.allreduce = allreduce;
    sh.user = user;
    const int rc = run_layer_batch(ctx, 1, k_i, k, layer->gt, layer->l, layer->r, z, dW, &out_coeffs, &out_len, &out_r, allreduce ? &sh : nullptr,
                                   &layer->lists);
    if (rc) {   // a failed first use may have left half-built lists behind
        layer->lists.ready = false;
    }
    return rc;
}

// gkr_resident_layer_sumcheck with W ALREADY in device memory (inside prover::prove the next layer's values come from the
// forward evaluation, prover.rs:38-43: they never were host data; a host that drives prove_sumcheck_opt itself keeps them on
// the device the same way): nothing but z and the transcript crosses PCIe.  W is checked where it lies; the status of that
// check is looked at after the sumcheck (no synchronisation before it).
int gkr_resident_layer_sumcheck_wdev(gkr_ctx* ctx, gkr_resident_layer* layer, const gkr_fr* z, const void* d_W, gkr_fr* out_coeffs, uint32_t* out_len,
                                     gkr_fr* out_r) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!layer || !d_W || !out_coeffs || !out_len || !out_r || (layer->k_i > 0 && !z)) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    const int k_i = layer->k_i, k = layer->k;
    if (layer->first != 0 || layer->count != ((uint64_t)1 << k_i)) return ctx->fail(GKR_ERR_INVALID, "the layer must be whole");
    if (k_i > 0 && !all_canonical(z, k_i)) return ctx->fail(GKR_ERR_NON_CANONICAL, "z entry >= r");
    GKR_ENTER(ctx);
    uint32_t* d_flag = nullptr;
    uint32_t* h_flag = nullptr;
    WS(ctx, "layer.Wflag", uint32_t, 1, d_flag);
    HIP_TRY(ctx, ctx->pinned_host("layer.hWflag", 64, reinterpret_cast<void**>(&h_flag)));
    *h_flag = 0;
    HIP_TRY(ctx, hipMemsetAsync(d_flag, 0, 4, ctx->stream));
    gkr::launch_check_canonical(static_cast<const Fr*>(d_W), (size_t)1 << k, d_flag, ctx->stream);
    gkr::launch_copy_words(d_flag, h_flag, 1, ctx->stream);
    const int rc = run_layer_batch(ctx, 1, k_i, k, layer->gt, layer->l, layer->r, z, static_cast<const Fr*>(d_W), &out_coeffs, &out_len, &out_r, nullptr,
                                   &layer->lists);
    if (rc) {
        layer->lists.ready = false;
        return rc;
    }
    if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE)) return ctx->fail(GKR_ERR_NON_CANONICAL, "W entry >= r");
    return GKR_OK;
}

size_t gkr_exchange_limbs(int k_next) {
    if (k_next < 0 || k_next > kMaxLayerK) return 0;
    return (((size_t)2 << k_next) + 1) * 8;
}

int gkr_resident_layer_sumcheck_dev(gkr_ctx* ctx, gkr_resident_layer* layer, const gkr_fr* z, const gkr_fr* W, const gkr_exchange_dev* exchange,
                                    gkr_fr* out_coeffs, uint32_t* out_len, gkr_fr* out_r) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!layer || !W || !out_coeffs || !out_len || !out_r || (layer->k_i > 0 && !z) || !exchange) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    const int k_i = layer->k_i, k = layer->k;
    if (k_i > 0 && !all_canonical(z, k_i)) return ctx->fail(GKR_ERR_NON_CANONICAL, "z entry >= r");
    GKR_ENTER(ctx);
    Fr* dW = nullptr;
    if (const int urc = upload_W(ctx, W, k, &dW)) return urc;
    LayerShardArgs sh;
    sh.gate_base = layer->first;
    sh.gate_count = layer->count;
    sh.dev = exchange;
    const int rc = run_layer_batch(ctx, 1, k_i, k, layer->gt, layer->l, layer->r, z, dW, &out_coeffs, &out_len, &out_r, &sh, &layer->lists);
    if (rc) layer->lists.ready = false;
    return rc;
}

int gkr_fr_widen(const gkr_fr* values, size_t count, int64_t* limbs) {
    if ((!values || !limbs) && count) return GKR_ERR_INVALID;
    for (size_t i = 0; i < count; ++i)
        for (int j = 0; j < 4; ++j) {
            limbs[8 * i + 2 * j] = (int64_t)(values[i].l[j] & 0xffffffffull);
            limbs[8 * i + 2 * j + 1] = (int64_t)(values[i].l[j] >> 32);
        }
    return GKR_OK;
}

int gkr_fr_narrow(const int64_t* limbs, size_t count, gkr_fr* values) {
    if ((!values || !limbs) && count) return GKR_ERR_INVALID;
    for (size_t i = 0; i < count; ++i) {
        gkr::Acc<10> a = gkr::acc_zero<10>();
        uint64_t carry = 0;
        for (int j = 0; j < 8; ++j) {
            if (limbs[8 * i + j] < 0) return GKR_ERR_INVALID;
            const uint64_t w = (uint64_t)limbs[8 * i + j];
            const uint64_t lo = (w & 0xffffffffull) + (carry & 0xffffffffull);
            a.l[j] = (uint32_t)lo;
            carry = (w >> 32) + (carry >> 32) + (lo >> 32);
        }
        a.l[8] = (uint32_t)carry;
        a.l[9] = (uint32_t)(carry >> 32);
        values[i] = to_abi(gkr::acc_reduce(a));
    }
    return GKR_OK;
}

int gkr_predicate_tables(gkr_ctx* ctx, int k_i, int k_next, const uint8_t* gate_type, const uint32_t* left,
                         const uint32_t* right, const gkr_fr* z, gkr_fr* out_A, gkr_fr* out_M) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!out_A || !out_M) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    int rc = check_layer_args(ctx, k_i, k_next, gate_type, left, right, z);
    if (rc) return rc;
    GKR_ENTER(ctx);
    const size_t N = (size_t)1 << (2 * k_next);
    DevBuf<uint8_t> dgt;
    DevBuf<uint32_t> dl, dr;
    DevBuf<Fr> A, M;
    rc = upload_gates(ctx, (size_t)1 << k_i, gate_type, left, right, dgt, dl, dr);
    if (rc) return rc;
    HIP_TRY(ctx, A.alloc(N));
    HIP_TRY(ctx, M.alloc(N));
    rc = build_predicates(ctx, k_i, k_next, dgt.p, dl.p, dr.p, z, A.p, M.p);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out_A, A.p, N * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out_M, M.p, N * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->drain_events();
    return GKR_OK;
}

int gkr_layer_eval(gkr_ctx* ctx, size_t gates, const uint8_t* gate_type, const uint32_t* left, const uint32_t* right,
                   const gkr_fr* prev, size_t n_prev, gkr_fr* out) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!gate_type || !left || !right || !prev || !out || !gates || !n_prev || gates > ((size_t)1 << 30))
        return ctx->fail(GKR_ERR_INVALID, "null pointer or empty layer");
    for (size_t g = 0; g < gates; ++g)
        if (gate_type[g] > 1 || left[g] >= n_prev || right[g] >= n_prev)
            return ctx->fail(GKR_ERR_INVALID, "gate type or operand index out of range");
    if (!all_canonical(prev, n_prev)) return ctx->fail(GKR_ERR_NON_CANONICAL, "prev entry >= r");
    GKR_ENTER(ctx);
    DevBuf<uint8_t> dgt;
    DevBuf<uint32_t> dl, dr;
    DevBuf<Fr> dprev, dout;
    int rc = upload_gates(ctx, gates, gate_type, left, right, dgt, dl, dr);
    if (rc) return rc;
    HIP_TRY(ctx, dprev.alloc(n_prev));
    HIP_TRY(ctx, dout.alloc(gates));
    HIP_TRY(ctx, hipMemcpyAsync(dprev.p, prev, n_prev * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    gkr::launch_layer_eval((uint32_t)gates, dgt.p, dl.p, dr.p, dprev.p, dout.p, 1, 0, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, gates * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GKR_OK;
}

// ---- step-wise sessions: one sumcheck split across GPUs ------------------------
//
// The hypercube is partitioned by its TRAILING log2(P) variables: rank p owns the entries whose
// low index bits are p.  Rounds bind the LEADING variable, so both members of every pair live on
// the same rank for the first v - log2(P) rounds; each round every rank produces partial sums, one
// tiny all-reduce (<= 96 bytes of field elements) gives every rank the round polynomial, every
// rank derives the same challenge and folds its shard.  The library does the table work per rank;
// the collective and the transcript sit in the caller (gkr_amd/parallel.py: torch.distributed over
// RCCL, or gloo in the CPU tests).  P = 1 is the whole sumcheck with an external transcript.

struct gkr_layer_session {
    int k = 0, kc = 0;          // W has 2^k entries; this shard's column index has kc = k - log2(P) bits
    uint32_t round = 0, rounds = 0;
    size_t cells = 0;           // current entries per table half pair (A, M each)
    Fr *A = nullptr, *M = nullptr, *Wb = nullptr, *Wc = nullptr;
    gkr::LayerPartial* partials = nullptr;
    uint32_t* d_dep = nullptr;
    uint32_t dep[32] = {0};
    gkr::LayerHostRec* rec = nullptr;
    gkr::FixedMul* rtab = nullptr;   // pinned
};


static void free_layer_session(gkr_layer_session* s) {
    if (!s) return;
    if (s->A) (void)hipFree(s->A);
    if (s->M) (void)hipFree(s->M);
    if (s->Wb) (void)hipFree(s->Wb);
    if (s->Wc) (void)hipFree(s->Wc);
    if (s->partials) (void)hipFree(s->partials);
    if (s->d_dep) (void)hipFree(s->d_dep);
    if (s->rec) (void)hipHostFree(s->rec);
    if (s->rtab) (void)hipHostFree(s->rtab);
    delete s;
}

static int alloc_layer_session(gkr_ctx* ctx, gkr_layer_session* S, size_t cells, size_t wb, size_t wc) {
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&S->A), cells * sizeof(Fr)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&S->M), cells * sizeof(Fr)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&S->Wb), wb * sizeof(Fr)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&S->Wc), wc * sizeof(Fr)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&S->partials), gkr::kMaxLayerBlocks * sizeof(gkr::LayerPartial)));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&S->d_dep), 32 * sizeof(uint32_t)));
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&S->rec), sizeof(gkr::LayerHostRec), hipHostMallocCoherent | hipHostMallocMapped));
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&S->rtab), sizeof(gkr::FixedMul), hipHostMallocCoherent | hipHostMallocMapped));
    memset(S->rec, 0, sizeof(gkr::LayerHostRec));
    return GKR_OK;
}

int gkr_layer_session_open(gkr_ctx* ctx, int k_i, int k_next, const uint8_t* gate_type, const uint32_t* left,
                           const uint32_t* right, const gkr_fr* z, const gkr_fr* W, uint32_t nshards, uint32_t shard,
                           gkr_layer_session** out) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!out || !W) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    *out = nullptr;
    int rc = check_layer_args(ctx, k_i, k_next, gate_type, left, right, z);
    if (rc) return rc;
    uint32_t log_p = 0;
    while ((1u << log_p) < nshards) ++log_p;
    if (nshards == 0 || (1u << log_p) != nshards || (int)log_p > k_next || shard >= nshards)
        return ctx->fail(GKR_ERR_INVALID, "shard count must be a power of two <= 2^k_next and shard < count");
    if (!all_canonical(W, (size_t)1 << k_next)) return ctx->fail(GKR_ERR_NON_CANONICAL, "W entry >= r");
    GKR_ENTER(ctx);
    hipStream_t s = ctx->stream;
    gkr_layer_session* S = new gkr_layer_session();
    S->k = k_next;
    S->kc = k_next - (int)log_p;
    S->rounds = (uint32_t)(2 * k_next) - log_p;
    S->cells = (size_t)1 << (2 * k_next - log_p);
    DevBuf<uint8_t> dgt;
    DevBuf<uint32_t> dl, dr;
    DevBuf<Fr> dW;
    rc = alloc_layer_session(ctx, S, S->cells, (size_t)1 << k_next, (size_t)1 << S->kc);
    if (!rc) rc = upload_gates(ctx, (size_t)1 << k_i, gate_type, left, right, dgt, dl, dr);
    if (rc) {
        free_layer_session(S);
        return rc;
    }
    hipError_t e = dW.alloc((size_t)1 << k_next);
    if (e == hipSuccess) e = hipMemcpyAsync(dW.p, W, sizeof(Fr) << k_next, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) {
        free_layer_session(S);
        return ctx->hip_fail(e, "upload W");
    }
    rc = build_predicates(ctx, k_i, k_next, dgt.p, dl.p, dr.p, z, S->A, S->M, log_p, shard);
    if (rc) {
        free_layer_session(S);
        return rc;
    }
    (void)hipMemsetAsync(S->d_dep, 0, 32 * sizeof(uint32_t), s);
    gkr::launch_to_mont(dW.p, S->Wb, 1u << k_next, s);
    gkr::launch_to_mont_strided(dW.p, S->Wc, 1u << S->kc, nshards, shard, s);
    gkr::launch_depends(dW.p, k_next, S->d_dep, 1, s);
    e = hipMemcpyAsync(S->dep, S->d_dep, 32 * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        free_layer_session(S);
        return ctx->hip_fail(e, "layer session setup");
    }
    *out = S;
    return GKR_OK;
}

// the redundant tail after the all-gather: explicit tables of 2^kc entries (A, M, Wc) and the scalar W(b*)
int gkr_layer_session_open_tables(gkr_ctx* ctx, int kc, const gkr_fr* A, const gkr_fr* M, const gkr_fr* wb,
                                  const gkr_fr* Wc, gkr_layer_session** out) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!A || !M || !wb || !Wc || !out || kc < 1 || kc > 14) return ctx->fail(GKR_ERR_INVALID, "bad tail tables");
    const size_t n = (size_t)1 << kc;
    if (!all_canonical(A, n) || !all_canonical(M, n) || !all_canonical(Wc, n) || !all_canonical(wb, 1))
        return ctx->fail(GKR_ERR_NON_CANONICAL, "tail table entry >= r");
    GKR_ENTER(ctx);
    hipStream_t s = ctx->stream;
    gkr_layer_session* S = new gkr_layer_session();
    S->k = kc;           // only c-variables remain: phase 1 from the first round
    S->kc = kc;
    S->round = (uint32_t)kc;   // counts as if k = kc b-rounds were already done
    S->rounds = (uint32_t)(2 * kc);
    S->cells = n;
    int rc = alloc_layer_session(ctx, S, n, 1, n);
    if (rc) {
        free_layer_session(S);
        return rc;
    }
    // W copies are kept in Montgomery form
    std::vector<Fr> wcm(n);
    for (size_t i = 0; i < n; ++i) wcm[i] = gkr::to_mont(to_dev(Wc[i]));
    Fr wbm = gkr::to_mont(to_dev(*wb));
    hipError_t e = hipMemcpyAsync(S->A, A, n * sizeof(Fr), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(S->M, M, n * sizeof(Fr), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(S->Wc, wcm.data(), n * sizeof(Fr), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(S->Wb, &wbm, sizeof(Fr), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        free_layer_session(S);
        return ctx->hip_fail(e, "tail session upload");
    }
    *out = S;
    return GKR_OK;
}

int gkr_layer_session_dep(gkr_ctx* ctx, const gkr_layer_session* S, uint32_t* out_dep, uint32_t count) {
    if (!ctx || !S || !out_dep || count > 32) return GKR_ERR_INVALID;
    for (uint32_t i = 0; i < count; ++i) out_dep[i] = S->dep[i];
    return GKR_OK;
}

int gkr_layer_session_rounds(const gkr_layer_session* S, uint32_t* done, uint32_t* total) {
    if (!S) return GKR_ERR_INVALID;
    if (done) *done = S->round;
    if (total) *total = S->rounds;
    return GKR_OK;
}

// partial sums of the current round over this shard: out = {c0, g(1), c2}, canonical
int gkr_layer_session_sums(gkr_ctx* ctx, gkr_layer_session* S, gkr_fr* out) {
    if (!ctx || !S || !out) return GKR_ERR_INVALID;
    if (S->round >= S->rounds) return ctx->fail(GKR_ERR_INVALID, "no round left in this session");
    GKR_ENTER(ctx);
    hipStream_t s = ctx->stream;
    const uint32_t h = (uint32_t)(S->cells / 2);
    const uint32_t phase = S->round < (uint32_t)S->k ? 0u : 1u;
    const uint32_t hb = phase == 0 ? (h >> S->kc) : 0u;
    const uint32_t nblk = gkr::layer_blocks(h);
    gkr::launch_layer_round(S->A, S->M, h, (uint32_t)S->kc, phase, hb, S->Wb, S->Wc, nblk, S->partials, gkr::single_layer(), s);
    const uint32_t ticket = ++ctx->ticket;
    gkr::launch_layer_round_reduce(S->partials, nblk, S->rec, ticket, gkr::single_layer(), s);
    HIP_TRY(ctx, hipGetLastError());
    int rc = wait_records(ctx, S->rec, 1, ticket);
    if (rc) return rc;
    memcpy(&out[0], &S->rec->c0, 32);
    memcpy(&out[1], &S->rec->g1, 32);
    memcpy(&out[2], &S->rec->c2, 32);
    return GKR_OK;
}

// bind the current variable to r
int gkr_layer_session_bind(gkr_ctx* ctx, gkr_layer_session* S, const gkr_fr* r) {
    if (!ctx || !S || !r) return GKR_ERR_INVALID;
    if (S->round >= S->rounds) return ctx->fail(GKR_ERR_INVALID, "no round left in this session");
    if (!all_canonical(r, 1)) return ctx->fail(GKR_ERR_NON_CANONICAL, "r >= modulus");
    GKR_ENTER(ctx);
    hipStream_t s = ctx->stream;
    gkr::h64::F r64;
    memcpy(&r64, r, 32);
    gkr::h64::make_fixed_mul(r64, S->rtab->w);
    const uint32_t h = (uint32_t)(S->cells / 2);
    const bool bphase = S->round < (uint32_t)S->k;
    // the W copy bound in this round: b-rounds fold Wb (2^k entries at the start), c-rounds fold Wc
    const uint32_t idx = bphase ? S->round : S->round - (uint32_t)S->k;
    const uint32_t hw = bphase ? (1u << (S->k - 1 - idx)) : (1u << (S->kc - 1 - idx));
    gkr::launch_fold_small(bphase ? S->Wb : S->Wc, hw, S->rtab, gkr::single_layer(), s);
    gkr::launch_layer_fold(S->A, S->M, h, S->rtab, gkr::single_layer(), s);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s));   // rtab is reused by the next bind
    S->cells = h;
    S->round += 1;
    return GKR_OK;
}

// when every local round is done: out = {A, M, Wc (canonical), W(b*) (canonical)} of this shard
int gkr_layer_session_tail(gkr_ctx* ctx, gkr_layer_session* S, gkr_fr* out) {
    if (!ctx || !S || !out) return GKR_ERR_INVALID;
    if (S->round != S->rounds || S->cells != 1) return ctx->fail(GKR_ERR_INVALID, "session still has rounds to run");
    GKR_ENTER(ctx);
    hipStream_t s = ctx->stream;
    Fr a, m, wc, wb;
    HIP_TRY(ctx, hipMemcpyAsync(&a, S->A, sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(&m, S->M, sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(&wc, S->Wc, sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(&wb, S->Wb, sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    out[0] = to_abi(a);
    out[1] = to_abi(m);
    out[2] = to_abi(gkr::from_mont(wc));
    out[3] = to_abi(gkr::from_mont(wb));
    return GKR_OK;
}

void gkr_layer_session_close(gkr_ctx* ctx, gkr_layer_session* S) {
    if (ctx) (void)hipSetDevice(ctx->device);
    free_layer_session(S);
}


}  // extern "C"
