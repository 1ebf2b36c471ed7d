// The plain multilinear sumcheck (prove_sumcheck, rust/src/gkr/sumcheck.rs:158-214): the multi-round passes with the host
// transcript, the per-round schedule (device transcript), one table split over ranks, the step-wise sessions.  C ABI: include/gkr_amd.h.
#include "capi_internal.h"

namespace gkr_host {

// ------------------------------------------------------------- plain MLE sumcheck, multi-round passes
// The host's share of one multi-round pass, scalar form (the IFMA-lane form is gkr_ifma_pass, mimc_ifma.cpp; same
// arguments, same results): per lane k and round t the round polynomial's coefficients from the sub-block sums, the
// vector's length, the challenge, then the sums with that variable bound; at the end the 2^J weights of the fold pass
// that binds the J variables, w_b = prod_t (bit_t(b) ? r_t : 1 - r_t), bit_0 = most significant, Montgomery form.
void host_pass_scalar(const uint64_t* sums, size_t sums_row_words, int count, int J, const uint32_t* final_len,
                             uint64_t (*c0)[16][4], uint64_t (*c1)[16][4], uint64_t (*r)[16][4], uint32_t (*len)[16],
                             uint64_t* weights, size_t w_row_words) {
    using gkr::h64::F;
    const F* cts = host_mimc_constants64();
    const F one_m = gkr::h64::to_mont(F{{1, 0, 0, 0}});
    for (int k = 0; k < count; ++k) {
        F S[gkr::kMleMaxSub], rm[gkr::kMlePassMaxRounds];
        memcpy(S, sums + (size_t)k * sums_row_words, sizeof(F) << J);
        for (int t = 0; t < J; ++t) {
            const int half = 1 << (J - t - 1);
            F lo = S[0], hi = S[half];
            for (int b = 1; b < half; ++b) {
                lo = gkr::h64::add(lo, S[b]);
                hi = gkr::h64::add(hi, S[half + b]);
            }
            const F d = gkr::h64::sub(hi, lo);
            const uint32_t ln = (final_len && t == J - 1) ? final_len[k] : (gkr::h64::is_zero(d) ? 1u : 2u);
            const F vec[2] = {d, lo};
            const F rc = host_multi_hash(vec + (2 - ln), (int)ln, cts);
            memcpy(c0[t][k], &lo, 32);
            memcpy(c1[t][k], &d, 32);
            memcpy(r[t][k], &rc, 32);
            len[t][k] = ln;
            rm[t] = gkr::h64::to_mont(rc);
            for (int b = 0; b < half; ++b) S[b] = gkr::h64::add(S[b], gkr::h64::mont_mul(gkr::h64::sub(S[half + b], S[b]), rm[t]));
        }
        if (!weights) continue;
        F* w = reinterpret_cast<F*>(weights + (size_t)k * w_row_words);
        F tmp[gkr::kMleMaxSub];
        tmp[0] = one_m;
        int cur = 1;
        for (int t = 0; t < J; ++t) {
            const F nr = gkr::h64::sub(one_m, rm[t]);
            for (int b = cur; b-- > 0;) {
                tmp[2 * b + 1] = gkr::h64::mont_mul(tmp[b], rm[t]);
                tmp[2 * b] = gkr::h64::mont_mul(tmp[b], nr);
            }
            cur <<= 1;
        }
        memcpy(w, tmp, sizeof(F) << J);
    }
}

// The host's share of one product pass of the layer sumcheck, scalar form (the IFMA-lane form is gkr_ifma_prod_pass,
// mimc_ifma.cpp; same arguments, same results).  Lane k: the 8 x 8 cross-sum matrix m[a][b] (W sub-block a times X
// sub-block b) and the Y sums sy[a] of its 2^J sub-blocks.  Round t (half = 2^(J-t-1)): with
//     P_xy = sum_{a < half} m[x half + a][y half + a],   S_x = sum_{a < half} sy[x half + a]
// the round polynomial is c2 X^2 + lin X + c0,  c0 = P_00 + S_0,  g(1) = P_11 + S_1,  c2 = P_11 - P_10 - P_01 + P_00,
// lin = g(1) - c0 - c2; the challenge is the hash of [c2, lin, c0] (2 + dep entries); binding the variable folds the
// matrix along both indices and sy along its one.  At the end the 2^J weights of the fold that binds the J variables.
int run_mle_batch_passes(gkr_ctx* ctx, const Fr* d_tables, int n, int batch, gkr_fr* out_coeffs, uint32_t* out_len,
                         gkr_fr* out_r, const MleTailArgs* tail) {
    using gkr::h64::F;
    const int n_out = tail ? tail->n_total : n, r_off = tail ? tail->round_offset : 0;
    const bool dbg = gkr::debug_timing();
    const auto dbg_t0 = std::chrono::steady_clock::now();
    auto dbg_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_t0).count(); };
    double dbg_a = 0, dbg_b = 0, dbg_c = 0, dbg_d = 0, dbg_e = 0;
    std::atomic<uint64_t> dbg_busy_ns{0};   // time inside process_chunk, all threads
    const size_t len = (size_t)1 << n;
    hipStream_t s = ctx->stream;
    // rounds per pass: up to 5 with the matrix-core fold (fewer passes, ~2.07 N elements moved instead of 2.29 N),
    // up to 3 with the v_mad_u64_u32 fold (option no_mfma_fold)
    const int jcap = gkr::opt(gkr::OPT_no_mfma_fold) ? 3 : gkr::kMlePassMaxRounds;
    const int jwant = gkr::opt(gkr::OPT_rounds_per_pass) > 0 ? (int)gkr::opt(gkr::OPT_rounds_per_pass) : jcap;
    const int jmax = jwant > jcap ? jcap : jwant;
    auto rounds_for = [&](int m) { return mle_pass_rounds(m, n, jmax); };
    const int j_first = rounds_for(n);
    const size_t work_len = len >> j_first;   // the first folded table
    Fr* work = nullptr;
    gkr::MleSubPartial* partials = nullptr;
    gkr::MleHostRecSub* rec = nullptr;
    Fr* h_w = nullptr;   // pinned: up to 32 Montgomery weights per sumcheck
    WS(ctx, "mlep.work", Fr, (size_t)batch * (work_len ? work_len : 1), work);
    WS(ctx, "mlep.partials", gkr::MleSubPartial, (size_t)batch * gkr::kMaxBlocksPerTable, partials);
    unsigned char* plans = nullptr;   // per sumcheck: the digit matrix of the matrix-core fold pass
    WS(ctx, "mlep.plans", unsigned char, (size_t)batch * gkr::mle_fold_plan_bytes(), plans);
    HIP_TRY(ctx, ctx->pinned_host("mlep.rec", sizeof(gkr::MleHostRecSub) * batch, reinterpret_cast<void**>(&rec)));
    HIP_TRY(ctx, ctx->pinned_host("mlep.w", sizeof(Fr) * gkr::kMleMaxSub * batch, reinterpret_cast<void**>(&h_w)));
    // Latency-bound passes (a few sumchecks of moderate size: at most kFusedPublishBytes read per launch) publish from their
    // last block instead of through k_mle_sub_reduce: one launch and one dependent-launch gap less per pass on the round
    // path of a lone sumcheck.  Streaming passes keep the second launch (see mle_publish_from_last_block).
    const bool no_fused = gkr::opt(gkr::OPT_no_fused_reduce) != 0;
    constexpr double kFusedPublishBytes = 64.0 * 1024 * 1024;
    uint32_t* d_arrivals = nullptr;
    if (!no_fused && (double)len * 32.0 <= kFusedPublishBytes) {
        constexpr size_t kCounters = 4096;   // (one size: zeroed once per allocation, every pass leaves them zero)
        WS(ctx, "mlep.arrivals", uint32_t, kCounters, d_arrivals);
        if (ctx->mle_arrivals_zeroed != d_arrivals) {
            HIP_TRY(ctx, hipMemsetAsync(d_arrivals, 0, sizeof(uint32_t) * kCounters, s));
            ctx->mle_arrivals_zeroed = d_arrivals;
        }
    }
    constexpr uint32_t kFusedPublishBlocks = 256;   // blocks per launch: each pays one L2 write-back (~30 ns, one after the other)
    auto fused_publish = [&](int b0, int nb, uint32_t nblk, double bytes_read, uint32_t ticket, int jout, gkr::MlePublish& pub) {
        if (!d_arrivals || b0 + nb > 4096 || bytes_read > kFusedPublishBytes || (uint64_t)nblk * nb > kFusedPublishBlocks || nblk > 128u) return false;
        pub.rec = rec + b0;
        pub.arrivals = d_arrivals + b0;
        pub.ticket = ticket;
        pub.jout = (uint32_t)jout;
        return true;
    };
    const bool ifma = host_ifma_ready();
    static const bool scalar_book = gkr::process_switch("GKR_HOST_PASS_SCALAR");   // A/B switch: host_pass_scalar even where the CPU has IFMA
    std::vector<uint32_t> dep_last(batch, 0);
    gkr::SpinPool* pool = ctx->host_pool();
    // Sumchecks a hashing thread takes at a time, per group: sixteen (full IFMA calls: throughput) when the group has plenty
    // for every thread; otherwise ONE chunk per thread where that fits the sixteen lanes -- a pass's J hashes of a sumcheck
    // are a serial chain, so a group of 128 on 14 threads is done in one chain of 16-lane calls filled to 10 (J x 20 us)
    // instead of two chains of 8-lane calls (2 x J x 16 us), at the same cost per hash; the option hash_chunk forces 8 or 16
    const int hash_threads = pool->workers() + 1;
    const int forced = gkr::opt(gkr::OPT_hash_chunk) == 8 || gkr::opt(gkr::OPT_hash_chunk) == 16 ? (int)gkr::opt(gkr::OPT_hash_chunk) : 0;
    auto group_chunk = [hash_threads, forced](int nb) -> uint32_t {
        if (forced) return (uint32_t)forced;
        if (nb >= 32 * hash_threads) return 16u;
        const int per = (nb + hash_threads - 1) / hash_threads;
        return (uint32_t)(per <= 8 ? 8 : (per <= 16 ? per : 16));
    };

    // Groups of ~4 GiB of tables, at least four and at most eight (1024 x 2^20: eight groups of 128); sixteen for batches
    // beyond 96 GiB (4096 x 2^20: 4.64e11 field-ops/s with sixteen groups of 256, 4.48e11 with eight of 512).  Larger launches
    // stream slightly better, smaller groups feed the host's hashing more evenly and leave a shorter exposed tail (the
    // last group's late passes); measured on MI355X, 1024 x 2^20, interleaved repeats on one box, ms per step with
    // 14 / 3 / 2 host threads: 4 groups, all pass 0s queued first 12.3-12.9 / 15.0-16.3 / 18.5-19.0; 8 groups, pass 0
    // queue depth 2 (below) 12.0-12.5 / 13.5-14.0 / 16.5-17.8; 6, 10 and 12 groups in between.
    // Sumchecks hashed ON THE DEVICE (kernels_transcript.hip): the first n_dev of the batch run as one chain of kernels on
    // a stream of their own -- pass, the pass's rounds with MiMC7 on eight lanes per element, fold, ... -- without the host;
    // the host hashes the rest as always.  A device-hashed pass takes 0.33 ms per round whatever the number of sumchecks
    // (the chain of 2 x 91 x 4 dependent products), so this is for steps that are bound by the host's hashing: a rank with
    // two or three host threads, tables so small that the GPU is mostly idle.  Option device_hash_percent = share of the batch
    // (0 = none, the default).
    const long long dp = gkr::opt(gkr::OPT_device_hash_percent);
    int dev_percent = dp < 0 ? 0 : (dp > 90 ? 90 : (int)dp);
    int n_dev = 0;
    if (dev_percent > 0 && !tail && batch >= 64 && j_first >= 1) {
        n_dev = (int)((long long)batch * dev_percent / 100) & ~7;
        if (batch - n_dev < 16) n_dev = (batch - 16) & ~7;
        if (n_dev < 8) n_dev = 0;
    }
    const int host_b0 = n_dev, host_batch = batch - n_dev;
    const double batch_bytes = (double)host_batch * (double)len * 32.0;
    int want_groups = (int)(batch_bytes / (4.0 * 1024 * 1024 * 1024));
    want_groups = want_groups < 4 ? 4 : (want_groups > 8 ? (batch_bytes > 96.0 * 1024 * 1024 * 1024 ? 16 : 8) : want_groups);
    // Small tables (BASELINE configs[1]: 4096 x 2^16) are bound by the host's hashing, not by the stream: sixteen groups
    // with pass 0 of four of them queued ahead keep the hashing threads fed from start to end (MI355X, 14 threads, ms per
    // 4096 x 2^16: 4 groups 8.1 - 8.2, 8 groups 8.1, 16 groups 7.2, 16 groups / depth 4 7.0 - 7.2, 32 groups / depth 8 7.1;
    // profiles/r03/f_n16_groups*.jsonl)
    const bool small_tables = n <= 17 && host_batch >= 256;
    if (small_tables) want_groups = 16;
    // A rank with two or three host threads (eight ranks on a 16-core host) is bound by its hashing: smaller groups shorten
    // the stretch before the first hashes and after the last fold (1024 x 2^20, two threads: 16.3 - 16.7 ms with eight
    // groups, 16.0 with sixteen; profiles/r03/w_two_host_threads_group_size.jsonl)
    if (hash_threads <= 3 && host_batch >= 256 && want_groups < 16) want_groups = 16;
    int group_size = host_batch >= 128 ? (host_batch + want_groups - 1) / want_groups : (host_batch >= 16 ? (host_batch + 1) / 2 : host_batch);
    if (n_dev && hash_threads <= 3 && host_batch >= 256) group_size = 64;   // (whole sixteen-lane chunks for both threads, as without a device share)
    if (gkr::opt(gkr::OPT_group_size) > 0) group_size = (int)gkr::opt(gkr::OPT_group_size);
    int groups = (host_batch + group_size - 1) / group_size;
    if (groups > kMaxGroups) groups = kMaxGroups;
    struct Group {
        int b0 = 0, nb = 0;
        int m = 0;          // variables left in the current table
        int j = 0;          // rounds the landed sums cover (the pass in flight produces 2^j sums)
        int round0 = 0;     // global index of the first of those rounds
        int state = 0;      // 0 waiting for the GPU, 1 hashing, 2 finished
        uint32_t ticket = 0;
        std::atomic<uint64_t> claim{0};   // (generation << 32) | next sumcheck; generation = pass number + 1
        std::atomic<int> done{0};
        int pass = 0;
        int index = 0;
        hipStream_t chain = nullptr;   // a device-hashed group: the stream its whole chain runs on
        bool on_host = false;          // the host tail: the group's tables (2^m entries each) are in h_tail, the device is done with them
    };
    std::vector<Group> grp(groups);
    HIP_TRY(ctx, ctx->aux_stream(groups));
    // The host tail: the last fold pass of a sumcheck works on a table of 2^7 entries and fewer -- 128 products, and ~30 us as a
    // device pass (launch, 15 us of kernel, the record's way back).  For a few sumchecks at a time (a latency chain, not a
    // throughput problem) the pass before it leaves its folded table in pinned memory as well, and the host binds the remaining
    // variables itself: exact field arithmetic, the same canonical sums.
    constexpr uint32_t kMleTailLog2 = 7;
    const bool tail_on = gkr::opt(gkr::OPT_host_tail_log2) >= 0 && n_dev == 0 &&
                         batch <= (gkr::opt(gkr::OPT_host_tail_max_batch) > 0 ? gkr::opt(gkr::OPT_host_tail_max_batch) : 8);
    Fr* h_tail = nullptr;
    if (tail_on) HIP_TRY(ctx, ctx->pinned_host("mlep.tail", sizeof(Fr) * ((size_t)batch << kMleTailLog2), reinterpret_cast<void**>(&h_tail)));
    {
        int start = 0;
        for (int g = 0; g < groups; ++g) {
            grp[g].index = g;
            const int end = (int)((long long)host_batch * (g + 1) / groups);
            grp[g].b0 = host_b0 + start;
            grp[g].nb = end - start;
            start = end;
            grp[g].m = n;
            grp[g].j = j_first;
        }
    }
    // pass 0: sub-block sums of the input tables
    auto launch_first = [&](Group& G) {
        const int b0 = G.b0, nb = G.nb;
        G.ticket = ++ctx->ticket;
        if (len <= gkr::kSmallPassEntries) {
            Timed t(ctx, "mle_pass_small", (double)nb * len * 32.0, G.chain, true);
            gkr::launch_mle_multifold_small(0, d_tables + (size_t)b0 * len, len, nullptr, 0, (uint32_t)len, (uint32_t)G.j, nb,
                                            h_w + (size_t)b0 * gkr::kMleMaxSub, rec + b0, G.ticket, G.chain ? G.chain : s);
            return;
        }
        const uint32_t nblk = gkr::mle_pass_blocks((uint32_t)len, (uint32_t)G.j, nb);
        gkr::MleSubPartial* part = partials + (size_t)b0 * gkr::kMaxBlocksPerTable;
        gkr::MlePublish pub;
        const bool fused = fused_publish(b0, nb, nblk, (double)nb * len * 32.0, G.ticket, G.j, pub);
        hipStream_t st0 = G.chain ? G.chain : s;
        {
            Timed t(ctx, G.chain ? "mle_sub_sums_dev" : "mle_sub_sums", (double)nb * len * 32.0, st0, fused);
            gkr::launch_mle_sub_sums(d_tables + (size_t)b0 * len, len, (uint32_t)len, nb, nblk, part, st0, fused ? &pub : nullptr);
        }
        if (fused) return;
        Timed t(ctx, "mle_sub_reduce", 0.0, st0, true);
        gkr::launch_mle_sub_reduce(part, nblk, (uint32_t)G.j, nb, rec + b0, G.ticket, st0);
    };
    // a fold pass: bind the jin variables just hashed, produce the sums of the next jout rounds
    const bool no_late = gkr::opt(gkr::OPT_no_late_stream) != 0;
    hipStream_t late = s;
    if (!no_late && groups > 1) HIP_TRY(ctx, ctx->late_stream(&late));
    auto launch_fold = [&](Group& G, int jin) {
        const int b0 = G.b0, nb = G.nb;
        const size_t src_len = (size_t)1 << G.m, S = src_len >> jin;
        const bool from_input = (G.m == n);
        const Fr* src = from_input ? d_tables + (size_t)b0 * len : work + (size_t)b0 * work_len;
        const size_t src_stride = from_input ? len : work_len;
        Fr* dst = work + (size_t)b0 * work_len;
        // small source tables: a latency-bound late pass, not to be queued behind other groups' streaming passes
        hipStream_t st = G.chain ? G.chain : ((!from_input && src_len <= ((size_t)1 << 16)) ? late : s);
        G.m -= jin;
        G.round0 += jin;
        G.j = rounds_for(G.m);
        G.ticket = ++ctx->ticket;
        const double bytes = (double)nb * ((double)src_len + (double)S) * 32.0;
        if (S <= gkr::kSmallPassEntries) {
            // (the table it leaves is small enough for the host to finish, and there is a pass left to save)
            const bool exports = tail_on && !G.chain && S <= ((size_t)1 << kMleTailLog2) && G.m - G.j > 0;
            Timed t(ctx, "mle_pass_small", bytes, st, true);
            gkr::launch_mle_multifold_small(jin, src, src_stride, dst, work_len, (uint32_t)S, (uint32_t)G.j, nb,
                                            h_w + (size_t)b0 * gkr::kMleMaxSub, rec + b0, G.ticket, st,
                                            exports ? h_tail + ((size_t)b0 << kMleTailLog2) : nullptr, 1u << kMleTailLog2);
            G.on_host = exports;
            return;
        }
        const uint32_t nblk = gkr::mle_multifold_blocks((uint32_t)S, (uint32_t)G.j, nb);
        gkr::MleSubPartial* part = partials + (size_t)b0 * gkr::kMaxBlocksPerTable;
        unsigned char* plan = plans + (size_t)b0 * gkr::mle_fold_plan_bytes();
        if (gkr::mle_multifold_uses_mfma((uint32_t)S, nblk)) {
            // the digit matrices only depend on the weights the host just wrote: built on the side stream, so the
            // main stream (busy with another group's pass) pays one event wait, not a launch round trip
            // (one group: nothing else is streaming, and the event between the two streams costs the round path ~10 us
            // more than a second launch on the same stream -- 15 us against 5 between the plan and the fold)
            const bool plan_inline = gkr::opt(gkr::OPT_plan_main) != 0;
            if (plan_inline || st != s || groups == 1) {
                gkr::launch_mle_fold_plan(jin, h_w + (size_t)b0 * gkr::kMleMaxSub, plan, nb, st);
            } else {
                {
                    Timed t(ctx, "mle_fold_plan", 0.0, ctx->aux, true);
                    gkr::launch_mle_fold_plan(jin, h_w + (size_t)b0 * gkr::kMleMaxSub, plan, nb, ctx->aux);
                }
                (void)hipEventRecord(ctx->aux_events[G.index], ctx->aux);
                (void)hipStreamWaitEvent(s, ctx->aux_events[G.index], 0);
            }
        }
        gkr::MlePublish pub;
        const bool fused = fused_publish(b0, nb, nblk, (double)nb * (double)src_len * 32.0, G.ticket, G.j, pub);
        {
            // late passes run beside other groups' streaming passes: their elapsed time is not their own cost, so they
            // are booked under their own name and stay out of the streaming fold pass's bandwidth figure
            Timed t(ctx, G.chain ? "mle_multifold_dev" : (st == s ? "mle_multifold" : "mle_multifold_late"), bytes, st, fused);
            gkr::launch_mle_multifold(jin, src, src_stride, dst, work_len, (uint32_t)S, nb, nblk, h_w + (size_t)b0 * gkr::kMleMaxSub,
                                      plan, part, st, fused ? &pub : nullptr);
        }
        if (fused) return;
        Timed t(ctx, "mle_sub_reduce", 0.0, st, true);
        gkr::launch_mle_sub_reduce(part, nblk, (uint32_t)G.j, nb, rec + b0, G.ticket, st);
    };
    // the same pass on the host (the group's tables are in h_tail): T'[i] = sum_t w_t T[t S + i], then the sub-block sums of the
    // next rounds into the record the device pass would have written
    auto host_fold = [&](Group& G, int jin) {
        using gkr::h64::F;
        const size_t S = ((size_t)1 << G.m) >> jin;
        G.m -= jin;
        G.round0 += jin;
        G.j = rounds_for(G.m);
        G.ticket = ++ctx->ticket;
        const size_t nsub = (size_t)1 << G.j, sub = S >> G.j;
        for (int b = G.b0; b < G.b0 + G.nb; ++b) {
            F* T = reinterpret_cast<F*>(h_tail + ((size_t)b << kMleTailLog2));
            const F* w = reinterpret_cast<const F*>(h_w + (size_t)b * gkr::kMleMaxSub);
            for (size_t i = 0; i < S; ++i) {
                gkr::h64::Wide acc = gkr::h64::wide_zero();
                for (size_t t = 0; t < ((size_t)1 << jin); ++t) gkr::h64::wide_mac(acc, T[t * S + i], w[t]);
                T[i] = gkr::h64::wide_reduce(acc);
            }
            F* sums = reinterpret_cast<F*>(rec[b].sums);
            for (size_t a = 0; a < nsub; ++a) {
                F v = T[a * sub];
                for (size_t i = 1; i < sub; ++i) v = gkr::h64::add(v, T[a * sub + i]);
                sums[a] = v;
            }
            __atomic_store_n(&rec[b].seq, G.ticket, __ATOMIC_RELEASE);
        }
    };
    // the J rounds of up to sixteen sumchecks whose sub-block sums have landed
    auto process_chunk = [&](const Group& G, int b_first, int count) {
        const int J = G.j;
        uint64_t c0[gkr::kMlePassMaxRounds][16][4], c1[gkr::kMlePassMaxRounds][16][4], r[gkr::kMlePassMaxRounds][16][4];
        uint32_t ln[gkr::kMlePassMaxRounds][16], final_len[16];
        const bool final_pass = G.round0 + J == n;
        for (int i = 0; i < count; ++i) {
            if (G.round0 == 0) dep_last[b_first + i] = tail && tail->dep_last ? tail->dep_last[b_first + i] : rec[b_first + i].dep;
            final_len[i] = dep_last[b_first + i] ? 2u : 1u;
        }
        static_assert(sizeof(gkr::MleHostRecSub) % 8 == 0, "hand-off records are addressed in 64-bit words");
        const uint64_t* sums = reinterpret_cast<const uint64_t*>(rec[b_first].sums);
        uint64_t* weights = G.m - J > 0 ? reinterpret_cast<uint64_t*>(h_w + (size_t)b_first * gkr::kMleMaxSub) : nullptr;
        (ifma && count >= 3 && !scalar_book ? gkr::gkr_ifma_pass : host_pass_scalar)(
            sums, sizeof(gkr::MleHostRecSub) / 8, count, J, final_pass ? final_len : nullptr, c0, c1, r, ln, weights, 4 * gkr::kMleMaxSub);
        for (int i = 0; i < count; ++i) {
            const int b = b_first + i;
            for (int t = 0; t < J; ++t) {
                const int round = r_off + G.round0 + t;
                gkr_fr* oc = out_coeffs + ((size_t)b * n_out + round) * 2;
                memset(&oc[0], 0, 32);
                if (ln[t][i] == 2) memcpy(&oc[0], c1[t][i], 32);
                memcpy(&oc[1], c0[t][i], 32);
                out_len[(size_t)b * n_out + round] = ln[t][i];
                memcpy(&out_r[(size_t)b * n_out + round], r[t][i], 32);
            }
        }
    };
    // A hashing thread takes its next chunk from the group that is EARLIEST in its schedule (generation = pass number):
    // the hashes of an early pass release the next streaming pass, whose results are most of the host work still to
    // come, while the late passes' hashes release microseconds of GPU work -- they fill the time in between.
    const std::function<bool()> try_work = [&]() -> bool {
        for (;;) {
            int best = -1;
            uint64_t best_c = 0;
            for (int g = 0; g < groups; ++g) {
                const uint64_t c = grp[g].claim.load(std::memory_order_acquire);
                if ((c >> 32) == 0 || (uint32_t)c >= (uint32_t)grp[g].nb) continue;
                if (best < 0 || (c >> 32) < (best_c >> 32)) {
                    best = g;
                    best_c = c;
                }
            }
            if (best < 0) return false;
            Group& G = grp[best];
            const uint32_t first = (uint32_t)best_c;
            const uint32_t left = (uint32_t)G.nb - first;
            const uint32_t chunk_tables = group_chunk(G.nb);
            const uint32_t take = left < chunk_tables ? left : chunk_tables;
            if (!G.claim.compare_exchange_strong(best_c, best_c + take, std::memory_order_acq_rel)) continue;   // lost a race: look again
            const double t_in = dbg ? dbg_us() : 0.0;
            process_chunk(G, G.b0 + (int)first, (int)take);
            G.done.fetch_add((int)take, std::memory_order_release);
            if (dbg) dbg_busy_ns.fetch_add((uint64_t)((dbg_us() - t_in) * 1e3), std::memory_order_relaxed);
            return true;
        }
    };
    auto records_landed = [&](const Group& G) {
        for (int i = G.nb - 1; i >= 0; --i)
            if (__atomic_load_n(&rec[G.b0 + i].seq, __ATOMIC_ACQUIRE) != G.ticket) return false;
        return true;
    };
    dbg_a = dbg_us();
    gkr::SpinPool::Session session(pool, &try_work);
    int rc = GKR_OK;
    // Pass 0 of the first `depth` groups is queued up front, pass 0 of a later group right behind the first fold of an
    // earlier one: the stream then alternates between pass 0 of later groups and the first fold of earlier ones
    // (P0 P0 F0 P0 F1 P0 F2 F3 with four groups), and the host's hashing -- which with few threads takes as long as the
    // GPU's work -- is fed from the first millisecond to the last instead of in one burst after all the pass 0s.
    // (All pass 0s first: 2 host threads 19.0 ms per 1024 x 2^20 at 77 % hashing occupancy, 3 threads 15.6 ms at 63 %.)
    const int depth_env = gkr::opt(gkr::OPT_pass_queue_depth) > 0 ? (int)gkr::opt(gkr::OPT_pass_queue_depth) : 0;
    const int depth = depth_env ? depth_env : (small_tables ? 4 : 2);
    int next_first = 0;   // groups [next_first, groups): pass 0 still to launch
    if (next_first < groups) launch_first(grp[next_first++]);   // (the host's first sums before the device chain's first pass)
    // the device-hashed sumchecks [0, n_dev): their whole chain is queued here, on its own stream
    Group dev;
    hipStream_t chain = nullptr;
    gkr_fr* stage_c = nullptr;
    gkr_fr* stage_r = nullptr;
    uint32_t* stage_len = nullptr;
    if (n_dev) {
        HIP_TRY(ctx, ctx->chain_stream(&chain));
        uint32_t* dep_dev = nullptr;
        WS(ctx, "mlep.dev_dep", uint32_t, (size_t)n_dev, dep_dev);
        HIP_TRY(ctx, ctx->pinned_host("mlep.stage_c", sizeof(gkr_fr) * 2 * (size_t)n_dev * n, reinterpret_cast<void**>(&stage_c)));
        HIP_TRY(ctx, ctx->pinned_host("mlep.stage_r", sizeof(gkr_fr) * (size_t)n_dev * n, reinterpret_cast<void**>(&stage_r)));
        HIP_TRY(ctx, ctx->pinned_host("mlep.stage_len", sizeof(uint32_t) * (size_t)n_dev * n, reinterpret_cast<void**>(&stage_len)));
        dev.b0 = 0;
        dev.nb = n_dev;
        dev.m = n;
        dev.j = j_first;
        dev.index = groups;
        dev.chain = chain;
        launch_first(dev);
        for (bool first = true;; first = false) {
            const int J = dev.j;
            const bool final_pass = dev.round0 + J == n;
            {
                Timed t(ctx, "mle_pass_hash_dev", 0.0, chain);
                gkr::launch_mle_pass_hash_lanes(rec, (uint32_t)n_dev, (uint32_t)J, (uint32_t)dev.round0, (uint32_t)n, final_pass, first, ctx->d_cts, dep_dev,
                                                dev.m - J > 0 ? h_w : nullptr, reinterpret_cast<Fr*>(stage_c), stage_len, reinterpret_cast<Fr*>(stage_r), chain);
            }
            if (dev.m - J <= 0) break;
            launch_fold(dev, J);
        }
    }
    while (next_first < groups && next_first < depth) launch_first(grp[next_first++]);
    dbg_b = dbg_us();
    int active = groups;
    auto t0 = std::chrono::steady_clock::now();
    uint32_t idle = 0;
    while (active > 0 && rc == GKR_OK) {
        bool progress = false;
        for (int g = 0; g < next_first; ++g) {
            Group& G = grp[g];
            if (G.state == 0 && records_landed(G)) {
                G.done.store(0, std::memory_order_relaxed);
                G.claim.store(((uint64_t)(++G.pass) << 32), std::memory_order_release);
                G.state = 1;
                progress = true;
            } else if (G.state == 1 && G.done.load(std::memory_order_acquire) == G.nb) {
                G.claim.store(0, std::memory_order_release);
                if (G.m - G.j > 0) {
                    const bool first_fold = G.m == n;
                    if (G.on_host)
                        host_fold(G, G.j);
                    else
                        launch_fold(G, G.j);
                    G.state = 0;
                    if (first_fold && next_first < groups) launch_first(grp[next_first++]);
                } else {
                    G.state = 2;
                    --active;
                    if (next_first < groups) launch_first(grp[next_first++]);   // single-pass sumchecks: no fold to ride on
                }
                progress = true;
            }
        }
        if (progress) {
            idle = 0;
            t0 = std::chrono::steady_clock::now();   // the limit is on time without progress, not on the whole call
            if (hipError_t le = hipGetLastError(); le != hipSuccess) rc = ctx->hip_fail(le, "launch of a sumcheck pass");
            continue;
        }
        if (try_work()) continue;
        GKR_CPU_RELAX();
        if ((++idle & 0xFFFF) == 0) {
            hipError_t q = hipStreamQuery(s);
            if (q != hipSuccess && q != hipErrorNotReady) rc = ctx->hip_fail(q, "stream failed during a sumcheck pass");
            else if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
                rc = ctx->fail(GKR_ERR_HIP, "timed out waiting for the device to publish a pass");
        }
    }
    dbg_c = dbg_us();
    session.close();
    if (rc) {
        (void)hipStreamSynchronize(s);
        if (late != s) (void)hipStreamSynchronize(late);
        if (chain) (void)hipStreamSynchronize(chain);
        ctx->mle_arrivals_zeroed = nullptr;   // (a pass that was given up may have left its counters half way)
        return rc;
    }
    dbg_d = dbg_us();
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (late != s) HIP_TRY(ctx, hipStreamSynchronize(late));
    if (n_dev) {
        HIP_TRY(ctx, hipStreamSynchronize(chain));
        // (tail == nullptr here: n_out = n, r_off = 0 -- the staging arrays have the caller's layout)
        memcpy(out_coeffs, stage_c, sizeof(gkr_fr) * 2 * (size_t)n_dev * n);
        memcpy(out_r, stage_r, sizeof(gkr_fr) * (size_t)n_dev * n);
        memcpy(out_len, stage_len, sizeof(uint32_t) * (size_t)n_dev * n);
    }
    dbg_e = dbg_us();
    if (ctx->pending.size() > 8192) ctx->drain_events();   // otherwise when the profile is read
    if (dbg)
        fprintf(stderr,
                "[gkr timing] setup %.0f us, first launches %.0f, loop %.0f, end_session %.0f, sync %.0f, drain %.0f; hashing %.0f us "
                "over %d threads = %.0f%% of the loop\n",
                dbg_a, dbg_b - dbg_a, dbg_c - dbg_b, dbg_d - dbg_c, dbg_e - dbg_d, dbg_us() - dbg_e, dbg_busy_ns.load() * 1e-3,
                pool->workers() + 1, dbg_busy_ns.load() * 1e-3 / ((dbg_c - dbg_b) * (pool->workers() + 1)) * 100.0);
    return GKR_OK;
}

// ------------------------------------------------------------- plain MLE sumcheck
// Length rule of prove_sumcheck (sumcheck.rs:158-214): rounds 1..n-1 drop a zero
// linear coefficient (add_poly, poly.rs:324-327); the last round has two
// coefficients iff the table depends on x_n (no merge, sumcheck.rs:206-207).
int run_mle_batch(gkr_ctx* ctx, const Fr* d_tables, int n, int batch, gkr_fr* out_coeffs, uint32_t* out_len,
                  gkr_fr* out_r) {
    const size_t len = (size_t)1 << n;
    const size_t rounds = (size_t)batch * n;
    const bool host_tx = ctx->transcript == GKR_TRANSCRIPT_HOST;
    const bool per_round = gkr::opt(gkr::OPT_mle_per_round) != 0;
    if (host_tx && !per_round) return run_mle_batch_passes(ctx, d_tables, n, batch, out_coeffs, out_len, out_r);
    Fr *work = nullptr, *d_coeffs = nullptr, *d_r = nullptr;
    uint32_t *d_len = nullptr, *d_dep = nullptr;
    gkr::MlePartial* partials = nullptr;
    // groups of the host pipeline are smaller than the batch and may use more blocks per table
    const uint32_t max_nblk = gkr::mle_blocks_per_table((uint32_t)(len / 2), 1u);
    WS(ctx, "mle.work", Fr, (size_t)batch * (len / 2), work);
    WS(ctx, "mle.partials", gkr::MlePartial, (size_t)batch * max_nblk, partials);
    hipStream_t s = ctx->stream;

    if (host_tx) {
        gkr::MleHostRec* rec = nullptr;
        gkr::FixedMul* h_rtab = nullptr;   // pinned: host writes r_j's multiplier table, the next fold kernel reads it
        HIP_TRY(ctx, ctx->pinned_host("mle.rec", sizeof(gkr::MleHostRec) * batch, reinterpret_cast<void**>(&rec)));
        HIP_TRY(ctx, ctx->pinned_host("mle.rtab", sizeof(gkr::FixedMul) * batch, reinterpret_cast<void**>(&h_rtab)));
        const gkr::h64::F* cts = host_mimc_constants64();
        std::vector<uint32_t> dep_last(batch, 0);
        gkr::SpinPool* pool = ctx->host_pool();
        const uint32_t chunk_tables = (uint32_t)hash_chunk_size(batch, pool->workers() + 1);

        // The batch is cut into groups that advance through their rounds independently:
        //   GPU (one in-order stream):  sums/fold of group g, round j  ->  reduce -> pinned records
        //   host workers:               MiMC7 of every sumcheck of a group whose records have landed
        //   this thread:                notices landed records, hands them to the workers, launches the
        //                               next round of a group as soon as its hashes are done
        // so one group's hash-bound late rounds overlap another group's bandwidth-bound early rounds.
        // Few, large groups: every group-round costs two launches.  A group starts once its
        // predecessor has left the bandwidth-bound rounds (round >= stagger).
        int group_size = batch >= 128 ? (batch + 3) / 4 : (batch >= 16 ? (batch + 1) / 2 : batch);
        if (gkr::opt(gkr::OPT_group_size) > 0) group_size = (int)gkr::opt(gkr::OPT_group_size);
        const int stagger = 0;   // measured on MI355X + 16 host CPUs: starting every group at once is best
        int groups = (batch + group_size - 1) / group_size;
        if (groups > kMaxGroups) groups = kMaxGroups;
        struct Group {
            int b0 = 0, nb = 0, round = 0;
            int state = 0;                     // 0 waiting for the GPU, 1 hashing, 2 finished
            uint32_t ticket = 0;
            std::atomic<uint64_t> claim{0};    // (generation << 32) | next table to hash; generation = round + 1
            std::atomic<int> done{0};
        };
        std::vector<Group> grp(groups);
        for (int g = 0; g < groups; ++g) {
            grp[g].b0 = (int)((long long)batch * g / groups);
            grp[g].nb = (int)((long long)batch * (g + 1) / groups) - grp[g].b0;
        }
        auto launch_round = [&](Group& G, int round) {
            const int b0 = G.b0, nb = G.nb;
            uint32_t nblk;
            gkr::MlePartial* part = partials + (size_t)b0 * max_nblk;
            if (round == 0) {
                const uint32_t h = (uint32_t)(len / 2);
                nblk = gkr::mle_blocks_per_table(h, nb);
                Timed t(ctx, "mle_sum_first", (double)nb * len * 32.0);
                gkr::launch_mle_sum_first(d_tables + (size_t)b0 * len, len, h, nb, nblk, part, s);
            } else {
                const uint32_t q = (uint32_t)(len >> (round + 1));
                const Fr* src = (round == 1) ? d_tables + (size_t)b0 * len : work + (size_t)b0 * (len / 2);
                const size_t src_stride = (round == 1) ? len : len / 2;
                if (q <= gkr::kSmallFoldQuarter) {
                    // small table: fold + sums + publish in one launch
                    G.ticket = ++ctx->ticket;
                    Timed t(ctx, "mle_fold_sum_small", (double)nb * 6.0 * q * 32.0);
                    gkr::launch_mle_fold_sum_small(src, src_stride, work + (size_t)b0 * (len / 2), len / 2, q, nb, h_rtab + b0,
                                                   rec + b0, G.ticket, s);
                    return;
                }
                nblk = gkr::mle_blocks_per_table(q, nb);
                Timed t(ctx, "mle_fold_sum", (double)nb * 6.0 * q * 32.0);
                gkr::launch_mle_fold_sum(src, src_stride, work + (size_t)b0 * (len / 2), len / 2, q, nb, nblk, h_rtab + b0, 1,
                                         part, s);
            }
            G.ticket = ++ctx->ticket;
            Timed t(ctx, "mle_round_reduce", 0.0);
            gkr::launch_mle_round_reduce(part, nblk, nb, rec + b0, G.ticket, s);
        };
        const bool ifma = host_ifma_ready();
        // length rule + outputs of one sumcheck's round, given its challenge
        auto round_len = [&](int b, int round, const gkr::h64::F& c1) -> uint32_t {
            if (round + 1 < n) return gkr::h64::is_zero(c1) ? 1u : 2u;
            return dep_last[b] ? 2u : 1u;
        };
        auto publish = [&](int b, int round, const gkr::h64::F& c0, const gkr::h64::F& c1, uint32_t ln, const gkr::h64::F& r) {
            gkr_fr* oc = out_coeffs + ((size_t)b * n + round) * 2;
            memset(&oc[0], 0, 32);
            if (ln == 2) memcpy(&oc[0], &c1, 32);
            memcpy(&oc[1], &c0, 32);
            out_len[(size_t)b * n + round] = ln;
            memcpy(&out_r[(size_t)b * n + round], &r, 32);
            if (round + 1 < n) gkr::h64::make_fixed_mul(r, h_rtab[b].w);
        };
        // up to eight sumchecks of one group: eight-lane IFMA hash when there are enough lanes to pay
        // for it, the scalar 4x64-bit code otherwise
        auto hash_chunk = [&](int b_first, int count, int round) {
            gkr::h64::F c0[kHashChunkMax], c1[kHashChunkMax];
            uint32_t ln[kHashChunkMax] = {};
            for (int i = 0; i < count; ++i) {
                const int b = b_first + i;
                memcpy(&c0[i], &rec[b].c0, 32);
                memcpy(&c1[i], &rec[b].c1, 32);
                if (round == 0) dep_last[b] = rec[b].dep;
                ln[i] = round_len(b, round, c1[i]);
            }
            if (ifma && count >= 3) {
                uint64_t vec[kHashChunkMax][3][4], out[kHashChunkMax][4];
                memset(vec, 0, sizeof vec);
                for (int i = 0; i < count; ++i) {
                    memcpy(vec[i][1], &c1[i], 32);
                    memcpy(vec[i][2], &c0[i], 32);
                }
                ifma_hash_chunk(vec, ln, count, out);
                for (int i = 0; i < count; ++i) {
                    gkr::h64::F r;
                    memcpy(&r, out[i], 32);
                    publish(b_first + i, round, c0[i], c1[i], ln[i], r);
                }
            } else {
                for (int i = 0; i < count; ++i) {
                    gkr::h64::F vec[2] = {c1[i], c0[i]};
                    const gkr::h64::F r = host_multi_hash(vec + (2 - ln[i]), (int)ln[i], cts);
                    publish(b_first + i, round, c0[i], c1[i], ln[i], r);
                }
            }
        };
        // one unit of work = up to eight sumchecks' hashes of the round their group is in
        const std::function<bool()> try_work = [&]() -> bool {
            for (int g = 0; g < groups; ++g) {
                Group& G = grp[g];
                uint64_t c = G.claim.load(std::memory_order_acquire);
                while ((uint32_t)c < (uint32_t)G.nb && (c >> 32) != 0) {
                    const uint32_t first = (uint32_t)c;
                    const uint32_t left = (uint32_t)G.nb - first;
                const uint32_t take = left < chunk_tables ? left : chunk_tables;
                    if (G.claim.compare_exchange_weak(c, c + take, std::memory_order_acq_rel)) {
                        hash_chunk(G.b0 + (int)first, (int)take, (int)(c >> 32) - 1);
                        G.done.fetch_add((int)take, std::memory_order_release);
                        return true;
                    }
                }
            }
            return false;
        };
        auto records_landed = [&](const Group& G) {
            for (int i = G.nb - 1; i >= 0; --i)
                if (__atomic_load_n(&rec[G.b0 + i].seq, __ATOMIC_ACQUIRE) != G.ticket) return false;
            return true;
        };
        gkr::SpinPool::Session session(pool, &try_work);
        int rc = GKR_OK;
        int started = 1;
        launch_round(grp[0], 0);
        int active = groups;
        auto t0 = std::chrono::steady_clock::now();
        uint32_t idle = 0;
        while (active > 0 && rc == GKR_OK) {
            bool progress = false;
            if (started < groups && (grp[started - 1].round >= stagger || grp[started - 1].state == 2)) {
                launch_round(grp[started], 0);
                ++started;
                progress = true;
            }
            for (int g = 0; g < started; ++g) {
                Group& G = grp[g];
                if (G.state == 0 && records_landed(G)) {
                    G.done.store(0, std::memory_order_relaxed);
                    G.claim.store(((uint64_t)(G.round + 1) << 32), std::memory_order_release);
                    G.state = 1;
                    progress = true;
                } else if (G.state == 1 && G.done.load(std::memory_order_acquire) == G.nb) {
                    G.claim.store(0, std::memory_order_release);
                    if (++G.round < n) {
                        launch_round(G, G.round);
                        G.state = 0;
                    } else {
                        G.state = 2;
                        --active;
                    }
                    progress = true;
                }
            }
            if (progress) {
                idle = 0;
                t0 = std::chrono::steady_clock::now();   // the limit is on time without progress
                if (hipError_t le = hipGetLastError(); le != hipSuccess) rc = ctx->hip_fail(le, "launch of a sumcheck round");
                continue;
            }
            if (try_work()) continue;   // nothing to schedule: help with the hashing
            GKR_CPU_RELAX();
            if ((++idle & 0xFFFF) == 0) {
                hipError_t q = hipStreamQuery(s);
                if (q != hipSuccess && q != hipErrorNotReady) rc = ctx->hip_fail(q, "stream failed during a sumcheck round");
                else if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
                    rc = ctx->fail(GKR_ERR_HIP, "timed out waiting for the device to publish a round");
            }
        }
        session.close();
        if (rc) {
            (void)hipStreamSynchronize(s);
            return rc;
        }
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(s));
        ctx->drain_events();
        return GKR_OK;
    }

    gkr::FixedMul* d_rtab = nullptr;
    WS(ctx, "mle.coeffs", Fr, rounds * 2, d_coeffs);
    WS(ctx, "mle.r", Fr, rounds, d_r);
    WS(ctx, "mle.rtab", gkr::FixedMul, rounds, d_rtab);
    WS(ctx, "mle.len", uint32_t, rounds, d_len);
    WS(ctx, "mle.dep", uint32_t, batch, d_dep);
    // round 1: sums only
    {
        const uint32_t h = (uint32_t)(len / 2);
        const uint32_t nblk = gkr::mle_blocks_per_table(h, batch);
        {
            Timed t(ctx, "mle_sum_first", (double)batch * len * 32.0);
            gkr::launch_mle_sum_first(d_tables, len, h, batch, nblk, partials, s);
        }
        {
            Timed t(ctx, "mle_round_hash", 0.0);
            gkr::launch_mle_round_hash(partials, nblk, 0, n, batch, ctx->d_cts, d_coeffs, d_len, d_r, d_rtab, d_dep, s);
        }
    }
    // rounds 2..n: fold with r_{j-1}, sum T_j in the same pass
    for (int round = 1; round < n; ++round) {
        const uint32_t q = (uint32_t)(len >> (round + 1));  // quarter of the source table
        const uint32_t nblk = gkr::mle_blocks_per_table(q, batch);
        const Fr* src = (round == 1) ? d_tables : work;
        const size_t src_stride = (round == 1) ? len : len / 2;
        {
            Timed t(ctx, "mle_fold_sum", (double)batch * 6.0 * q * 32.0);
            gkr::launch_mle_fold_sum(src, src_stride, work, len / 2, q, batch, nblk, d_rtab + (round - 1), n, partials,
                                     s);
        }
        {
            Timed t(ctx, "mle_round_hash", 0.0);
            gkr::launch_mle_round_hash(partials, nblk, round, n, batch, ctx->d_cts, d_coeffs, d_len, d_r, d_rtab, d_dep,
                                       s);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_coeffs, d_coeffs, rounds * 2 * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(out_len, d_len, rounds * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(out_r, d_r, rounds * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    ctx->drain_events();
    return GKR_OK;
}


}  // namespace gkr_host

// =========================================================================== C ABI

extern "C" {

// ---- plain multilinear sumcheck -------------------------------------------------

int gkr_sumcheck_mle_batch_device(gkr_ctx* ctx, const void* d_tables, int n, int batch, gkr_fr* out_coeffs,
                                  uint32_t* out_len, gkr_fr* out_r) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!d_tables || !out_coeffs || !out_len || !out_r || batch < 1 || batch > 65535)
        return ctx->fail(GKR_ERR_INVALID, "null pointer or batch out of range [1, 65535]");
    if (n < 2 || n > 30) return ctx->fail(GKR_ERR_INVALID, "n must be in [2, 30]");
    GKR_ENTER(ctx);
    return run_mle_batch(ctx, static_cast<const Fr*>(d_tables), n, batch, out_coeffs, out_len, out_r);
}

int gkr_sumcheck_mle(gkr_ctx* ctx, const gkr_fr* table, int n, gkr_fr* out_coeffs, uint32_t* out_len, gkr_fr* out_r) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!table || !out_coeffs || !out_len || !out_r) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    if (n < 2 || n > 30) return ctx->fail(GKR_ERR_INVALID, "n must be in [2, 30]");
    const size_t len = (size_t)1 << n;
    if (!all_canonical(table, len)) return ctx->fail(GKR_ERR_NON_CANONICAL, "table entry >= r");
    GKR_ENTER(ctx);
    DevBuf<Fr> d;
    HIP_TRY(ctx, d.alloc(len));
    HIP_TRY(ctx, hipMemcpyAsync(d.p, table, len * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    return run_mle_batch(ctx, d.p, n, 1, out_coeffs, out_len, out_r);
}

// ---- one plain sumcheck split over ranks, on the multi-round schedule -----------------------------------------------
// prove_sumcheck (sumcheck.rs:158-214) with the reduce over the hypercube (the rayon reduce of :62) split over P = 2^lp
// ranks.  Rank p holds, of every table T (2^n entries, variable 1 = most significant index bit), the shard
//     T_p[h * 2 + x_n] = T[h * 2P + 2p + x_n],   h < 2^(n - lp - 1):
// the index bits lp .. 1 are the rank, the last variable stays inside every shard.  Rounds bind the leading variable,
// so every pair (i, i + half) is rank-local while bits of h are bound; the sub-block sums a pass hands the host are
// linear in the table, so the whole table's 2^J sums are the sums over ranks of the shards' -- ONE all-reduce of
// 2^J (+ 2 flags) field elements per pass of J <= 5 rounds (n = 20 on 8 ranks: 3 exchanges + the gather, not 20), queued
// on the library's stream through the caller's gkr_exchange_dev; every rank then runs the same J rounds on the same
// sums and derives the same weights, no broadcast.  When 2^6 entries per shard are left they are gathered (one more
// all-reduce, of zero-padded buffers) into a tail table of 2^(6 + lp) entries on which every rank finishes the last
// rounds redundantly.  "Does T depend on x_n" (the last round's length, sumcheck.rs:206-207) is the OR over ranks of
// a neighbour compare inside each shard -- exact, no shard is compared across ranks.
size_t gkr_exchange_limbs_mle(int n, int log2_shards, int batch) {
    if (n < 2 || log2_shards < 0 || log2_shards > 16 || n - log2_shards < 1 || n - log2_shards > GKR_MAX_MLE_N || batch < 1) return 0;
    const int nl = n - log2_shards, t = nl < kMleShardTailLog2 ? nl : kMleShardTailLog2;
    const size_t per_pass = (size_t)batch * (gkr::kMleMaxSub + 2) * 8, gather = ((size_t)batch << (t + log2_shards)) * 8 + 8;
    return per_pass > gather ? per_pass : gather;
}

int gkr_sumcheck_mle_sharded_dev(gkr_ctx* ctx, const void* d_shards, int n, int log2_shards, int shard, int batch,
                                 const gkr_exchange_dev* exchange, gkr_fr* out_coeffs, uint32_t* out_len, gkr_fr* out_r,
                                 uint32_t* out_exchanges) {
    using gkr::h64::F;
    if (!ctx) return GKR_ERR_INVALID;
    if (!d_shards || !exchange || !exchange->fn || !exchange->d_limbs || !out_coeffs || !out_len || !out_r || batch < 1 || batch > 65535)
        return ctx->fail(GKR_ERR_INVALID, "null pointer or batch out of range [1, 65535]");
    const int lp = log2_shards, nl = n - lp;
    if (lp < 0 || lp > 16 || shard < 0 || shard >= (1 << lp)) return ctx->fail(GKR_ERR_INVALID, "shard must be in [0, 2^log2_shards), log2_shards in [0, 16]");
    if (n < 2 || nl < 1 || nl > GKR_MAX_MLE_N) return ctx->fail(GKR_ERR_INVALID, "n >= 2 and 1 <= n - log2_shards <= GKR_MAX_MLE_N needed");
    if (ctx->transcript != GKR_TRANSCRIPT_HOST) return ctx->fail(GKR_ERR_INVALID, "a sumcheck split over ranks needs the host transcript");
    if (exchange->capacity < gkr_exchange_limbs_mle(n, lp, batch)) return ctx->fail(GKR_ERR_INVALID, "the exchange buffer is smaller than gkr_exchange_limbs_mle(n, log2_shards, batch) int64");
    GKR_ENTER(ctx);
    hipStream_t s = ctx->stream;
    const Fr* shards = static_cast<const Fr*>(d_shards);
    const size_t len = (size_t)1 << nl;
    const int t_stop = nl < kMleShardTailLog2 ? nl : kMleShardTailLog2;   // variables every shard keeps for the gathered tail
    const int jmax = gkr::opt(gkr::OPT_no_mfma_fold) ? 3 : gkr::kMlePassMaxRounds;
    auto rounds_for = [&](int m) {
        int j = mle_pass_rounds(m, nl, jmax);
        if (m - j < t_stop) j = m - t_stop;
        return j;
    };
    long long* limbs = reinterpret_cast<long long*>(exchange->d_limbs);
    // everything that can fail locally is set up BEFORE the first exchange; from there on a failure is carried through the
    // remaining exchanges as a flag, so that no rank is left waiting inside a collective
    Fr *work = nullptr, *d_tail = nullptr;
    gkr::MleSubPartial* partials = nullptr;
    gkr::MleHostRecSub *rec = nullptr, *d_rec = nullptr;
    Fr* h_w = nullptr;
    unsigned char* plans = nullptr;
    uint32_t* h_fail = nullptr;
    const int j_first = nl > t_stop ? rounds_for(nl) : 0;
    const size_t work_len = j_first ? len >> j_first : 1;
    WS(ctx, "mlex.work", Fr, (size_t)batch * work_len, work);
    WS(ctx, "mlex.tail", Fr, (size_t)batch << (t_stop + lp), d_tail);
    WS(ctx, "mlex.partials", gkr::MleSubPartial, (size_t)batch * gkr::kMaxBlocksPerTable, partials);
    WS(ctx, "mlex.plans", unsigned char, (size_t)batch * gkr::mle_fold_plan_bytes(), plans);
    WS(ctx, "mlex.drec", gkr::MleHostRecSub, (size_t)batch, d_rec);
    HIP_TRY(ctx, ctx->pinned_host("mlex.rec", sizeof(gkr::MleHostRecSub) * batch, reinterpret_cast<void**>(&rec)));
    HIP_TRY(ctx, ctx->pinned_host("mlex.w", sizeof(Fr) * gkr::kMleMaxSub * batch, reinterpret_cast<void**>(&h_w)));
    HIP_TRY(ctx, ctx->pinned_host("mlex.fail", 64, reinterpret_cast<void**>(&h_fail)));
    *h_fail = 0;
    std::vector<uint32_t> dep_last(batch, 0);
    const bool ifma = host_ifma_ready();
    gkr::SpinPool* pool = batch >= 32 ? ctx->host_pool() : nullptr;
    int rc = GKR_OK;          // this rank's own failure, carried through the remaining exchanges
    uint32_t exchanges = 0;
    auto exchange_sums = [&](int J, uint32_t ticket) {
        Timed t(ctx, "exchange", 0.0);
        gkr::launch_mle_xwiden(d_rec, (uint32_t)J, (uint32_t)batch, rc ? 1u : 0u, limbs, s);
        const int arc = exchange->fn(exchange->user, (size_t)batch * (((size_t)1 << J) + 2) * 8, static_cast<void*>(s));
        gkr::launch_mle_xnarrow(limbs, (uint32_t)J, (uint32_t)batch, rec, ticket, h_fail, s);
        ++exchanges;
        if (arc && !rc) rc = ctx->fail(GKR_ERR_INVALID, "the device sum-over-ranks hook failed (status " + std::to_string(arc) + ")");
    };
    auto some_rank_failed = [&]() { return __atomic_load_n(h_fail, __ATOMIC_ACQUIRE) != 0; };
    // the J rounds of every table on the summed sub-block sums (the same on every rank), and the fold weights
    auto host_rounds = [&](int J, int round0) {
        const int chunk = 16;
        std::atomic<int> next{0};
        const std::function<bool()> work_fn = [&]() -> bool {
            const int first = next.fetch_add(chunk, std::memory_order_relaxed);
            if (first >= batch) return false;
            const int count = batch - first < chunk ? batch - first : chunk;
            uint64_t c0[gkr::kMlePassMaxRounds][16][4], c1[gkr::kMlePassMaxRounds][16][4], r[gkr::kMlePassMaxRounds][16][4];
            uint32_t ln[gkr::kMlePassMaxRounds][16];
            (ifma && count >= 3 ? gkr::gkr_ifma_pass : host_pass_scalar)(reinterpret_cast<const uint64_t*>(rec[first].sums), sizeof(gkr::MleHostRecSub) / 8, count, J,
                                                                          nullptr, c0, c1, r, ln, reinterpret_cast<uint64_t*>(h_w + (size_t)first * gkr::kMleMaxSub),
                                                                          4 * gkr::kMleMaxSub);
            for (int i = 0; i < count; ++i) {
                const int b = first + i;
                if (round0 == 0) dep_last[b] = rec[b].dep;
                for (int tt = 0; tt < J; ++tt) {
                    const size_t row = (size_t)b * n + round0 + tt;
                    memset(&out_coeffs[row * 2], 0, 32);
                    if (ln[tt][i] == 2) memcpy(&out_coeffs[row * 2], c1[tt][i], 32);
                    memcpy(&out_coeffs[row * 2 + 1], c0[tt][i], 32);
                    out_len[row] = ln[tt][i];
                    memcpy(&out_r[row], r[tt][i], 32);
                }
            }
            return true;
        };
        gkr::SpinPool::Session session(pool, nullptr);
        run_pieces(pool, &work_fn, batch > chunk);
    };
    // ---- the rank-local rounds: n - lp - t_stop of them, in passes
    int m = nl, round0 = 0, jin = 0;
    while (m - jin > t_stop) {
        m -= jin;
        const int J = rounds_for(m);
        const uint32_t ticket = ++ctx->ticket;
        if (!rc) {
            const size_t src_len = (size_t)1 << (m + jin), S = (size_t)1 << m;
            const bool from_input = round0 == jin;   // pass 0 (sums only) and the first fold read the input shards
            const Fr* src = from_input ? shards : work;
            const size_t src_stride = from_input ? len : work_len;
            if (jin == 0) {
                if (len <= gkr::kSmallPassEntries) {
                    gkr::launch_mle_multifold_small(0, shards, len, nullptr, 0, (uint32_t)len, (uint32_t)J, batch, h_w, d_rec, ticket, s);
                } else {
                    const uint32_t nblk = gkr::mle_pass_blocks((uint32_t)len, (uint32_t)J, batch);
                    {
                        Timed t(ctx, "mle_sub_sums", (double)batch * len * 32.0);
                        gkr::launch_mle_sub_sums(shards, len, (uint32_t)len, batch, nblk, partials, s);
                    }
                    gkr::launch_mle_sub_reduce(partials, nblk, (uint32_t)J, batch, d_rec, ticket, s);
                }
            } else if (S <= gkr::kSmallPassEntries) {
                gkr::launch_mle_multifold_small(jin, src, src_stride, work, work_len, (uint32_t)S, (uint32_t)J, batch, h_w, d_rec, ticket, s);
            } else {
                const uint32_t nblk = gkr::mle_multifold_blocks((uint32_t)S, (uint32_t)J, batch);
                if (gkr::mle_multifold_uses_mfma((uint32_t)S, nblk)) gkr::launch_mle_fold_plan(jin, h_w, plans, batch, s);
                {
                    Timed t(ctx, "mle_multifold", (double)batch * ((double)src_len + (double)S) * 32.0);
                    gkr::launch_mle_multifold(jin, src, src_stride, work, work_len, (uint32_t)S, batch, nblk, h_w, plans, partials, s);
                }
                gkr::launch_mle_sub_reduce(partials, nblk, (uint32_t)J, batch, d_rec, ticket, s);
            }
            if (hipError_t le = hipGetLastError(); le != hipSuccess) rc = ctx->hip_fail(le, "launch of a sumcheck pass");
        }
        exchange_sums(J, ticket);
        if (!rc) rc = wait_records(ctx, rec, batch, ticket);
        if (!rc && some_rank_failed()) rc = ctx->fail(GKR_ERR_HIP, "another rank failed during the sumcheck");
        if (!rc) host_rounds(J, round0);
        round0 += J;
        jin = J;
    }
    // ---- bind the last pass's variables (2^t_stop entries per shard are left), gather the tail
    m -= jin;
    const Fr* rest = shards;
    size_t rest_stride = len;
    if (jin && !rc) {
        const uint32_t ticket = ++ctx->ticket;
        const size_t S = (size_t)1 << m;
        const bool from_input = round0 == jin;   // one pass so far: its sums came from the input shards
        gkr::launch_mle_multifold_small(jin, from_input ? shards : work, from_input ? len : work_len, work, work_len, (uint32_t)S, 1u, batch, h_w, d_rec,
                                        ticket, s);
        rest = work;
        rest_stride = work_len;
        if (hipError_t le = hipGetLastError(); le != hipSuccess) rc = ctx->hip_fail(le, "launch of the last rank-local fold");
    }
    {
        Timed t(ctx, "exchange", 0.0);
        gkr::launch_mle_gather_widen(rest, rest_stride, (uint32_t)m, (uint32_t)lp, (uint32_t)shard, rc ? 1u : 0u, (uint32_t)batch, limbs, s);
        const int arc = exchange->fn(exchange->user, ((size_t)batch << (m + lp)) * 8 + 8, static_cast<void*>(s));
        gkr::launch_mle_gather_narrow(limbs, (uint32_t)(m + lp), (uint32_t)batch, d_tail, h_fail, s);
        ++exchanges;
        if (arc && !rc) rc = ctx->fail(GKR_ERR_INVALID, "the device sum-over-ranks hook failed (status " + std::to_string(arc) + ")");
    }
    if (out_exchanges) *out_exchanges = exchanges;
    {
        const hipError_t se = hipStreamSynchronize(s);   // the tail is complete, the flag has landed
        if (se != hipSuccess && !rc) rc = ctx->hip_fail(se, "hipStreamSynchronize after the gather");
    }
    if (!rc && some_rank_failed()) rc = ctx->fail(GKR_ERR_HIP, "another rank failed during the sumcheck");
    if (rc) return rc;
    ctx->drain_events();
    // ---- the last t_stop + lp rounds on the gathered tail, the same on every rank
    MleTailArgs tail;
    tail.n_total = n;
    tail.round_offset = round0;
    tail.dep_last = round0 ? dep_last.data() : nullptr;   // (no rank-local round: the tail is the whole table, its own neighbour compare decides)
    const int n_tail = m + lp;
    return run_mle_batch_passes(ctx, d_tail, n_tail, batch, out_coeffs, out_len, out_r, &tail);
}

struct gkr_mle_session {
    int n = 0;                  // variables of this shard's table
    uint32_t round = 0;
    const Fr* input = nullptr;  // not owned
    Fr* work = nullptr;
    gkr::MlePartial* partials = nullptr;
    gkr::MleHostRec* rec = nullptr;
    gkr::FixedMul* rtab = nullptr;
    uint32_t dep = 0;
    bool have_sums = false;
};


// ---- plain MLE sumcheck, step-wise ----

static void free_mle_session(gkr_mle_session* S) {
    if (!S) return;
    if (S->work) (void)hipFree(S->work);
    if (S->partials) (void)hipFree(S->partials);
    if (S->rec) (void)hipHostFree(S->rec);
    if (S->rtab) (void)hipHostFree(S->rtab);
    delete S;
}

// d_table: 2^n entries in device memory (this rank's shard, or the whole table); not modified
int gkr_mle_session_open(gkr_ctx* ctx, const void* d_table, int n, gkr_mle_session** out) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!d_table || !out || n < 1 || n > 30) return ctx->fail(GKR_ERR_INVALID, "null pointer or n out of [1, 30]");
    GKR_ENTER(ctx);
    gkr_mle_session* S = new gkr_mle_session();
    S->n = n;
    S->input = static_cast<const Fr*>(d_table);
    const size_t len = (size_t)1 << n;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&S->work), (len / 2 ? len / 2 : 1) * sizeof(Fr));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&S->partials), gkr::kMaxBlocksPerTable * sizeof(gkr::MlePartial));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&S->rec), sizeof(gkr::MleHostRec), hipHostMallocCoherent | hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&S->rtab), sizeof(gkr::FixedMul), hipHostMallocCoherent | hipHostMallocMapped);
    if (e != hipSuccess) {
        free_mle_session(S);
        return ctx->hip_fail(e, "mle session allocation");
    }
    memset(S->rec, 0, sizeof(gkr::MleHostRec));
    *out = S;
    return GKR_OK;
}

// out = {sum of the low half, sum of the high half} of the current table (canonical);
// *out_dep (first round only, may be null): does this shard's table depend on its own last variable
int gkr_mle_session_sums(gkr_ctx* ctx, gkr_mle_session* S, gkr_fr* out, uint32_t* out_dep) {
    if (!ctx || !S || !out) return GKR_ERR_INVALID;
    if ((int)S->round >= S->n) return ctx->fail(GKR_ERR_INVALID, "no round left in this session");
    GKR_ENTER(ctx);
    hipStream_t s = ctx->stream;
    if (!S->have_sums) {   // only the very first round computes sums without a fold
        const size_t len = (size_t)1 << S->n;
        const uint32_t h = (uint32_t)(len / 2);
        const uint32_t nblk = gkr::mle_blocks_per_table(h, 1);
        gkr::launch_mle_sum_first(S->input, len, h, 1, nblk, S->partials, s);
        const uint32_t ticket = ++ctx->ticket;
        gkr::launch_mle_round_reduce(S->partials, nblk, 1, S->rec, ticket, s);
        HIP_TRY(ctx, hipGetLastError());
        int rc = wait_records(ctx, S->rec, 1, ticket);
        if (rc) return rc;
        {
            // a 2-entry table has no neighbour pairs inside a half: it depends on its variable iff T[1] != T[0]
            gkr::h64::F d1;
            memcpy(&d1, &S->rec->c1, 32);
            S->dep = S->n == 1 ? (gkr::h64::is_zero(d1) ? 0u : 1u) : S->rec->dep;
        }
        S->have_sums = true;
    }
    gkr::h64::F c0, c1;
    memcpy(&c0, &S->rec->c0, 32);
    memcpy(&c1, &S->rec->c1, 32);
    gkr::h64::F hi = gkr::h64::add(c0, c1);   // the record holds (low sum, high - low)
    memcpy(&out[0], &c0, 32);
    memcpy(&out[1], &hi, 32);
    if (out_dep) *out_dep = S->dep;
    return GKR_OK;
}

// bind the leading variable to r; the sums of the folded table are ready for the next _sums call
int gkr_mle_session_bind(gkr_ctx* ctx, gkr_mle_session* S, const gkr_fr* r) {
    if (!ctx || !S || !r) return GKR_ERR_INVALID;
    if ((int)S->round >= S->n) return ctx->fail(GKR_ERR_INVALID, "no round left in this session");
    if (!all_canonical(r, 1)) return ctx->fail(GKR_ERR_NON_CANONICAL, "r >= modulus");
    GKR_ENTER(ctx);
    hipStream_t s = ctx->stream;
    gkr::h64::F r64;
    memcpy(&r64, r, 32);
    gkr::h64::make_fixed_mul(r64, S->rtab->w);
    const size_t len = (size_t)1 << (S->n - S->round);   // current table
    const Fr* src = S->round == 0 ? S->input : S->work;
    if (len == 2) {
        gkr::launch_fold_pair(src, S->work, S->rtab, s);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(s));
    } else {
        const uint32_t q = (uint32_t)(len / 4);
        const uint32_t nblk = gkr::mle_blocks_per_table(q, 1);
        gkr::launch_mle_fold_sum(src, len, S->work, len / 2, q, 1, nblk, S->rtab, 0, S->partials, s);
        const uint32_t ticket = ++ctx->ticket;
        gkr::launch_mle_round_reduce(S->partials, nblk, 1, S->rec, ticket, s);
        HIP_TRY(ctx, hipGetLastError());
        int rc = wait_records(ctx, S->rec, 1, ticket);
        if (rc) return rc;
    }
    S->round += 1;
    return GKR_OK;
}

// the single remaining entry once all n local variables are bound
int gkr_mle_session_value(gkr_ctx* ctx, gkr_mle_session* S, gkr_fr* out) {
    if (!ctx || !S || !out) return GKR_ERR_INVALID;
    if ((int)S->round != S->n) return ctx->fail(GKR_ERR_INVALID, "session still has rounds to run");
    GKR_ENTER(ctx);
    Fr v;
    HIP_TRY(ctx, hipMemcpyAsync(&v, S->work, sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out = to_abi(v);
    return GKR_OK;
}

void gkr_mle_session_close(gkr_ctx* ctx, gkr_mle_session* S) {
    if (ctx) (void)hipSetDevice(ctx->device);
    free_mle_session(S);
}

// *out_differ = 1 iff the two device tables differ somewhere (a table's dependence on a variable
// that is a rank bit: compare the shards of ranks p and p ^ 1)
int gkr_device_tables_differ(gkr_ctx* ctx, const void* d_a, const void* d_b, size_t count, uint32_t* out_differ) {
    if (!ctx || !d_a || !d_b || !out_differ || !count) return GKR_ERR_INVALID;
    GKR_ENTER(ctx);
    DevBuf<uint32_t> flag;
    HIP_TRY(ctx, flag.alloc(1));
    HIP_TRY(ctx, hipMemsetAsync(flag.p, 0, 4, ctx->stream));
    gkr::launch_tables_differ(static_cast<const Fr*>(d_a), static_cast<const Fr*>(d_b), count, flag.p, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_differ, flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GKR_OK;
}


}  // extern "C"
