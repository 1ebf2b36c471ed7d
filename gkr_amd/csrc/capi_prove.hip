// Whole proofs (prover::prove, rust/src/gkr/prover.rs:6-96; its par_iter, aggregator.rs:350-355): gkr_prove / _batch / _many, the
// line restriction's host twin, the Moebius transform, circuit checks.  C ABI: include/gkr_amd.h.
#include "capi_internal.h"

namespace gkr_host {

void mobius_msb(std::vector<gkr::h64::F>& c, int k) {
    const size_t n = (size_t)1 << k;
    for (int b = 0; b < k; ++b) {
        const size_t bit = (size_t)1 << (k - 1 - b);
        for (size_t i = 0; i < n; ++i)
            if (i & bit) c[i] = gkr::h64::sub(c[i], c[i ^ bit]);
    }
}

// reduce_multiple_polynomial (poly.rs:469-500): q(t) = W(b + t (c - b)).
// vals: the evaluation table of W (canonical); coeffs: its monomial coefficients (only their support is used).
// out: k+1 slots right-aligned, highest first; *out_len = 1 + the largest total degree of a non-zero monomial of W
// (:484-497).  The reference expands every monomial along the line (2^k products of up to k linear factors); the
// same polynomial comes out of binding the variables one after the other on the evaluation table with the linear
// polynomial l_j(t) = b_j + t (c_j - b_j) in place of a challenge:
//     P'[i](t) = P[i](t) + l_j(t) (P[i + h](t) - P[i](t)),
// entries being coefficient vectors in t whose degree grows by one per variable -- about 4 * 2^k products instead
// of ~k^2 * 2^(k-1), and the coefficients above the largest monomial degree come out as the zeros they are.
void line_restriction(const std::vector<gkr::h64::F>& vals, const std::vector<gkr::h64::F>& coeffs, int k, const gkr_fr* b,
                      const gkr_fr* c, gkr_fr* out, uint32_t* out_len) {
    using gkr::h64::F;
    const F zero = {{0, 0, 0, 0}};
    int maxdeg = 0;
    const size_t n = (size_t)1 << k;
    for (size_t mono = 0; mono < n; ++mono)
        if (!gkr::h64::is_zero(coeffs[mono])) {
            const int deg = __builtin_popcountll((unsigned long long)mono);
            if (deg > maxdeg) maxdeg = deg;
        }
    // table of polynomials, stride k + 1 coefficients (lowest degree first); canonical values, Montgomery multipliers
    const size_t stride = (size_t)k + 1;
    std::vector<F> tab(n * stride, zero);
    for (size_t i = 0; i < n; ++i) tab[i * stride] = vals[i];
    size_t h = n >> 1;
    for (int j = 0; j < k; ++j, h >>= 1) {
        F bj, cj;
        memcpy(&bj, &b[j], 32);
        memcpy(&cj, &c[j], 32);
        const F grad = gkr::h64::to_mont(gkr::h64::sub(cj, bj)), cst = gkr::h64::to_mont(bj);
        for (size_t i = 0; i < h; ++i) {
            F* lo = &tab[i * stride];
            const F* hi = &tab[(i + h) * stride];
            F carry = zero;   // grad * d[m - 1]
            for (int m = 0; m <= j + 1; ++m) {
                const F d = m <= j ? gkr::h64::sub(hi[m], lo[m]) : zero;
                const F v = gkr::h64::add(gkr::h64::add(m <= j ? lo[m] : zero, gkr::h64::mont_mul(d, cst)), carry);
                carry = gkr::h64::mont_mul(d, grad);
                lo[m] = v;
            }
        }
    }
    *out_len = (uint32_t)(maxdeg + 1);
    for (int d = 0; d <= k; ++d) memcpy(&out[k - d], &tab[d], 32);
}

int check_circuit(gkr_ctx* ctx, const gkr_circuit_desc* c) {
    if (!c || !c->k || c->depth < 1 || !c->gate_type || !c->left || !c->right)
        return ctx ? ctx->fail(GKR_ERR_INVALID, "null circuit description") : GKR_ERR_INVALID;
    if (c->k[0] > (uint32_t)kMaxLayerKi) return ctx ? ctx->fail(GKR_ERR_INVALID, "output layer wider than 2^GKR_MAX_K_I") : GKR_ERR_INVALID;
    for (uint32_t i = 1; i <= c->depth; ++i) {
        if (c->k[i] == 0) return ctx ? ctx->fail(GKR_ERR_DEGENERATE, "k[i+1] == 0: v = 0 (sumcheck.rs:49)") : GKR_ERR_DEGENERATE;
        if (c->k[i] > (uint32_t)kMaxLayerK)
            return ctx ? ctx->fail(GKR_ERR_INVALID, "layer of more than 2^GKR_MAX_K_NEXT values (gkr_amd.h, limits)") : GKR_ERR_INVALID;
        if (ctx && ctx->transcript != GKR_TRANSCRIPT_HOST && c->k[i] > (uint32_t)kMaxDenseK)
            return ctx->fail(GKR_ERR_INVALID, "the device transcript needs k[i+1] <= GKR_MAX_K_NEXT_DEVICE_TRANSCRIPT (dense predicate tables)");
    }
    return GKR_OK;
}

}  // namespace gkr_host

// =========================================================================== C ABI

extern "C" {

// ---- full proof ---------------------------------------------------------------------

int gkr_proof_sizes(const gkr_circuit_desc* c, gkr_proof_sizes_t* out) {
    if (!out) return GKR_ERR_INVALID;
    int rc = check_circuit(nullptr, c);
    if (rc) return rc;
    memset(out, 0, sizeof *out);
    for (uint32_t i = 0; i < c->depth; ++i) {
        out->rounds += 2 * (size_t)c->k[i + 1];
        out->q_slots += (size_t)c->k[i + 1] + 1;
    }
    for (uint32_t i = 0; i <= c->depth; ++i) out->z_values += c->k[i];
    out->d_coeffs = (size_t)1 << c->k[0];
    out->input_coeffs = (size_t)1 << c->k[c->depth];
    return GKR_OK;
}

// The circuit on the device: from the context's cache when this circuit was proven before, else validated, uploaded and
// handed back in `fresh` (the caller caches it when its call succeeds, or drops it).  A cache hit is decided by two
// independent 64-bit hashes over the k list and the gate arrays AND a comparison with the gate arrays as they were when the
// entry was made: byte for byte for circuits of up to kExactCompareBytes of gate data (every circom-sized sub-circuit),
// kSampleBlocks evenly spaced 4 KiB blocks of every gate array beyond (a wide circuit's 12 MB of gate arrays: a full compare
// would double the cost of the lookup).  include/gkr_amd.h says what that means for callers that share a context.
static constexpr size_t kExactCompareBytes = (size_t)1 << 20, kSampleBlock = 4096, kSampleBlocks = 8;
static void circuit_gate_bytes(const gkr_circuit_desc* c, std::vector<std::pair<const unsigned char*, size_t>>& parts) {
    for (uint32_t i = 0; i < c->depth; ++i) {
        const size_t gates = (size_t)1 << c->k[i];
        parts.push_back({reinterpret_cast<const unsigned char*>(c->gate_type[i]), gates});
        parts.push_back({reinterpret_cast<const unsigned char*>(c->left[i]), gates * 4});
        parts.push_back({reinterpret_cast<const unsigned char*>(c->right[i]), gates * 4});
    }
}
// the bytes the retained copy holds of one array: all of it, or its sampled blocks
static void for_each_retained_range(size_t total_bytes, size_t n, const std::function<void(size_t, size_t)>& f) {
    if (total_bytes <= kExactCompareBytes || n <= kSampleBlock * 2) {
        f((size_t)0, n);
        return;
    }
    const size_t blocks = n / kSampleBlock < kSampleBlocks ? n / kSampleBlock : kSampleBlocks;
    for (size_t b = 0; b < blocks; ++b) {
        const size_t off = b + 1 == blocks ? n - kSampleBlock : (n - kSampleBlock) / (blocks - 1 ? blocks - 1 : 1) * b;
        f(off, kSampleBlock);
    }
}
static void retain_gate_bytes(const gkr_circuit_desc* c, std::vector<unsigned char>& out) {
    std::vector<std::pair<const unsigned char*, size_t>> parts;
    circuit_gate_bytes(c, parts);
    size_t total = 0;
    for (auto& p : parts) total += p.second;
    out.clear();
    for (auto& p : parts) for_each_retained_range(total, p.second, [&](size_t off, size_t len) { out.insert(out.end(), p.first + off, p.first + off + len); });
}
static bool retained_gate_bytes_match(const gkr_circuit_desc* c, const std::vector<unsigned char>& kept) {
    std::vector<std::pair<const unsigned char*, size_t>> parts;
    circuit_gate_bytes(c, parts);
    size_t total = 0, at = 0;
    for (auto& p : parts) total += p.second;
    bool same = true;
    for (auto& p : parts)
        for_each_retained_range(total, p.second, [&](size_t off, size_t len) {
            if (at + len > kept.size() || memcmp(kept.data() + at, p.first + off, len) != 0) same = false;
            at += len;
        });
    return same && at == kept.size();
}

// two independent 64-bit hashes over the k list and the gate arrays (pure: a group's members are hashed side by side)
// The cache key of a circuit: two 64-bit hashes over its k list and gate arrays.  The arrays are hashed in SEGMENTS of at most
// kHashSegment bytes, each on its own (seeded by its position), and the segments' hashes are folded in order -- so that the
// megabytes of a wide circuit (12 MB for k = 18, 20, 20: 0.54 ms on one core, an eighth of the proof) can be hashed by several
// threads at once.  circuit_hash_segments lists them, hash_segment hashes one, fold_segment_hashes makes the key.
constexpr size_t kHashSegment = (size_t)256 << 10;
struct HashSegment {
    const void* p;
    size_t n;
};
static void circuit_hash_segments(const gkr_circuit_desc* c, std::vector<HashSegment>& out) {
    const uint32_t L = c->depth;
    out.push_back({c->k, (L + 1) * sizeof(uint32_t)});
    for (uint32_t i = 0; i < L; ++i) {
        const size_t gates = (size_t)1 << c->k[i];
        const void* arrays[3] = {c->gate_type[i], c->left[i], c->right[i]};
        const size_t bytes[3] = {gates, gates * 4, gates * 4};
        for (int a = 0; a < 3; ++a)
            for (size_t off = 0; off < bytes[a]; off += kHashSegment)
                out.push_back({static_cast<const unsigned char*>(arrays[a]) + off, bytes[a] - off < kHashSegment ? bytes[a] - off : kHashSegment});
    }
}
static void hash_segment(const HashSegment& seg, uint64_t index, uint64_t* out_h1, uint64_t* out_h2) {
    uint64_t h1 = 0xcbf29ce484222325ULL ^ (index * 0x9E3779B97F4A7C15ULL), h2 = 0x9E3779B97F4A7C15ULL + index;
    const unsigned char* q = static_cast<const unsigned char*>(seg.p);
    const size_t n = seg.n;
    size_t i = 0;
    if (n >= 4096) {
        // long arrays: four independent multiply chains over 32-byte blocks, folded into the two running hashes at the end -- a
        // single chain is latency-bound (1.8 ms for 12 MB of gate arrays per proof)
        uint64_t a[4] = {h1, h1 ^ 0x9E3779B97F4A7C15ULL, h1 + 0x632BE59BD9B4E019ULL, ~h1}, b[4] = {h2, ~h2, h2 ^ 0xD6E8FEB86659FD93ULL, h2 + 1};
        for (; i + 32 <= n; i += 32) {
            uint64_t w[4];
            memcpy(w, q + i, 32);
            for (int t = 0; t < 4; ++t) {
                a[t] = (a[t] ^ w[t]) * 0x100000001b3ULL;
                b[t] = (b[t] + w[t]) * 0xBF58476D1CE4E5B9ULL;
                b[t] ^= b[t] >> 29;
            }
        }
        for (int t = 0; t < 4; ++t) {
            h1 = (h1 ^ a[t]) * 0x100000001b3ULL;
            h2 = (h2 + b[t]) * 0xBF58476D1CE4E5B9ULL;
            h2 ^= h2 >> 29;
        }
    }
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, q + i, 8);
        h1 = (h1 ^ w) * 0x100000001b3ULL;
        h2 = (h2 + w) * 0xBF58476D1CE4E5B9ULL;
        h2 ^= h2 >> 29;
    }
    for (; i < n; ++i) {
        h1 = (h1 ^ q[i]) * 0x100000001b3ULL;
        h2 = (h2 + q[i]) * 0x94D049BB133111EBULL;
    }
    *out_h1 = h1;
    *out_h2 = h2;
}
static void fold_segment_hashes(uint32_t depth, const uint64_t* seg_hashes, size_t n_segments, uint64_t* out_h1, uint64_t* out_h2) {
    uint64_t h1 = 0xcbf29ce484222325ULL ^ depth, h2 = 0x9E3779B97F4A7C15ULL + n_segments;
    for (size_t i = 0; i < n_segments; ++i) {
        h1 = (h1 ^ seg_hashes[2 * i]) * 0x100000001b3ULL;
        h2 = (h2 + seg_hashes[2 * i + 1]) * 0xBF58476D1CE4E5B9ULL;
        h2 ^= h2 >> 29;
    }
    *out_h1 = h1;
    *out_h2 = h2;
}

static int find_or_upload_circuit(gkr_ctx* ctx, const gkr_circuit_desc* c, uint64_t h1, uint64_t h2, PreparedCircuit** out,
                                  std::unique_ptr<PreparedCircuit>& fresh) {
    const uint32_t L = c->depth;
    hipStream_t s = ctx->stream;
    *out = nullptr;
    for (size_t i = 0; i < ctx->circuits.size(); ++i)
        if (ctx->circuits[i]->h1 == h1 && ctx->circuits[i]->h2 == h2 && ctx->circuits[i]->k.size() == L + 1 &&
            memcmp(ctx->circuits[i]->k.data(), c->k, (L + 1) * sizeof(uint32_t)) == 0 && retained_gate_bytes_match(c, ctx->circuits[i]->retained)) {
            std::unique_ptr<PreparedCircuit> hit = std::move(ctx->circuits[i]);
            ctx->circuits.erase(ctx->circuits.begin() + i);
            ctx->circuits.push_back(std::move(hit));   // most recently used last
            *out = ctx->circuits.back().get();
            return GKR_OK;
        }
    for (uint32_t i = 0; i < L; ++i) {
        const size_t gates = (size_t)1 << c->k[i];
        for (size_t g = 0; g < gates; ++g)
            if (c->gate_type[i][g] > 1 || (c->left[i][g] >> c->k[i + 1]) || (c->right[i][g] >> c->k[i + 1]))
                return ctx->fail(GKR_ERR_INVALID, "gate type or operand index out of range");
    }
    fresh.reset(new PreparedCircuit());
    fresh->h1 = h1;
    fresh->h2 = h2;
    fresh->k.assign(c->k, c->k + L + 1);
    fresh->lists.resize(L);
    retain_gate_bytes(c, fresh->retained);
    for (uint32_t i = 0; i < L; ++i) {
        const size_t gates = (size_t)1 << c->k[i];
        uint8_t* dg = nullptr;
        uint32_t *dl_ = nullptr, *dr_ = nullptr;
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dg), gates));
        fresh->gt.push_back(dg);
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dl_), gates * 4));
        fresh->l.push_back(dl_);
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dr_), gates * 4));
        fresh->r.push_back(dr_);
        HIP_TRY(ctx, hipMemcpyAsync(dg, c->gate_type[i], gates, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(dl_, c->left[i], gates * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(dr_, c->right[i], gates * 4, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));   // the caller's gate arrays may go away after the call
    *out = fresh.get();
    return GKR_OK;
}

// Can the layers of a circuit with this k list run in a lockstep group (proofs of DIFFERENT circuits in one launch per
// pass)?  Every form of the gate passes takes per-proof gate lists except the segment form of very large layers.
static bool k_list_groupable(const uint32_t* k, uint32_t depth) {
    for (uint32_t i = 0; i < depth; ++i) {
        const gkr::GateSpan span{0, (uint64_t)1 << k[i]};
        if (gkr::gate_segs_words(span, k[i], k[i + 1]) != 0) return false;
    }
    return true;
}

// The proofs of `n_members` (circuit, witnesses) pairs advanced together: every layer's sumcheck runs as ONE batched
// sumcheck (run_layer_batch), so a round costs one set of launches and one host round trip for all proofs.
//   one member:   `batch` witnesses of ONE circuit -- the multi-proof form of the reference's rayon par_iter over independent
//                 (circuit, input) pairs (aggregator.rs:350-355) where the circuits coincide (BASELINE configs[3]);
//   several:      DIFFERENT circuits with the same k list (the <= 20 sub-circuits of one compiled R1CS come in a few shapes:
//                 aggregator.rs:411-416) -- a LOCKSTEP GROUP: the gate passes and the layer evaluation take a per-proof table
//                 of gate lists (gkr::GateSet), everything else is indexed by proof anyway.  One chain of launches and
//                 hand-offs for the whole group instead of one per circuit, its round vectors hashed together.
static int prove_group_impl(gkr_ctx* ctx, const gkr_prove_item* members, int n_members) {
    using gkr::h64::F;
    const auto t_entry = std::chrono::steady_clock::now();
    if (!members || n_members < 1) return ctx->fail(GKR_ERR_INVALID, "no members");
    GKR_ENTER(ctx);   // (before anything that reads an option: the crew's threads enter their context's scope here)
    const gkr_circuit_desc* c = members[0].circuit;
    int rc = GKR_OK;
    int batch = 0;
    for (int m = 0; m < n_members; ++m) {
        const gkr_prove_item& it = members[m];
        rc = check_circuit(ctx, it.circuit);
        if (rc) return rc;
        if (!it.input_values || !it.outs || it.batch < 1 || it.batch > 4096) return ctx->fail(GKR_ERR_INVALID, "null pointer or batch out of [1, 4096]");
        if (it.circuit->depth != c->depth || memcmp(it.circuit->k, c->k, (c->depth + 1) * sizeof(uint32_t)) != 0)
            return ctx->fail(GKR_ERR_INVALID, "the circuits of a lockstep group must share their k list");
        batch += it.batch;
    }
    if (batch > 4096) return ctx->fail(GKR_ERR_INVALID, "more than 4096 proofs in one group");
    // per proof: its output buffers, its member
    std::vector<gkr_proof_buf> outs_v((size_t)batch);
    std::vector<int> member_of((size_t)batch), first_of((size_t)n_members);
    {
        int b = 0;
        for (int m = 0; m < n_members; ++m) {
            first_of[m] = b;
            for (int i = 0; i < members[m].batch; ++i, ++b) {
                outs_v[b] = members[m].outs[i];
                member_of[b] = m;
            }
        }
    }
    gkr_proof_buf* const outs = outs_v.data();
    for (int b = 0; b < batch; ++b) {
        const gkr_proof_buf* out = &outs[b];
        if (!out->sumcheck_coeffs || !out->sumcheck_len || !out->sumcheck_r || !out->q || !out->q_len || !out->z || !out->r ||
            !out->d_coeffs || !out->input_coeffs)
            return ctx->fail(GKR_ERR_INVALID, "null pointer in proof buffers");
    }
    const uint32_t L = c->depth;
    for (int m = 0; m < n_members; ++m)
        for (uint32_t i = 0; i < L; ++i)
            if (!members[m].circuit->gate_type[i] || !members[m].circuit->left[i] || !members[m].circuit->right[i])
                return ctx->fail(GKR_ERR_INVALID, "null gate array");
    if (n_members > 1 && (ctx->transcript != GKR_TRANSCRIPT_HOST || !k_list_groupable(c->k, L)))
        return ctx->fail(GKR_ERR_INVALID, "a lockstep group needs the host transcript and layers below the segment form's size");
    const size_t n_in = (size_t)1 << c->k[L];
    const bool dbg_pre = gkr::debug_timing();
    auto us_since_entry = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_entry).count(); };
    // (a large input layer is validated where it lands, on the device: the host loop over 2^20 values took 1.6 ms of a 12 ms proof)
    const bool check_on_device = n_in * (size_t)batch >= ((size_t)1 << 16);
    if (!check_on_device)
        for (int m = 0; m < n_members; ++m)
            if (!all_canonical(members[m].input_values, n_in * members[m].batch)) return ctx->fail(GKR_ERR_NON_CANONICAL, "input value >= r");
    const double us_canon = us_since_entry();
    hipStream_t s = ctx->stream;

    // every member's circuit on the device (cache, or upload)
    const bool no_cache = gkr::opt(gkr::OPT_no_circuit_cache) != 0;
    std::vector<PreparedCircuit*> pcs((size_t)n_members, nullptr);
    std::vector<std::unique_ptr<PreparedCircuit>> fresh((size_t)n_members);
    struct DropFresh {   // an uncached or failed circuit's device arrays do not outlive the call
        gkr_ctx* ctx;
        std::vector<std::unique_ptr<PreparedCircuit>>& v;
        ~DropFresh() {
            bool any = false;
            for (auto& p : v) any |= (bool)p;
            if (any) (void)hipStreamSynchronize(ctx->stream);
            for (auto& p : v)
                if (p) p->release();
        }
    } drop_fresh{ctx, fresh};
    // (hashing a circuit's gate arrays costs ~0.1 ms per MB: the members of a group are hashed side by side, by whichever
    // threads of the crew have nothing of their own -- one after the other they were 0.9 ms at the head of a 9 ms step)
    std::vector<uint64_t> hs((size_t)2 * n_members);
    {
        // (the segments of all members' arrays in one list: a piece is a segment -- the members of a group, and the megabytes of one
        // wide circuit, are hashed by whichever threads are free: the crew's, or the context's own pool when it proves alone)
        std::vector<HashSegment> segs;
        std::vector<size_t> first_seg((size_t)n_members + 1, 0);
        size_t total_bytes = 0;
        for (int m = 0; m < n_members; ++m) {
            circuit_hash_segments(members[m].circuit, segs);
            first_seg[(size_t)m + 1] = segs.size();
        }
        for (const HashSegment& sg : segs) total_bytes += sg.n;
        std::vector<uint64_t> seg_hashes(2 * segs.size());
        std::atomic<size_t> next{0};
        const std::function<bool()> work = [&]() -> bool {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= segs.size()) return false;
            const size_t m = (size_t)(std::upper_bound(first_seg.begin(), first_seg.end(), i) - first_seg.begin()) - 1;
            hash_segment(segs[i], i - first_seg[m], &seg_hashes[2 * i], &seg_hashes[2 * i + 1]);
            return true;
        };
        gkr::SpinPool* hash_pool = (!ctx->crew_member && total_bytes >= ((size_t)2 << 20)) ? ctx->host_pool() : nullptr;
        gkr::SpinPool::Session session(hash_pool, nullptr);
        run_pieces(hash_pool, &work, segs.size() > (size_t)n_members);
        session.close();
        for (int m = 0; m < n_members; ++m)
            fold_segment_hashes(members[m].circuit->depth, &seg_hashes[2 * first_seg[m]], first_seg[(size_t)m + 1] - first_seg[m], &hs[2 * (size_t)m],
                                &hs[2 * (size_t)m + 1]);
    }
    for (int m = 0; m < n_members; ++m) {
        // (a member's circuit may be the very circuit of an earlier member: found in `fresh` then, not uploaded twice)
        for (int e = 0; e < m && !pcs[m]; ++e)
            if (members[e].circuit == members[m].circuit) pcs[m] = pcs[e];
        if (pcs[m]) continue;
        rc = find_or_upload_circuit(ctx, members[m].circuit, hs[2 * (size_t)m], hs[2 * (size_t)m + 1], &pcs[m], fresh[m]);
        if (rc) return rc;
    }
    PreparedCircuit* pc = pcs[0];
    const bool any_fresh = [&] {
        for (auto& p : fresh)
            if (p) return true;
        return false;
    }();
    const double us_hash = us_since_entry();

    // A lockstep group: every member's gate lists must exist before the group's first launch (a lone circuit builds them
    // inside its first layer sumcheck), and the passes over the gates get their per-proof table.
    gkr::GateSet* d_sets = nullptr;            // [layer][proof]
    std::vector<LayerGroup> groups;
    if (n_members > 1) {
        for (int m = 0; m < n_members; ++m)
            for (uint32_t i = 0; i < L; ++i)
                if (!pcs[m]->lists[i].ready) {
                    rc = build_cached_gate_lists(ctx, (int)c->k[i], (int)c->k[i + 1], pcs[m]->gt[i], pcs[m]->l[i], pcs[m]->r[i], &pcs[m]->lists[i]);
                    if (rc) return rc;
                }
        gkr::GateSet* h_sets = nullptr;
        const size_t n_sets = (size_t)L * batch;
        HIP_TRY(ctx, ctx->pinned_host("prove.sets", n_sets * sizeof(gkr::GateSet), reinterpret_cast<void**>(&h_sets)));
        WS(ctx, "prove.dsets", gkr::GateSet, n_sets, d_sets);
        groups.resize(L);
        for (uint32_t i = 0; i < L; ++i) {
            groups[i].d_sets = d_sets + (size_t)i * batch;
            groups[i].plan_counts.known = true;
            for (int m = 0; m < n_members; ++m) {
                const gkr::GatePlanCounts& pcn = pcs[m]->lists[i].plan_counts;
                groups[i].plan_counts.known = groups[i].plan_counts.known && pcn.known;
                for (int half = 0; half < 2; ++half)
                    for (int w = 0; w < 8; ++w)
                        if (pcn.hdr[half][w] > groups[i].plan_counts.hdr[half][w]) groups[i].plan_counts.hdr[half][w] = pcn.hdr[half][w];
            }
            for (int b = 0; b < batch; ++b) {
                const PreparedCircuit* p = pcs[member_of[b]];
                const GateLists& gl = p->lists[i];
                h_sets[(size_t)i * batch + b] = gkr::GateSet{gl.offsets, gl.cursor, gl.list, gl.plan, p->gt[i], p->l[i], p->r[i], 0};
            }
        }
        gkr::launch_copy_words(h_sets, d_sets, n_sets * sizeof(gkr::GateSet) / 4, s);
    }

    // forward-evaluate every layer of every proof on the device (calculate_input, convert.rs:787-831)
    std::vector<Fr*> dW(L + 1, nullptr);
    for (uint32_t i = 0; i <= L; ++i) {
        const std::string slot = "prove.W" + std::to_string(i);
        HIP_TRY(ctx, ctx->workspace(slot.c_str(), ((size_t)batch << c->k[i]) * sizeof(Fr), reinterpret_cast<void**>(&dW[i])));
    }
    // Small transfers go through pinned buffers and a copy kernel, not through the runtime's transfer calls (see
    // k_copy_words); large ones (a 2^20-value input layer) keep the copy engine's bandwidth.
    constexpr size_t kKernelCopyLimit = (size_t)4 << 20, kGroupCopyLimit = (size_t)64 << 20;
    const size_t in_bytes = n_in * batch * sizeof(Fr);
    if (in_bytes <= (n_members > 1 ? kGroupCopyLimit : kKernelCopyLimit)) {
        gkr_fr* h_in = nullptr;
        HIP_TRY(ctx, ctx->pinned_host("prove.in", in_bytes, reinterpret_cast<void**>(&h_in)));
        // (a group's members are staged side by side by the crew's free threads: seven 1 MiB input layers one after the other
        // were 0.3 ms at the head of the group's chain)
        std::atomic<int> next{0};
        const std::function<bool()> work = [&]() -> bool {
            const int m = next.fetch_add(1, std::memory_order_relaxed);
            if (m >= n_members) return false;
            memcpy(h_in + (size_t)first_of[m] * n_in, members[m].input_values, n_in * members[m].batch * sizeof(gkr_fr));
            return true;
        };
        run_pieces(nullptr, &work, n_members > 1 && in_bytes >= ((size_t)1 << 20));
        gkr::launch_copy_words(h_in, dW[L], in_bytes / 4, s);
    } else {
        for (int m = 0; m < n_members; ++m)
            HIP_TRY(ctx, hipMemcpyAsync(dW[L] + (size_t)first_of[m] * n_in, members[m].input_values, n_in * members[m].batch * sizeof(Fr), hipMemcpyHostToDevice, s));
    }
    uint32_t* d_in_flag = nullptr;
    uint32_t h_in_flag = 0;
    if (check_on_device) {
        WS(ctx, "prove.inflag", uint32_t, 1, d_in_flag);
        HIP_TRY(ctx, hipMemsetAsync(d_in_flag, 0, 4, s));
        gkr::launch_check_canonical(dW[L], n_in * (size_t)batch, d_in_flag, s);
    }
    for (int i = (int)L - 1; i >= 0; --i)
        gkr::launch_layer_eval(1u << c->k[i], pc->gt[i], pc->l[i], pc->r[i], dW[i + 1], dW[i], (uint32_t)batch, 1u << c->k[i + 1], s,
                               n_members > 1 ? groups[i].d_sets : nullptr);
    HIP_TRY(ctx, hipGetLastError());
    // Everything below up to the layers is OUTPUT ONLY (d, input_func, "output 0 must be zero", "the inputs were canonical"):
    // it runs on the side stream behind the evaluation, beside the first layer's sumcheck, and the host looks at it after the
    // last layer.  (It used to sit on the main stream with a synchronisation before the first layer: 0.14 ms of a lone
    // 4.6 ms proof, 0.6 ms at the head of a seven-circuit group's chain.)
    HIP_TRY(ctx, ctx->aux_stream(1));
    hipStream_t side = ctx->aux;
    // The side stream's copies below land in locals of this call (h_in_flag, hW_big) and in the caller's buffers: whatever path
    // leaves the call -- the HIP_TRY / WS early returns included -- waits for the stream first.  Declared after those locals'
    // storage, so it runs before they go.
    std::vector<F> hW_big[2];
    struct SideDrain {
        hipStream_t st;
        ~SideDrain() {
            if (st) (void)hipStreamSynchronize(st);
        }
    } side_drain{side};
    HIP_TRY(ctx, hipEventRecord(ctx->aux_events[0], s));
    HIP_TRY(ctx, hipStreamWaitEvent(side, ctx->aux_events[0], 0));
    if (check_on_device) HIP_TRY(ctx, hipMemcpyAsync(&h_in_flag, d_in_flag, 4, hipMemcpyDeviceToHost, side));
    // the host needs the outputs and the inputs (d, input_func); the layers in between stay on the device
    const F* hW[2] = {nullptr, nullptr};   // [0]: W_0, [1]: W_L
    const F* hW0_first = nullptr;           // or only output 0 of every proof
    // d and input_func are the monomial forms of W_0 and W_L (get_multi_ext, poly.rs:502-536): tables beyond 2^12 values are
    // transformed on the device (k launches over a grid) and land in the proof buffers directly; small ones on the host
    constexpr uint32_t kDeviceMobiusMinK = 13;
    bool coeffs_done[2] = {false, false};
    const Fr* coeff_src[2] = {nullptr, nullptr};
    for (int e = 0; e < 2; ++e) {
        const uint32_t i = e ? L : 0;
        if (c->k[i] < kDeviceMobiusMinK) continue;
        const size_t n = (size_t)1 << c->k[i];
        Fr* mono = nullptr;
        HIP_TRY(ctx, ctx->workspace(e ? "prove.monoL" : "prove.mono0", n * batch * sizeof(Fr), reinterpret_cast<void**>(&mono)));
        HIP_TRY(ctx, hipMemcpyAsync(mono, dW[i], n * batch * sizeof(Fr), hipMemcpyDeviceToDevice, side));
        gkr::launch_mobius(mono, c->k[i], n, (uint32_t)batch, side);
        coeff_src[e] = mono;
        coeffs_done[e] = true;
    }
    // Their way to the caller's (pageable) buffers -- tens of MiB at the staged-copy rate, 2.5 ms for a 2^20-value input
    // layer -- runs on a helper thread beside the layers' sumchecks: the coefficients are outputs only, nothing waits for
    // them but the end of the call.  The thread waits for the transforms through an event, then copies synchronously.
    struct CoeffCopier {
        AsyncWorker* worker = nullptr;
        bool started = false;
        hipError_t err = hipSuccess;
        hipStream_t side = nullptr;
        // every path out of the call: the copies are all queued (the helper is done) and have landed (the side stream is
        // waited for) before the caller sees its buffers again
        void finish() {
            if (started) {
                worker->wait();
                if (side) (void)hipStreamSynchronize(side);
                started = false;
            }
        }
        ~CoeffCopier() { finish(); }
    } copier;
    if (coeffs_done[0] || coeffs_done[1]) {
        const int device = ctx->device;
        const uint32_t k0 = c->k[0], kL = c->k[L];
        // (on the context's side stream, which the line restrictions use later in the call: stream order keeps them apart; a
        // synchronous hipMemcpy on the null stream held every other thread's HIP calls up for its whole duration -- 23 ms)
        if (!ctx->copier) ctx->copier.reset(new AsyncWorker());
        copier.worker = ctx->copier.get();
        copier.side = side;
        copier.started = true;
        CoeffCopier* cp = &copier;
        const gkr::Options* const call_options = &ctx->options;
        ctx->copier->run([=]() {
            gkr::OptionScope option_scope(call_options);
            hipError_t e = hipSetDevice(device);   // (the transforms are queued on the side stream already: stream order)
            for (int which = 0; which < 2 && e == hipSuccess; ++which) {
                if (!coeff_src[which]) continue;
                const size_t n = (size_t)1 << (which ? kL : k0);
                for (int b = 0; b < batch && e == hipSuccess; ++b)
                    e = hipMemcpyAsync(which ? outs[b].input_coeffs : outs[b].d_coeffs, coeff_src[which] + (size_t)b * n, n * sizeof(Fr), hipMemcpyDeviceToHost, side);
            }
            cp->err = e;
        });
    }
    for (int e = 0; e < 2; ++e) {
        const uint32_t i = e ? L : 0;
        const size_t bytes = ((size_t)batch << c->k[i]) * sizeof(Fr);
        if (coeffs_done[e] && e == 1) continue;
        if (coeffs_done[e] && batch <= 64) {
            // (of a W_0 whose coefficients come from the device only output 0 of every proof is looked at: "must be zero")
            F* dst = nullptr;
            HIP_TRY(ctx, ctx->pinned_host("prove.hW0first", sizeof(F) * (size_t)batch, reinterpret_cast<void**>(&dst)));
            gkr::launch_copy_rows(dW[0], (size_t)8 << c->k[0], dst, 8, 8, (uint32_t)batch, side);   // (one launch: 8 words of every proof's table)
            hW0_first = dst;
            continue;
        }
        if (bytes <= kKernelCopyLimit) {
            F* dst = nullptr;
            HIP_TRY(ctx, ctx->pinned_host(e ? "prove.hWL" : "prove.hW0", bytes, reinterpret_cast<void**>(&dst)));
            gkr::launch_copy_words(dW[i], dst, bytes / 4, side);
            hW[e] = dst;
        } else {
            hW_big[e].resize((size_t)batch << c->k[i]);
            HIP_TRY(ctx, hipMemcpyAsync(hW_big[e].data(), dW[i], bytes, hipMemcpyDeviceToHost, side));
            hW[e] = hW_big[e].data();
        }
    }
    const bool dbg_pb = gkr::debug_timing();
    const auto tpb0 = std::chrono::steady_clock::now();
    const bool account = accounting_on();
    if (dbg_pb || account) t_account = ThreadTimeAccount();
    if (dbg_pb) fprintf(stderr, "[gkr timing] prove: %d circuit(s) %s, forward evaluation + readback done\n", n_members, any_fresh ? "uploaded" : "from cache");
    if (dbg_pre)
        fprintf(stderr, "[gkr timing] prove, before the layers: input check %.0f us, circuit hash %.0f us, upload + evaluation + Moebius + readback %.0f us\n", us_canon,
                us_hash - us_canon, us_since_entry() - us_hash);
    // z[0] = 0 (prover.rs:16-21)
    for (int b = 0; b < batch; ++b)
        for (uint32_t j = 0; j < c->k[0]; ++j) memset(&outs[b].z[j], 0, sizeof(gkr_fr));
    std::vector<gkr_fr> z_cur((size_t)batch * (c->k[0] ? c->k[0] : 1));
    memset(z_cur.data(), 0, z_cur.size() * sizeof(gkr_fr));
    std::vector<gkr_fr*> scp(batch), srp(batch);
    std::vector<uint32_t*> slp(batch);
    size_t row_off = 0, q_off = 0, z_off = 0;
    gkr::SpinPool* pool = batch >= 16 ? ctx->host_pool() : nullptr;
    // q_i (W_{i+1} on the line b* -> c*, prover.rs:70) is output only -- nothing later in the proof depends on it -- so
    // it is computed on the side stream while the next layers' sumchecks run, and read back once at the end
    uint32_t kmax = 0;
    size_t q_total = 0;
    for (uint32_t i = 1; i <= L; ++i) {
        kmax = c->k[i] > kmax ? c->k[i] : kmax;
        q_total += (size_t)c->k[i] + 1;
    }
    gkr_fr* h_lines = nullptr;   // pinned: per layer and proof b*_1..b*_k, c*_1..c*_k; the kernel reads it in place
    Fr *d_q = nullptr, *d_lr = nullptr;
    uint32_t* d_qlen = nullptr;
    HIP_TRY(ctx, ctx->pinned_host("prove.lines", (size_t)L * batch * 2 * kmax * sizeof(gkr_fr), reinterpret_cast<void**>(&h_lines)));
    // (the kernel stores q and its length straight into pinned host memory: read after the side stream's last kernel)
    HIP_TRY(ctx, ctx->pinned_host("prove.q", q_total * batch * sizeof(Fr), reinterpret_cast<void**>(&d_q)));
    HIP_TRY(ctx, ctx->pinned_host("prove.qlen", (size_t)L * batch * sizeof(uint32_t), reinterpret_cast<void**>(&d_qlen)));
    WS(ctx, "prove.lr", Fr, (size_t)batch * 3 * ((size_t)1 << kmax), d_lr);
    uint32_t* d_lrdeg = nullptr;
    WS(ctx, "prove.lrdeg", uint32_t, (size_t)batch, d_lrdeg);
    Fr* d_lrbc = nullptr;   // per layer: the next layer's restriction may start before this one's last step has read them
    WS(ctx, "prove.lrbc", Fr, (size_t)L * batch * 2 * kmax, d_lrbc);
    HIP_TRY(ctx, ctx->aux_stream(0));
    struct LineWorker {   // every path out of the call: the helper has issued what it was handed before the side stream is waited for
        AsyncWorker* worker = nullptr;
        bool started = false;
        void finish() {
            if (started) worker->wait();
            started = false;
        }
        ~LineWorker() { finish(); }
    } liner;
    for (uint32_t i = 0; i < L; ++i) {
        const int k_i = c->k[i], k = c->k[i + 1];
        for (int b = 0; b < batch; ++b) {
            scp[b] = outs[b].sumcheck_coeffs + row_off * 3;
            slp[b] = outs[b].sumcheck_len + row_off;
            srp[b] = outs[b].sumcheck_r + row_off;
        }
        // The LAST layer's line restriction has no later sumcheck to run beside: the part of it that needs W alone (the copy, the
        // Moebius transform, the largest degree: 0.1 ms of the 0.3 at 2^20 values) is issued before the layer's sumcheck, on the
        // side stream, behind the previous layer's restriction.
        const bool line_prepared = i + 1 == L && k > 12 && ctx->profile == 0;
        if (line_prepared) {
            if (!ctx->liner) ctx->liner.reset(new AsyncWorker());
            liner.worker = ctx->liner.get();
            liner.started = true;
            const int device = ctx->device;
            const Fr* Wn = dW[i + 1];
            const uint32_t ub = (uint32_t)batch, uk = (uint32_t)k;
            hipStream_t aux = ctx->aux;
            const gkr::Options* const call_options = &ctx->options;
            ctx->liner->run([=]() {
                gkr::OptionScope option_scope(call_options);   // (the helper thread proves with the context's options, not the process defaults)
                if (hipSetDevice(device) != hipSuccess) return;
                gkr::launch_line_restriction(Wn, uk, nullptr, d_lr, d_lrdeg, nullptr, nullptr, nullptr, ub, aux, gkr::LinePart::prepare);
            });
        }
        const auto tl0 = std::chrono::steady_clock::now();
        ctx->rounds_ahead = 0;
        for (uint32_t later = i + 1; later < L; ++later) ctx->rounds_ahead += 2 * (int)c->k[later + 1];
        rc = run_layer_batch(ctx, batch, k_i, k, pc->gt[i], pc->l[i], pc->r[i], z_cur.data(), dW[i + 1], scp.data(), slp.data(),
                             srp.data(), nullptr, &pc->lists[i], n_members > 1 ? &groups[i] : nullptr);
        ctx->rounds_ahead = 0;
        if (rc) {
            liner.finish();
            (void)hipStreamSynchronize(ctx->aux);   // earlier layers' line restrictions still write the pinned q buffers the next call reuses
            return rc;
        }
        const auto tl1 = std::chrono::steady_clock::now();
        std::vector<gkr_fr> z_next((size_t)batch * k);
        {
            gkr_fr* lines = h_lines + (size_t)i * batch * 2 * kmax;
            for (int b = 0; b < batch; ++b) memcpy(lines + (size_t)b * 2 * k, srp[b], (size_t)2 * k * sizeof(gkr_fr));
            const Fr* Wn = dW[i + 1];
            Fr* bcm = d_lrbc + (size_t)i * batch * 2 * kmax;
            Fr* qdst = d_q + q_off * batch;
            uint32_t* qlen_dst = d_qlen + (size_t)i * batch;
            hipStream_t aux = ctx->aux;
            if (k > 12 && ctx->profile == 0) {
                // A wide layer's restriction is 4 + (k - 6) + 1 launches on the side stream -- output only, but issued by THIS
                // thread they were ~0.15 ms per layer of launch calls on the proof's round path (five layers of 2^15 values:
                // 0.7 ms of a 5 ms proof).  A helper thread of the context issues them; the side stream keeps their order.
                if (!ctx->liner) ctx->liner.reset(new AsyncWorker());
                liner.worker = ctx->liner.get();
                liner.started = true;
                const int device = ctx->device;
                const uint32_t ub = (uint32_t)batch, uk = (uint32_t)k;
                const gkr::Options* const call_options = &ctx->options;
                ctx->liner->run([=]() {
                    gkr::OptionScope option_scope(call_options);
                    if (hipSetDevice(device) != hipSuccess) return;
                    gkr::launch_line_restriction(Wn, uk, reinterpret_cast<const Fr*>(lines), d_lr, d_lrdeg, bcm, qdst, qlen_dst, ub, aux,
                                                 line_prepared ? gkr::LinePart::finish : gkr::LinePart::all);
                });
            } else {
                Timed t(ctx, "line_restriction", 0.0, ctx->aux, true);
                gkr::launch_line_restriction(Wn, (uint32_t)k, reinterpret_cast<const Fr*>(lines), d_lr, d_lrdeg, bcm, qdst, qlen_dst, (uint32_t)batch, aux);
            }
        }
        auto finish = [&](int b) {
            const gkr_fr* sr = srp[b];
            const gkr_fr* b_star = sr;
            const gkr_fr* c_star = sr + k;
            // r* = multi_hash(last round vector) (prover.rs:74-78) -- the same hash, vector and key as the
            // sumcheck's last challenge, so it is that challenge
            const gkr_fr r_star = sr[2 * k - 1];
            outs[b].r[i] = r_star;
            // z_{i+1} = b* + r* (c* - b*) (l_function, poly.rs:538-551)
            F rs;
            memcpy(&rs, &r_star, 32);
            rs = gkr::h64::to_mont(rs);
            gkr_fr* zn = outs[b].z + z_off + k_i;
            for (int j = 0; j < k; ++j) {
                F bj, cj;
                memcpy(&bj, &b_star[j], 32);
                memcpy(&cj, &c_star[j], 32);
                const F v = gkr::h64::add(bj, gkr::h64::mont_mul(gkr::h64::sub(cj, bj), rs));
                memcpy(&zn[j], &v, 32);
                memcpy(&z_next[(size_t)b * k + j], &v, 32);
            }
        };
        {
            std::atomic<int> next{0};
            const std::function<bool()> work = [&]() -> bool {
                const int b = next.fetch_add(1, std::memory_order_relaxed);
                if (b >= batch) return false;
                finish(b);
                return true;
            };
            gkr::SpinPool::Session session(pool, nullptr);
            run_pieces(pool, &work, batch > 1);
        }
        if (dbg_pb)
            fprintf(stderr, "[gkr timing] prove layer %u: sumcheck %.0f us, q / z on the host %.0f us (since entry of the hand-off: %.0f us)\n", i,
                    std::chrono::duration<double, std::micro>(tl1 - tl0).count(),
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tl1).count(),
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tpb0).count());
        z_cur.swap(z_next);
        row_off += (size_t)2 * k;
        q_off += (size_t)k + 1;
        z_off += (size_t)k_i;
    }
    {
        HIP_TRY(ctx, hipGetLastError());
        copier.finish();   // (all coefficient copies are queued on the side stream before it is waited for)
        liner.finish();    // (and all line restrictions)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->aux));
        // what the side stream brought back from before the layers: the checks of the inputs and outputs, and -- for tables
        // the host transforms -- d and input_func
        if (h_in_flag) return ctx->fail(GKR_ERR_NON_CANONICAL, "input value >= r");
        for (int b = 0; b < batch; ++b) {
            if (members[member_of[b]].require_zero_output && !gkr::h64::is_zero(hW0_first ? hW0_first[b] : hW[0][(size_t)b << c->k[0]]))
                return ctx->fail(GKR_ERR_INVALID, "output 0 is not zero (convert.rs:838 asserts d_values[0] == 0)");
            // monomial forms the Proof carries (get_multi_ext): d = W_0, input_func = W_L
            std::vector<F> co;
            if (!coeffs_done[0]) {
                co.assign(hW[0] + ((size_t)b << c->k[0]), hW[0] + ((size_t)(b + 1) << c->k[0]));
                mobius_msb(co, c->k[0]);
                memcpy(outs[b].d_coeffs, co.data(), co.size() * sizeof(F));
            }
            if (!coeffs_done[1]) {
                co.assign(hW[1] + ((size_t)b << c->k[L]), hW[1] + ((size_t)(b + 1) << c->k[L]));
                mobius_msb(co, c->k[L]);
                memcpy(outs[b].input_coeffs, co.data(), co.size() * sizeof(F));
            }
        }

        const F* hq = reinterpret_cast<const F*>(d_q);
        const uint32_t* hqlen = d_qlen;
        size_t off = 0;
        for (uint32_t i = 0; i < L; ++i) {
            const size_t kq = (size_t)c->k[i + 1] + 1;
            for (int b = 0; b < batch; ++b) {
                memcpy(outs[b].q + off, &hq[off * batch + (size_t)b * kq], kq * sizeof(F));
                outs[b].q_len[i] = hqlen[(size_t)i * batch + b];
            }
            off += kq;
        }
    }
    if (dbg_pb) {
        const auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "[gkr timing] prove batch=%d depth=%u: %.0f us before the layers (circuit lookup, forward evaluation, readback), "
                        "%.0f us layers + q readback, since entry %.0f us\n", batch, L, us(t_entry, tpb0), us(tpb0, t_end), us(t_entry, t_end));
        fprintf(stderr, "[gkr timing] this thread: own hashing pieces (incl. waiting for helpers) %.0f us, others' pieces %.0f us, spinning with nothing to take %.0f us, "
                        "the rest (launches, set-up, copies) %.0f us\n", t_account.own_pieces_us, t_account.helped_us, t_account.spin_us,
                us(t_entry, t_end) - t_account.own_pieces_us - t_account.helped_us - t_account.spin_us);
    }
    if (account) {
        HostAccountTotals& tot = host_account_totals();
        const double all_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_entry).count();
        const double rest = all_us - t_account.own_pieces_us - t_account.helped_us - t_account.spin_us;
        tot.own_ns.fetch_add((uint64_t)(t_account.own_pieces_us * 1e3), std::memory_order_relaxed);
        tot.helped_ns.fetch_add((uint64_t)(t_account.helped_us * 1e3), std::memory_order_relaxed);
        tot.spin_ns.fetch_add((uint64_t)(t_account.spin_us * 1e3), std::memory_order_relaxed);
        tot.rest_ns.fetch_add((uint64_t)((rest > 0 ? rest : 0) * 1e3), std::memory_order_relaxed);
        tot.calls.fetch_add(1, std::memory_order_relaxed);
    }
    copier.finish();
    if (copier.err != hipSuccess) return ctx->hip_fail(copier.err, "copy of the d / input_func coefficients to the proof buffers");
    for (auto& f : fresh) {
        if (!f || no_cache) continue;
        constexpr size_t kMaxCachedCircuits = 64;   // three aggregation steps' worth of sub-circuits
        if (ctx->circuits.size() >= kMaxCachedCircuits) {
            // (never an entry this call is still using -- a group's members are all in use until here)
            ctx->circuits.front()->release();
            ctx->circuits.erase(ctx->circuits.begin());
        }
        ctx->circuits.push_back(std::move(f));
    }
    return GKR_OK;
}

static int prove_batch_impl(gkr_ctx* ctx, const gkr_circuit_desc* c, const gkr_fr* input_values, int batch, int require_zero_output,
                            gkr_proof_buf* outs) {
    gkr_prove_item one{c, input_values, batch, require_zero_output, outs, 0};
    return prove_group_impl(ctx, &one, 1);
}

int gkr_prove(gkr_ctx* ctx, const gkr_circuit_desc* c, const gkr_fr* input_values, int require_zero_output,
              gkr_proof_buf* out) {
    if (!ctx) return GKR_ERR_INVALID;
    return prove_batch_impl(ctx, c, input_values, 1, require_zero_output, out);
}

int gkr_prove_batch(gkr_ctx* ctx, const gkr_circuit_desc* c, const gkr_fr* input_values, int batch, int require_zero_output,
                    gkr_proof_buf* outs) {
    if (!ctx) return GKR_ERR_INVALID;
    return prove_batch_impl(ctx, c, input_values, batch, require_zero_output, outs);
}

// ---- gkr_prove_many: the items of one aggregation step proven side by side ----------------------------------------
static void crew_prove_items(ProveCrew* crew, ProveCrew::Member* m) {
    if (accounting_on()) host_account_totals().wake_ns.fetch_add((uint64_t)((now_us_dbg() - crew->t_call_us) * 1e3), std::memory_order_relaxed);
    // threads per unit in this call: how finely a unit cuts its hashing into pieces (capi_layer.hip, the product passes)
    m->ctx->help_share = crew->units->empty() ? 1 : (crew->active + (int)crew->units->size() - 1) / (int)crew->units->size();
    for (int u : m->items) {
        const std::vector<int>& unit = (*crew->units)[u];
        auto prove_one = [&](int idx) {
            gkr_prove_item& it = crew->items[idx];
            if (!it.circuit || !it.input_values || !it.outs) {
                it.status = m->ctx->fail(GKR_ERR_INVALID, "null pointer in a prove item");
                return;
            }
            it.status = prove_batch_impl(m->ctx, it.circuit, it.input_values, it.batch, it.require_zero_output, it.outs);
        };
        if (unit.size() == 1) {
            prove_one(unit[0]);
            continue;
        }
        // a lockstep group: its items' proofs advance together, one launch per pass
        std::vector<gkr_prove_item> group;
        for (int idx : unit) group.push_back(crew->items[idx]);
        const int rc = prove_group_impl(m->ctx, group.data(), (int)group.size());
        if (rc == GKR_OK) {
            for (int idx : unit) crew->items[idx].status = GKR_OK;
        } else {
            // something in the group failed (a bad gate, a witness that does not satisfy its circuit ...): the items one by
            // one, so that every item gets its own status and the others are still proven
            for (int idx : unit) prove_one(idx);
        }
    }
    __atomic_fetch_sub(&crew->busy, 1, __ATOMIC_ACQ_REL);
    (void)gkr_host_help_while(&crew->busy);   // out of items: pieces of the others' host work until all are done
}

static void crew_thread(ProveCrew* crew, int index) {
    ProveCrew::Member* m = crew->members[index].get();
    (void)hipSetDevice(m->ctx->device);
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> g(crew->mu);
            crew->cv_start.wait(g, [&] { return crew->stop || crew->generation != seen; });
            if (crew->stop) return;
            seen = crew->generation;
            if (index >= crew->active) continue;   // not needed in this call
        }
        crew_prove_items(crew, m);
        {
            std::lock_guard<std::mutex> g(crew->mu);
            ++crew->finished;
        }
        crew->cv_done.notify_one();
    }
}

static void destroy_crew(ProveCrew* crew) {
    if (!crew) return;
    {
        std::lock_guard<std::mutex> g(crew->mu);
        crew->stop = true;
    }
    crew->cv_start.notify_all();
    for (size_t i = 1; i < crew->members.size(); ++i) {
        if (crew->members[i]->th.joinable()) crew->members[i]->th.join();
        gkr_ctx_destroy(crew->members[i]->ctx);
    }
    delete crew;
}

int gkr_prove_many(gkr_ctx* ctx, gkr_prove_item* items, size_t n_items, int max_concurrent) {
    if (!ctx) return GKR_ERR_INVALID;
    if ((!items && n_items) || max_concurrent < 0 || n_items > (size_t)1 << 20) return ctx->fail(GKR_ERR_INVALID, "null item list or negative thread count");
    if (n_items == 0) return GKR_OK;
    if (ctx->crew_member) return ctx->fail(GKR_ERR_INVALID, "gkr_prove_many from inside a crew");
    gkr::OptionScope option_scope(&ctx->options);
    // (child contexts are created on their devices below, and member 0 may end up without an item: the caller's current
    // device is put back on every path out)
    struct RestoreDevice {
        int prev = -1;
        RestoreDevice() {
            if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        }
        ~RestoreDevice() {
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    } restore_device;
    int want = max_concurrent;
    if (!want) {
        int share = usable_cpus();
        if (const int ranks = gkr::process_int("LOCAL_WORLD_SIZE", 1); ranks > 1) share = share / ranks > 1 ? share / ranks : 1;   // ranks of one node share its CPUs
        want = share >= 6 ? share - 2 : (share >= 3 ? share - 1 : share);   // two (one, none) left to the runtime's own threads
    }
    // (members beyond the number of items have nothing to prove: they lend themselves from the start -- only if asked for)
    {
        // (members beyond what the items -- cut in two where they are large, below -- can occupy have nothing to prove)
        size_t can_use = n_items;
        if (gkr::opt(gkr::OPT_prove_many_pieces) > 0)
            for (size_t i = 0; i < n_items; ++i) can_use += items[i].batch >= 32 ? (size_t)items[i].batch / 32 : 0;
        if (!max_concurrent && (size_t)want > can_use) want = (int)can_use;
    }
    // several devices: at least one member per device (as far as there are items), or a device would sit idle
    if (!max_concurrent && !ctx->devices.empty() && want < (int)ctx->devices.size())
        want = n_items < ctx->devices.size() ? (int)n_items : (int)ctx->devices.size();
    if (want > 64) want = 64;
    if (!ctx->crew) {
        ctx->crew = std::unique_ptr<ProveCrew, void (*)(ProveCrew*)>(new ProveCrew(), destroy_crew);
        ctx->crew->members.emplace_back(new ProveCrew::Member());
        ctx->crew->members[0]->ctx = ctx;
    }
    ProveCrew* crew = ctx->crew.get();
    for (size_t m = 1; m < crew->members.size(); ++m) crew->members[m]->ctx->options = ctx->options;   // (idle between calls)
    while ((int)crew->members.size() < want) {
        gkr_ctx* child = nullptr;
        // member m lives on device devices[m mod #devices] (member 0 = this context, on devices[0])
        const int member_device = ctx->devices.empty() ? ctx->device : ctx->devices[crew->members.size() % ctx->devices.size()];
        const int rc = gkr_ctx_create(member_device, &child);
        if (rc) return ctx->fail(rc, "child context of gkr_prove_many");
        child->crew_member = true;
        child->transcript = GKR_TRANSCRIPT_HOST;
        child->options = ctx->options;
        crew->members.emplace_back(new ProveCrew::Member());
        crew->members.back()->ctx = child;
        const int index = (int)crew->members.size() - 1;
        crew->members.back()->th = std::thread(crew_thread, crew, index);
    }
    // GKR_PROVE_MANY_PIECES = n (opt-in): the costliest items with >= 32 witnesses are cut in two until there are n items --
    // the halves are independent proving chains like any other item.  Meant for the deep sub-circuits of an R1CS, which run
    // alone for the last third of a step; measured on MI355X (64 inputs x 12 sub-circuits, 14 threads, ms per step, two
    // runs each): no cut 9.0 / 9.2, 14 items 8.4 / 10.1, 16: 9.2 / 10.1, 19 (all seven deep ones cut): 11.1 / 12.2,
    // 24: 12.1 / 12.8 (profiles/r04/e_prove_many_item_split_ab.txt) -- every extra chain adds its launches and hand-offs
    // (~750 launches per step already) and the step gets SLOWER; the default is no cut.
    auto item_cost = [](const gkr_prove_item& it) {
        double rounds = 0;
        const gkr_circuit_desc* c = it.circuit;
        if (c && c->k && c->depth <= 4096)
            for (uint32_t l = 1; l <= c->depth; ++l) rounds += 2.0 * c->k[l];
        return rounds * (50.0 + 2.0 * (it.batch > 0 ? it.batch : 1));
    };
    const int pieces_env = (int)gkr::opt(gkr::OPT_prove_many_pieces);
    std::vector<gkr_prove_item> work(items, items + n_items);
    std::vector<int> origin(n_items);
    for (size_t i = 0; i < n_items; ++i) origin[i] = (int)i;
    const size_t aim = pieces_env > 0 ? (size_t)pieces_env : 0;
    while (work.size() < aim) {
        int best = -1;
        double best_cost = 0;
        for (size_t i = 0; i < work.size(); ++i) {
            const gkr_prove_item& it = work[i];
            if (it.batch < 32 || !it.circuit || !it.circuit->k || !it.input_values || !it.outs) continue;
            const double c = item_cost(it);
            if (c > best_cost) {
                best_cost = c;
                best = (int)i;
            }
        }
        if (best < 0) break;
        gkr_prove_item a = work[best], b = work[best];
        const int half = ((a.batch / 2 + 15) / 16) * 16;   // whole sixteen-lane hash calls in the first half
        a.batch = half;
        b.batch = work[best].batch - half;
        b.input_values = a.input_values + ((size_t)half << a.circuit->k[a.circuit->depth]);
        b.outs = a.outs + half;
        work[best] = a;
        work.push_back(b);
        origin.push_back(origin[best]);
    }
    gkr_prove_item* const caller_items = items;
    const size_t caller_n = n_items;
    items = work.data();
    n_items = work.size();
    // Units: items whose circuits share their k list advance in LOCKSTEP (prove_group_impl) -- the <= 20 sub-circuits of one
    // compiled R1CS come in a few shapes (the 16 of a 262 144-constraint R1CS: 8 x [14,15,15,15], 7 x [14,15,16,15,15,15], one
    // other), and 16 chains of one-block kernels are bound by the four hardware queues they share, not by the chip.
    std::vector<std::vector<int>> units;
    {
        const bool lockstep = gkr::opt(gkr::OPT_prove_many_lockstep) != 0;
        const long long cap_opt = gkr::opt(gkr::OPT_lockstep_max_proofs);
        // Lockstep pays where a chain is LATENCY-bound -- a few proofs per circuit: one launch and one hand-off per pass for the
        // group, its handful of round vectors hashed side by side by the crew.  Many proofs per item are a THROUGHPUT problem:
        // the host hashes for a millisecond per pass while the GPU waits and the other way round, and independent chains
        // overlap each other where one big group cannot (MI355X, 14 threads: 64 inputs x 12 sub-circuits 8.5 ms as twelve
        // chains, 12.2 ms as two groups of 320 and 448 proofs; 3 inputs x 12: 3.7 against 3.2 ms; the 16 sub-circuits of the
        // 262 144-constraint R1CS, one input: 10.8 against 7.1 ms -- profiles/r05/a_*).  Hence the cap on a group's proofs.
        const int cap = cap_opt > 0 && cap_opt <= 4096 ? (int)cap_opt : 32;
        std::vector<int> unit_proofs;
        std::vector<bool> taken(n_items, false);
        for (size_t i = 0; i < n_items; ++i) {
            if (taken[i]) continue;
            taken[i] = true;
            units.push_back({(int)i});
            unit_proofs.push_back(items[i].batch);
            const gkr_circuit_desc* c = items[i].circuit;
            if (!lockstep || !c || !c->k || c->depth < 1 || c->depth > 4096 || !items[i].input_values || !items[i].outs || items[i].batch < 1) continue;
            if (check_circuit(nullptr, c) != GKR_OK || !k_list_groupable(c->k, c->depth)) continue;
            for (size_t j = i + 1; j < n_items; ++j) {
                const gkr_circuit_desc* d = items[j].circuit;
                if (taken[j] || !d || !d->k || d->depth != c->depth || !items[j].input_values || !items[j].outs || items[j].batch < 1) continue;
                if (memcmp(d->k, c->k, (c->depth + 1) * sizeof(uint32_t)) != 0) continue;
                // (the group's proofs: at most `cap`; a member larger than the rest of the room starts a group of its own later)
                if (unit_proofs.back() + items[j].batch > cap) continue;
                taken[j] = true;
                units.back().push_back((int)j);
                unit_proofs.back() += items[j].batch;
            }
        }
    }
    auto unit_cost = [&](const std::vector<int>& u) {
        gkr_prove_item all = items[u[0]];
        for (size_t e = 1; e < u.size(); ++e) all.batch += items[u[e]].batch;
        return item_cost(all);
    };
    // deal the units out by estimated cost, longest first, each to the member with the least so far (deterministic)
    std::vector<std::pair<double, int>> cost(units.size());
    for (size_t i = 0; i < n_items; ++i) items[i].status = GKR_OK;
    for (size_t u = 0; u < units.size(); ++u) cost[u] = {unit_cost(units[u]), (int)u};
    // (members beyond the number of units prove nothing themselves: they take pieces of the units' host work from the start --
    // a lockstep group's round vectors are hashed by the whole crew, not by the one thread that launches its kernels)
    std::stable_sort(cost.begin(), cost.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; });
    std::vector<double> load(want, 0.0);
    for (int m = 0; m < want; ++m) crew->members[m]->items.clear();
    for (const auto& ci : cost) {
        int best = 0;
        for (int m = 1; m < want; ++m)
            if (load[m] < load[best]) best = m;
        load[best] += ci.first;
        crew->members[best]->items.push_back(ci.second);
    }
    const int saved_transcript = ctx->transcript;
    ctx->transcript = GKR_TRANSCRIPT_HOST;
    ctx->crew_member = true;   // the parent proves its share like the others: one thread, no workers of its own
    {
        std::lock_guard<std::mutex> g(crew->mu);
        crew->items = items;
        crew->units = &units;
        crew->active = want;
        crew->finished = 0;
        __atomic_store_n(&crew->busy, want, __ATOMIC_RELEASE);
        crew->t_call_us = now_us_dbg();
        ++crew->generation;
    }
    crew->cv_start.notify_all();
    crew_prove_items(crew, crew->members[0].get());
    {
        std::unique_lock<std::mutex> g(crew->mu);
        crew->cv_done.wait(g, [&] { return crew->finished == want - 1; });
        crew->items = nullptr;
        crew->units = nullptr;
    }
    ctx->crew_member = false;
    ctx->transcript = saved_transcript;
    for (size_t i = 0; i < caller_n; ++i) caller_items[i].status = GKR_OK;
    int first_bad = GKR_OK;
    for (int m = 0; m < want; ++m)
        for (int u : crew->members[m]->items)
            for (int idx : units[u])
                if (items[idx].status != GKR_OK) {
                    if (caller_items[origin[idx]].status == GKR_OK) caller_items[origin[idx]].status = items[idx].status;
                    if (first_bad == GKR_OK) {
                        if (m) ctx->err = crew->members[m]->ctx->err;
                        first_bad = items[idx].status;
                    }
                }
    return first_bad;
}


}  // extern "C"
