// C ABI (include/gkr_amd.h) and host driver of the MI355X GKR sumcheck prover.
//
// Host-side mirror of the reference's prover loop (rust/src/gkr/prover.rs:6-96):
// per layer  predicate build -> 2k sumcheck rounds -> q_i -> r* -> z_{i+1}.
// All table work is launched on the context's HIP stream; the per-round MiMC7
// hash runs on the device by default, so a whole sumcheck is one uninterrupted
// stream of launches with a single copy-back at the end.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/gkr_amd.h"
#include "kernels.h"
#include "keccak.h"
#include "mimc7.h"

using gkr::Fr;

static_assert(sizeof(gkr_fr) == sizeof(Fr), "gkr_fr and the device element share one 32-byte layout");

namespace {

struct ProfileRow {
    uint64_t launches = 0;
    double total_ms = 0.0;
    double bytes = 0.0;
};

struct PendingEvent {
    hipEvent_t start, stop;
    const char* name;
    double bytes;
};

const Fr* host_mimc_constants() {
    static Fr cts[gkr::kMimcRounds];
    static bool ready = false;
    if (!ready) {  // idempotent; first use happens under ctx creation or a host hash call
        gkr::mimc7_make_constants(cts);
        ready = true;
    }
    return cts;
}

inline Fr to_dev(const gkr_fr& x) {
    Fr f;
    memcpy(&f, &x, 32);
    return f;
}

inline gkr_fr to_abi(const Fr& x) {
    gkr_fr f;
    memcpy(&f, &x, 32);
    return f;
}

bool all_canonical(const gkr_fr* v, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (!gkr::fr_is_canonical(to_dev(v[i]))) return false;
    return true;
}

}  // namespace

struct gkr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    Fr* d_cts = nullptr;
    int transcript = GKR_TRANSCRIPT_DEVICE;
    std::string err;
    bool profile = false;
    std::map<std::string, ProfileRow> prof;
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
    char name[256] = {0};

    int fail(int status, const std::string& what) {
        err = what;
        return status;
    }
    int hip_fail(hipError_t e, const char* what) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return GKR_ERR_HIP;
    }
    hipEvent_t get_event() {
        if (!event_pool.empty()) {
            hipEvent_t e = event_pool.back();
            event_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        hipEventCreate(&e);
        return e;
    }
    void drain_events() {
        for (auto& p : pending) {
            hipEventSynchronize(p.stop);
            float ms = 0.f;
            hipEventElapsedTime(&ms, p.start, p.stop);
            ProfileRow& r = prof[p.name];
            r.launches += 1;
            r.total_ms += ms;
            r.bytes += p.bytes;
            event_pool.push_back(p.start);
            event_pool.push_back(p.stop);
        }
        pending.clear();
    }
};

// RAII timing bracket around one launch (only when profiling is on)
struct Timed {
    gkr_ctx* c;
    PendingEvent ev;
    bool on;
    Timed(gkr_ctx* ctx, const char* name, double bytes) : c(ctx), on(ctx->profile) {
        if (on) {
            ev.start = c->get_event();
            ev.stop = c->get_event();
            ev.name = name;
            ev.bytes = bytes;
            hipEventRecord(ev.start, c->stream);
        }
    }
    ~Timed() {
        if (on) {
            hipEventRecord(ev.stop, c->stream);
            c->pending.push_back(ev);
        }
    }
};

#define HIP_TRY(ctx, expr)                                   \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return (ctx)->hip_fail(_e, #expr); \
    } while (0)

namespace {

// device buffer that frees itself
template <typename T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() {
        if (p) hipFree(p);
    }
    hipError_t alloc(size_t count) { return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)); }
};

// ------------------------------------------------------------- plain MLE sumcheck
int run_mle_batch(gkr_ctx* ctx, const Fr* d_tables, int n, int batch, gkr_fr* out_coeffs, uint32_t* out_len,
                  gkr_fr* out_r) {
    const size_t len = (size_t)1 << n;
    const size_t rounds = (size_t)batch * n;
    DevBuf<Fr> work, d_coeffs, d_r, d_rmont;
    DevBuf<uint32_t> d_len, d_dep;
    DevBuf<gkr::MlePartial> partials;
    const uint32_t max_nblk = gkr::mle_blocks_per_table((uint32_t)(len / 2), (uint32_t)batch);
    HIP_TRY(ctx, work.alloc((size_t)batch * (len / 2)));
    HIP_TRY(ctx, d_coeffs.alloc(rounds * 2));
    HIP_TRY(ctx, d_r.alloc(rounds));
    HIP_TRY(ctx, d_rmont.alloc(rounds));
    HIP_TRY(ctx, d_len.alloc(rounds));
    HIP_TRY(ctx, d_dep.alloc(batch));
    HIP_TRY(ctx, partials.alloc((size_t)batch * max_nblk));
    hipStream_t s = ctx->stream;

    if (ctx->transcript != GKR_TRANSCRIPT_DEVICE)
        return ctx->fail(GKR_ERR_INVALID, "host transcript mode is not available for this entry point yet");

    // round 1: sums only
    {
        const uint32_t h = (uint32_t)(len / 2);
        const uint32_t nblk = gkr::mle_blocks_per_table(h, batch);
        {
            Timed t(ctx, "mle_sum_first", (double)batch * len * 32.0);
            gkr::launch_mle_sum_first(d_tables, len, h, batch, nblk, partials.p, s);
        }
        {
            Timed t(ctx, "mle_round_hash", 0.0);
            gkr::launch_mle_round_hash(partials.p, nblk, 0, n, batch, ctx->d_cts, d_coeffs.p, d_len.p, d_r.p, d_rmont.p,
                                       d_dep.p, s);
        }
    }
    // rounds 2..n: fold with r_{j-1}, sum T_j in the same pass
    for (int round = 1; round < n; ++round) {
        const uint32_t q = (uint32_t)(len >> (round + 1));  // quarter of the source table
        const uint32_t nblk = gkr::mle_blocks_per_table(q, batch);
        const Fr* src = (round == 1) ? d_tables : work.p;
        const size_t src_stride = (round == 1) ? len : len / 2;
        {
            Timed t(ctx, "mle_fold_sum", (double)batch * 6.0 * q * 32.0);
            gkr::launch_mle_fold_sum(src, src_stride, work.p, len / 2, q, batch, nblk, d_rmont.p + (round - 1), n,
                                     partials.p, s);
        }
        {
            Timed t(ctx, "mle_round_hash", 0.0);
            gkr::launch_mle_round_hash(partials.p, nblk, round, n, batch, ctx->d_cts, d_coeffs.p, d_len.p, d_r.p,
                                       d_rmont.p, d_dep.p, s);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_coeffs, d_coeffs.p, rounds * 2 * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(out_len, d_len.p, rounds * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(out_r, d_r.p, rounds * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    ctx->drain_events();
    return GKR_OK;
}

// ------------------------------------------------------------- predicate tables
// builds canonical A, M (2^{2k} each) in device memory from device gate arrays
int build_predicates(gkr_ctx* ctx, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r,
                     const gkr_fr* z, Fr* d_A, Fr* d_M) {
    const size_t N = (size_t)1 << (2 * k);
    hipStream_t s = ctx->stream;
    DevBuf<unsigned long long> wideA, wideM;
    DevBuf<Fr> zfac;
    DevBuf<uint32_t> bad;
    HIP_TRY(ctx, wideA.alloc(N * 8));
    HIP_TRY(ctx, wideM.alloc(N * 8));
    HIP_TRY(ctx, zfac.alloc(2 * (size_t)(k_i > 0 ? k_i : 1)));
    HIP_TRY(ctx, bad.alloc(1));
    std::vector<Fr> hz(2 * (size_t)(k_i > 0 ? k_i : 1), gkr::fr_zero());
    Fr one = gkr::fr_zero();
    one.l[0] = 1;
    for (int i = 0; i < k_i; ++i) {
        Fr zi = to_dev(z[i]);
        hz[2 * i] = gkr::to_mont(gkr::fr_sub(one, zi));
        hz[2 * i + 1] = gkr::to_mont(zi);
    }
    HIP_TRY(ctx, hipMemcpyAsync(zfac.p, hz.data(), hz.size() * sizeof(Fr), hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(wideA.p, 0, N * 64, s));
    HIP_TRY(ctx, hipMemsetAsync(wideM.p, 0, N * 64, s));
    HIP_TRY(ctx, hipMemsetAsync(bad.p, 0, 4, s));
    {
        Timed t(ctx, "predicate_scatter", (double)((size_t)1 << k_i) * (9.0 + 64.0));
        gkr::launch_predicate_scatter(k_i, k, d_gt, d_l, d_r, zfac.p, wideA.p, wideM.p, bad.p, s);
    }
    {
        Timed t(ctx, "predicate_normalise", (double)N * 2.0 * (64.0 + 32.0));
        gkr::launch_predicate_normalise(wideA.p, d_A, N, s);
        gkr::launch_predicate_normalise(wideM.p, d_M, N, s);
    }
    uint32_t hbad = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&hbad, bad.p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));   // also keeps hz alive until the upload is done
    if (hbad) return ctx->fail(GKR_ERR_INVALID, "gate operand index out of range for 2^k_next");
    return GKR_OK;
}

// ------------------------------------------------------------- layer sumcheck
// d_W: canonical values of layer i+1 (2^k) in device memory
int run_layer(gkr_ctx* ctx, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r,
              const gkr_fr* z, const Fr* d_W, gkr_fr* out_coeffs, uint32_t* out_len, gkr_fr* out_r) {
    const size_t N = (size_t)1 << (2 * k);
    const uint32_t v = 2 * k;
    hipStream_t s = ctx->stream;
    DevBuf<Fr> A, M, Wb, Wc, d_coeffs, d_r_out, d_rmont;
    DevBuf<uint32_t> d_len, dep;
    DevBuf<gkr::LayerPartial> partials;
    HIP_TRY(ctx, A.alloc(N));
    HIP_TRY(ctx, M.alloc(N));
    HIP_TRY(ctx, Wb.alloc((size_t)1 << k));
    HIP_TRY(ctx, Wc.alloc((size_t)1 << k));
    HIP_TRY(ctx, d_coeffs.alloc((size_t)v * 3));
    HIP_TRY(ctx, d_r_out.alloc(v));
    HIP_TRY(ctx, d_rmont.alloc(v));
    HIP_TRY(ctx, d_len.alloc(v));
    HIP_TRY(ctx, dep.alloc(k));
    HIP_TRY(ctx, partials.alloc(gkr::kMaxLayerBlocks));
    int rc = build_predicates(ctx, k_i, k, d_gt, d_l, d_r, z, A.p, M.p);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemsetAsync(dep.p, 0, sizeof(uint32_t) * k, s));
    gkr::launch_to_mont(d_W, Wb.p, 1u << k, s);
    HIP_TRY(ctx, hipMemcpyAsync(Wc.p, Wb.p, sizeof(Fr) << k, hipMemcpyDeviceToDevice, s));
    gkr::launch_depends(d_W, k, dep.p, s);
    for (uint32_t round = 0; round < v; ++round) {
        const uint32_t h = (uint32_t)(N >> (round + 1));
        const uint32_t phase = round < (uint32_t)k ? 0u : 1u;
        const uint32_t hb = phase == 0 ? (h >> k) : 0u;
        const uint32_t nblk = gkr::layer_blocks(h);
        {
            Timed t(ctx, "layer_round", (double)h * 4.0 * 32.0);
            gkr::launch_layer_round(A.p, M.p, h, k, phase, hb, Wb.p, Wc.p, nblk, partials.p, s);
        }
        {
            Timed t(ctx, "layer_round_hash", 0.0);
            gkr::launch_layer_round_hash(partials.p, nblk, round, k, dep.p, ctx->d_cts, d_coeffs.p, d_len.p, d_r_out.p,
                                         d_rmont.p, Wb.p, Wc.p, s);
        }
        if (round + 1 < v) {
            Timed t(ctx, "layer_fold", (double)h * 6.0 * 32.0);
            gkr::launch_layer_fold(A.p, M.p, h, d_rmont.p + round, s);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_coeffs, d_coeffs.p, (size_t)v * 3 * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(out_len, d_len.p, v * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(out_r, d_r_out.p, v * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    ctx->drain_events();
    return GKR_OK;
}

// evaluation table -> monomial coefficients, variable 1 = most significant bit
// (what get_multi_ext stores, poly.rs:502-536)
void mobius_msb(std::vector<Fr>& c, int k) {
    const size_t n = (size_t)1 << k;
    for (int b = 0; b < k; ++b) {
        const size_t bit = (size_t)1 << (k - 1 - b);
        for (size_t i = 0; i < n; ++i)
            if (i & bit) c[i] = gkr::fr_sub(c[i], c[i ^ bit]);
    }
}

// reduce_multiple_polynomial (poly.rs:469-500): q(t) = W(b + t (c - b)).
// coeffs: monomial coefficients of W.  out: k+1 slots right-aligned, highest first.
void line_restriction(const std::vector<Fr>& coeffs, int k, const gkr_fr* b, const gkr_fr* c, gkr_fr* out,
                      uint32_t* out_len) {
    std::vector<Fr> grad(k), cst(k), res(k + 1, gkr::fr_zero()), poly(k + 2);
    for (int j = 0; j < k; ++j) {
        grad[j] = gkr::to_mont(gkr::fr_sub(to_dev(c[j]), to_dev(b[j])));
        cst[j] = gkr::to_mont(to_dev(b[j]));
    }
    int maxdeg = 0;
    const size_t n = (size_t)1 << k;
    for (size_t mono = 0; mono < n; ++mono) {
        if (gkr::fr_is_zero(coeffs[mono])) continue;
        int deg = 0;
        poly[0] = coeffs[mono];
        for (int j = 0; j < k; ++j) {
            if (!((mono >> (k - 1 - j)) & 1)) continue;
            poly[deg + 1] = gkr::fr_zero();
            for (int d = deg + 1; d >= 1; --d)
                poly[d] = gkr::fr_add(gkr::mont_mul(poly[d - 1], grad[j]), gkr::mont_mul(poly[d], cst[j]));
            poly[0] = gkr::mont_mul(poly[0], cst[j]);
            ++deg;
        }
        if (deg > maxdeg) maxdeg = deg;
        for (int d = 0; d <= deg; ++d) res[d] = gkr::fr_add(res[d], poly[d]);
    }
    *out_len = (uint32_t)(maxdeg + 1);
    for (int d = 0; d <= k; ++d) out[k - d] = to_abi(res[d]);
}

int check_circuit(gkr_ctx* ctx, const gkr_circuit_desc* c) {
    if (!c || !c->k || c->depth < 1 || !c->gate_type || !c->left || !c->right)
        return ctx ? ctx->fail(GKR_ERR_INVALID, "null circuit description") : GKR_ERR_INVALID;
    for (uint32_t i = 0; i <= c->depth; ++i)
        if (c->k[i] > 28) return ctx ? ctx->fail(GKR_ERR_INVALID, "layer wider than 2^28") : GKR_ERR_INVALID;
    for (uint32_t i = 1; i <= c->depth; ++i) {
        if (c->k[i] == 0) return ctx ? ctx->fail(GKR_ERR_DEGENERATE, "k[i+1] == 0: v = 0 (sumcheck.rs:49)") : GKR_ERR_DEGENERATE;
        if (c->k[i] > 14) return ctx ? ctx->fail(GKR_ERR_INVALID, "dense predicate tables need k[i+1] <= 14") : GKR_ERR_INVALID;
    }
    return GKR_OK;
}

}  // namespace

// =========================================================================== C ABI

extern "C" {

const char* gkr_strerror(int status) {
    switch (status) {
        case GKR_OK: return "ok";
        case GKR_ERR_INVALID: return "invalid argument";
        case GKR_ERR_NON_CANONICAL: return "field element is not canonical (>= r)";
        case GKR_ERR_NO_DEVICE: return "no gfx950 device available";
        case GKR_ERR_HIP: return "HIP runtime error";
        case GKR_ERR_NOMEM: return "out of memory";
        case GKR_ERR_DEGENERATE: return "degenerate sumcheck (v == 0)";
        default: return "unknown status";
    }
}

const char* gkr_version(void) { return "gkr_amd 0.1 (gfx950)"; }

int gkr_ctx_create(int device_id, gkr_ctx** out) {
    if (!out) return GKR_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return GKR_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return GKR_ERR_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return GKR_ERR_HIP;
    gkr_ctx* c = new gkr_ctx();
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess)
        snprintf(c->name, sizeof c->name, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return GKR_ERR_HIP;
    }
    if (hipMalloc(reinterpret_cast<void**>(&c->d_cts), sizeof(Fr) * gkr::kMimcRounds) != hipSuccess ||
        hipMemcpy(c->d_cts, host_mimc_constants(), sizeof(Fr) * gkr::kMimcRounds, hipMemcpyHostToDevice) != hipSuccess) {
        gkr_ctx_destroy(c);
        return GKR_ERR_HIP;
    }
    *out = c;
    return GKR_OK;
}

void gkr_ctx_destroy(gkr_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    ctx->drain_events();
    for (hipEvent_t e : ctx->event_pool) hipEventDestroy(e);
    if (ctx->d_cts) hipFree(ctx->d_cts);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* gkr_last_error(const gkr_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int gkr_ctx_set_transcript(gkr_ctx* ctx, int mode) {
    if (!ctx || (mode != GKR_TRANSCRIPT_DEVICE && mode != GKR_TRANSCRIPT_HOST)) return GKR_ERR_INVALID;
    ctx->transcript = mode;
    return GKR_OK;
}

int gkr_ctx_device_name(const gkr_ctx* ctx, char* buf, size_t len) {
    if (!ctx || !buf || !len) return GKR_ERR_INVALID;
    snprintf(buf, len, "%s", ctx->name);
    return GKR_OK;
}

int gkr_ctx_profile(gkr_ctx* ctx, int enable) {
    if (!ctx) return GKR_ERR_INVALID;
    ctx->profile = enable != 0;
    return GKR_OK;
}

int gkr_ctx_profile_get(gkr_ctx* ctx, const char* kernel, uint64_t* launches, double* total_ms, double* bytes) {
    if (!ctx || !kernel) return GKR_ERR_INVALID;
    ctx->drain_events();
    auto it = ctx->prof.find(kernel);
    ProfileRow r = it == ctx->prof.end() ? ProfileRow() : it->second;
    if (launches) *launches = r.launches;
    if (total_ms) *total_ms = r.total_ms;
    if (bytes) *bytes = r.bytes;
    return GKR_OK;
}

int gkr_ctx_profile_reset(gkr_ctx* ctx) {
    if (!ctx) return GKR_ERR_INVALID;
    ctx->drain_events();
    ctx->prof.clear();
    return GKR_OK;
}

// ---- MiMC7 on the host ------------------------------------------------------

int gkr_mimc7_multi_hash(const gkr_fr* arr, size_t n, const gkr_fr* key, gkr_fr* out) {
    if ((!arr && n) || !out) return GKR_ERR_INVALID;
    if (!all_canonical(arr, n) || (key && !all_canonical(key, 1))) return GKR_ERR_NON_CANONICAL;
    const Fr* cts = host_mimc_constants();
    Fr r = key ? gkr::to_mont(to_dev(*key)) : gkr::fr_zero();
    for (size_t i = 0; i < n; ++i) {
        Fr a = gkr::to_mont(to_dev(arr[i]));
        Fr h = gkr::mimc7_hash_mont(a, r, cts);
        r = gkr::fr_add(gkr::fr_add(r, a), h);
    }
    *out = to_abi(gkr::from_mont(r));
    return GKR_OK;
}

int gkr_mimc7_hash(const gkr_fr* x, const gkr_fr* k, gkr_fr* out) {
    if (!x || !k || !out) return GKR_ERR_INVALID;
    if (!all_canonical(x, 1) || !all_canonical(k, 1)) return GKR_ERR_NON_CANONICAL;
    Fr h = gkr::mimc7_hash_mont(gkr::to_mont(to_dev(*x)), gkr::to_mont(to_dev(*k)), host_mimc_constants());
    *out = to_abi(gkr::from_mont(h));
    return GKR_OK;
}

int gkr_mimc7_constant(int i, gkr_fr* out) {
    if (i < 0 || i >= gkr::kMimcRounds || !out) return GKR_ERR_INVALID;
    *out = to_abi(gkr::from_mont(host_mimc_constants()[i]));
    return GKR_OK;
}

int gkr_selftest_mul(const gkr_fr* a, const gkr_fr* b, gkr_fr* out) {
    if (!a || !b || !out) return GKR_ERR_INVALID;
    if (!all_canonical(a, 1) || !all_canonical(b, 1)) return GKR_ERR_NON_CANONICAL;
    *out = to_abi(gkr::fr_mul(to_dev(*a), to_dev(*b)));
    return GKR_OK;
}

int gkr_selftest_wide_sum(const gkr_fr* vals, size_t n, gkr_fr* out) {
    if ((!vals && n) || !out) return GKR_ERR_INVALID;
    if (!all_canonical(vals, n)) return GKR_ERR_NON_CANONICAL;
    // the kernels' accumulation scheme: 288-bit partials, 320-bit totals, one reduction
    gkr::Acc<10> total = gkr::acc_zero<10>();
    gkr::Acc<9> part = gkr::acc_zero<9>();
    for (size_t i = 0; i < n; ++i) {
        gkr::acc_add_fr(part, to_dev(vals[i]));
        if ((i & 1023) == 1023) {
            gkr::acc_add_acc(total, part);
            part = gkr::acc_zero<9>();
        }
    }
    gkr::acc_add_acc(total, part);
    *out = to_abi(gkr::acc_reduce(total));
    return GKR_OK;
}

// ---- plain multilinear sumcheck -------------------------------------------------

int gkr_sumcheck_mle_batch_device(gkr_ctx* ctx, const void* d_tables, int n, int batch, gkr_fr* out_coeffs,
                                  uint32_t* out_len, gkr_fr* out_r) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!d_tables || !out_coeffs || !out_len || !out_r || batch < 1 || batch > 65535)
        return ctx->fail(GKR_ERR_INVALID, "null pointer or batch out of range [1, 65535]");
    if (n < 2 || n > 30) return ctx->fail(GKR_ERR_INVALID, "n must be in [2, 30]");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return run_mle_batch(ctx, static_cast<const Fr*>(d_tables), n, batch, out_coeffs, out_len, out_r);
}

int gkr_sumcheck_mle(gkr_ctx* ctx, const gkr_fr* table, int n, gkr_fr* out_coeffs, uint32_t* out_len, gkr_fr* out_r) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!table || !out_coeffs || !out_len || !out_r) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    if (n < 2 || n > 30) return ctx->fail(GKR_ERR_INVALID, "n must be in [2, 30]");
    const size_t len = (size_t)1 << n;
    if (!all_canonical(table, len)) return ctx->fail(GKR_ERR_NON_CANONICAL, "table entry >= r");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf<Fr> d;
    HIP_TRY(ctx, d.alloc(len));
    HIP_TRY(ctx, hipMemcpyAsync(d.p, table, len * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    return run_mle_batch(ctx, d.p, n, 1, out_coeffs, out_len, out_r);
}

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
    if (k_next < 0 || k_next > 14) return ctx->fail(GKR_ERR_INVALID, "k_next must be in [1, 14] (dense predicate tables)");
    if (k_i > 0 && !all_canonical(z, k_i)) return ctx->fail(GKR_ERR_NON_CANONICAL, "z entry >= r");
    const size_t gates = (size_t)1 << k_i;
    for (size_t g = 0; g < gates; ++g) {
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf<uint8_t> dgt;
    DevBuf<uint32_t> dl, dr;
    DevBuf<Fr> dW;
    rc = upload_gates(ctx, (size_t)1 << k_i, gate_type, left, right, dgt, dl, dr);
    if (rc) return rc;
    HIP_TRY(ctx, dW.alloc((size_t)1 << k_next));
    HIP_TRY(ctx, hipMemcpyAsync(dW.p, W, sizeof(Fr) << k_next, hipMemcpyHostToDevice, ctx->stream));
    return run_layer(ctx, k_i, k_next, dgt.p, dl.p, dr.p, z, dW.p, out_coeffs, out_len, out_r);
}

int gkr_predicate_tables(gkr_ctx* ctx, int k_i, int k_next, const uint8_t* gate_type, const uint32_t* left,
                         const uint32_t* right, const gkr_fr* z, gkr_fr* out_A, gkr_fr* out_M) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!out_A || !out_M) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    int rc = check_layer_args(ctx, k_i, k_next, gate_type, left, right, z);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf<uint8_t> dgt;
    DevBuf<uint32_t> dl, dr;
    DevBuf<Fr> dprev, dout;
    int rc = upload_gates(ctx, gates, gate_type, left, right, dgt, dl, dr);
    if (rc) return rc;
    HIP_TRY(ctx, dprev.alloc(n_prev));
    HIP_TRY(ctx, dout.alloc(gates));
    HIP_TRY(ctx, hipMemcpyAsync(dprev.p, prev, n_prev * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    gkr::launch_layer_eval((uint32_t)gates, dgt.p, dl.p, dr.p, dprev.p, dout.p, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, gates * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GKR_OK;
}

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

int gkr_prove(gkr_ctx* ctx, const gkr_circuit_desc* c, const gkr_fr* input_values, int require_zero_output,
              gkr_proof_buf* out) {
    if (!ctx) return GKR_ERR_INVALID;
    int rc = check_circuit(ctx, c);
    if (rc) return rc;
    if (!input_values || !out || !out->sumcheck_coeffs || !out->sumcheck_len || !out->sumcheck_r || !out->q ||
        !out->q_len || !out->z || !out->r || !out->d_coeffs || !out->input_coeffs)
        return ctx->fail(GKR_ERR_INVALID, "null pointer in proof buffers");
    const uint32_t L = c->depth;
    for (uint32_t i = 0; i < L; ++i) {
        if (!c->gate_type[i] || !c->left[i] || !c->right[i]) return ctx->fail(GKR_ERR_INVALID, "null gate array");
        const size_t gates = (size_t)1 << c->k[i];
        for (size_t g = 0; g < gates; ++g)
            if (c->gate_type[i][g] > 1 || (c->left[i][g] >> c->k[i + 1]) || (c->right[i][g] >> c->k[i + 1]))
                return ctx->fail(GKR_ERR_INVALID, "gate type or operand index out of range");
    }
    const size_t n_in = (size_t)1 << c->k[L];
    if (!all_canonical(input_values, n_in)) return ctx->fail(GKR_ERR_NON_CANONICAL, "input value >= r");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;

    // upload the circuit; forward-evaluate every layer on the device (calculate_input, convert.rs:787-831)
    std::vector<DevBuf<uint8_t>> dgt(L);
    std::vector<DevBuf<uint32_t>> dl(L), dr(L);
    std::vector<DevBuf<Fr>> dW(L + 1);
    for (uint32_t i = 0; i < L; ++i) {
        rc = upload_gates(ctx, (size_t)1 << c->k[i], c->gate_type[i], c->left[i], c->right[i], dgt[i], dl[i], dr[i]);
        if (rc) return rc;
        HIP_TRY(ctx, dW[i].alloc((size_t)1 << c->k[i]));
    }
    HIP_TRY(ctx, dW[L].alloc(n_in));
    HIP_TRY(ctx, hipMemcpyAsync(dW[L].p, input_values, n_in * sizeof(Fr), hipMemcpyHostToDevice, s));
    for (int i = (int)L - 1; i >= 0; --i)
        gkr::launch_layer_eval(1u << c->k[i], dgt[i].p, dl[i].p, dr[i].p, dW[i + 1].p, dW[i].p, s);
    HIP_TRY(ctx, hipGetLastError());
    std::vector<std::vector<Fr>> hW(L + 1);
    for (uint32_t i = 0; i <= L; ++i) {
        hW[i].resize((size_t)1 << c->k[i]);
        HIP_TRY(ctx, hipMemcpyAsync(hW[i].data(), dW[i].p, sizeof(Fr) << c->k[i], hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (require_zero_output && !gkr::fr_is_zero(hW[0][0]))
        return ctx->fail(GKR_ERR_INVALID, "output 0 is not zero (convert.rs:838 asserts d_values[0] == 0)");

    // monomial forms the Proof carries (get_multi_ext): d = W_0, input_func = W_L
    {
        std::vector<Fr> co = hW[0];
        mobius_msb(co, c->k[0]);
        memcpy(out->d_coeffs, co.data(), co.size() * sizeof(Fr));
        co = hW[L];
        mobius_msb(co, c->k[L]);
        memcpy(out->input_coeffs, co.data(), co.size() * sizeof(Fr));
    }

    // z[0] = 0 (prover.rs:16-21)
    gkr_fr* z_cur = out->z;
    for (uint32_t j = 0; j < c->k[0]; ++j) memset(&z_cur[j], 0, sizeof(gkr_fr));
    gkr_fr* sc = out->sumcheck_coeffs;
    uint32_t* sl = out->sumcheck_len;
    gkr_fr* sr = out->sumcheck_r;
    gkr_fr* q = out->q;
    for (uint32_t i = 0; i < L; ++i) {
        const int k_i = c->k[i], k = c->k[i + 1];
        rc = run_layer(ctx, k_i, k, dgt[i].p, dl[i].p, dr[i].p, z_cur, dW[i + 1].p, sc, sl, sr);
        if (rc) return rc;
        const gkr_fr* b_star = sr;
        const gkr_fr* c_star = sr + k;
        // q_i = W_{i+1} restricted to the line b* -> c* (prover.rs:70)
        std::vector<Fr> co = hW[i + 1];
        mobius_msb(co, k);
        line_restriction(co, k, b_star, c_star, q, &out->q_len[i]);
        // r* = multi_hash(last round vector) (prover.rs:74-78) -- the same hash, vector
        // and key as the sumcheck's last challenge, so it is that challenge
        const gkr_fr r_star = sr[2 * k - 1];
        out->r[i] = r_star;
        // z_{i+1} = b* + r* (c* - b*) (l_function, poly.rs:538-551)
        gkr_fr* z_next = z_cur + k_i;
        const Fr rs = gkr::to_mont(to_dev(r_star));
        for (int j = 0; j < k; ++j)
            z_next[j] = to_abi(gkr::fr_fold(to_dev(b_star[j]), to_dev(c_star[j]), rs));
        z_cur = z_next;
        sc += (size_t)2 * k * 3;
        sl += 2 * k;
        sr += 2 * k;
        q += k + 1;
    }
    return GKR_OK;
}

// ---- device memory helpers ---------------------------------------------------------------

int gkr_device_alloc(gkr_ctx* ctx, size_t bytes, void** d_ptr) {
    if (!ctx || !d_ptr || !bytes) return GKR_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(d_ptr, bytes);
    if (e == hipErrorOutOfMemory) return ctx->fail(GKR_ERR_NOMEM, "hipMalloc: out of memory");
    HIP_TRY(ctx, e);
    return GKR_OK;
}

int gkr_device_free(gkr_ctx* ctx, void* d_ptr) {
    if (!ctx) return GKR_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipFree(d_ptr));
    return GKR_OK;
}

int gkr_device_upload(gkr_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!ctx || !d_dst || !h_src) return GKR_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GKR_OK;
}

int gkr_device_download(gkr_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!ctx || !h_dst || !d_src) return GKR_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GKR_OK;
}

int gkr_device_fill_table(gkr_ctx* ctx, void* d_table, size_t count, uint64_t seed) {
    if (!ctx || !d_table || !count) return GKR_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    gkr::launch_fill_table(static_cast<Fr*>(d_table), count, seed, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    return GKR_OK;
}

int gkr_device_synchronize(gkr_ctx* ctx) {
    if (!ctx) return GKR_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GKR_OK;
}

}  // extern "C"
