// Internals shared by the translation units of the C ABI (gkr_capi.hip: contexts, transcript helpers, self-tests, device
// memory; capi_mle.hip: the plain sumcheck; capi_layer.hip: the layer sumcheck; capi_prove.hip: whole proofs): the context,
// its caches and workspaces, profiling brackets, the host transcript's helpers, the hand-off wait.  Not a public header.
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gkr_amd.h"
#include "fr64.h"
#include "host_cpus.h"
#include "hostpool.h"
#include "kernels.h"
#include "options.h"
#include "gate_seg.h"
#include "keccak.h"
#include "mimc7.h"
#include "mimc_adx.h"
#include "mimc_ifma.h"

using gkr::Fr;

static_assert(sizeof(gkr_fr) == sizeof(Fr), "gkr_fr and the device element share one 32-byte layout");


namespace gkr_host {


struct ProfileRow {
    uint64_t launches = 0;
    double total_ms = 0.0;
    double bytes = 0.0;
    std::vector<std::pair<float, double>> samples;   // (ms, algorithmic bytes) of single launches, the first kMaxSamples
};
constexpr size_t kMaxSamples = 4096;

struct PendingEvent {
    hipEvent_t start, stop;
    const char* name;
    double bytes;
};

// The three lazily built constant tables of the host transcript are reached by up to 64 crew threads at once on the
// first gkr_prove_many of a process: function-local statics (one thread builds, the others wait; the finished table
// is published with the guard's release).
inline const Fr* host_mimc_constants() {
    struct Table {
        Fr cts[gkr::kMimcRounds];
        Table() { gkr::mimc7_make_constants(cts); }
    };
    static const Table t;
    return t.cts;
}

// the same 91 constants for the 4 x 64-bit host arithmetic (identical bytes: same Montgomery radix)
inline const gkr::h64::F* host_mimc_constants64() {
    struct Table {
        gkr::h64::F cts[gkr::kMimcRounds];
        Table() { memcpy(cts, host_mimc_constants(), sizeof cts); }
    };
    static const Table t;
    return t.cts;
}

using gkr::usable_cpus;   // (host_cpus.h)

// true once the eight-lane IFMA hash is initialised (CPU has avx512ifma and GKR_NO_IFMA is unset)
inline bool host_ifma_ready() {
    static const bool ready = [] {   // one thread initialises the 52-bit tables, the others wait at the guard
        if (gkr::process_switch("GKR_NO_IFMA") || !gkr::gkr_ifma_available()) return false;
        const gkr::h64::F* cts = host_mimc_constants64();
        static uint64_t canon[gkr::kMimcRounds][4];
        for (int i = 0; i < gkr::kMimcRounds; ++i) {
            const gkr::h64::F c = gkr::h64::from_mont(cts[i]);
            memcpy(canon[i], &c, 32);
        }
        gkr::gkr_ifma_init(canon);
        return true;
    }();
    return ready;
}

// ONE transcript's hash of a round vector (canonical in, canonical out) on the calling thread: the mulx / adcx / adox
// code of mimc_adx.cpp where the CPU has it (GKR_NO_ADX unset), the portable 4 x 64-bit code of fr64.h otherwise
inline gkr::h64::F host_multi_hash(const gkr::h64::F* arr, int n, const gkr::h64::F* cts) {
    static const bool adx = !gkr::process_switch("GKR_NO_ADX") && gkr::gkr_adx_available();
    if (!adx) return gkr::h64::mimc7_multi_hash(arr, n, cts, nullptr);
    gkr::h64::F out;
    gkr::gkr_adx_multi_hash(reinterpret_cast<const uint64_t(*)[4]>(arr), n, reinterpret_cast<const uint64_t(*)[4]>(cts), out.l);
    return out;
}

// up to sixteen transcripts on the IFMA code: two interleaved groups of eight fill the FMA pipes (one group is
// a dependent chain), so chunks of sixteen cost ~1.3x a chunk of eight
constexpr int kHashChunkMax = 16;
// the size limits of include/gkr_amd.h
constexpr int kMaxLayerK = GKR_MAX_K_NEXT, kMaxLayerKi = GKR_MAX_K_I, kMaxDenseK = GKR_MAX_K_NEXT_DEVICE_TRANSCRIPT;
inline void ifma_hash_chunk(const uint64_t (*vec)[3][4], const uint32_t* ln, int count, uint64_t (*out)[4]) {
    if (count > 8)
        gkr::gkr_ifma_multi_hash16(vec, ln, 3, out);
    else
        gkr::gkr_ifma_multi_hash8(vec, ln, 3, out);
}
// chunk of sumchecks one host thread hashes at a time: sixteen when the host threads are the scarce resource
// (throughput: MI355X + 2 threads, 256 x 2^20: 7.1 ms per step against 8.5), eight otherwise (latency: with 15
// threads 5.04 ms against 5.19)
inline int hash_chunk_size(int tables, int threads, int help_share = 0) {
    const long long forced = gkr::opt(gkr::OPT_hash_chunk);   // (called on a context's own thread, before the pieces are posted)
    if (forced == 8 || forced == 16) return (int)forced;
    // One of a crew's units (gkr_prove_many) with `help_share` threads to itself on average: a FEW proofs in lockstep are a
    // latency chain -- a lone transcript's scalar hash takes 12 us per round vector, an IFMA call ~25 us whether one lane is
    // filled or sixteen -- so they go out one proof per piece to the threads that have no unit of their own;
    // many proofs are a throughput problem: whole eight- or sixteen-lane calls.
    if (help_share > 1 && tables <= 3 * help_share) return 1;
    return tables >= 32 * threads ? 16 : 8;
}

// Device allocations of the library: plain hipMalloc.  (Rounds 1-2 kept two experiment modes here -- contiguous physical
// memory, and reserved virtual ranges backed by 2 MiB .. 8 GiB handles -- to find out whether the streaming kernels'
// bandwidth depends on how an allocation is mapped.  It does not: profiles/r02/c_placement_map_modes.txt; what varied
// was the fold kernel's launch geometry, kernels.hip mle_multifold_blocks.  Retired in round 5.)
inline hipError_t device_malloc(void** p, size_t bytes) { return hipMalloc(p, bytes); }
inline hipError_t device_free(void* p) { return hipFree(p); }

inline int default_host_threads() {
    if (const int v = gkr::process_int("GKR_HOST_THREADS", 0); v >= 1) return v;
    int cpus = usable_cpus();
    int local = 1;   // one process per GPU: share the host cores between the ranks of this node
    if (const int ranks = gkr::process_int("LOCAL_WORLD_SIZE", 1); ranks > 0) local = ranks;
    // leave room for the HIP runtime's own threads and the interpreter -- unless the rank's share is so small that
    // the transcript needs all of it (a rank of the 2^20 workload needs ~2 hashing threads to keep its GPU fed)
    const int share = cpus / local;
    int t = share >= 6 ? share - 2 : (share >= 3 ? share - 1 : share);
    if (t < 1) t = 1;
    if (t > 64) t = 64;
    return t;
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

inline bool all_canonical(const gkr_fr* v, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (!gkr::fr_is_canonical(to_dev(v[i]))) return false;
    return true;
}


}  // namespace gkr_host
using namespace gkr_host;


// Gate arrays of a circuit kept on the device across gkr_prove / gkr_prove_batch calls, with each layer's gate
// lists (the counting sort by left / right operand the linear-time layer sumcheck sums over): they depend only on
// the circuit, and an aggregation step proves the same <= 20 circuits for input after input.
struct GateLists {
    uint32_t *offsets = nullptr, *cursor = nullptr, *list = nullptr;
    uint32_t* plan = nullptr;    // wide layers: the item plan of the gate passes (kernels_wide.hip, launch_gate_plan)
    gkr::GatePlanCounts plan_counts{};   // ... and its counts as read back when it was built (known: exact grids, no empty combine step)
    gkr::GateSegs segs;   // the lists' segments (large layers; segs.words is one more device allocation)
    bool ready = false;
    void release() {
        if (offsets) (void)hipFree(offsets);
        if (cursor) (void)hipFree(cursor);
        if (list) (void)hipFree(list);
        if (plan) (void)hipFree(plan);
        if (segs.words) (void)hipFree(segs.words);
        offsets = cursor = list = plan = nullptr;
        plan_counts = gkr::GatePlanCounts{};
        segs = gkr::GateSegs();
        ready = false;
    }
};
struct PreparedCircuit {
    uint64_t h1 = 0, h2 = 0;
    std::vector<uint32_t> k;
    std::vector<uint8_t*> gt;
    std::vector<uint32_t*> l, r;
    std::vector<GateLists> lists;
    std::vector<unsigned char> retained;   // the gate arrays as they were when the entry was made (all of them, or sampled blocks): compared on a hit
    void release() {
        for (auto p : gt) (void)hipFree(p);
        for (auto p : l) (void)hipFree(p);
        for (auto p : r) (void)hipFree(p);
        for (auto& g : lists) g.release();
        gt.clear();
        l.clear();
        r.clear();
        lists.clear();
    }
};

struct ProveCrew;

// One helper thread per context, started on first use and kept: work that must not hold the proving thread up (the staged
// copies of a wide proof's coefficient tables to the caller's pageable buffers).  One task at a time: run() hands it over,
// wait() returns when it is done.  (A thread per call would do, but costs its start-up on every proof and churns the
// per-thread state of whatever tool is attached to the process.)
struct AsyncWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> task;
    bool busy = false, stop = false;
    void loop() {
        std::unique_lock<std::mutex> g(mu);
        for (;;) {
            cv.wait(g, [this] { return stop || (busy && task); });
            if (stop) return;
            std::function<void()> t = std::move(task);
            task = nullptr;
            g.unlock();
            t();
            g.lock();
            busy = false;
            cv.notify_all();
        }
    }
    void run(std::function<void()> fn) {
        std::unique_lock<std::mutex> g(mu);
        if (!th.joinable()) th = std::thread([this] { loop(); });
        cv.wait(g, [this] { return !busy; });
        task = std::move(fn);
        busy = true;
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [this] { return !busy; });
    }
    ~AsyncWorker() {
        {
            std::unique_lock<std::mutex> g(mu);
            cv.wait(g, [this] { return !busy; });
            stop = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

struct gkr_ctx {
    int device = 0;
    std::vector<int> devices;                  // gkr_ctx_create_multi: the devices gkr_prove_many's child contexts are dealt over (empty: `device` only)
    int host_threads = 0;                      // 0: from GKR_HOST_THREADS / the usable CPUs; else this many (caller included)
    gkr::Options options = gkr::Options::from_environment();   // gkr_ctx_set_option; read through gkr::opt() inside GKR_ENTER's scope
    std::vector<std::unique_ptr<PreparedCircuit>> circuits;   // most recently used last; bounded
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;                 // side stream for tiny kernels that only depend on host-written data (lazy)
    hipStream_t late = nullptr;                // separate stream for a group's small late passes (lazy), see late_stream()
    hipStream_t chain = nullptr;               // the device-hashed sumchecks' chain of kernels (lazy), see chain_stream()
    std::vector<hipEvent_t> aux_events;        // one per group of a batch: "the side kernel of this group is done"
    Fr* d_cts = nullptr;
    int transcript = GKR_TRANSCRIPT_HOST;
    std::string err;
    int profile = 0;                           // 0 off, 1 every kernel, 2 the bandwidth-bound kernels only
    std::map<std::string, ProfileRow> prof;
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
    char name[256] = {0};
    uint32_t ticket = 0;                       // unique per hand-off, never reused within a context
    std::unique_ptr<gkr::SpinPool> pool;       // host transcript workers (lazy)
    std::unique_ptr<gkr::SpinPool> solo_pool;  // the empty pool a context uses while it is one of a crew (gkr_prove_many)
    std::unique_ptr<AsyncWorker> copier;       // helper thread for a wide proof's coefficient copies (lazy)
    std::unique_ptr<AsyncWorker> liner;        // helper thread that issues a wide layer's line-restriction launches on the side stream (lazy)
    void* mle_arrivals_zeroed = nullptr;       // the same for the plain sumcheck's latency-bound passes
    void* arrivals_zeroed = nullptr;           // the product passes' arrival counters: zeroed once per allocation (every pass leaves them zero)
    void* gate_arrive_zeroed = nullptr;        // the same for the wide gate passes' combine step
    bool crew_member = false;                  // one thread of several proving side by side: no workers of its own
    int help_share = 0;                        // crew threads per proving unit of the current gkr_prove_many call (0: not in one)
    int rounds_ahead = 0;                      // sumcheck rounds left in the proof being proven AFTER the current layer (help priority)
    std::unique_ptr<ProveCrew, void (*)(ProveCrew*)> crew{nullptr, nullptr};   // gkr_prove_many's threads and child contexts (lazy)
    std::map<std::string, std::pair<void*, size_t>> ws;        // grow-only device workspaces
    std::map<std::string, std::pair<void*, size_t>> pinned;    // grow-only pinned host buffers

    // cached device workspace: hipMalloc / hipFree of multi-GiB buffers costs milliseconds per call
    hipError_t workspace(const char* slot, size_t bytes, void** out) {
        auto& e = ws[slot];
        if (e.second < bytes) {
            if (e.first) (void)device_free(e.first);
            e.first = nullptr;
            e.second = 0;
            hipError_t rc = device_malloc(&e.first, bytes);
            if (rc != hipSuccess) return rc;
            e.second = bytes;
        }
        *out = e.first;
        return hipSuccess;
    }
    hipError_t pinned_host(const char* slot, size_t bytes, void** out) {
        auto& e = pinned[slot];
        if (e.second < bytes) {
            if (e.first) (void)hipHostFree(e.first);
            e.first = nullptr;
            e.second = 0;
            hipError_t rc = hipHostMalloc(&e.first, bytes, hipHostMallocCoherent | hipHostMallocMapped);
            if (rc != hipSuccess) return rc;
            memset(e.first, 0, bytes);
            e.second = bytes;
        }
        *out = e.first;
        return hipSuccess;
    }
    void release_buffers() {
        for (auto& c : circuits) c->release();
        circuits.clear();
        for (auto& kv : ws)
            if (kv.second.first) (void)device_free(kv.second.first);
        ws.clear();
        for (auto& kv : pinned)
            if (kv.second.first) (void)hipHostFree(kv.second.first);
        pinned.clear();
    }
    int threads() const { return host_threads > 0 ? host_threads : default_host_threads(); }
    gkr::SpinPool* host_pool() {
        if (crew_member) {
            if (!solo_pool) solo_pool.reset(new gkr::SpinPool(0));
            return solo_pool.get();
        }
        if (!pool) pool.reset(new gkr::SpinPool(threads() - 1));
        return pool.get();
    }
    // The passes of one group are ordered by the host (a pass is launched after the previous one's record has
    // landed), not by the stream.  Its LATE passes -- tables of a few thousand entries, latency-bound round trips --
    // go to their own stream (normal priority): on the main stream they would queue behind the other groups' multi-GiB
    // streaming passes launched earlier, and all groups' tails would pile up at the end of the call (measured: 1.9 of
    // 13.1 ms per 1024 sumchecks).
    hipError_t late_stream(hipStream_t* out) {
        if (!late) {
            // (normal priority: a HIGH-priority stream, once created in a process, doubled the round latency of every small
            // kernel launched later on other streams -- DESIGN.md section 3, "Round scheduling")
            const hipError_t rc = hipStreamCreateWithFlags(&late, hipStreamNonBlocking);
            if (rc != hipSuccess) return rc;
        }
        *out = late;
        return hipSuccess;
    }
    // The sumchecks hashed on the device run as ONE chain on this stream (kernels_transcript.hip): with the main, side and
    // late streams that makes four -- ROCm's default number of hardware queues; a fifth stream would share a queue with
    // one of them, and its kernels would queue behind that stream's.
    hipError_t chain_stream(hipStream_t* out) {
        if (!chain) {
            // (normal priority: at high priority the step got slower, 18.0 against 16.2 ms -- profiles/r04/p_*)
            hipError_t rc = hipStreamCreateWithFlags(&chain, hipStreamNonBlocking);
            if (rc != hipSuccess) return rc;
        }
        *out = chain;
        return hipSuccess;
    }
    hipError_t aux_stream(int events) {
        if (!aux) {
            hipError_t rc = hipStreamCreateWithFlags(&aux, hipStreamNonBlocking);
            if (rc != hipSuccess) return rc;
        }
        while ((int)aux_events.size() < events) {
            hipEvent_t e;
            hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
            if (rc != hipSuccess) return rc;
            aux_events.push_back(e);
        }
        return hipSuccess;
    }

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
        (void)hipEventCreate(&e);
        return e;
    }
    // a step timed on the host clock (an exchange through host memory: not a kernel) under the same profile names
    void add_host_sample(const char* name, double ms) {
        ProfileRow& r = prof[name];
        r.launches += 1;
        r.total_ms += ms;
        if (r.samples.size() < kMaxSamples) r.samples.emplace_back((float)ms, 0.0);
    }
    void drain_events() {
        for (auto& p : pending) {
            (void)hipEventSynchronize(p.stop);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, p.start, p.stop);
            ProfileRow& r = prof[p.name];
            r.launches += 1;
            r.total_ms += ms;
            r.bytes += p.bytes;
            if (r.samples.size() < kMaxSamples) r.samples.emplace_back(ms, p.bytes);
            event_pool.push_back(p.start);
            event_pool.push_back(p.stop);
        }
        pending.clear();
    }
};

// gkr_prove_many's crew: member 0 is the calling thread with the parent context, every other member a thread with a
// child context of its own.  The threads sleep between calls.
struct ProveCrew {
    struct Member {
        gkr_ctx* ctx = nullptr;
        std::thread th;
        std::vector<int> items;   // indices into the current call's UNIT list, in proving order
    };
    std::vector<std::unique_ptr<Member>> members;
    std::mutex mu;
    std::condition_variable cv_start, cv_done;
    uint64_t generation = 0;
    bool stop = false;
    int active = 0;               // members taking part in the current call (the first `active`)
    int finished = 0;             // of the threads (members 1 ..), in the current call
    gkr_prove_item* items = nullptr;
    const std::vector<std::vector<int>>* units = nullptr;   // a unit: one item, or the items of a lockstep group (same k list)
    int32_t busy = 0;             // members still proving (atomic access); the others lend themselves
    double t_call_us = 0;         // (accounting) when the current call woke the crew
};

// RAII timing bracket around one launch (only when profiling is on)
struct Timed {
    gkr_ctx* c;
    PendingEvent ev;
    bool on;
    hipStream_t st;
    // minor: a small latency-bound kernel on the round-trip path (left out at profile level 2, where the event
    // records themselves would show in the wall time)
    Timed(gkr_ctx* ctx, const char* name, double bytes, hipStream_t stream = nullptr, bool minor = false)
        : c(ctx), on(ctx->profile == 1 || (ctx->profile == 2 && !minor)), st(stream ? stream : ctx->stream) {
        if (on) {
            ev.start = c->get_event();
            ev.stop = c->get_event();
            ev.name = name;
            ev.bytes = bytes;
            (void)hipEventRecord(ev.start, st);
        }
    }
    ~Timed() {
        if (on) {
            (void)hipEventRecord(ev.stop, st);
            c->pending.push_back(ev);
        }
    }
};

// every compute entry point of the C ABI: the context's device becomes current and the context's options become the
// calling thread's (options.h) until the entry point returns
#define GKR_ENTER(ctx)                                        \
    HIP_TRY(ctx, hipSetDevice((ctx)->device));                \
    gkr::OptionScope gkr_option_scope_(&(ctx)->options)

#define HIP_TRY(ctx, expr)                                   \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return (ctx)->hip_fail(_e, #expr); \
    } while (0)


namespace gkr_host {


// device buffer that frees itself
template <typename T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t count) { return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)); }
};

// ------------------------------------------------------------- hand-off waiting
// Spin on the seq words the reduce kernel stores last (system-scope release) into
// pinned host memory.  Bounded: a device fault or a lost launch turns into an
// error status instead of a hang.
// Host work in pieces (`work` claims and runs one per call): this thread, this context's pool workers if a session is
// open, and -- while the job is on the process-wide board -- threads of OTHER contexts that are waiting for their GPU
// (wait_records).  Returns when every piece has been run to its end.  GKR_NO_HELP=1: no sharing between contexts.
// GKR_DEBUG_TIMING: where a proving thread's time goes (per thread, summed over a gkr_prove_batch call)
struct ThreadTimeAccount {
    double own_pieces_us = 0, helped_us = 0, spin_us = 0;
};
inline thread_local ThreadTimeAccount t_account;
// gkr_host_accounting (C ABI): the same accounts summed over all threads of the process, for a caller that wants the figures
// instead of the stderr lines -- bench.py puts them on its line for one extra, untimed proving step
struct HostAccountTotals {
    std::atomic<uint64_t> own_ns{0}, helped_ns{0}, spin_ns{0}, rest_ns{0}, lent_ns{0}, lent_idle_ns{0}, calls{0}, wake_ns{0};
    // the hashing pieces themselves (whoever ran them): how many, their time, the part of it inside the pass function (the J
    // hashes of the piece's transcripts in IFMA lanes or on the scalar code, and the field arithmetic between them), and how
    // many transcripts a piece carried (1 .. 16: the lanes of an IFMA call that were filled)
    std::atomic<uint64_t> pieces{0}, piece_ns{0}, piece_pass_ns{0}, lanes[17] = {};
    std::atomic<bool> on{false};
};
inline HostAccountTotals& host_account_totals() {
    static HostAccountTotals t;
    return t;
}
inline bool accounting_on() { return host_account_totals().on.load(std::memory_order_relaxed); }
inline void account_piece(int transcripts, double pass_us, double total_us) {
    HostAccountTotals& t = host_account_totals();
    t.pieces.fetch_add(1, std::memory_order_relaxed);
    t.piece_ns.fetch_add((uint64_t)(total_us * 1e3), std::memory_order_relaxed);
    t.piece_pass_ns.fetch_add((uint64_t)(pass_us * 1e3), std::memory_order_relaxed);
    t.lanes[transcripts < 1 ? 1 : (transcripts > 16 ? 16 : transcripts)].fetch_add(1, std::memory_order_relaxed);
}
inline double now_us_dbg() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

inline bool help_enabled() {
    static const bool on = !gkr::process_switch("GKR_NO_HELP");
    return on;
}
static void run_pieces(gkr::SpinPool* pool, const std::function<bool()>* work, bool several, int priority = 0) {
    auto run = [&] {
        if (pool)
            pool->run_now(work);
        else
            while ((*work)()) {
            }
    };
    const bool dbg = gkr::debug_timing() || accounting_on();
    const double t0 = dbg ? now_us_dbg() : 0.0;
    if (several && help_enabled()) {
        gkr::HelpBoard::Posted posted(work, priority);
        run();
    } else {
        run();
    }
    if (dbg) t_account.own_pieces_us += now_us_dbg() - t0;
}

template <typename Rec>
inline int wait_records(gkr_ctx* ctx, const volatile Rec* recs, int count, uint32_t ticket) {
    const auto t0 = std::chrono::steady_clock::now();
    const bool help = help_enabled();
    // The stream check and the deadline run every so often whether or not the wait was filled with other contexts'
    // pieces (a faulted stream must be noticed also while the help board stays busy); the deadline counts from this
    // context's last own progress -- a record of ITS round landing -- not from the entry, so time spent on others'
    // work does not run it down.
    auto last_progress = t0;
    for (int b = 0; b < count; ++b) {
        uint32_t spins = 0, helped = 0;
        while (__atomic_load_n(&recs[b].seq, __ATOMIC_ACQUIRE) != ticket) {
            // this thread has nothing to do until its round lands: a piece of another context's posted host work
            // (a 16-lane hash call, ~30 us) instead of spinning
            const bool dbg = gkr::debug_timing() || accounting_on();
            const double th0 = dbg ? now_us_dbg() : 0.0;
            const bool did_help = help && gkr::HelpBoard::instance().help();
            if (!did_help) GKR_CPU_RELAX();
            if (dbg) (did_help ? t_account.helped_us : t_account.spin_us) += now_us_dbg() - th0;
            if (did_help ? (++helped & 0x3F) == 0 : (++spins & 0xFFFF) == 0) {
                hipError_t q = hipStreamQuery(ctx->stream);
                if (q != hipSuccess && q != hipErrorNotReady) return ctx->hip_fail(q, "stream failed while waiting for a round");
                if (std::chrono::steady_clock::now() - last_progress > std::chrono::seconds(30)) {
                    if (q == hipSuccess && __atomic_load_n(&recs[b].seq, __ATOMIC_ACQUIRE) == ticket) break;
                    return ctx->fail(GKR_ERR_HIP, "timed out waiting for the device to publish a round");
                }
            }
        }
        if (spins | helped) last_progress = std::chrono::steady_clock::now();
    }
    return GKR_OK;
}

constexpr int kMaxGroups = 32;
constexpr int kMleShardTailLog2 = 6;   // entries (log2) every shard keeps for the gathered tail of a sumcheck split over ranks

#define WS(ctx, slot, type, count, ptr) \
    HIP_TRY(ctx, (ctx)->workspace(slot, (size_t)(count) * sizeof(type), reinterpret_cast<void**>(&(ptr))))

// Rounds the sub-block sums of a table of 2^m entries cover (= variables the next fold pass binds) in a sumcheck over
// 2^n points with at most jmax rounds per pass.  A fold pass over more than kSmallPassEntries outputs splits each
// sub-block over whole 64-entry chunks (so at most m - 6 rounds from its sums); and a fold should not leave 1024 or
// 2048 entries -- too many for the one-block kernel, too few to fill the chip with 64-entry wave tiles: stop at 4096
// and take the rest in the pass after.
inline int mle_pass_rounds(int m, int n, int jmax) {
    int j = m < jmax ? m : jmax;
    if (((size_t)1 << m) > gkr::kSmallPassEntries && m != n && j > m - 6) j = m - 6;
    if (jmax > 3 && (m - j == 10 || m - j == 11) && m - 12 >= 1) j = m - 12;
    return j < 1 ? 1 : j;
}

// ---- defined in capi_mle.hip
void host_pass_scalar(const uint64_t* sums, size_t sums_row_words, int count, int J, const uint32_t* final_len, uint64_t (*c0)[16][4],
                      uint64_t (*c1)[16][4], uint64_t (*r)[16][4], uint32_t (*len)[16], uint64_t* weights, size_t w_row_words);
// `tail` (may be null): the tables are the TAIL of longer sumchecks (gkr_sumcheck_mle_sharded_dev: what is left of a table
// split over ranks, gathered) -- round j of a tail is round round_offset + j of a sumcheck with n_total rounds (row stride
// of the outputs), and the last round's length follows the ORIGINAL table's dependence on its last variable (dep_last,
// null: the tail is the whole table), not the folded tail's.
struct MleTailArgs {
    int n_total = 0, round_offset = 0;
    const uint32_t* dep_last = nullptr;
};
int run_mle_batch_passes(gkr_ctx* ctx, const Fr* d_tables, int n, int batch, gkr_fr* out_coeffs, uint32_t* out_len, gkr_fr* out_r,
                         const MleTailArgs* tail = nullptr);
int run_mle_batch(gkr_ctx* ctx, const Fr* d_tables, int n, int batch, gkr_fr* out_coeffs, uint32_t* out_len, gkr_fr* out_r);
// ---- defined in capi_layer.hip
void host_tail_pass(gkr::h64::F* tables, size_t stride, uint32_t m, uint32_t jp, const gkr::h64::F* weights, uint32_t J, gkr::h64::F* rec);
void host_tail_pass_scalar(gkr::h64::F* tables, size_t stride, uint32_t m, uint32_t jp, const gkr::h64::F* weights, uint32_t J, gkr::h64::F* rec);
void host_prod_pass_scalar(const uint64_t* recs, size_t rec_row_words, int count, int J, const uint32_t (*vec_len)[16], uint64_t (*c2)[16][4],
                           uint64_t (*lin)[16][4], uint64_t (*c0)[16][4], uint64_t (*r)[16][4], uint64_t* weights, size_t w_row_words);
// One rank's share of a layer split across GPUs by GATES (gkr_sumcheck_layer_sharded): the device gate arrays hold
// gates gate_base .. gate_base + gate_count - 1, and the two tables that are sums over gates -- (U, V) before the
// b-rounds, the row (a_u, m_u) before the c-rounds -- are completed by the caller's sum-over-ranks hook.
struct LayerShardArgs {
    uint64_t gate_base = 0, gate_count = 0;
    gkr_allreduce_fn allreduce = nullptr;   // host hook (field elements in host memory), or
    void* user = nullptr;
    const gkr_exchange_dev* dev = nullptr;  // device exchange: limbs widened into the caller's device buffer, summed on the stream
};
// A lockstep group (gkr_prove_many): the proofs of a launch belong to DIFFERENT circuits whose layer has this shape; the
// passes over the gates read each proof's lists through d_sets (one gkr::GateSet per proof, device memory).
struct LayerGroup {
    const gkr::GateSet* d_sets = nullptr;
    gkr::GatePlanCounts plan_counts{};   // the largest counts over the members (known only if every member's are)
};
int run_layer_batch(gkr_ctx* ctx, int batch, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r, const gkr_fr* z,
                    const Fr* d_W, gkr_fr* const* out_coeffs, uint32_t* const* out_len, gkr_fr* const* out_r, const LayerShardArgs* shard = nullptr,
                    GateLists* cached = nullptr, const LayerGroup* group = nullptr);
// the cached gate lists of one circuit layer, built and validated as a call of its own (a lone circuit builds them inside its
// first layer sumcheck; a lockstep group needs every member's before its first launch)
int build_cached_gate_lists(gkr_ctx* ctx, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r, GateLists* cached);
int run_layer(gkr_ctx* ctx, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r, const gkr_fr* z, const Fr* d_W,
              gkr_fr* out_coeffs, uint32_t* out_len, gkr_fr* out_r);
// ---- defined in capi_prove.hip
void mobius_msb(std::vector<gkr::h64::F>& c, int k);
void line_restriction(const std::vector<gkr::h64::F>& vals, const std::vector<gkr::h64::F>& coeffs, int k, const gkr_fr* b, const gkr_fr* c,
                      gkr_fr* out, uint32_t* out_len);
int check_circuit(gkr_ctx* ctx, const gkr_circuit_desc* c);

}  // namespace gkr_host
