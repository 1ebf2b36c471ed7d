// C ABI (include/gkr_amd.h) and host driver of the MI355X GKR sumcheck prover.
//
// Host-side mirror of the reference's prover loop (rust/src/gkr/prover.rs:6-96):
// per layer  predicate build -> 2k sumcheck rounds -> q_i -> r* -> z_{i+1}.
// All table work is launched on the context's HIP stream; the per-round MiMC7
// hash runs on the device by default, so a whole sumcheck is one uninterrupted
// stream of launches with a single copy-back at the end.
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
#include "hostpool.h"
#include "kernels.h"
#include "gate_seg.h"
#include "keccak.h"
#include "mimc7.h"
#include "mimc_adx.h"
#include "mimc_ifma.h"

using gkr::Fr;

static_assert(sizeof(gkr_fr) == sizeof(Fr), "gkr_fr and the device element share one 32-byte layout");

namespace {

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
const Fr* host_mimc_constants() {
    struct Table {
        Fr cts[gkr::kMimcRounds];
        Table() { gkr::mimc7_make_constants(cts); }
    };
    static const Table t;
    return t.cts;
}

// the same 91 constants for the 4 x 64-bit host arithmetic (identical bytes: same Montgomery radix)
const gkr::h64::F* host_mimc_constants64() {
    struct Table {
        gkr::h64::F cts[gkr::kMimcRounds];
        Table() { memcpy(cts, host_mimc_constants(), sizeof cts); }
    };
    static const Table t;
    return t.cts;
}

// CPUs this process may really use: the affinity mask, capped by the cgroup CPU
// quota (cgroup v2 cpu.max / v1 cfs_quota).  Spinning on more threads than the
// quota allows gets the whole process throttled for the rest of a 100 ms period.
int usable_cpus() {
    int hw = (int)std::thread::hardware_concurrency();
    cpu_set_t mask;   // a process pinned to a few cores (taskset, per-rank core binding) must not spin on more threads
    CPU_ZERO(&mask);
    if (sched_getaffinity(0, sizeof mask, &mask) == 0 && CPU_COUNT(&mask) > 0) hw = CPU_COUNT(&mask);
    if (hw < 1) hw = 1;
    double quota = -1, period = -1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        double p = 0;
        if (fscanf(f, "%63s %lf", q, &p) == 2 && strcmp(q, "max") != 0) {
            quota = atof(q);
            period = p;
        }
        fclose(f);
    } else {
        FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
        FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fp && fscanf(fq, "%lf", &quota) == 1 && fscanf(fp, "%lf", &period) == 1) {
        } else {
            quota = -1;
        }
        if (fq) fclose(fq);
        if (fp) fclose(fp);
    }
    if (quota > 0 && period > 0) {
        int q = (int)(quota / period);
        if (q < 1) q = 1;
        if (q < hw) hw = q;
    }
    return hw;
}

// true once the eight-lane IFMA hash is initialised (CPU has avx512ifma and GKR_NO_IFMA is unset)
bool host_ifma_ready() {
    static const bool ready = [] {   // one thread initialises the 52-bit tables, the others wait at the guard
        if (getenv("GKR_NO_IFMA") || !gkr::gkr_ifma_available()) return false;
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
    static const bool adx = !getenv("GKR_NO_ADX") && gkr::gkr_adx_available();
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
inline int hash_chunk_size(int tables, int threads) {
    static const int forced = [] {
        const char* e = getenv("GKR_HASH_CHUNK");
        const int v = e ? atoi(e) : 0;
        return v == 8 || v == 16 ? v : 0;
    }();
    if (forced) return forced;
    return tables >= 32 * threads ? 16 : 8;
}

// Device allocations of the library: plain hipMalloc.  Two experiment modes stay behind GKR_ALLOC_MODE because the
// question they answered may come back on other driver versions: whether the streaming kernels' bandwidth depends
// on how an allocation is mapped (it seemed to: round 1's "allocation modes").  Measured in round 2
// (profiles/r02/c_placement_map_modes.txt): hipMemAddressReserve does not honour an alignment above 2 MiB here,
// physical handles of 2 MiB / 64 MiB / 1 GiB / 8 GiB behave like hipMalloc, and the modes were a property of the
// fold kernel's launch geometry, not of the mapping (kernels.hip, mle_multifold_blocks).
//   GKR_ALLOC_MODE=malloc      (default) hipMalloc
//   GKR_ALLOC_MODE=contiguous  hipExtMallocWithFlags(hipDeviceMallocContiguous)
//   GKR_ALLOC_MODE=vmm         buffers >= 256 MiB: reserved virtual range + physical handles of
//                              2^GKR_VMM_CHUNK_LOG2 bytes (default 1 GiB), alignment request 2^GKR_VMM_ALIGN_LOG2
struct VmmAllocation {
    size_t size = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> chunk_sizes;
};
std::mutex g_vmm_mu;
std::map<void*, VmmAllocation> g_vmm;

inline int alloc_mode() {   // 0 malloc, 1 contiguous, 2 vmm
    static const int mode = [] {
        if (getenv("GKR_ALLOC_CONTIGUOUS")) return 1;
        const char* e = getenv("GKR_ALLOC_MODE");
        if (!e || !strcmp(e, "malloc")) return 0;
        if (!strcmp(e, "contiguous")) return 1;
        return !strcmp(e, "vmm") ? 2 : 0;
    }();
    return mode;
}
constexpr size_t kVmmMin = (size_t)256 << 20;

hipError_t vmm_malloc(void** out, size_t bytes) {
    static const size_t align = (size_t)1 << [] { const char* e = getenv("GKR_VMM_ALIGN_LOG2"); const int v = e ? atoi(e) : 30; return v < 21 ? 21 : (v > 36 ? 36 : v); }();
    static const size_t chunk = (size_t)1 << [] { const char* e = getenv("GKR_VMM_CHUNK_LOG2"); const int v = e ? atoi(e) : 30; return v < 21 ? 21 : (v > 36 ? 36 : v); }();
    int dev = 0;
    hipError_t rc = hipGetDevice(&dev);
    if (rc != hipSuccess) return rc;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    rc = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (rc != hipSuccess || gran == 0) return rc != hipSuccess ? rc : hipErrorNotSupported;
    const size_t total = (bytes + gran - 1) / gran * gran;
    void* va = nullptr;
    rc = hipMemAddressReserve(&va, total, align, nullptr, 0);
    if (rc != hipSuccess) return rc;
    VmmAllocation a;
    a.size = total;
    size_t off = 0;
    while (off < total && rc == hipSuccess) {
        size_t sz = total - off < chunk ? total - off : chunk;
        hipMemGenericAllocationHandle_t h;
        rc = hipMemCreate(&h, sz, &prop, 0);
        if (rc != hipSuccess) break;
        rc = hipMemMap(static_cast<char*>(va) + off, sz, 0, h, 0);
        if (rc != hipSuccess) {
            (void)hipMemRelease(h);
            break;
        }
        a.handles.push_back(h);
        a.chunk_sizes.push_back(sz);
        off += sz;
    }
    if (rc == hipSuccess) {
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = dev;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        rc = hipMemSetAccess(va, total, &acc, 1);
    }
    if (rc != hipSuccess) {
        size_t o = 0;
        for (size_t i = 0; i < a.handles.size(); ++i) {
            (void)hipMemUnmap(static_cast<char*>(va) + o, a.chunk_sizes[i]);
            (void)hipMemRelease(a.handles[i]);
            o += a.chunk_sizes[i];
        }
        (void)hipMemAddressFree(va, total);
        return rc;
    }
    {
        std::lock_guard<std::mutex> g(g_vmm_mu);
        g_vmm[va] = std::move(a);
    }
    *out = va;
    return hipSuccess;
}

inline hipError_t device_malloc(void** p, size_t bytes) {
    const int mode = alloc_mode();
    if (mode == 1) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous);
    if (mode == 2 && bytes >= kVmmMin) {
        const hipError_t rc = vmm_malloc(p, bytes);
        if (rc == hipSuccess || rc == hipErrorOutOfMemory) return rc;
        (void)hipGetLastError();   // the virtual-memory API is not usable here: fall back to the plain allocator
    }
    return hipMalloc(p, bytes);
}

inline hipError_t device_free(void* p) {
    VmmAllocation a;
    bool vmm = false;
    {
        std::lock_guard<std::mutex> g(g_vmm_mu);
        auto it = g_vmm.find(p);
        if (it != g_vmm.end()) {
            a = std::move(it->second);
            g_vmm.erase(it);
            vmm = true;
        }
    }
    if (!vmm) return hipFree(p);
    hipError_t rc = hipDeviceSynchronize();
    size_t o = 0;
    for (size_t i = 0; i < a.handles.size(); ++i) {
        const hipError_t u = hipMemUnmap(static_cast<char*>(p) + o, a.chunk_sizes[i]);
        const hipError_t r = hipMemRelease(a.handles[i]);
        if (rc == hipSuccess) rc = u != hipSuccess ? u : r;
        o += a.chunk_sizes[i];
    }
    const hipError_t f = hipMemAddressFree(p, a.size);
    return rc != hipSuccess ? rc : f;
}

int default_host_threads() {
    if (const char* e = getenv("GKR_HOST_THREADS")) {
        int v = atoi(e);
        if (v >= 1) return v;
    }
    int cpus = usable_cpus();
    int local = 1;   // one process per GPU: share the host cores between the ranks of this node
    if (const char* e = getenv("LOCAL_WORLD_SIZE")) local = atoi(e) > 0 ? atoi(e) : 1;
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

bool all_canonical(const gkr_fr* v, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (!gkr::fr_is_canonical(to_dev(v[i]))) return false;
    return true;
}

}  // namespace

// Gate arrays of a circuit kept on the device across gkr_prove / gkr_prove_batch calls, with each layer's gate
// lists (the counting sort by left / right operand the linear-time layer sumcheck sums over): they depend only on
// the circuit, and an aggregation step proves the same <= 20 circuits for input after input.
struct GateLists {
    uint32_t *offsets = nullptr, *cursor = nullptr, *list = nullptr;
    uint32_t* heavy = nullptr;   // wide layers: the work lists of the buckets too long for a lane group (kernels_wide.hip)
    gkr::GateSegs segs;   // the lists' segments (large layers; segs.words is one more device allocation)
    bool ready = false;
    void release() {
        if (offsets) (void)hipFree(offsets);
        if (cursor) (void)hipFree(cursor);
        if (list) (void)hipFree(list);
        if (heavy) (void)hipFree(heavy);
        if (segs.words) (void)hipFree(segs.words);
        offsets = cursor = list = heavy = nullptr;
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

struct gkr_ctx {
    int device = 0;
    std::vector<int> devices;                  // gkr_ctx_create_multi: the devices gkr_prove_many's child contexts are dealt over (empty: `device` only)
    int host_threads = 0;                      // 0: from GKR_HOST_THREADS / the usable CPUs; else this many (caller included)
    std::vector<std::unique_ptr<PreparedCircuit>> circuits;   // most recently used last; bounded
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;                 // side stream for tiny kernels that only depend on host-written data (lazy)
    hipStream_t late = nullptr;                // separate stream for a group's small late passes (lazy), see late_stream()
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
    bool crew_member = false;                  // one thread of several proving side by side: no workers of its own
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
            static const bool high = getenv("GKR_LATE_HIGH_PRIORITY") != nullptr;
            hipError_t rc;
            if (high) {
                int lo = 0, hi = 0;
                (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // numerically lower = higher priority
                rc = hipStreamCreateWithPriority(&late, hipStreamNonBlocking, hi);
            } else {
                rc = hipStreamCreateWithFlags(&late, hipStreamNonBlocking);
            }
            if (rc != hipSuccess) return rc;
        }
        *out = late;
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
        std::vector<int> items;   // indices into the current call's item list, in proving order
    };
    std::vector<std::unique_ptr<Member>> members;
    std::mutex mu;
    std::condition_variable cv_start, cv_done;
    uint64_t generation = 0;
    bool stop = false;
    int active = 0;               // members taking part in the current call (the first `active`)
    int finished = 0;             // of the threads (members 1 ..), in the current call
    gkr_prove_item* items = nullptr;
    int32_t busy = 0;             // members still proving (atomic access); the others lend themselves
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
static thread_local ThreadTimeAccount t_account;
static inline double now_us_dbg() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static bool help_enabled() {
    static const bool on = getenv("GKR_NO_HELP") == nullptr;
    return on;
}
static void run_pieces(gkr::SpinPool* pool, const std::function<bool()>* work, bool several, int priority = 0) {
    static const bool flat = getenv("GKR_HELP_FLAT") != nullptr;   // A/B: every posted job alike
    auto run = [&] {
        if (pool)
            pool->run_now(work);
        else
            while ((*work)()) {
            }
    };
    static const bool dbg = getenv("GKR_DEBUG_TIMING") != nullptr;
    const double t0 = dbg ? now_us_dbg() : 0.0;
    if (several && help_enabled()) {
        gkr::HelpBoard::Posted posted(work, flat ? 0 : priority);
        run();
    } else {
        run();
    }
    if (dbg) t_account.own_pieces_us += now_us_dbg() - t0;
}

template <typename Rec>
int wait_records(gkr_ctx* ctx, const volatile Rec* recs, int count, uint32_t ticket) {
    const auto t0 = std::chrono::steady_clock::now();
    const bool help = help_enabled();
    static const int wait_mode = [] { const char* e = getenv("GKR_WAIT_MODE"); return e ? atoi(e) : 0; }();
    if (wait_mode == 1) {   // diagnostic: classic stream synchronisation instead of polling the records
        hipError_t q = hipStreamSynchronize(ctx->stream);
        if (q != hipSuccess) return ctx->hip_fail(q, "hipStreamSynchronize");
    }
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
            static const bool dbg = getenv("GKR_DEBUG_TIMING") != nullptr;
            const double th0 = dbg ? now_us_dbg() : 0.0;
            const bool did_help = help && gkr::HelpBoard::instance().help();
            if (!did_help) GKR_CPU_RELAX();
            if (dbg) (did_help ? t_account.helped_us : t_account.spin_us) += now_us_dbg() - th0;
            if (did_help ? (++helped & 0x3F) == 0 : (++spins & 0xFFFF) == 0) {
                hipError_t q = wait_mode == 2 ? hipErrorNotReady : hipStreamQuery(ctx->stream);
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
static int mle_pass_rounds(int m, int n, int jmax) {
    int j = m < jmax ? m : jmax;
    if (((size_t)1 << m) > gkr::kSmallPassEntries && m != n && j > m - 6) j = m - 6;
    if (jmax > 3 && (m - j == 10 || m - j == 11) && m - 12 >= 1) j = m - 12;
    return j < 1 ? 1 : j;
}

// ------------------------------------------------------------- plain MLE sumcheck, multi-round passes
// The host's share of one multi-round pass, scalar form (the IFMA-lane form is gkr_ifma_pass, mimc_ifma.cpp; same
// arguments, same results): per lane k and round t the round polynomial's coefficients from the sub-block sums, the
// vector's length, the challenge, then the sums with that variable bound; at the end the 2^J weights of the fold pass
// that binds the J variables, w_b = prod_t (bit_t(b) ? r_t : 1 - r_t), bit_0 = most significant, Montgomery form.
static void host_pass_scalar(const uint64_t* sums, size_t sums_row_words, int count, int J, const uint32_t* final_len,
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
static void host_prod_pass_scalar(const uint64_t* recs, size_t rec_row_words, int count, int J, const uint32_t (*vec_len)[16],
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

// Host transcript, default schedule (kernels.hip "Multi-round passes"): a pass hands the host the
// 2^J sub-block sums of the current table; the host runs J rounds on them (J <= 5 hashes in a row,
// eight or sixteen sumchecks per IFMA call), derives the 2^J fold weights, and the next pass binds all J
// variables at once.  Length rules as in run_mle_batch.
// `tail` (may be null): the tables are the TAIL of longer sumchecks (gkr_sumcheck_mle_sharded_dev: what is left of a table
// split over ranks, gathered) -- round j of a tail is round round_offset + j of a sumcheck with n_total rounds (row stride
// of the outputs), and the last round's length follows the ORIGINAL table's dependence on its last variable (dep_last,
// null: the tail is the whole table), not the folded tail's.
struct MleTailArgs {
    int n_total = 0, round_offset = 0;
    const uint32_t* dep_last = nullptr;
};
int run_mle_batch_passes(gkr_ctx* ctx, const Fr* d_tables, int n, int batch, gkr_fr* out_coeffs, uint32_t* out_len,
                         gkr_fr* out_r, const MleTailArgs* tail = nullptr) {
    using gkr::h64::F;
    const int n_out = tail ? tail->n_total : n, r_off = tail ? tail->round_offset : 0;
    static const bool dbg = getenv("GKR_DEBUG_TIMING") != nullptr;
    const auto dbg_t0 = std::chrono::steady_clock::now();
    auto dbg_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_t0).count(); };
    double dbg_a = 0, dbg_b = 0, dbg_c = 0, dbg_d = 0, dbg_e = 0;
    std::atomic<uint64_t> dbg_busy_ns{0};   // time inside process_chunk, all threads
    const size_t len = (size_t)1 << n;
    hipStream_t s = ctx->stream;
    // rounds per pass: up to 5 with the matrix-core fold (fewer passes, ~2.07 N elements moved instead of 2.29 N),
    // up to 3 with the v_mad_u64_u32 fold (GKR_NO_MFMA_FOLD)
    static const int jmax = [] {
        const int cap = getenv("GKR_NO_MFMA_FOLD") ? 3 : gkr::kMlePassMaxRounds;
        const char* e = getenv("GKR_ROUNDS_PER_PASS");
        const int v = e ? atoi(e) : cap;
        return v < 1 ? 1 : (v > cap ? cap : v);
    }();
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
    const bool ifma = host_ifma_ready();
    const bool scalar_book = getenv("GKR_HOST_PASS_SCALAR") != nullptr;   // A/B switch: host_pass_scalar even where the CPU has IFMA
    std::vector<uint32_t> dep_last(batch, 0);
    gkr::SpinPool* pool = ctx->host_pool();
    // Sumchecks a hashing thread takes at a time, per group: sixteen (full IFMA calls: throughput) when the group has plenty
    // for every thread; otherwise ONE chunk per thread where that fits the sixteen lanes -- a pass's J hashes of a sumcheck
    // are a serial chain, so a group of 128 on 14 threads is done in one chain of 16-lane calls filled to 10 (J x 20 us)
    // instead of two chains of 8-lane calls (2 x J x 16 us), at the same cost per hash; GKR_HASH_CHUNK forces 8 or 16
    const int hash_threads = pool->workers() + 1;
    auto group_chunk = [hash_threads](int nb) -> uint32_t {
        static const int forced = [] {
            const char* e = getenv("GKR_HASH_CHUNK");
            const int v = e ? atoi(e) : 0;
            return v == 8 || v == 16 ? v : 0;
        }();
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
    const double batch_bytes = (double)batch * (double)len * 32.0;
    int want_groups = (int)(batch_bytes / (4.0 * 1024 * 1024 * 1024));
    want_groups = want_groups < 4 ? 4 : (want_groups > 8 ? (batch_bytes > 96.0 * 1024 * 1024 * 1024 ? 16 : 8) : want_groups);
    // Small tables (BASELINE configs[1]: 4096 x 2^16) are bound by the host's hashing, not by the stream: sixteen groups
    // with pass 0 of four of them queued ahead keep the hashing threads fed from start to end (MI355X, 14 threads, ms per
    // 4096 x 2^16: 4 groups 8.1 - 8.2, 8 groups 8.1, 16 groups 7.2, 16 groups / depth 4 7.0 - 7.2, 32 groups / depth 8 7.1;
    // profiles/r03/f_n16_groups*.jsonl)
    const bool small_tables = n <= 17 && batch >= 256;
    if (small_tables) want_groups = 16;
    // A rank with two or three host threads (eight ranks on a 16-core host) is bound by its hashing: smaller groups shorten
    // the stretch before the first hashes and after the last fold (1024 x 2^20, two threads: 16.3 - 16.7 ms with eight
    // groups, 16.0 with sixteen; profiles/r03/w_two_host_threads_group_size.jsonl)
    if (hash_threads <= 3 && batch >= 256 && want_groups < 16) want_groups = 16;
    int group_size = batch >= 128 ? (batch + want_groups - 1) / want_groups : (batch >= 16 ? (batch + 1) / 2 : batch);
    if (const char* e = getenv("GKR_GROUP_SIZE")) group_size = atoi(e) > 0 ? atoi(e) : group_size;
    int groups = (batch + group_size - 1) / group_size;
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
    };
    // Shares of the batch per group, in per cent (GKR_GROUP_SPLIT="40,30,20,10"): the groups finish in order, and the
    // LAST one's latency-bound late passes (four host round trips with nothing left to overlap them) are the exposed
    // tail of the call -- a smaller last group has a shorter tail (fewer hash chunks per round trip).
    static const std::vector<int> split = [] {
        std::vector<int> v;
        if (const char* e = getenv("GKR_GROUP_SPLIT")) {
            int sum = 0;
            for (const char* p = e; *p;) {
                const int x = atoi(p);
                if (x > 0) {
                    v.push_back(x);
                    sum += x;
                }
                while (*p && *p != ',') ++p;
                if (*p == ',') ++p;
            }
            if (sum != 100 || v.size() > (size_t)kMaxGroups) v.clear();
        }
        return v;
    }();
    if (!split.empty() && batch >= 16 * (int)split.size()) groups = (int)split.size();
    std::vector<Group> grp(groups);
    HIP_TRY(ctx, ctx->aux_stream(groups));
    {
        int start = 0, acc = 0;
        for (int g = 0; g < groups; ++g) {
            grp[g].index = g;
            int end;
            if (!split.empty() && groups == (int)split.size()) {
                acc += split[g];
                end = g + 1 == groups ? batch : (int)((long long)batch * acc / 100);
            } else {
                end = (int)((long long)batch * (g + 1) / groups);
            }
            grp[g].b0 = start;
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
            Timed t(ctx, "mle_pass_small", (double)nb * len * 32.0, nullptr, true);
            gkr::launch_mle_multifold_small(0, d_tables + (size_t)b0 * len, len, nullptr, 0, (uint32_t)len, (uint32_t)G.j, nb,
                                            h_w + (size_t)b0 * gkr::kMleMaxSub, rec + b0, G.ticket, s);
            return;
        }
        const uint32_t nblk = gkr::mle_pass_blocks((uint32_t)len, (uint32_t)G.j, nb);
        gkr::MleSubPartial* part = partials + (size_t)b0 * gkr::kMaxBlocksPerTable;
        {
            Timed t(ctx, "mle_sub_sums", (double)nb * len * 32.0);
            gkr::launch_mle_sub_sums(d_tables + (size_t)b0 * len, len, (uint32_t)len, nb, nblk, part, s);
        }
        Timed t(ctx, "mle_sub_reduce", 0.0, nullptr, true);
        gkr::launch_mle_sub_reduce(part, nblk, (uint32_t)G.j, nb, rec + b0, G.ticket, s);
    };
    // a fold pass: bind the jin variables just hashed, produce the sums of the next jout rounds
    static const bool no_late = getenv("GKR_NO_LATE_STREAM") != nullptr;
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
        hipStream_t st = (!from_input && src_len <= ((size_t)1 << 16)) ? late : s;
        G.m -= jin;
        G.round0 += jin;
        G.j = rounds_for(G.m);
        G.ticket = ++ctx->ticket;
        const double bytes = (double)nb * ((double)src_len + (double)S) * 32.0;
        if (S <= gkr::kSmallPassEntries) {
            Timed t(ctx, "mle_pass_small", bytes, st, true);
            gkr::launch_mle_multifold_small(jin, src, src_stride, dst, work_len, (uint32_t)S, (uint32_t)G.j, nb,
                                            h_w + (size_t)b0 * gkr::kMleMaxSub, rec + b0, G.ticket, st);
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
            static const bool plan_inline = getenv("GKR_PLAN_MAIN") != nullptr;
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
        {
            // late passes run beside other groups' streaming passes: their elapsed time is not their own cost, so they
            // are booked under their own name and stay out of the streaming fold pass's bandwidth figure
            Timed t(ctx, st == s ? "mle_multifold" : "mle_multifold_late", bytes, st);
            gkr::launch_mle_multifold(jin, src, src_stride, dst, work_len, (uint32_t)S, nb, nblk, h_w + (size_t)b0 * gkr::kMleMaxSub,
                                      plan, part, st);
        }
        Timed t(ctx, "mle_sub_reduce", 0.0, st, true);
        gkr::launch_mle_sub_reduce(part, nblk, (uint32_t)G.j, nb, rec + b0, G.ticket, st);
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
    static const int depth_env = [] {
        const char* e = getenv("GKR_PASS_QUEUE_DEPTH");
        return e && atoi(e) > 0 ? atoi(e) : 0;
    }();
    const int depth = depth_env ? depth_env : (small_tables ? 4 : 2);
    int next_first = 0;   // groups [next_first, groups): pass 0 still to launch
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
        return rc;
    }
    dbg_d = dbg_us();
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (late != s) HIP_TRY(ctx, hipStreamSynchronize(late));
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
    static const bool per_round = getenv("GKR_MLE_PER_ROUND") != nullptr;
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
        if (const char* e = getenv("GKR_GROUP_SIZE")) group_size = atoi(e) > 0 ? atoi(e) : group_size;
        int stagger = 0;   // measured on MI355X + 16 host CPUs: starting every group at once is best
        if (const char* e = getenv("GKR_STAGGER")) stagger = atoi(e);
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
    static const bool use_atomics = getenv("GKR_PREDICATE_ATOMICS") != nullptr;
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
struct LayerShardArgs {
    uint64_t gate_base = 0, gate_count = 0;
    gkr_allreduce_fn allreduce = nullptr;   // host hook (field elements in host memory), or
    void* user = nullptr;
    const gkr_exchange_dev* dev = nullptr;  // device exchange: limbs widened into the caller's device buffer, summed on the stream
};

int run_layer_batch_impl(gkr_ctx* ctx, int batch, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r,
                         const gkr_fr* z, const Fr* d_W, gkr_fr* const* out_coeffs, uint32_t* const* out_len, gkr_fr* const* out_r,
                         const LayerShardArgs* shard, GateLists* cached);

// Gate lists that this call built (cached->ready false on entry) count as ready only if the whole call succeeded: a bad
// gate, a HIP error or a timeout after the sort was queued must not leave half-validated lists marked usable.
int run_layer_batch(gkr_ctx* ctx, int batch, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r,
                    const gkr_fr* z, const Fr* d_W, gkr_fr* const* out_coeffs, uint32_t* const* out_len, gkr_fr* const* out_r,
                    const LayerShardArgs* shard = nullptr, GateLists* cached = nullptr) {
    const bool was_ready = cached && cached->ready;
    int rc = GKR_OK;
    if (ctx->transcript != GKR_TRANSCRIPT_HOST && batch > 1 && !shard) {
        // The device transcript hashes on one lane per sumcheck and its round kernels take one proof: the proofs of a
        // batch go through one after the other (complete and host-free, not fast: ~1 ms per round and proof).
        const size_t wlen = (size_t)1 << k;
        for (int b = 0; b < batch && rc == GKR_OK; ++b)
            rc = run_layer_batch_impl(ctx, 1, k_i, k, d_gt, d_l, d_r, z + (size_t)b * k_i, d_W + (size_t)b * wlen, out_coeffs + b, out_len + b,
                                      out_r + b, nullptr, cached);
    } else {
        rc = run_layer_batch_impl(ctx, batch, k_i, k, d_gt, d_l, d_r, z, d_W, out_coeffs, out_len, out_r, shard, cached);
    }
    if (rc && cached && !was_ready) cached->ready = false;
    return rc;
}

int run_layer_batch_impl(gkr_ctx* ctx, int batch, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r,
                         const gkr_fr* z, const Fr* d_W, gkr_fr* const* out_coeffs, uint32_t* const* out_len, gkr_fr* const* out_r,
                         const LayerShardArgs* shard, GateLists* cached) {
    const size_t N = (size_t)1 << (2 * k);
    const size_t wlen = (size_t)1 << k;
    const uint32_t v = 2 * k;
    const double t_entry_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const bool host_tx = ctx->transcript == GKR_TRANSCRIPT_HOST;
    if (!host_tx && batch != 1) return ctx->fail(GKR_ERR_INVALID, "batched proving needs the host transcript");
    if (shard && (!host_tx || batch != 1)) return ctx->fail(GKR_ERR_INVALID, "a gate-sharded layer needs the host transcript and one proof");
    if (k > kMaxLayerK || k_i > kMaxLayerKi) return ctx->fail(GKR_ERR_INVALID, "layer wider than the library's limits (gkr_amd.h: GKR_MAX_K_NEXT, GKR_MAX_K_I)");
    if (!host_tx && k > kMaxDenseK)
        return ctx->fail(GKR_ERR_INVALID, "the device transcript works on dense 2^(2 k_next)-entry predicate tables: k_next <= 14 (GKR_MAX_K_NEXT_DEVICE_TRANSCRIPT)");
    const gkr::GateSpan span{shard ? shard->gate_base : 0, shard ? shard->gate_count : (uint64_t)1 << k_i};
    hipStream_t s = ctx->stream;
    Fr *A = nullptr, *M = nullptr, *Wb = nullptr, *Wc = nullptr, *d_coeffs = nullptr, *d_r_out = nullptr;
    gkr::FixedMul* d_rtab = nullptr;
    uint32_t *d_len = nullptr, *dep = nullptr;
    gkr::LayerPartial* partials = nullptr;
    // Three forms of the b-phase (all the same transcript):
    //   gate lists (default with the host transcript; the only form for k > 13): no 2^{2k}-entry tables at all -- U, V and the
    //     c-phase row are summed straight from the gates grouped by left / right operand (kernels.hip, k_gate_*);
    //   dense predicate tables, U, V and the row from two passes over them (layers with more than 2^{2k-2} gates;
    //     GKR_LAYER_DENSE_TABLES forces it, GKR_LAYER_GATE_LISTS forces the gate lists);
    //   GKR_LAYER_DENSE_B (and the device transcript): k passes over the dense tables.
    static const bool dense_b = getenv("GKR_LAYER_DENSE_B") != nullptr;
    static const bool dense_tables = getenv("GKR_LAYER_DENSE_TABLES") != nullptr;
    const bool lin_b = host_tx && !dense_b && k >= 1;
    // gate lists pay when the layer is sparse in its 2^{2k} cells (every circom layer is); for a layer with a gate in
    // (nearly) every cell the dense tables' counting sort is the cheaper grouping (k_i = 24, k = 12: 8.9 ms against 9.1)
    static const bool gate_lists_always = getenv("GKR_LAYER_GATE_LISTS") != nullptr;
    // Small layers (every layer of a circom-sized circuit) can run their whole sumcheck as ONE resident kernel, tables
    // in LDS, rounds handed over through pinned memory (kernels.hip, k_layer_persistent): GKR_LAYER_PERSISTENT=1.
    // Opt-in: measured on MI355X it saves the launch per round but a round stays at 70 - 80 us, because what
    // dominates is the host's 24 - 30 us hash call and ~20 us of PCIe latency per hand-off in either form (64 inputs
    // x 12 sub-circuits: 63 instead of 75 ms from one context, 19.6 instead of 22 ms from six) -- not enough to make a
    // kernel that waits on the host the default.
    static const bool want_persistent = getenv("GKR_LAYER_PERSISTENT") != nullptr;
    const bool persistent = lin_b && !shard && !dense_tables && want_persistent && !gate_lists_always && k <= (int)gkr::kPersistentMaxK &&
                            k_i <= k + 4;
    // gate lists also for dense layers when the block-private sort applies (k <= 12, >= 2^16 gates: 2^24 gates sort in
    // ~0.5 ms, against 2.7 ms for the dense tables' cell sort)
    const bool lds_sort = gkr::gate_lists_lds_blocks(span.count, (uint32_t)k) != 0;
    const bool sparse = shard || persistent || (lin_b && (k > 13 || (!dense_tables && (gate_lists_always || lds_sort || k_i + 2 <= 2 * k))));
    // Wide layers (2^13 buckets and more per half, each with a few gates): the gate passes run with a group of lanes per
    // bucket and the rare long buckets in units (kernels_wide.hip) -- a block per bucket would be 2^20 blocks for a gate apiece.
    // GKR_GATE_GROUPS_MIN_K moves the switch (tests run the form on small layers too).
    static const int wide_min_k = [] { const char* e = getenv("GKR_GATE_GROUPS_MIN_K"); return e ? atoi(e) : (int)gkr::kWideMinK; }();
    const bool wide = sparse && !persistent && k >= wide_min_k && gkr::gate_segs_words(span, (uint32_t)k_i, (uint32_t)k) == 0;
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
    Fr *U = nullptr, *V = nullptr, *d_eq = nullptr, *collapse = nullptr;
    gkr_fr* h_u = nullptr;   // pinned: u = (r_1 .. r_k) of every proof, from which the device builds eq(u, .)
    Fr *e_hi = nullptr, *e_lo = nullptr;
    uint32_t *g_offsets = nullptr, *g_cursor = nullptr, *g_list = nullptr, *g_heavy = nullptr;
    Fr* heavy_partials = nullptr;
    // where eq(z, g) is split into E_hi, E_lo: in the middle, or -- large layers, whose gate passes run over segments
    // of the sorted lists (gate_seg.h) -- where the segments are cut
    const uint32_t kl = gkr::gate_seg_shift(span, (uint32_t)k_i, (uint32_t)k);
    gkr::GateSegs local_segs;
    gkr::GateSegs* segs = cached ? &cached->segs : &local_segs;
    Fr* seg_partials = nullptr;
    if (lin_b) {
        WS(ctx, "layer.U", Fr, wlen * batch, U);
        WS(ctx, "layer.V", Fr, wlen * batch, V);
        WS(ctx, "layer.eq", Fr, wlen * batch, d_eq);
        if (!sparse)
            WS(ctx, "layer.collapse", Fr, (size_t)2 * batch * gkr::layer_collapse_chunks((uint32_t)k, (uint32_t)batch) * wlen, collapse);
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
            gkr::launch_layer_prologue(reinterpret_cast<const Fr*>(hz), (uint32_t)k_i, (uint32_t)kh, kl, e_hi, e_lo, d_W, persistent ? nullptr : Wb,
                                       persistent ? nullptr : Wc, (uint32_t)k, dep_wide ? nullptr : dep, h_dep, (uint32_t)batch, s);
            if (dep_wide) {
                uint32_t* dep_bits = nullptr;
                WS(ctx, "layer.depbits", uint32_t, (size_t)batch, dep_bits);
                gkr::launch_depends_wide(d_W, (uint32_t)k, dep_bits, dep, h_dep, (uint32_t)batch, s);
            }
        }
        if (wide) WS(ctx, "gates.heavypart", Fr, gkr::gate_heavy_partial_elems(span.count, (uint32_t)k) * batch, heavy_partials);
        if (!(cached && cached->ready)) HIP_TRY(ctx, hipMemsetAsync(bad, 0, 4, s));   // (only the list build writes it)
        if (const size_t pe = gkr::gate_seg_partial_elems(span, (uint32_t)k_i, (uint32_t)k)) WS(ctx, "gates.segpart", Fr, pe * batch, seg_partials);
        if (cached && cached->ready) {
            lists_fresh = false;
            g_offsets = cached->offsets;   // the circuit's lists from an earlier call (validated then)
            g_cursor = cached->cursor;
            g_list = cached->list;
            g_heavy = cached->heavy;
            if (wide && !g_heavy) return ctx->fail(GKR_ERR_INVALID, "cached gate lists were built without the wide layer's work lists");
        } else {
            WS(ctx, "gates.counts", uint32_t, nb2, g_counts);
            WS(ctx, "gates.bsums", uint32_t, (nb2 + 2047) / 2048 + 1, g_bsums);
            if (cached) {
                if (!cached->offsets) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&cached->offsets), nb2 * sizeof(uint32_t)));
                if (!cached->cursor) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&cached->cursor), nb2 * sizeof(uint32_t)));
                if (!cached->list) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&cached->list), 2 * gkr::gate_list_words(span.count) * sizeof(uint32_t)));
                if (wide && !cached->heavy)
                    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&cached->heavy), gkr::gate_heavy_words(span.count, (uint32_t)k) * sizeof(uint32_t)));
                g_offsets = cached->offsets;
                g_cursor = cached->cursor;
                g_list = cached->list;
                g_heavy = cached->heavy;
            } else {
                WS(ctx, "gates.offsets", uint32_t, nb2, g_offsets);
                WS(ctx, "gates.cursor", uint32_t, nb2, g_cursor);
                WS(ctx, "gates.list", uint32_t, 2 * gkr::gate_list_words(span.count), g_list);
                if (wide) WS(ctx, "gates.heavy", uint32_t, gkr::gate_heavy_words(span.count, (uint32_t)k), g_heavy);
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
            if (wide) gkr::launch_gate_heavy_lists(span, (uint32_t)k, g_offsets, g_cursor, g_heavy, s);
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
    if (persistent) {
        // U, V, the rounds and the row are all inside the one kernel launched below
    } else if (sparse) {
        Timed t(ctx, "gate_uv", (double)span.count * 8.0 * batch);   // HBM: the 8-byte list entry per gate (operands are L2 gathers)
        if (wide)
            gkr::launch_gate_uv_wide(span, (uint32_t)k_i, (uint32_t)k, g_offsets, g_cursor, g_list, e_hi, e_lo, kl, Wc, U, V, lb, g_heavy, heavy_partials, s);
        else
            gkr::launch_gate_uv(span, (uint32_t)k_i, (uint32_t)k, g_offsets, g_cursor, g_list, e_hi, e_lo, kl, Wc, U, V, lb, segs, seg_partials, s);
    } else if (lin_b) {
        Timed t(ctx, "layer_uv", (double)N * 2.0 * 32.0 * batch);
        gkr::launch_layer_uv(A, M, Wc, U, V, (uint32_t)k, lb, s);
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
        if (shard->dev->capacity < gkr_exchange_limbs(k) || !shard->dev->d_limbs || !shard->dev->fn)
            return ctx->fail(GKR_ERR_INVALID, "the exchange buffer is smaller than gkr_exchange_limbs(k_next) int64");
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

    gkr::LayerHostRec* rec = nullptr;
    gkr::FixedMul* h_rtab = nullptr;   // pinned, two slots of `batch` tables used alternately
    const gkr::h64::F* cts64 = host_mimc_constants64();
    const bool ifma = host_ifma_ready();
    gkr::SpinPool* pool = nullptr;
    if (host_tx) {
        HIP_TRY(ctx, ctx->pinned_host("layer.rec", sizeof(gkr::LayerHostRec) * batch, reinterpret_cast<void**>(&rec)));
        HIP_TRY(ctx, ctx->pinned_host("layer.rtab", 2 * sizeof(gkr::FixedMul) * batch, reinterpret_cast<void**>(&h_rtab)));
        // h_dep is read when round 0 is hashed, i.e. after a LATER kernel of this stream has released that round's record
        // (the gate-list form's prologue launch wrote it)
        if (!sparse) gkr::launch_copy_words(dep, h_dep, (size_t)32 * batch, s);
        // (gate-sharded with the device exchange: the flag travels with the first exchange and is looked at after the
        // first round's record, on every rank alike -- a rank that left here would leave its peers inside a collective)
        if (sparse && lists_fresh && !(shard && shard->dev)) {   // lists found in the circuit cache were validated when they were built
            uint32_t hbad = 0;
            HIP_TRY(ctx, hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
            if (hbad) return ctx->fail(GKR_ERR_INVALID, "gate type or operand index out of range");
        }
        if (batch >= 16) pool = ctx->host_pool();
    }
    if (persistent) {
        static const bool dbg_s = getenv("GKR_DEBUG_TIMING") != nullptr;
        if (dbg_s)
            fprintf(stderr, "[gkr timing] resident layer set-up (eq upload, gate lists, dep readback): %.0f us\n",
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_entry_us);
        gkr::LayerChallenge* chal = nullptr;
        uint32_t* abort_flag = nullptr;
        HIP_TRY(ctx, ctx->pinned_host("layer.chal", sizeof(gkr::LayerChallenge) * batch, reinterpret_cast<void**>(&chal)));
        HIP_TRY(ctx, ctx->pinned_host("layer.abort", 64, reinterpret_cast<void**>(&abort_flag)));
        __atomic_store_n(abort_flag, 0u, __ATOMIC_RELEASE);
        const uint32_t base = ctx->ticket + 1;
        ctx->ticket += v;
        static const bool dbg_p = getenv("GKR_DEBUG_TIMING") != nullptr;
        const auto tp0 = std::chrono::steady_clock::now();
        {
            Timed t(ctx, "layer_persistent", 0.0);
            gkr::launch_layer_persistent(span, (uint32_t)k_i, (uint32_t)k, g_offsets, g_cursor, g_list, d_gt, d_l, d_r, e_hi, e_lo, kl, d_W,
                                         rec, chal, abort_flag, base, lb, s);
        }
        HIP_TRY(ctx, hipGetLastError());
        struct AbortGuard {   // whatever path leaves this scope early, the resident kernel is told to stop waiting
            uint32_t* flag;
            bool armed = true;
            ~AbortGuard() {
                if (armed) __atomic_store_n(flag, 1u, __ATOMIC_RELEASE);
            }
        } guard{abort_flag};
        std::atomic<int> slice_rc{GKR_OK};
        // a slice of proofs, each advanced independently: whichever records have landed are answered (hashed up to
        // sixteen at a time), so a proof never waits for another one's block to become resident
        static const bool dbg_r = getenv("GKR_DEBUG_TIMING") != nullptr;
        auto run_slice = [&](int first, int count) {
            std::vector<uint32_t> round(count, 0);
            int done = 0;
            double us_gather = 0, us_hash = 0, us_write = 0, us_wait = 0;
            long calls = 0, lanes = 0;
            auto t_mark = std::chrono::steady_clock::now();
            auto lap_us = [&](double& bucket) {
                const auto t = std::chrono::steady_clock::now();
                bucket += std::chrono::duration<double, std::micro>(t - t_mark).count();
                t_mark = t;
            };
            auto last_progress = std::chrono::steady_clock::now();
            uint32_t idle = 0;
            while (done < count && slice_rc.load(std::memory_order_relaxed) == GKR_OK) {
                int idx[kHashChunkMax], nr = 0;
                auto scan = [&] {
                    nr = 0;
                    for (int i = 0; i < count && nr < kHashChunkMax; ++i)
                        if (round[i] < v && __atomic_load_n(&rec[first + i].seq, __ATOMIC_ACQUIRE) == base + round[i]) idx[nr++] = i;
                };
                scan();
                if (nr > 0 && dbg_r) lap_us(us_wait);
                // A hash call costs the same for one lane as for sixteen (one serial chain of ~1100 products either
                // way: 24 - 30 us), so answering a few records now and the rest in a second call doubles every proof's
                // round time -- and the two cohorts then stay out of phase for the rest of the layer.  So a record waits
                // for the slice's other unfinished proofs, up to about two hash calls' time (blocks that are not resident
                // yet must not hold the others up for ever).
                if (nr > 0 && nr < (count - done < kHashChunkMax ? count - done : kHashChunkMax)) {
                    const int want = count - done < kHashChunkMax ? count - done : kHashChunkMax;
                    const auto t_gather = std::chrono::steady_clock::now();
                    while (nr < want && std::chrono::steady_clock::now() - t_gather < std::chrono::microseconds(50)) {
                        GKR_CPU_RELAX();
                        scan();
                    }
                }
                if (nr == 0) {
                    GKR_CPU_RELAX();
                    // No HIP call in here: another context's thread may sit inside the runtime waiting for work that is
                    // queued BEHIND this context's resident kernel (streams share hardware queues), and a runtime lock
                    // taken by this loop would then wait for a kernel that waits for this loop.  Only the clock.
                    if ((++idle & 0x3FFF) == 0 && std::chrono::steady_clock::now() - last_progress > std::chrono::seconds(30))
                        slice_rc.store(-2);   // no record for 30 s
                    continue;
                }
                idle = 0;
                last_progress = std::chrono::steady_clock::now();
                if (dbg_r) {
                    lap_us(us_gather);
                    ++calls;
                    lanes += nr;
                }
                gkr::h64::F c0[kHashChunkMax], lin[kHashChunkMax], c2[kHashChunkMax], r[kHashChunkMax];
                uint32_t ln[kHashChunkMax] = {};
                for (int j = 0; j < nr; ++j) {
                    const int b = first + idx[j];
                    gkr::h64::F g1;
                    memcpy(&c0[j], &rec[b].c0, 32);
                    memcpy(&g1, &rec[b].g1, 32);
                    memcpy(&c2[j], &rec[b].c2, 32);
                    lin[j] = gkr::h64::sub(gkr::h64::sub(g1, c0[j]), c2[j]);
                    ln[j] = 2u + (h_dep[(size_t)b * 32 + round[idx[j]] % k] ? 1u : 0u);
                }
                if (ifma && nr >= 3) {
                    uint64_t vec[kHashChunkMax][3][4], out[kHashChunkMax][4];
                    memset(vec, 0, sizeof vec);
                    for (int j = 0; j < nr; ++j) {
                        memcpy(vec[j][0], &c2[j], 32);
                        memcpy(vec[j][1], &lin[j], 32);
                        memcpy(vec[j][2], &c0[j], 32);
                    }
                    ifma_hash_chunk(vec, ln, nr, out);
                    for (int j = 0; j < nr; ++j) memcpy(&r[j], out[j], 32);
                } else {
                    for (int j = 0; j < nr; ++j) {
                        gkr::h64::F vec[3] = {c2[j], lin[j], c0[j]};
                        r[j] = host_multi_hash(vec + (3 - ln[j]), (int)ln[j], cts64);
                    }
                }
                if (dbg_r) lap_us(us_hash);
                for (int j = 0; j < nr; ++j) {
                    const int i = idx[j], b = first + i;
                    const uint32_t rd = round[i];
                    const gkr::h64::F rm = gkr::h64::to_mont(r[j]);
                    memcpy(&chal[b].r_mont, &rm, 32);
                    __atomic_store_n(&chal[b].seq, base + rd, __ATOMIC_RELEASE);   // the device folds while the host writes out
                    gkr_fr* oc = out_coeffs[b] + (size_t)rd * 3;
                    memset(&oc[0], 0, 32);
                    if (ln[j] == 3) memcpy(&oc[0], &c2[j], 32);
                    memcpy(&oc[1], &lin[j], 32);
                    memcpy(&oc[2], &c0[j], 32);
                    out_len[b][rd] = ln[j];
                    memcpy(&out_r[b][rd], &r[j], 32);
                    if (++round[i] == v) ++done;
                }
                if (dbg_r) lap_us(us_write);
            }
            if (dbg_r)
                fprintf(stderr, "[gkr timing] slice of %d proofs: %ld hash calls, %.1f lanes each; per call: waiting %.1f us, gathering %.1f, hashing %.1f, writing %.1f\n",
                        count, calls, calls ? (double)lanes / calls : 0.0, calls ? us_wait / calls : 0.0, calls ? us_gather / calls : 0.0,
                        calls ? us_hash / calls : 0.0, calls ? us_write / calls : 0.0);
        };
        if (pool) {
            const int want = (batch + 15) / 16, most = pool->workers() + 1;
            const int slices = want < most ? want : most;
            std::atomic<int> next{0};
            const std::function<bool()> work = [&]() -> bool {
                const int sidx = next.fetch_add(1, std::memory_order_relaxed);
                if (sidx >= slices) return false;
                const int f = (int)((long long)batch * sidx / slices), e = (int)((long long)batch * (sidx + 1) / slices);
                run_slice(f, e - f);
                return true;
            };
            gkr::SpinPool::Session session(pool, nullptr);
            pool->run_now(&work);
        } else {
            run_slice(0, batch);
        }
        if (const int src = slice_rc.load()) {
            __atomic_store_n(abort_flag, 1u, __ATOMIC_RELEASE);
            (void)hipStreamSynchronize(s);
            (void)src;
            return ctx->fail(GKR_ERR_HIP, "timed out waiting for the resident layer kernel to publish a round");
        }
        guard.armed = false;
        const auto tp1 = std::chrono::steady_clock::now();
        HIP_TRY(ctx, hipStreamSynchronize(s));
        ctx->drain_events();
        if (dbg_p) {
            const auto tp2 = std::chrono::steady_clock::now();
            fprintf(stderr, "[gkr timing] resident layer k_i=%d k=%d batch=%d: launch + %u rounds %.0f us, final sync %.0f us\n", k_i, k, batch, v,
                    std::chrono::duration<double, std::micro>(tp1 - tp0).count(), std::chrono::duration<double, std::micro>(tp2 - tp1).count());
        }
        return GKR_OK;
    }
    // round vectors of up to eight proofs: g = [c2, c1, c0] with c1 = g(1) - c0 - c2, length 2 + dep
    // (get_univariate_coeff, poly.rs:388-420), hashed together (eight-lane IFMA where available)
    auto hash_chunk = [&](int first, int count, uint32_t round, gkr::FixedMul* slot) {
        gkr::h64::F c0[kHashChunkMax], lin[kHashChunkMax], c2[kHashChunkMax], r[kHashChunkMax];
        uint32_t ln[kHashChunkMax] = {};
        for (int i = 0; i < count; ++i) {
            const int b = first + i;
            gkr::h64::F g1;
            memcpy(&c0[i], &rec[b].c0, 32);
            memcpy(&g1, &rec[b].g1, 32);
            memcpy(&c2[i], &rec[b].c2, 32);
            lin[i] = gkr::h64::sub(gkr::h64::sub(g1, c0[i]), c2[i]);
            ln[i] = 2u + (h_dep[(size_t)b * 32 + round % k] ? 1u : 0u);
        }
        if (ifma && count >= 3) {
            uint64_t vec[kHashChunkMax][3][4], out[kHashChunkMax][4];
            memset(vec, 0, sizeof vec);
            for (int i = 0; i < count; ++i) {
                memcpy(vec[i][0], &c2[i], 32);
                memcpy(vec[i][1], &lin[i], 32);
                memcpy(vec[i][2], &c0[i], 32);
            }
            ifma_hash_chunk(vec, ln, count, out);
            for (int i = 0; i < count; ++i) memcpy(&r[i], out[i], 32);
        } else {
            for (int i = 0; i < count; ++i) {
                gkr::h64::F vec[3] = {c2[i], lin[i], c0[i]};
                r[i] = host_multi_hash(vec + (3 - ln[i]), (int)ln[i], cts64);
            }
        }
        for (int i = 0; i < count; ++i) {
            const int b = first + i;
            gkr_fr* oc = out_coeffs[b] + (size_t)round * 3;
            memset(&oc[0], 0, 32);
            if (ln[i] == 3) memcpy(&oc[0], &c2[i], 32);
            memcpy(&oc[1], &lin[i], 32);
            memcpy(&oc[2], &c0[i], 32);
            out_len[b][round] = ln[i];
            memcpy(&out_r[b][round], &r[i], 32);
            gkr::h64::make_fixed_mul(r[i], slot[b].w);
        }
    };
    // Product passes (kernels.hip): both phases as sumchecks of W X + Y over three small tables, up to three rounds per
    // device round trip.  The default for the linear-time form over gate lists; GKR_LAYER_PER_ROUND=1: one round per trip.
    static const bool per_round = getenv("GKR_LAYER_PER_ROUND") != nullptr;
    if (lin_b && sparse && (k > 13 || !per_round)) {   // (the per-round kernels' c-phase keeps a row in one block: k <= 13)
        gkr::ProdPassRec* prec = nullptr;
        Fr *h_pw = nullptr, *d_ppart = nullptr, *Xc = nullptr, *Yc = nullptr;
        const uint32_t max_blocks = gkr::prod_pass_max_blocks((uint32_t)k);
        HIP_TRY(ctx, ctx->pinned_host("layer.prec", sizeof(gkr::ProdPassRec) * batch, reinterpret_cast<void**>(&prec)));
        HIP_TRY(ctx, ctx->pinned_host("layer.pw", sizeof(Fr) * 8 * batch, reinterpret_cast<void**>(&h_pw)));
        WS(ctx, "layer.ppart", Fr, (size_t)batch * gkr::prod_pass_scratch_values((uint32_t)k), d_ppart);
        (void)max_blocks;
        WS(ctx, "layer.X", Fr, wlen * batch, Xc);
        WS(ctx, "layer.Y", Fr, wlen * batch, Yc);
        gkr::SpinPool::Session session(pool, nullptr);
        uint32_t round0 = 0, jp = 0;
        bool second_exchange_done = false;
        for (int phase = 0; phase < 2 && rc == GKR_OK; ++phase) {
            Fr *Tw = Wb, *Tx = U, *Ty = V;
            if (phase == 1) {
                // all of b is bound: the rows of a, m at u = (r_1 .. r_k), then the c-phase's tables X = a_u + W(u) m_u,
                // Y = W(u) a_u (W(u): the last b pass's fold of what is left of Wb)
                for (int b = 0; b < batch; ++b) memcpy(h_u + (size_t)b * k, out_r[b], sizeof(gkr_fr) * k);
                gkr::launch_eq_table(reinterpret_cast<const Fr*>(h_u), (uint32_t)k, 0u, (uint32_t)k, d_eq, true, (uint32_t)batch, s);
                bool c_tables_done = false;   // (one rank holds all gates and the rows come from segments: the row pass writes X, Y too)
                {
                    Timed t(ctx, "gate_rows", (double)span.count * 8.0 * batch);
                    const gkr::CPhaseFuse fuse{Wb, h_pw, Xc, Yc, jp};
                    if (wide)
                        gkr::launch_gate_rows_wide(span, (uint32_t)k_i, (uint32_t)k, g_offsets, g_cursor, g_list, e_hi, e_lo, kl, d_eq, A, M, lb, g_heavy,
                                                   heavy_partials, s);
                    else
                        c_tables_done = gkr::launch_gate_rows(span, (uint32_t)k_i, (uint32_t)k, g_offsets, g_cursor, g_list, e_hi, e_lo, kl, d_eq, A, M, lb,
                                                              segs, seg_partials, s, shard ? nullptr : &fuse);
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
            uint32_t m = (uint32_t)k;   // log2 of the tables' length before the pending fold
            for (uint32_t rem = (uint32_t)k; rem > 0 && rc == GKR_OK;) {
                const uint32_t J = rem < (uint32_t)gkr::kProdMaxJ ? rem : (uint32_t)gkr::kProdMaxJ;
                const uint32_t ticket = ++ctx->ticket;
                {
                    Timed t(ctx, "layer_prod_pass", 0.0);
                    gkr::launch_prod_pass(Tw, Tx, Ty, m, jp, h_pw, J, d_ppart, (uint32_t)wlen, prec, ticket, (uint32_t)batch, s);
                }
                if (hipError_t le = hipGetLastError(); le != hipSuccess) {
                    rc = ctx->hip_fail(le, "launch of a layer pass");
                    break;
                }
                m -= jp;
                rc = wait_records(ctx, prec, batch, ticket);
                if (!rc) rc = xflag_check();
                if (rc) break;
                const int chunk = hash_chunk_size(batch, pool ? pool->workers() + 1 : 1);
                std::atomic<int> next{0};
                const std::function<bool()> work = [&]() -> bool {
                    const int first = next.fetch_add(chunk, std::memory_order_relaxed);
                    if (first >= batch) return false;
                    const int cnt = batch - first < chunk ? batch - first : chunk;
                    uint64_t c2[gkr::kProdMaxJ][16][4], lin[gkr::kProdMaxJ][16][4], c0[gkr::kProdMaxJ][16][4], rr[gkr::kProdMaxJ][16][4];
                    uint32_t vl[gkr::kProdMaxJ][16];
                    for (uint32_t t = 0; t < J; ++t)
                        for (int i = 0; i < cnt; ++i) vl[t][i] = 2u + (h_dep[(size_t)(first + i) * 32 + (round0 + t) % k] ? 1u : 0u);
                    (ifma && cnt >= 3 ? gkr::gkr_ifma_prod_pass : host_prod_pass_scalar)(
                        reinterpret_cast<const uint64_t*>(prec + first), sizeof(gkr::ProdPassRec) / 8, cnt, (int)J, vl, c2, lin, c0, rr,
                        reinterpret_cast<uint64_t*>(h_pw + (size_t)first * 8), 32);
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
                    return true;
                };
                run_pieces(pool, &work, batch > chunk, ctx->rounds_ahead + (int)(v - round0));
                jp = J;
                round0 += J;
                rem -= J;
            }
        }
        session.close();
        // a rank that failed between the exchanges still enters the second one (flag set): its peers are waiting in it.
        // (Not when the failure is the travelling flag itself: then every rank is leaving at this very point.)
        if (rc && shard && !second_exchange_done && !(h_xflag && __atomic_load_n(h_xflag, __ATOMIC_ACQUIRE)))
            (void)sum_over_ranks(A, M, wlen, nullptr, rc);
        if (rc) {
            (void)hipStreamSynchronize(s);
            return rc;
        }
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(s));
        rc = xflag_check();
        if (rc) return rc;
        ctx->drain_events();
        return GKR_OK;
    }
    // The fold with r_j is deferred into the pass that computes round j+1's sums (b-phase: the
    // fused kernel; c-phase: a separate fold of the single remaining row).
    const gkr::FixedMul* pending = nullptr;   // challenge tables not yet applied to A, M
    static const bool no_fused = getenv("GKR_LAYER_NO_FUSED") != nullptr;
    static const bool dbg = getenv("GKR_DEBUG_TIMING") != nullptr;
    double t_launch = 0, t_wait = 0, t_hash = 0, worst_lap = 0;
    int worst_round = -1, worst_kind = 0, cur_round = 0;
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now_us();
    auto lap = [&](double& bucket) {
        const double t = now_us();
        bucket += t - t_mark;
        if (t - t_mark > worst_lap) {
            worst_lap = t - t_mark;
            worst_round = cur_round;
            worst_kind = &bucket == &t_launch ? 0 : (&bucket == &t_wait ? 1 : 2);
        }
        t_mark = t;
    };
    gkr::SpinPool::Session session(pool, nullptr);   // closed on every path out of the round loop
    bool second_exchange_done = false;
    for (uint32_t round = 0; round < v; ++round) {
        cur_round = (int)round;
        const uint32_t h = (uint32_t)(N >> (round + 1));   // half of the table this round sums over
        const uint32_t phase = round < (uint32_t)k ? 0u : 1u;
        const uint32_t hb = phase == 0 ? (h >> k) : 0u;
        uint32_t nblk = 0;
        bool published = false;   // the round's kernel wrote the host record itself
        if (lin_b && round == (uint32_t)k) {
            // all of b is bound: collapse the rows of A, M at u = (r_1 .. r_k) into the single row the c-phase works on
            for (int b = 0; b < batch; ++b) memcpy(h_u + (size_t)b * k, out_r[b], sizeof(gkr_fr) * k);
            gkr::launch_eq_table(reinterpret_cast<const Fr*>(h_u), (uint32_t)k, 0u, (uint32_t)k, d_eq, true, (uint32_t)batch, s);
            if (sparse) {
                {
                    Timed t(ctx, "gate_rows", (double)span.count * 8.0 * batch);
                    if (wide)
                        gkr::launch_gate_rows_wide(span, (uint32_t)k_i, (uint32_t)k, g_offsets, g_cursor, g_list, e_hi, e_lo, kl, d_eq, A, M, lb, g_heavy,
                                                   heavy_partials, s);
                    else
                        gkr::launch_gate_rows(span, (uint32_t)k_i, (uint32_t)k, g_offsets, g_cursor, g_list, e_hi, e_lo, kl, d_eq,
                                              A, M, lb, segs, seg_partials, s);
                }
                if (shard) {
                    second_exchange_done = true;
                    rc = sum_over_ranks(A, M, wlen, nullptr, GKR_OK);
                    if (rc) break;
                }
            } else {
                Timed t(ctx, "layer_collapse", (double)N * 2.0 * 32.0 * batch);
                gkr::launch_layer_collapse(A, M, d_eq, collapse, (uint32_t)k, lb, s);
            }
            pending = nullptr;   // U, V are done with; the row is already taken at u
        }
        if (phase == 0 && lin_b) {
            const uint32_t ticket = ++ctx->ticket;
            {
                Timed t(ctx, "layer_uv_round", 0.0);
                gkr::launch_uv_round(pending != nullptr, Wb, U, V, 1u << (k - 1 - (int)round), pending, rec, ticket, lb, s);
            }
            pending = nullptr;
            published = true;
            lap(t_launch);
            rc = wait_records(ctx, rec, batch, ticket);
            if (!rc) rc = xflag_check();
            lap(t_wait);
            if (rc) break;
        } else if (phase == 1 && lin_b && k <= 13) {
            // the single remaining row: one small block per proof folds it and publishes the round's sums
            const uint32_t ticket = ++ctx->ticket;
            {
                Timed t(ctx, "layer_c_round", 0.0);
                gkr::launch_c_round(pending != nullptr, A, M, Wc, Wb, h, pending, rec, ticket, lb, s);
            }
            pending = nullptr;
            published = true;
            lap(t_launch);
            rc = wait_records(ctx, rec, batch, ticket);
            lap(t_wait);
            if (rc) break;
        } else if (phase == 0 && !no_fused) {
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
        if (host_tx) {
            if (!published) {
                const uint32_t ticket = ++ctx->ticket;
                {
                    Timed t(ctx, "layer_round_reduce", 0.0);
                    gkr::launch_layer_round_reduce(partials, nblk, rec, ticket, lb, s);
                }
                lap(t_launch);
                rc = wait_records(ctx, rec, batch, ticket);
                lap(t_wait);
                if (rc) break;
            }
            // two sets of pinned tables used alternately: the deferred fold of round j reads set j % 2
            // while the host already writes round j+1's
            gkr::FixedMul* slot = h_rtab + (size_t)(round & 1) * batch;
            const int chunk = hash_chunk_size(batch, pool ? pool->workers() + 1 : 1);
            {
                std::atomic<int> next{0};
                const std::function<bool()> work = [&]() -> bool {
                    const int first = next.fetch_add(chunk, std::memory_order_relaxed);
                    if (first >= batch) return false;
                    hash_chunk(first, batch - first < chunk ? batch - first : chunk, round, slot);
                    return true;
                };
                run_pieces(pool, &work, batch > chunk, ctx->rounds_ahead + (int)(v - round));
            }
            lap(t_hash);
            // fold the W copy bound in this round (rounds 0..k-1 bind b -> Wb, then c -> Wc); the linear-time
            // round kernels fold W themselves with the pending challenge, except for the last b round, whose
            // fold leaves the scalar W(u) the c-phase multiplies with
            const bool fused_w = lin_b && (phase == 0 ? round + 1 < (uint32_t)k : k <= 13);
            if (!fused_w) gkr::launch_fold_small(phase == 0 ? Wb : Wc, 1u << (k - 1 - (round % k)), slot, lb, s);
            pending = slot;
        } else {
            Timed t(ctx, "layer_round_hash", 0.0);
            gkr::launch_layer_round_hash(partials, nblk, round, k, dep, ctx->d_cts, d_coeffs, d_len, d_r_out, d_rtab, Wb,
                                         Wc, s);
            pending = d_rtab + round;
        }
    }
    session.close();
    if (rc && shard && lin_b && !second_exchange_done && !(h_xflag && __atomic_load_n(h_xflag, __ATOMIC_ACQUIRE)))
        (void)sum_over_ranks(A, M, wlen, nullptr, rc);   // the peers are waiting in the second exchange
    if (dbg)
        fprintf(stderr, "[gkr timing] layer k=%d batch=%d: %u rounds, launch %.0f us, wait %.0f us, hash %.0f us; longest single step %.0f us (%s, round %d)\n",
                k, batch, v, t_launch, t_wait, t_hash, worst_lap, worst_kind == 0 ? "launch" : (worst_kind == 1 ? "wait" : "hash"), worst_round);
    if (rc) {
        (void)hipStreamSynchronize(s);
        return rc;
    }
    HIP_TRY(ctx, hipGetLastError());
    if (!host_tx) {
        HIP_TRY(ctx, hipMemcpyAsync(out_coeffs[0], d_coeffs, (size_t)v * 3 * sizeof(Fr), hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipMemcpyAsync(out_len[0], d_len, v * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipMemcpyAsync(out_r[0], d_r_out, v * sizeof(Fr), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (const int xr = xflag_check()) return xr;
    ctx->drain_events();
    return GKR_OK;
}

int run_layer(gkr_ctx* ctx, int k_i, int k, const uint8_t* d_gt, const uint32_t* d_l, const uint32_t* d_r, const gkr_fr* z,
              const Fr* d_W, gkr_fr* out_coeffs, uint32_t* out_len, gkr_fr* out_r) {
    return run_layer_batch(ctx, 1, k_i, k, d_gt, d_l, d_r, z, d_W, &out_coeffs, &out_len, &out_r);
}

// evaluation table -> monomial coefficients, variable 1 = most significant bit
// (what get_multi_ext stores, poly.rs:502-536); host, 4x64-bit arithmetic
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
        case GKR_ERR_UNSUPPORTED: return "R1CS shape the reference's compiler does not support";
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

// One host process driving several GPUs -- the reference is ONE process whose par_iter fans prover::prove out over the
// (circuit, input) pairs of a step (aggregator.rs:350-355, 411-416): the context lives on device_ids[0], and
// gkr_prove_many deals its items over child contexts created round-robin on ALL the listed devices (a device may be
// listed more than once: that many child contexts on it per round).  Every other entry point runs on device_ids[0].
int gkr_ctx_create_multi(const int* device_ids, int n_devices, gkr_ctx** out) {
    if (!out) return GKR_ERR_INVALID;
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return GKR_ERR_INVALID;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return GKR_ERR_NO_DEVICE;
    for (int i = 0; i < n_devices; ++i)
        if (device_ids[i] < 0 || device_ids[i] >= count) return GKR_ERR_INVALID;
    gkr_ctx* c = nullptr;
    const int rc = gkr_ctx_create(device_ids[0], &c);
    if (rc) return rc;
    c->devices.assign(device_ids, device_ids + n_devices);
    *out = c;
    return GKR_OK;
}

int gkr_ctx_device_count(const gkr_ctx* ctx) { return ctx ? (ctx->devices.empty() ? 1 : (int)ctx->devices.size()) : 0; }

void gkr_ctx_destroy(gkr_ctx* ctx) {
    if (!ctx) return;
    ctx->crew.reset();   // joins gkr_prove_many's threads and destroys their contexts
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux) (void)hipStreamSynchronize(ctx->aux);
    if (ctx->late) (void)hipStreamSynchronize(ctx->late);
    ctx->drain_events();
    ctx->drain_events();
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    ctx->release_buffers();
    if (ctx->d_cts) (void)hipFree(ctx->d_cts);
    for (hipEvent_t e : ctx->aux_events) (void)hipEventDestroy(e);
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    if (ctx->late) (void)hipStreamDestroy(ctx->late);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* gkr_last_error(const gkr_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int gkr_ctx_set_transcript(gkr_ctx* ctx, int mode) {
    if (!ctx || (mode != GKR_TRANSCRIPT_DEVICE && mode != GKR_TRANSCRIPT_HOST)) return GKR_ERR_INVALID;
    ctx->transcript = mode;
    return GKR_OK;
}

int gkr_ctx_set_host_threads(gkr_ctx* ctx, int threads) {
    if (!ctx || threads < 0 || threads > 256) return GKR_ERR_INVALID;
    if (threads != ctx->host_threads) {
        ctx->pool.reset();   // joins the workers; the next call that needs them starts the new number
        ctx->host_threads = threads;
    }
    return GKR_OK;
}

long gkr_host_help_while(const volatile int32_t* busy) {
    if (!busy) return 0;
    long pieces = 0;
    unsigned idle = 0;
    while (__atomic_load_n(busy, __ATOMIC_ACQUIRE) != 0) {
        if (help_enabled() && gkr::HelpBoard::instance().help()) {
            ++pieces;
            idle = 0;
            continue;
        }
        GKR_CPU_RELAX();
        if (++idle > 4096) {   // nothing posted for a while: give the core away for a moment rather than spin at full rate
            std::this_thread::yield();
            idle = 0;
        }
    }
    return pieces;
}

int gkr_ctx_device_name(const gkr_ctx* ctx, char* buf, size_t len) {
    if (!ctx || !buf || !len) return GKR_ERR_INVALID;
    snprintf(buf, len, "%s", ctx->name);
    return GKR_OK;
}

int gkr_ctx_profile(gkr_ctx* ctx, int enable) {
    if (!ctx) return GKR_ERR_INVALID;
    ctx->profile = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
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

int gkr_ctx_profile_samples(gkr_ctx* ctx, const char* kernel, double* ms, double* bytes, size_t capacity, size_t* count) {
    if (!ctx || !kernel || !count) return GKR_ERR_INVALID;
    ctx->drain_events();
    auto it = ctx->prof.find(kernel);
    const size_t n = it == ctx->prof.end() ? 0 : it->second.samples.size();
    *count = n;
    for (size_t i = 0; i < n && i < capacity; ++i) {
        if (ms) ms[i] = it->second.samples[i].first;
        if (bytes) bytes[i] = it->second.samples[i].second;
    }
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

int gkr_selftest_fold(const gkr_fr* lo, const gkr_fr* hi, const gkr_fr* r, gkr_fr* out) {
    if (!lo || !hi || !r || !out) return GKR_ERR_INVALID;
    if (!all_canonical(lo, 1) || !all_canonical(hi, 1) || !all_canonical(r, 1)) return GKR_ERR_NON_CANONICAL;
    // the round's fixed-multiplier table built both ways (8x32-bit and the host's 4x64-bit code) must agree
    const gkr::FixedMul T = gkr::make_fixed_mul(to_dev(*r));
    gkr::FixedMul T64;
    gkr::h64::F r64;
    memcpy(&r64, r, 32);
    gkr::h64::make_fixed_mul(r64, T64.w);
    if (memcmp(&T, &T64, sizeof T) != 0) return GKR_ERR_INVALID;
    const Fr single = gkr::fr_fold_fixed(to_dev(*lo), to_dev(*hi), T);
    Fr y0, y1;   // the paired form the kernels use: (lo, hi) and (hi, lo) together
    gkr::fr_fold_fixed2(to_dev(*lo), to_dev(*hi), to_dev(*hi), to_dev(*lo), T, y0, y1);
    if (!gkr::fr_eq(y0, single) || !gkr::fr_eq(y1, gkr::fr_fold_fixed(to_dev(*hi), to_dev(*lo), T))) return GKR_ERR_INVALID;
    *out = to_abi(single);
    return GKR_OK;
}

// the pass schedule of a 2^n-point sumcheck (host logic only): rounds[i] = rounds covered by pass i, *passes = how many.
// mfma != 0: the default schedule (up to 5 rounds per pass), 0: the v_mad_u64_u32 fold's (up to 3).
int gkr_selftest_pass_schedule(int n, int mfma, uint32_t* rounds, size_t capacity, size_t* passes) {
    if (n < 1 || n > 40 || !passes) return GKR_ERR_INVALID;
    const int jmax = mfma ? gkr::kMlePassMaxRounds : 3;
    size_t count = 0;
    for (int m = n; m > 0;) {
        const int j = mle_pass_rounds(m, n, jmax);
        if (rounds && count < capacity) rounds[count] = (uint32_t)j;
        ++count;
        m -= j;
    }
    *passes = count;
    return (rounds && count > capacity) ? GKR_ERR_NOMEM : GKR_OK;
}

// The host's share of one multi-round pass (host logic only): `count` <= 16 sumchecks, each with 2^J sub-block sums
// (rows of 32: sums[k * 32 + b]) -> per round t < J the coefficients c0[t * count + k], c1[..], the vector length
// len[..] (final_len non-null: the last round's lengths are given, as in a sumcheck's final round), the challenge
// r[..]; and w[k * 32 + b], b < 2^J: the weights eq((r_0..r_{J-1}), b) of the fold pass that follows, canonical.
// Scalar code always; the IFMA-lane form runs beside it when the CPU has it (*used_ifma = 1) and any difference is
// GKR_ERR_INVALID.
int gkr_selftest_host_pass(const gkr_fr* sums, int count, int J, const uint32_t* final_len, gkr_fr* c0, gkr_fr* c1, uint32_t* len,
                           gkr_fr* r, gkr_fr* w, int* used_ifma) {
    using gkr::h64::F;
    if (!sums || !c0 || !c1 || !len || !r || !w || count < 1 || count > kHashChunkMax || J < 1 || J > gkr::kMlePassMaxRounds)
        return GKR_ERR_INVALID;
    static_assert(gkr::kMleMaxSub == 32 && kHashChunkMax == 16, "shapes of the self-test's arrays");
    if (!all_canonical(sums, (size_t)count * 32)) return GKR_ERR_NON_CANONICAL;
    if (final_len)
        for (int k = 0; k < count; ++k)
            if (final_len[k] != 1 && final_len[k] != 2) return GKR_ERR_INVALID;
    struct Out {
        uint64_t c0[gkr::kMlePassMaxRounds][16][4], c1[gkr::kMlePassMaxRounds][16][4], r[gkr::kMlePassMaxRounds][16][4];
        uint32_t len[gkr::kMlePassMaxRounds][16];
        std::vector<F> w;
    } a, b;
    a.w.assign((size_t)count * 32, F{{0, 0, 0, 0}});
    host_pass_scalar(reinterpret_cast<const uint64_t*>(sums), 4 * 32, count, J, final_len, a.c0, a.c1, a.r, a.len, &a.w[0].l[0], 4 * 32);
    const bool ifma = host_ifma_ready();
    if (used_ifma) *used_ifma = ifma ? 1 : 0;
    if (ifma) {
        b.w.assign((size_t)count * 32, F{{0, 0, 0, 0}});
        gkr::gkr_ifma_pass(reinterpret_cast<const uint64_t*>(sums), 4 * 32, count, J, final_len, b.c0, b.c1, b.r, b.len, &b.w[0].l[0], 4 * 32);
        for (int t = 0; t < J; ++t)
            for (int k = 0; k < count; ++k)
                if (memcmp(a.c0[t][k], b.c0[t][k], 32) || memcmp(a.c1[t][k], b.c1[t][k], 32) || memcmp(a.r[t][k], b.r[t][k], 32) ||
                    a.len[t][k] != b.len[t][k])
                    return GKR_ERR_INVALID;
        if (memcmp(a.w.data(), b.w.data(), sizeof(F) * a.w.size()) != 0) return GKR_ERR_INVALID;
    }
    for (int t = 0; t < J; ++t)
        for (int k = 0; k < count; ++k) {
            memcpy(&c0[(size_t)t * count + k], a.c0[t][k], 32);
            memcpy(&c1[(size_t)t * count + k], a.c1[t][k], 32);
            memcpy(&r[(size_t)t * count + k], a.r[t][k], 32);
            len[(size_t)t * count + k] = a.len[t][k];
        }
    for (size_t i = 0; i < a.w.size(); ++i) {
        const F wc = gkr::h64::from_mont(a.w[i]);
        memcpy(&w[i], &wc, 32);
    }
    return GKR_OK;
}

// The host's share of one product pass of the layer sumcheck (host logic only): `count` <= 16 sumchecks; recs: per
// sumcheck 72 values (cross sums m[a * 8 + b], a, b < 2^J, then the Y sums at 64 + a); vec_len[t * count + k]: 2 or 3.
// -> per round t < J (index t * count + k) the coefficients c2, lin, c0 and the challenge; w[k * 8 + b], b < 2^J: the
// weights eq(r, b) of the fold that follows, canonical.  Scalar code always; the IFMA-lane form runs beside it when
// the CPU has it (*used_ifma = 1) and any difference is GKR_ERR_INVALID.
int gkr_selftest_host_prod_pass(const gkr_fr* recs, int count, int J, const uint32_t* vec_len, gkr_fr* c2, gkr_fr* lin, gkr_fr* c0,
                                gkr_fr* r, gkr_fr* w, int* used_ifma) {
    using gkr::h64::F;
    if (!recs || !vec_len || !c2 || !lin || !c0 || !r || !w || count < 1 || count > kHashChunkMax || J < 1 || J > gkr::kProdMaxJ)
        return GKR_ERR_INVALID;
    if (!all_canonical(recs, (size_t)count * gkr::kProdRecValues)) return GKR_ERR_NON_CANONICAL;
    uint32_t vl[gkr::kProdMaxJ][16] = {};
    for (int t = 0; t < J; ++t)
        for (int k = 0; k < count; ++k) {
            vl[t][k] = vec_len[(size_t)t * count + k];
            if (vl[t][k] != 2 && vl[t][k] != 3) return GKR_ERR_INVALID;
        }
    struct Out {
        uint64_t c2[gkr::kProdMaxJ][16][4], lin[gkr::kProdMaxJ][16][4], c0[gkr::kProdMaxJ][16][4], r[gkr::kProdMaxJ][16][4];
        F w[16][8];
    } a, b;
    memset(&a, 0, sizeof a);
    memset(&b, 0, sizeof b);
    host_prod_pass_scalar(reinterpret_cast<const uint64_t*>(recs), 4 * gkr::kProdRecValues, count, J, vl, a.c2, a.lin, a.c0, a.r, &a.w[0][0].l[0], 32);
    const bool ifma = host_ifma_ready();
    if (used_ifma) *used_ifma = ifma ? 1 : 0;
    if (ifma) {
        gkr::gkr_ifma_prod_pass(reinterpret_cast<const uint64_t*>(recs), 4 * gkr::kProdRecValues, count, J, vl, b.c2, b.lin, b.c0, b.r, &b.w[0][0].l[0], 32);
        for (int t = 0; t < J; ++t)
            for (int k = 0; k < count; ++k)
                if (memcmp(a.c2[t][k], b.c2[t][k], 32) || memcmp(a.lin[t][k], b.lin[t][k], 32) || memcmp(a.c0[t][k], b.c0[t][k], 32) ||
                    memcmp(a.r[t][k], b.r[t][k], 32))
                    return GKR_ERR_INVALID;
        for (int k = 0; k < count; ++k)
            if (memcmp(a.w[k], b.w[k], sizeof(F) << J) != 0) return GKR_ERR_INVALID;
    }
    for (int t = 0; t < J; ++t)
        for (int k = 0; k < count; ++k) {
            memcpy(&c2[(size_t)t * count + k], a.c2[t][k], 32);
            memcpy(&lin[(size_t)t * count + k], a.lin[t][k], 32);
            memcpy(&c0[(size_t)t * count + k], a.c0[t][k], 32);
            memcpy(&r[(size_t)t * count + k], a.r[t][k], 32);
        }
    for (int k = 0; k < count; ++k)
        for (int bb = 0; bb < 8; ++bb) {
            const F wc = bb < (1 << J) ? gkr::h64::from_mont(a.w[k][bb]) : F{{0, 0, 0, 0}};
            memcpy(&w[(size_t)k * 8 + bb], &wc, 32);
        }
    return GKR_OK;
}

// eight right-aligned round vectors (3 slots each, the last len[k] slots count) hashed the way the
// host transcript does: the eight-lane IFMA code when the CPU has it (*used_ifma = 1; its sixteen-lane
// form is cross-checked on the way), else scalar
int gkr_selftest_hash8(const gkr_fr* vecs, const uint32_t* len, gkr_fr* out, int* used_ifma) {
    if (!vecs || !len || !out) return GKR_ERR_INVALID;
    if (!all_canonical(vecs, 24)) return GKR_ERR_NON_CANONICAL;
    for (int k = 0; k < 8; ++k)
        if (len[k] > 3) return GKR_ERR_INVALID;
    const bool ifma = host_ifma_ready();
    if (used_ifma) *used_ifma = ifma ? 1 : 0;
    if (ifma) {
        uint64_t v[8][3][4], o[8][4];
        memcpy(v, vecs, sizeof v);
        gkr::gkr_ifma_multi_hash8(v, len, 3, o);
        memcpy(out, o, sizeof o);
        // the sixteen-lane interleaved form on the same vectors (second half in reverse order) must agree
        uint64_t v16[16][3][4], o16[16][4];
        uint32_t len16[16];
        for (int k = 0; k < 8; ++k) {
            memcpy(v16[k], v[k], sizeof v[k]);
            memcpy(v16[15 - k], v[k], sizeof v[k]);
            len16[k] = len16[15 - k] = len[k];
        }
        gkr::gkr_ifma_multi_hash16(v16, len16, 3, o16);
        for (int k = 0; k < 8; ++k)
            if (memcmp(o16[k], o[k], 32) != 0 || memcmp(o16[15 - k], o[k], 32) != 0) return GKR_ERR_INVALID;
    } else {
        const gkr::h64::F* cts = host_mimc_constants64();
        for (int k = 0; k < 8; ++k) {
            gkr::h64::F v[3];
            memcpy(v, vecs + 3 * k, 96);
            const gkr::h64::F r = host_multi_hash(v + (3 - len[k]), (int)len[k], cts);
            memcpy(&out[k], &r, 32);
        }
    }
    return GKR_OK;
}

// the transcript's own ceiling on this host: microseconds per MiMC7 multi_hash of a `len`-element round vector (len 2:
// plain sumcheck; 3: layer sumcheck) on one thread -- sixteen transcripts side by side in IFMA lanes (per hash, i.e.
// call time / 16; 0 where the CPU has no IFMA) and one transcript on the scalar 4 x 64-bit code
int gkr_ubench_host_hash(int len, double* us_per_hash_lanes16, double* us_per_hash_scalar) {
    if (len < 1 || len > 3 || !us_per_hash_lanes16 || !us_per_hash_scalar) return GKR_ERR_INVALID;
    uint64_t v16[16][3][4], o16[16][4];
    uint32_t len16[16];
    for (int k = 0; k < 16; ++k) {
        len16[k] = (uint32_t)len;
        for (int e = 0; e < 3; ++e)
            for (int j = 0; j < 4; ++j) v16[k][e][j] = j == 3 ? 0x0123456789abcdefull >> 4 : 0x9E3779B97F4A7C15ull * (uint64_t)(k * 12 + e * 4 + j + 1);
    }
    auto now = [] { return std::chrono::steady_clock::now(); };
    *us_per_hash_lanes16 = 0.0;
    if (host_ifma_ready()) {
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            const auto t0 = now();
            for (int i = 0; i < 64; ++i) {
                gkr::gkr_ifma_multi_hash16(v16, len16, 3, o16);
                memcpy(v16[i & 15][2], o16[(i + 1) & 15], 32);   // (keeps the calls dependent)
            }
            const double us = std::chrono::duration<double, std::micro>(now() - t0).count() / (64.0 * 16.0);
            if (us < best) best = us;
        }
        *us_per_hash_lanes16 = best;
    }
    const gkr::h64::F* cts = host_mimc_constants64();
    gkr::h64::F v[3];
    memcpy(v, v16[0], 96);
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        const auto t0 = now();
        for (int i = 0; i < 32; ++i) v[2] = host_multi_hash(v + (3 - len), len, cts);
        const double us = std::chrono::duration<double, std::micro>(now() - t0).count() / 32.0;
        if (us < best) best = us;
    }
    *us_per_hash_scalar = best;
    return GKR_OK;
}

int gkr_selftest_dot(const gkr_fr* a, const gkr_fr* b, size_t n, gkr_fr* out) {
    if ((!a || !b) && n) return GKR_ERR_INVALID;
    if (!out) return GKR_ERR_INVALID;
    if (!all_canonical(a, n) || !all_canonical(b, n)) return GKR_ERR_NON_CANONICAL;
    // sum a_i b_i the way the fused layer kernel does: b in Montgomery form, full products
    // accumulated unreduced, one reduction at the end
    gkr::Lazy17 acc = gkr::lazy_zero(), t0 = gkr::lazy_zero(), t1 = gkr::lazy_zero(), t2 = gkr::lazy_zero();
    for (size_t i = 0; i < n; ++i) {
        const Fr x = to_dev(a[i]), y = gkr::to_mont(to_dev(b[i]));
        gkr::lazy_mac_s(acc, x, y);
        gkr::lazy_mac3_s(t0, x, y, t1, y, x, t2, x, y);   // the tripled form the fused layer kernel uses
    }
    const Fr r = gkr::lazy_reduce(acc);
    if (!gkr::fr_eq(gkr::lazy_reduce(t0), r) || !gkr::fr_eq(gkr::lazy_reduce(t1), r) || !gkr::fr_eq(gkr::lazy_reduce(t2), r))
        return GKR_ERR_INVALID;
    if (n <= 8) {   // the multi-round fold's form: four accumulators advanced together, short reduction
        gkr::Lazy17 q[4] = {gkr::lazy_zero(), gkr::lazy_zero(), gkr::lazy_zero(), gkr::lazy_zero()};
        Fr xs[8], ys[8];
        for (size_t i = 0; i < 8; ++i) {
            xs[i] = i < n ? to_dev(a[i]) : gkr::fr_zero();
            ys[i] = i < n ? gkr::to_mont(to_dev(b[i])) : gkr::fr_zero();
        }
        gkr::lazy_mac4_s(q[0], xs[0], ys[0], q[1], xs[1], ys[1], q[2], xs[2], ys[2], q[3], xs[3], ys[3]);
        gkr::lazy_mac4_s(q[0], xs[4], ys[4], q[1], xs[5], ys[5], q[2], xs[6], ys[6], q[3], xs[7], ys[7]);
        gkr::lazy_add(q[0], q[1]);
        gkr::lazy_add(q[2], q[3]);
        gkr::lazy_add(q[0], q[2]);
        if (!gkr::fr_eq(gkr::lazy_reduce_k8(q[0]), r)) return GKR_ERR_INVALID;
        gkr::Lazy17 ws;   // the one-shot column form the multifold kernel uses
        gkr::weighted_sum_s<8>(xs, ys, ws);
        if (!gkr::fr_eq(gkr::lazy_reduce_k8(ws), r)) return GKR_ERR_INVALID;
        if (n <= 4) {
            Fr x4[4] = {xs[0], xs[1], xs[2], xs[3]};
            gkr::weighted_sum_s<4>(x4, ys, ws);
            if (!gkr::fr_eq(gkr::lazy_reduce_k8(ws), r)) return GKR_ERR_INVALID;
        }
    }
    *out = to_abi(r);
    return GKR_OK;
}

// One item of a gate-list segment and its bucket's combine step, on the host twins of the device code (gate_seg.h):
// out0 / out1 = E_hi * sum over the item's gates, canonical.
int gkr_selftest_seg_item(const gkr_fr* e_lo, const gkr_fr* t, const uint8_t* is_mult, size_t n, const gkr_fr* e_hi, int rows,
                          gkr_fr* out0, gkr_fr* out1) {
    if (!e_lo || !t || !is_mult || !e_hi || !out0 || !out1 || n > gkr::kSegCap) return GKR_ERR_INVALID;
    if (!all_canonical(e_lo, n) || !all_canonical(t, n) || !all_canonical(e_hi, 1)) return GKR_ERR_NON_CANONICAL;
    // both forms of the item arithmetic (gate_seg.h): a select per gate in the order given, and -- what k_seg_pass runs -- the
    // add gates first with one exchange of the accumulators; they must leave the same two sums
    gkr::Lazy17 L0 = gkr::lazy_zero(), L1 = gkr::lazy_zero(), R0 = gkr::lazy_zero(), R1 = gkr::lazy_zero(), O0, O1;
    bool sw = false;
    for (int pass = 0; pass < 2; ++pass)
        for (size_t i = 0; i < n; ++i) {
            const Fr e = gkr::to_mont(to_dev(e_lo[i])), w = gkr::to_mont(to_dev(t[i]));
            const bool m = is_mult[i] != 0;
            if (pass == 0) {
                if (rows)
                    gkr::seg_gate<true>(L0, L1, e, w, m);
                else
                    gkr::seg_gate<false>(L0, L1, e, w, m);
            }
            if (m != (pass == 1)) continue;   // pass 0: the add gates, pass 1: the mult gates
            if (rows)
                gkr::seg_gate_ordered<true>(R0, R1, sw, e, w, m);
            else
                gkr::seg_gate_ordered<false>(R0, R1, sw, e, w, m);
        }
    if (rows)
        gkr::seg_item_sums<true>(R0, R1, sw, O0, O1);
    else
        gkr::seg_item_sums<false>(R0, R1, sw, O0, O1);
    for (int c = 0; c < 17; ++c)
        if (O0.l[c] != L0.l[c] || O1.l[c] != L1.l[c]) return GKR_ERR_INVALID;
    const Fr x = gkr::lazy_reduce_partial32(L0), y = gkr::lazy_reduce_partial32(L1);
    gkr::Lazy17 A = gkr::lazy_zero(), B = gkr::lazy_zero();
    gkr::lazy_mac_v(A, x, to_dev(*e_hi));
    gkr::lazy_mac_v(B, y, to_dev(*e_hi));
    *out0 = to_abi(gkr::lazy_reduce(A));
    *out1 = to_abi(gkr::lazy_reduce(B));
    return GKR_OK;
}

// q(t) = W(b + t (c - b)) the way gkr_prove computes it on the host (Moebius transform for the length, variable-by-
// variable binding for the coefficients); out: k + 1 slots right-aligned
int gkr_selftest_line_restriction(int k, const gkr_fr* W, const gkr_fr* b, const gkr_fr* c, gkr_fr* out, uint32_t* out_len) {
    if (k < 1 || k > 20 || !W || !b || !c || !out || !out_len) return GKR_ERR_INVALID;
    const size_t n = (size_t)1 << k;
    if (!all_canonical(W, n) || !all_canonical(b, k) || !all_canonical(c, k)) return GKR_ERR_NON_CANONICAL;
    std::vector<gkr::h64::F> vals(n);
    memcpy(vals.data(), W, n * 32);
    std::vector<gkr::h64::F> co(vals);
    mobius_msb(co, k);
    line_restriction(vals, co, k, b, c, out, out_len);
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const Fr* shards = static_cast<const Fr*>(d_shards);
    const size_t len = (size_t)1 << nl;
    const int t_stop = nl < kMleShardTailLog2 ? nl : kMleShardTailLog2;   // variables every shard keeps for the gathered tail
    static const int jmax = getenv("GKR_NO_MFMA_FOLD") ? 3 : gkr::kMlePassMaxRounds;
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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

int gkr_resident_layer_sumcheck(gkr_ctx* ctx, gkr_resident_layer* layer, const gkr_fr* z, const gkr_fr* W, gkr_allreduce_fn allreduce,
                                void* user, gkr_fr* out_coeffs, uint32_t* out_len, gkr_fr* out_r) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!layer || !W || !out_coeffs || !out_len || !out_r || (layer->k_i > 0 && !z)) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    const int k_i = layer->k_i, k = layer->k;
    if (!allreduce && (layer->first != 0 || layer->count != ((uint64_t)1 << k_i)))
        return ctx->fail(GKR_ERR_INVALID, "without an exchange hook the layer must be whole");
    if (k_i > 0 && !all_canonical(z, k_i)) return ctx->fail(GKR_ERR_NON_CANONICAL, "z entry >= r");
    if (!all_canonical(W, (size_t)1 << k)) return ctx->fail(GKR_ERR_NON_CANONICAL, "W entry >= r");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Fr* dW = nullptr;
    HIP_TRY(ctx, ctx->workspace("layer.Win", sizeof(Fr) << k, reinterpret_cast<void**>(&dW)));
    gkr_fr* hW = nullptr;   // through pinned memory and a copy kernel (no transfer call on the path, see k_copy_words)
    HIP_TRY(ctx, ctx->pinned_host("layer.hWin", sizeof(gkr_fr) << k, reinterpret_cast<void**>(&hW)));
    memcpy(hW, W, sizeof(gkr_fr) << k);
    gkr::launch_copy_words(hW, dW, ((size_t)8) << k, ctx->stream);
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
    if (!all_canonical(W, (size_t)1 << k)) return ctx->fail(GKR_ERR_NON_CANONICAL, "W entry >= r");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Fr* dW = nullptr;
    HIP_TRY(ctx, ctx->workspace("layer.Win", sizeof(Fr) << k, reinterpret_cast<void**>(&dW)));
    gkr_fr* hW = nullptr;
    HIP_TRY(ctx, ctx->pinned_host("layer.hWin", sizeof(gkr_fr) << k, reinterpret_cast<void**>(&hW)));
    memcpy(hW, W, sizeof(gkr_fr) << k);
    gkr::launch_copy_words(hW, dW, ((size_t)8) << k, ctx->stream);
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
    gkr::launch_layer_eval((uint32_t)gates, dgt.p, dl.p, dr.p, dprev.p, dout.p, 1, 0, ctx->stream);
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

// `batch` proofs of ONE circuit (different witnesses) advanced together: every layer's sumcheck runs as
// one batched sumcheck (run_layer_batch), so a round costs one set of launches and one host round trip
// for all proofs.  This is the multi-proof form of the reference's rayon par_iter over independent
// (circuit, input) pairs (aggregator.rs:350-355) for the case where the circuits coincide
// (BASELINE configs[3]: 64 inputs of one circom circuit).
static int prove_batch_impl(gkr_ctx* ctx, const gkr_circuit_desc* c, const gkr_fr* input_values, int batch,
                            int require_zero_output, gkr_proof_buf* outs) {
    using gkr::h64::F;
    const auto t_entry = std::chrono::steady_clock::now();
    int rc = check_circuit(ctx, c);
    if (rc) return rc;
    if (!input_values || !outs || batch < 1 || batch > 4096) return ctx->fail(GKR_ERR_INVALID, "null pointer or batch out of [1, 4096]");
    for (int b = 0; b < batch; ++b) {
        const gkr_proof_buf* out = &outs[b];
        if (!out->sumcheck_coeffs || !out->sumcheck_len || !out->sumcheck_r || !out->q || !out->q_len || !out->z || !out->r ||
            !out->d_coeffs || !out->input_coeffs)
            return ctx->fail(GKR_ERR_INVALID, "null pointer in proof buffers");
    }
    const uint32_t L = c->depth;
    for (uint32_t i = 0; i < L; ++i)
        if (!c->gate_type[i] || !c->left[i] || !c->right[i]) return ctx->fail(GKR_ERR_INVALID, "null gate array");
    const size_t n_in = (size_t)1 << c->k[L];
    if (!all_canonical(input_values, n_in * batch)) return ctx->fail(GKR_ERR_NON_CANONICAL, "input value >= r");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;

    // the circuit on the device: from the context's cache when this circuit was proven before (two independent
    // 64-bit hashes over the k list and the gate arrays decide), else validated, uploaded and remembered
    uint64_t h1 = 0xcbf29ce484222325ULL, h2 = 0x9E3779B97F4A7C15ULL;
    auto mix = [&](const void* p, size_t n) {
        const unsigned char* q = static_cast<const unsigned char*>(p);
        size_t i = 0;
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
    };
    mix(&L, sizeof L);
    mix(c->k, (L + 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < L; ++i) {
        const size_t gates = (size_t)1 << c->k[i];
        mix(c->gate_type[i], gates);
        mix(c->left[i], gates * 4);
        mix(c->right[i], gates * 4);
    }
    PreparedCircuit* pc = nullptr;
    for (size_t i = 0; i < ctx->circuits.size(); ++i)
        if (ctx->circuits[i]->h1 == h1 && ctx->circuits[i]->h2 == h2 && ctx->circuits[i]->k.size() == L + 1 &&
            memcmp(ctx->circuits[i]->k.data(), c->k, (L + 1) * sizeof(uint32_t)) == 0) {
            std::unique_ptr<PreparedCircuit> hit = std::move(ctx->circuits[i]);
            ctx->circuits.erase(ctx->circuits.begin() + i);
            ctx->circuits.push_back(std::move(hit));   // most recently used last
            pc = ctx->circuits.back().get();
            break;
        }
    static const bool no_cache = getenv("GKR_NO_CIRCUIT_CACHE") != nullptr;
    std::unique_ptr<PreparedCircuit> fresh;
    struct DropFresh {   // an uncached or failed circuit's device arrays do not outlive the call
        gkr_ctx* ctx;
        std::unique_ptr<PreparedCircuit>& p;
        ~DropFresh() {
            if (p) {
                (void)hipStreamSynchronize(ctx->stream);
                p->release();
            }
        }
    } drop_fresh{ctx, fresh};
    if (!pc) {
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
        pc = fresh.get();
    }

    // forward-evaluate every layer of every proof on the device (calculate_input, convert.rs:787-831)
    std::vector<Fr*> dW(L + 1, nullptr);
    for (uint32_t i = 0; i <= L; ++i) {
        const std::string slot = "prove.W" + std::to_string(i);
        HIP_TRY(ctx, ctx->workspace(slot.c_str(), ((size_t)batch << c->k[i]) * sizeof(Fr), reinterpret_cast<void**>(&dW[i])));
    }
    // Small transfers go through pinned buffers and a copy kernel, not through the runtime's transfer calls (see
    // k_copy_words); large ones (a 2^20-value input layer) keep the copy engine's bandwidth.
    constexpr size_t kKernelCopyLimit = (size_t)4 << 20;
    const size_t in_bytes = n_in * batch * sizeof(Fr);
    if (in_bytes <= kKernelCopyLimit) {
        gkr_fr* h_in = nullptr;
        HIP_TRY(ctx, ctx->pinned_host("prove.in", in_bytes, reinterpret_cast<void**>(&h_in)));
        memcpy(h_in, input_values, in_bytes);
        gkr::launch_copy_words(h_in, dW[L], in_bytes / 4, s);
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(dW[L], input_values, in_bytes, hipMemcpyHostToDevice, s));
    }
    for (int i = (int)L - 1; i >= 0; --i)
        gkr::launch_layer_eval(1u << c->k[i], pc->gt[i], pc->l[i], pc->r[i], dW[i + 1], dW[i], (uint32_t)batch, 1u << c->k[i + 1], s);
    HIP_TRY(ctx, hipGetLastError());
    // the host needs the outputs and the inputs (d, input_func); the layers in between stay on the device
    const F* hW[2] = {nullptr, nullptr};   // [0]: W_0, [1]: W_L
    std::vector<F> hW_big[2];
    // d and input_func are the monomial forms of W_0 and W_L (get_multi_ext, poly.rs:502-536): tables beyond 2^12 values are
    // transformed on the device (k launches over a grid) and land in the proof buffers directly; small ones on the host
    constexpr uint32_t kDeviceMobiusMinK = 13;
    bool coeffs_done[2] = {false, false};
    for (int e = 0; e < 2; ++e) {
        const uint32_t i = e ? L : 0;
        if (c->k[i] < kDeviceMobiusMinK) continue;
        const size_t n = (size_t)1 << c->k[i];
        Fr* mono = nullptr;
        HIP_TRY(ctx, ctx->workspace(e ? "prove.monoL" : "prove.mono0", n * batch * sizeof(Fr), reinterpret_cast<void**>(&mono)));
        HIP_TRY(ctx, hipMemcpyAsync(mono, dW[i], n * batch * sizeof(Fr), hipMemcpyDeviceToDevice, s));
        gkr::launch_mobius(mono, c->k[i], n, (uint32_t)batch, s);
        for (int b = 0; b < batch; ++b)
            HIP_TRY(ctx, hipMemcpyAsync(e ? outs[b].input_coeffs : outs[b].d_coeffs, mono + (size_t)b * n, n * sizeof(Fr), hipMemcpyDeviceToHost, s));
        coeffs_done[e] = true;
    }
    for (int e = 0; e < 2; ++e) {
        const uint32_t i = e ? L : 0;
        const size_t bytes = ((size_t)batch << c->k[i]) * sizeof(Fr);
        if (coeffs_done[e] && e == 1) continue;   // (W_0 is still read below: output 0 must be zero)
        if (bytes <= kKernelCopyLimit) {
            F* dst = nullptr;
            HIP_TRY(ctx, ctx->pinned_host(e ? "prove.hWL" : "prove.hW0", bytes, reinterpret_cast<void**>(&dst)));
            gkr::launch_copy_words(dW[i], dst, bytes / 4, s);
            hW[e] = dst;
        } else {
            hW_big[e].resize((size_t)batch << c->k[i]);
            HIP_TRY(ctx, hipMemcpyAsync(hW_big[e].data(), dW[i], bytes, hipMemcpyDeviceToHost, s));
            hW[e] = hW_big[e].data();
        }
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    static const bool dbg_pb = getenv("GKR_DEBUG_TIMING") != nullptr;
    const auto tpb0 = std::chrono::steady_clock::now();
    if (dbg_pb) t_account = ThreadTimeAccount();
    if (dbg_pb) fprintf(stderr, "[gkr timing] prove: circuit %s, forward evaluation + readback done\n", fresh ? "uploaded" : "from cache");
    for (int b = 0; b < batch; ++b) {
        if (require_zero_output && !gkr::h64::is_zero(hW[0][(size_t)b << c->k[0]]))
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
        // z[0] = 0 (prover.rs:16-21)
        for (uint32_t j = 0; j < c->k[0]; ++j) memset(&outs[b].z[j], 0, sizeof(gkr_fr));
    }
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
    HIP_TRY(ctx, ctx->aux_stream(0));
    for (uint32_t i = 0; i < L; ++i) {
        const int k_i = c->k[i], k = c->k[i + 1];
        for (int b = 0; b < batch; ++b) {
            scp[b] = outs[b].sumcheck_coeffs + row_off * 3;
            slp[b] = outs[b].sumcheck_len + row_off;
            srp[b] = outs[b].sumcheck_r + row_off;
        }
        const auto tl0 = std::chrono::steady_clock::now();
        ctx->rounds_ahead = 0;
        for (uint32_t later = i + 1; later < L; ++later) ctx->rounds_ahead += 2 * (int)c->k[later + 1];
        rc = run_layer_batch(ctx, batch, k_i, k, pc->gt[i], pc->l[i], pc->r[i], z_cur.data(), dW[i + 1], scp.data(), slp.data(),
                             srp.data(), nullptr, &pc->lists[i]);
        ctx->rounds_ahead = 0;
        if (rc) {
            (void)hipStreamSynchronize(ctx->aux);   // earlier layers' line restrictions still write the pinned q buffers the next call reuses
            return rc;
        }
        const auto tl1 = std::chrono::steady_clock::now();
        std::vector<gkr_fr> z_next((size_t)batch * k);
        {
            gkr_fr* lines = h_lines + (size_t)i * batch * 2 * kmax;
            for (int b = 0; b < batch; ++b) memcpy(lines + (size_t)b * 2 * k, srp[b], (size_t)2 * k * sizeof(gkr_fr));
            Timed t(ctx, "line_restriction", 0.0, ctx->aux, true);
            gkr::launch_line_restriction(dW[i + 1], (uint32_t)k, reinterpret_cast<const Fr*>(lines), d_lr, d_lrdeg, d_q + q_off * batch,
                                         d_qlen + (size_t)i * batch, (uint32_t)batch, ctx->aux);
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
        HIP_TRY(ctx, hipStreamSynchronize(ctx->aux));
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
    if (fresh && !no_cache) {
        constexpr size_t kMaxCachedCircuits = 64;   // three aggregation steps' worth of sub-circuits
        if (ctx->circuits.size() >= kMaxCachedCircuits) {
            ctx->circuits.front()->release();
            ctx->circuits.erase(ctx->circuits.begin());
        }
        ctx->circuits.push_back(std::move(fresh));
    }
    return GKR_OK;
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
    for (int idx : m->items) {
        gkr_prove_item& it = crew->items[idx];
        if (!it.circuit || !it.input_values || !it.outs) {
            it.status = m->ctx->fail(GKR_ERR_INVALID, "null pointer in a prove item");
            continue;
        }
        it.status = prove_batch_impl(m->ctx, it.circuit, it.input_values, it.batch, it.require_zero_output, it.outs);
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
    int want = max_concurrent;
    if (!want) {
        int share = usable_cpus();
        if (const char* e = getenv("LOCAL_WORLD_SIZE")) {   // ranks of one node share its CPUs
            const int ranks = atoi(e);
            if (ranks > 1) share = share / ranks > 1 ? share / ranks : 1;
        }
        want = share >= 6 ? share - 2 : (share >= 3 ? share - 1 : share);   // two (one, none) left to the runtime's own threads
    }
    // (members beyond the number of items have nothing to prove: they lend themselves from the start -- only if asked for)
    {
        // (members beyond what the items -- cut in two where they are large, below -- can occupy have nothing to prove)
        size_t can_use = n_items;
        if (getenv("GKR_PROVE_MANY_PIECES"))
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
    while ((int)crew->members.size() < want) {
        gkr_ctx* child = nullptr;
        // member m lives on device devices[m mod #devices] (member 0 = this context, on devices[0])
        const int member_device = ctx->devices.empty() ? ctx->device : ctx->devices[crew->members.size() % ctx->devices.size()];
        const int rc = gkr_ctx_create(member_device, &child);
        if (rc) return ctx->fail(rc, "child context of gkr_prove_many");
        child->crew_member = true;
        child->transcript = GKR_TRANSCRIPT_HOST;
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
    static const int pieces_env = [] { const char* e = getenv("GKR_PROVE_MANY_PIECES"); return e ? atoi(e) : -1; }();
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
    // deal the items out by estimated cost, longest first, each to the member with the least so far (deterministic)
    std::vector<std::pair<double, int>> cost(n_items);
    for (size_t i = 0; i < n_items; ++i) {
        cost[i] = {item_cost(items[i]), (int)i};
        items[i].status = GKR_OK;
    }
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
        crew->active = want;
        crew->finished = 0;
        __atomic_store_n(&crew->busy, want, __ATOMIC_RELEASE);
        ++crew->generation;
    }
    crew->cv_start.notify_all();
    crew_prove_items(crew, crew->members[0].get());
    {
        std::unique_lock<std::mutex> g(crew->mu);
        crew->cv_done.wait(g, [&] { return crew->finished == want - 1; });
        crew->items = nullptr;
    }
    ctx->crew_member = false;
    ctx->transcript = saved_transcript;
    for (size_t i = 0; i < caller_n; ++i) caller_items[i].status = GKR_OK;
    int first_bad = GKR_OK;
    for (int m = 0; m < want; ++m)
        for (int idx : crew->members[m]->items)
            if (items[idx].status != GKR_OK) {
                if (caller_items[origin[idx]].status == GKR_OK) caller_items[origin[idx]].status = items[idx].status;
                if (first_bad == GKR_OK) {
                    if (m) ctx->err = crew->members[m]->ctx->err;
                    first_bad = items[idx].status;
                }
            }
    return first_bad;
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf<uint32_t> flag;
    HIP_TRY(ctx, flag.alloc(1));
    HIP_TRY(ctx, hipMemsetAsync(flag.p, 0, 4, ctx->stream));
    gkr::launch_tables_differ(static_cast<const Fr*>(d_a), static_cast<const Fr*>(d_b), count, flag.p, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out_differ, flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GKR_OK;
}

// ---- device memory helpers ---------------------------------------------------------------

int gkr_device_alloc(gkr_ctx* ctx, size_t bytes, void** d_ptr) {
    if (!ctx || !d_ptr || !bytes) return GKR_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = device_malloc(d_ptr, bytes);
    if (e == hipErrorOutOfMemory) return ctx->fail(GKR_ERR_NOMEM, "hipMalloc: out of memory");
    HIP_TRY(ctx, e);
    return GKR_OK;
}

int gkr_device_free(gkr_ctx* ctx, void* d_ptr) {
    if (!ctx) return GKR_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, device_free(d_ptr));
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

int gkr_device_fill_shard(gkr_ctx* ctx, void* d_shard, int n, int log2_shards, int shard, uint64_t seed) {
    if (!ctx || !d_shard || log2_shards < 0 || log2_shards > 16 || n - log2_shards < 1 || n - log2_shards > GKR_MAX_MLE_N || shard < 0 ||
        shard >= (1 << log2_shards))
        return GKR_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    gkr::launch_fill_shard(static_cast<Fr*>(d_shard), (size_t)1 << (n - log2_shards), (uint32_t)log2_shards, (uint32_t)shard, seed, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    return GKR_OK;
}

int gkr_ubench_ceilings(gkr_ctx* ctx, size_t bytes, double* copy_GBps, double* read_GBps, double* modmul_per_sec) {
    if (!ctx) return GKR_ERR_INVALID;
    if (!copy_GBps || !read_GBps || !modmul_per_sec) return ctx->fail(GKR_ERR_INVALID, "null pointer");
    if (bytes < ((size_t)64 << 20)) bytes = (size_t)64 << 20;
    bytes &= ~(size_t)4095;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    char *a = nullptr, *b = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&a), bytes));
    if (hipMalloc(reinterpret_cast<void**>(&b), bytes) != hipSuccess) {
        (void)hipFree(a);
        return ctx->fail(GKR_ERR_NOMEM, "device memory for the copy probe");
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = GKR_OK;
    auto best_ms = [&](auto&& launch, int reps) -> double {
        double best = 1e30;
        for (int r = 0; r < reps; ++r) {
            if (hipEventRecord(e0, s) != hipSuccess) return -1.0;
            launch();
            if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return -1.0;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.0;
            if (r > 0 && ms < best) best = ms;   // the first repetition touches the pages
        }
        return best;
    };
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = ctx->fail(GKR_ERR_HIP, "hipEventCreate");
    if (!rc) {
        gkr::launch_fill_table(reinterpret_cast<Fr*>(a), bytes / 32, 1, s);
        const double c = best_ms([&] { gkr::launch_ubench_copy(a, b, bytes, s); }, 6);
        const double rd = best_ms([&] { gkr::launch_ubench_read(a, b, bytes, s); }, 6);
        const int waves = 16384, reps = 128;   // 16 waves per SIMD: the rate no longer grows with more
        const double mm = best_ms([&] { gkr::launch_ubench_modmul(reinterpret_cast<Fr*>(a), 1u << 20, waves, reps, s); }, 4);
        if (c <= 0 || rd <= 0 || mm <= 0 || hipGetLastError() != hipSuccess) {
            rc = ctx->fail(GKR_ERR_HIP, "the ceiling probes failed");
        } else {
            *copy_GBps = 2.0 * (double)bytes / (c * 1e-3) / 1e9;
            *read_GBps = (double)bytes / (rd * 1e-3) / 1e9;
            *modmul_per_sec = (double)waves * 64.0 * reps / (mm * 1e-3);
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    return rc;
}

int gkr_device_synchronize(gkr_ctx* ctx) {
    if (!ctx) return GKR_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GKR_OK;
}

}  // extern "C"
