// The sum over ranks as a collective the LIBRARY owns: gkr_exchange_dev backed by RCCL (ncclAllReduce of int64 limbs
// with ncclSum, queued on the stream the library hands the hook -- its own), so that a host that is not Python (the
// reference's Rust gkr-aggregator) needs nothing but this C ABI to prove across GPUs: no torch, no
// callback of its own.  What it replaces in the reference: the rayon reduce of the per-gate / per-assignment terms
// (rust/src/gkr/sumcheck.rs:50-63, 97-124; :62 for prove_sumcheck), across GPUs instead of across cores.
//
// librccl is loaded on first use (dlopen), not linked: a process that proves on one GPU never maps it.
// One communicator per (device, rank); several ranks may live in one process (a thread per device, the calls to
// gkr_exchange_rccl_create made concurrently -- ncclCommInitRank blocks until all ranks have arrived) or in one process
// each.  The unique id (128 bytes) is made on one rank and handed to the others by whatever the host has (a file, a
// socket, MPI, torch.distributed's store).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>

#include "../../include/gkr_amd.h"

// The five entry points and the handful of types of RCCL's C ABI this unit uses, declared HERE: librccl is optional at run
// time (dlopen), and so is its development header at build time -- the library builds on a ROCm install without RCCL and
// reports GKR_ERR_UNSUPPORTED when the collective is asked for.  Values as in nccl.h (stable across NCCL 2.x / RCCL);
// where the header is present the build checks them against it.
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define GKR_HAVE_RCCL_HEADER 1
#endif
#endif
namespace rccl_abi {
typedef int Result;                      // ncclResult_t; ncclSuccess = 0
struct UniqueId {
    char internal[128];                  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES)
};
typedef struct CommOpaque* Comm;         // ncclComm_t
constexpr int kSuccess = 0, kInt64 = 4, kSum = 0;   // ncclSuccess, ncclInt64, ncclSum
}  // namespace rccl_abi
#ifdef GKR_HAVE_RCCL_HEADER
static_assert(sizeof(rccl_abi::UniqueId) == sizeof(ncclUniqueId) && (int)ncclSuccess == rccl_abi::kSuccess && (int)ncclInt64 == rccl_abi::kInt64 &&
                  (int)ncclSum == rccl_abi::kSum && sizeof(ncclComm_t) == sizeof(rccl_abi::Comm),
              "the locally declared RCCL ABI differs from <rccl/rccl.h>");
#endif

namespace {

struct RcclApi {
    void* handle = nullptr;
    rccl_abi::Result (*GetUniqueId)(rccl_abi::UniqueId*) = nullptr;
    rccl_abi::Result (*CommInitRank)(rccl_abi::Comm*, int, rccl_abi::UniqueId, int) = nullptr;
    rccl_abi::Result (*CommDestroy)(rccl_abi::Comm) = nullptr;
    rccl_abi::Result (*AllReduce)(const void*, void*, size_t, int /* ncclDataType_t */, int /* ncclRedOp_t */, rccl_abi::Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(rccl_abi::Result) = nullptr;
    std::string error;
};

thread_local std::string t_error;

RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            const char* e = dlerror();
            api.error = std::string("librccl could not be loaded: ") + (e ? e : "?");
            return;
        }
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.handle, "ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString)
            api.error = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce / ncclGetErrorString";
    });
    return &api;
}

int fail(int status, const std::string& what) {
    t_error = what;
    return status;
}

}  // namespace

struct gkr_rccl_exchange {
    int device = 0, rank = 0, nranks = 1;
    rccl_abi::Comm comm = nullptr;
    gkr_exchange_dev dev{};
    std::atomic<uint64_t> calls{0};
    std::atomic<int> last_status{0};
};

static int rccl_hook(void* user, size_t count, void* hip_stream) {
    gkr_rccl_exchange* x = static_cast<gkr_rccl_exchange*>(user);
    x->calls.fetch_add(1, std::memory_order_relaxed);
    const rccl_abi::Result rc = rccl_api()->AllReduce(x->dev.d_limbs, x->dev.d_limbs, count, rccl_abi::kInt64, rccl_abi::kSum, x->comm, static_cast<hipStream_t>(hip_stream));
    if (rc != rccl_abi::kSuccess) x->last_status.store((int)rc, std::memory_order_relaxed);
    return rc == rccl_abi::kSuccess ? 0 : (int)rc;
}

extern "C" {

const char* gkr_exchange_rccl_error(void) { return t_error.c_str(); }

int gkr_exchange_rccl_unique_id(void* id_out) {
    if (!id_out) return fail(GKR_ERR_INVALID, "null pointer");
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(GKR_ERR_UNSUPPORTED, api->error);
    static_assert(GKR_RCCL_ID_BYTES == sizeof(rccl_abi::UniqueId), "the ABI's id size is RCCL's");
    rccl_abi::UniqueId id;
    const rccl_abi::Result rc = api->GetUniqueId(&id);
    if (rc != rccl_abi::kSuccess) return fail(GKR_ERR_HIP, std::string("ncclGetUniqueId: ") + api->GetErrorString(rc));
    memcpy(id_out, &id, sizeof id);
    return GKR_OK;
}

int gkr_exchange_rccl_create(int device_id, const void* unique_id, int rank, int nranks, size_t capacity_limbs, gkr_rccl_exchange** out) {
    if (!out) return fail(GKR_ERR_INVALID, "null pointer");
    *out = nullptr;
    if (!unique_id || nranks < 1 || rank < 0 || rank >= nranks || capacity_limbs == 0) return fail(GKR_ERR_INVALID, "bad rank, world size or capacity");
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(GKR_ERR_UNSUPPORTED, api->error);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(GKR_ERR_NO_DEVICE, "no device");
    if (device_id < 0 || device_id >= count) return fail(GKR_ERR_INVALID, "device id out of range");
    if (hipSetDevice(device_id) != hipSuccess) return fail(GKR_ERR_HIP, "hipSetDevice");
    gkr_rccl_exchange* x = new gkr_rccl_exchange();
    x->device = device_id;
    x->rank = rank;
    x->nranks = nranks;
    void* buf = nullptr;
    if (hipMalloc(&buf, capacity_limbs * sizeof(int64_t)) != hipSuccess) {
        delete x;
        return fail(GKR_ERR_NOMEM, "hipMalloc of the limb buffer");
    }
    (void)hipMemset(buf, 0, capacity_limbs * sizeof(int64_t));
    rccl_abi::UniqueId id;
    memcpy(&id, unique_id, sizeof id);
    // ncclCommInitRank BLOCKS until all `nranks` ranks have called it with this id (there is no timeout in RCCL's API): a
    // rank that never arrives leaves the others here.  The host owns that failure mode -- it starts the ranks -- exactly as
    // with a bare RCCL program; gkr_amd.h says so, tests/test_gpu_sharded.py shows it (the parent kills the waiting child).
    const rccl_abi::Result rc = api->CommInitRank(&x->comm, nranks, id, rank);
    if (rc != rccl_abi::kSuccess) {
        (void)hipFree(buf);
        delete x;
        return fail(GKR_ERR_HIP, std::string("ncclCommInitRank: ") + api->GetErrorString(rc));
    }
    x->dev.fn = rccl_hook;
    x->dev.user = x;
    x->dev.d_limbs = static_cast<int64_t*>(buf);
    x->dev.capacity = capacity_limbs;
    *out = x;
    return GKR_OK;
}

const gkr_exchange_dev* gkr_exchange_rccl_dev(const gkr_rccl_exchange* x) { return x ? &x->dev : nullptr; }

uint64_t gkr_exchange_rccl_calls(const gkr_rccl_exchange* x) { return x ? x->calls.load(std::memory_order_relaxed) : 0; }

void gkr_exchange_rccl_destroy(gkr_rccl_exchange* x) {
    if (!x) return;
    (void)hipSetDevice(x->device);
    (void)hipDeviceSynchronize();
    if (x->comm) (void)rccl_api()->CommDestroy(x->comm);
    if (x->dev.d_limbs) (void)hipFree(x->dev.d_limbs);
    delete x;
}

}  // extern "C"
