// Feasibility probe: can the HOST write device memory directly (large BAR) so that a waiting kernel sees it without a PCIe
// read of host memory?  hipExtMallocWithFlags(hipDeviceMallocFinegrained) / hipMallocManaged candidates.  A kernel spins on a
// word in device memory; the host stores to it through the pointer; measured: time from the host's store to the kernel's
// exit stamp landing in pinned memory, against the same with the flag in pinned HOST memory (what GKR_LAUNCH_AHEAD polls).
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/bar_write.hip -o /tmp/bar && /tmp/bar
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <signal.h>
#include <setjmp.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_wait(const volatile uint32_t* flag, uint32_t want, volatile uint32_t* done) {
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != want) __builtin_amdgcn_s_sleep(2);
    __hip_atomic_store(done, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }

static double trial(volatile uint32_t* flag_host_view, const uint32_t* flag_dev_view, volatile uint32_t* done, int reps) {
    double total = 0;
    for (int r = 1; r <= reps; ++r) {
        k_wait<<<1, 64>>>(flag_dev_view, (uint32_t)r, done);
        // let the kernel start spinning
        auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 200.0) {}
        auto t1 = std::chrono::steady_clock::now();
        __atomic_store_n(const_cast<uint32_t*>(flag_host_view), (uint32_t)r, __ATOMIC_RELEASE);
        while (__atomic_load_n(const_cast<uint32_t*>(done), __ATOMIC_ACQUIRE) != (uint32_t)r) {}
        total += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
        if (hipDeviceSynchronize() != hipSuccess) return -1;
    }
    return total / reps;
}

int main() {
    uint32_t *done = nullptr, *hflag = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void**>(&done), 64, hipHostMallocCoherent | hipHostMallocMapped));
    CK(hipHostMalloc(reinterpret_cast<void**>(&hflag), 64, hipHostMallocCoherent | hipHostMallocMapped));
    *done = 0;
    *hflag = 0;
    printf("flag in pinned host memory (kernel polls over PCIe): %.2f us from the host's store to the kernel's answer\n", trial(hflag, hflag, done, 200));
    uint32_t* dflag = nullptr;
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&dflag), 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        CK(hipMemset(dflag, 0, 4096));
        CK(hipDeviceSynchronize());
        signal(SIGSEGV, on_segv);
        signal(SIGBUS, on_segv);
        if (sigsetjmp(jb, 1) == 0) {
            volatile uint32_t probe = *reinterpret_cast<volatile uint32_t*>(dflag);   // host load from device memory
            (void)probe;
            *done = 0;
            printf("flag in fine-grained DEVICE memory written by the host through the BAR: %.2f us\n", trial(dflag, dflag, done, 200));
        } else {
            printf("the host cannot touch fine-grained device memory here (fault)\n");
        }
    }
    return 0;
}
