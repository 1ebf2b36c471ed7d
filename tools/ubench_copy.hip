// HBM bandwidth probe: read-only, write-only and copy streams with several launch shapes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int U>
__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * stride < n) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * stride < n) out[i + u * stride] = v[u];
    }
}
template <int U>
__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * stride < n) { uint4 v = in[i + u * stride]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    }
    if (acc.x == 0x12345 && acc.y == 0x777) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(uint4* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) out[i] = make_uint4(i, 1, 2, 3);
}
// one element per thread, no loop
__global__ void __launch_bounds__(256) k_copy1(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// blocked distribution: block b owns the contiguous chunk [b*C, (b+1)*C); blocks are dispatched in
// order, so the chip sweeps each stream as one compact moving window
__global__ void __launch_bounds__(256) k_copy_blocked(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, uint32_t C) {
    const size_t base = (size_t)blockIdx.x * C;
    for (uint32_t t = threadIdx.x; t < C; t += 256) { const size_t i = base + t; if (i < n) out[i] = in[i]; }
}
__global__ void __launch_bounds__(256) k_read_blocked(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n, uint32_t C) {
    const size_t base = (size_t)blockIdx.x * C;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t t = threadIdx.x; t < C; t += 256) { const size_t i = base + t; if (i < n) { uint4 v = in[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; } }
    if (acc.x == 0x12345 && acc.y == 0x777) out[0] = acc;
}
// the fold access pattern (32-byte elements: two uint4 per lane; 4 read streams, 2 write streams), blocked
__global__ void __launch_bounds__(256) k_fold_blocked(const uint4* __restrict__ in, uint4* __restrict__ out, size_t q, uint32_t C) {
    const size_t base = (size_t)blockIdx.x * C;
    for (uint32_t t = threadIdx.x; t < C; t += 256) {
        const size_t i = base + t;
        if (i < q) {
            uint4 a0 = in[2 * i], a1 = in[2 * i + 1], b0 = in[2 * (i + 2 * q)], b1 = in[2 * (i + 2 * q) + 1];
            uint4 c0 = in[2 * (i + q)], c1 = in[2 * (i + q) + 1], d0 = in[2 * (i + 3 * q)], d1 = in[2 * (i + 3 * q) + 1];
            a0.x ^= b0.x; a1.x ^= b1.x; c0.x ^= d0.x; c1.x ^= d1.x;
            out[2 * i] = a0; out[2 * i + 1] = a1; out[2 * (i + q)] = c0; out[2 * (i + q) + 1] = c1;
        }
    }
}
__global__ void __launch_bounds__(256) k_fold_strided(const uint4* __restrict__ in, uint4* __restrict__ out, size_t q) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
        uint4 a0 = in[2 * i], a1 = in[2 * i + 1], b0 = in[2 * (i + 2 * q)], b1 = in[2 * (i + 2 * q) + 1];
        uint4 c0 = in[2 * (i + q)], c1 = in[2 * (i + q) + 1], d0 = in[2 * (i + 3 * q)], d1 = in[2 * (i + 3 * q) + 1];
        a0.x ^= b0.x; a1.x ^= b1.x; c0.x ^= d0.x; c1.x ^= d1.x;
        out[2 * i] = a0; out[2 * i + 1] = a1; out[2 * (i + q)] = c0; out[2 * (i + q) + 1] = c1;
    }
}

int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    uint4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define TIME(label, traffic, launch) { float best = 1e9; for (int r = 0; r < 5; ++r) { CK(hipEventRecord(e0)); launch; CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } CK(hipGetLastError()); printf("%-36s %.3f ms  %.1f GB/s\n", label, best, (double)(traffic) / best / 1e6); }
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        char l[64];
        snprintf(l, 64, "copy U1 grid %d", grid); TIME(l, 2 * bytes, hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, a, b, n));
        snprintf(l, 64, "copy U4 grid %d", grid); TIME(l, 2 * bytes, hipLaunchKernelGGL(k_copy<4>, dim3(grid), dim3(256), 0, 0, a, b, n));
        snprintf(l, 64, "read U1 grid %d", grid); TIME(l, bytes, hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(256), 0, 0, a, b, n));
        snprintf(l, 64, "read U4 grid %d", grid); TIME(l, bytes, hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, a, b, n));
        snprintf(l, 64, "write grid %d", grid); TIME(l, bytes, hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n));
    }
    for (uint32_t C : {256u, 1024u, 4096u, 16384u, 65536u}) {
        char l[64];
        snprintf(l, 64, "copy blocked C=%u", C); TIME(l, 2 * bytes, hipLaunchKernelGGL(k_copy_blocked, dim3((unsigned)((n + C - 1) / C)), dim3(256), 0, 0, a, b, n, C));
        snprintf(l, 64, "read blocked C=%u", C); TIME(l, bytes, hipLaunchKernelGGL(k_read_blocked, dim3((unsigned)((n + C - 1) / C)), dim3(256), 0, 0, a, b, n, C));
    }
    {
        const size_t q = n / 2 / 4;   // 32-byte elements: source table of 4q elements = 2 GiB
        for (int grid : {1024, 2048, 4096}) { char l[64]; snprintf(l, 64, "fold pattern strided grid %d", grid); TIME(l, q * 192, hipLaunchKernelGGL(k_fold_strided, dim3(grid), dim3(256), 0, 0, a, b, q)); }
        for (uint32_t C : {256u, 512u, 1024u, 2048u, 4096u, 16384u}) { char l[64]; snprintf(l, 64, "fold pattern blocked C=%u", C); TIME(l, q * 192, hipLaunchKernelGGL(k_fold_blocked, dim3((unsigned)((q + C - 1) / C)), dim3(256), 0, 0, a, b, q, C)); }
    }
    TIME("copy one element per thread", 2 * bytes, hipLaunchKernelGGL(k_copy1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n));
    TIME("hipMemcpyDtoD", 2 * bytes, CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)));
    return 0;
}
