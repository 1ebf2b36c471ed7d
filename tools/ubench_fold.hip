// Micro-benchmark: variants of the fold arithmetic inside the k_mle_fold_sum access
// pattern (4 coalesced read streams, 2 write streams).  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gkr_amd/csrc tools/ubench_fold.hip -o /tmp/ubench_fold && /tmp/ubench_fold
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include "fr32.h"
using namespace gkr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ Fr ld(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    Fr f; f.l[0]=a.x; f.l[1]=a.y; f.l[2]=a.z; f.l[3]=a.w; f.l[4]=b.x; f.l[5]=b.y; f.l[6]=b.z; f.l[7]=b.w; return f;
}
__device__ __forceinline__ void st(Fr* p, const Fr& f) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(f.l[0], f.l[1], f.l[2], f.l[3]); q[1] = make_uint4(f.l[4], f.l[5], f.l[6], f.l[7]);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ Fr ld_nt(const Fr* p) {
    const u32x4* q = reinterpret_cast<const u32x4*>(p);
    u32x4 a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1);
    Fr f; f.l[0]=a.x; f.l[1]=a.y; f.l[2]=a.z; f.l[3]=a.w; f.l[4]=b.x; f.l[5]=b.y; f.l[6]=b.z; f.l[7]=b.w; return f;
}
__device__ __forceinline__ void st_nt(Fr* p, const Fr& f) {
    u32x4* q = reinterpret_cast<u32x4*>(p);
    u32x4 a = {f.l[0], f.l[1], f.l[2], f.l[3]}, b = {f.l[4], f.l[5], f.l[6], f.l[7]};
    __builtin_nontemporal_store(a, q);
    __builtin_nontemporal_store(b, q + 1);
}

// ---- variant 1: asm mac with the 2-wait-state hazard padded inside
__device__ __forceinline__ void mac96n(uint64_t& acc, uint32_t& ex, uint32_t x, uint32_t y) {
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2"
        : "+v"(acc), "+v"(ex), "=&s"(carry) : "v"(x), "v"(y));
}
__device__ __forceinline__ void mac96ns(uint64_t& acc, uint32_t& ex, uint32_t x, uint32_t y) {
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\ts_nop 1\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2"
        : "+v"(acc), "+v"(ex), "=&s"(carry) : "v"(x), "s"(y));
}
__device__ __forceinline__ Fr mont_mul_v1(const Fr& a, const Fr& b) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    uint32_t m[8], t[8];
    uint64_t acc = 0; uint32_t ex = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int i = 0; i <= c; ++i) mac96n(acc, ex, a.l[i], b.l[c - i]);
#pragma unroll
        for (int i = 0; i < c; ++i) mac96ns(acc, ex, m[i], p[c - i]);
        m[c] = (uint32_t)acc * GKR_INV32;
        mac96ns(acc, ex, m[c], p[0]);
        acc = (acc >> 32) | ((uint64_t)ex << 32); ex = 0;
    }
#pragma unroll
    for (int c = 8; c < 15; ++c) {
#pragma unroll
        for (int i = c - 7; i < 8; ++i) mac96n(acc, ex, a.l[i], b.l[c - i]);
#pragma unroll
        for (int i = c - 7; i < 8; ++i) mac96ns(acc, ex, m[i], p[c - i]);
        t[c - 8] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)ex << 32); ex = 0;
    }
    t[7] = (uint32_t)acc;
    Fr out;
#pragma unroll
    for (int i = 0; i < 8; ++i) out.l[i] = t[i];
    return fr_reduce_once(out);
}

// ---- asm carry chains (hazard padded): out = a + b (256-bit, no carry out), out = a - b (returns borrow mask)
__device__ __forceinline__ void add256(uint32_t (&o)[8], const uint32_t (&a)[8], const uint32_t (&b)[8]) {
    asm("v_add_co_u32_e32 %0, vcc, %8, %16\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %1, vcc, %9, %17, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %2, vcc, %10, %18, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %3, vcc, %11, %19, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %4, vcc, %12, %20, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %5, vcc, %13, %21, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %6, vcc, %14, %22, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %7, vcc, %15, %23, vcc"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
        : "vcc");
}
// o = a - b; borrow_mask = all ones if a < b
__device__ __forceinline__ void sub256(uint32_t (&o)[8], uint32_t& borrow_mask, const uint32_t (&a)[8], const uint32_t (&b)[8]) {
    uint32_t bm;
    asm("v_sub_co_u32_e32 %0, vcc, %9, %17\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %1, vcc, %10, %18, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %2, vcc, %11, %19, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %3, vcc, %12, %20, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %4, vcc, %13, %21, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %5, vcc, %14, %22, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %6, vcc, %15, %23, vcc\n\ts_nop 1\n\t"
        "v_subb_co_u32_e32 %7, vcc, %16, %24, vcc\n\ts_nop 1\n\t"
        "v_cndmask_b32_e64 %8, 0, -1, vcc"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(bm)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])
        : "vcc");
    borrow_mask = bm;
}
__device__ __forceinline__ Fr fr_sub_v2(const Fr& a, const Fr& b) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    uint32_t d[8], bm, pm[8], o[8];
    sub256(d, bm, a.l, b.l);
#pragma unroll
    for (int i = 0; i < 8; ++i) pm[i] = p[i] & bm;
    add256(o, d, pm);
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = o[i];
    return r;
}
__device__ __forceinline__ Fr fr_cond_sub_v2(const uint32_t (&s)[8]) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    uint32_t pl[8], d[8], bm;
#pragma unroll
    for (int i = 0; i < 8; ++i) pl[i] = p[i];
    sub256(d, bm, s, pl);
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = bm ? s[i] : d[i];
    return r;
}
__device__ __forceinline__ Fr fr_add_v2(const Fr& a, const Fr& b) {
    uint32_t s[8];
    add256(s, a.l, b.l);
    return fr_cond_sub_v2(s);
}
__device__ __forceinline__ Fr mont_mul_v2(const Fr& a, const Fr& b) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    uint32_t m[8], t[8];
    uint64_t acc = 0; uint32_t ex = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int i = 0; i <= c; ++i) mac96n(acc, ex, a.l[i], b.l[c - i]);
#pragma unroll
        for (int i = 0; i < c; ++i) mac96ns(acc, ex, m[i], p[c - i]);
        m[c] = (uint32_t)acc * GKR_INV32;
        mac96ns(acc, ex, m[c], p[0]);
        acc = (acc >> 32) | ((uint64_t)ex << 32); ex = 0;
    }
#pragma unroll
    for (int c = 8; c < 15; ++c) {
#pragma unroll
        for (int i = c - 7; i < 8; ++i) mac96n(acc, ex, a.l[i], b.l[c - i]);
#pragma unroll
        for (int i = c - 7; i < 8; ++i) mac96ns(acc, ex, m[i], p[c - i]);
        t[c - 8] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)ex << 32); ex = 0;
    }
    t[7] = (uint32_t)acc;
    return fr_cond_sub_v2(t);
}
template <int NL>
__device__ __forceinline__ void acc_add_v2(Acc<NL>& a, const Fr& x) {
    static_assert(NL == 9, "9-limb accumulator");
    asm("v_add_co_u32_e32 %0, vcc, %0, %9\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %1, vcc, %1, %10, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %11, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %3, vcc, %3, %12, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %4, vcc, %4, %13, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %5, vcc, %5, %14, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %6, vcc, %6, %15, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %7, vcc, %7, %16, vcc\n\ts_nop 1\n\t"
        "v_addc_co_u32_e32 %8, vcc, 0, %8, vcc"
        : "+v"(a.l[0]), "+v"(a.l[1]), "+v"(a.l[2]), "+v"(a.l[3]), "+v"(a.l[4]), "+v"(a.l[5]), "+v"(a.l[6]), "+v"(a.l[7]), "+v"(a.l[8])
        : "v"(x.l[0]), "v"(x.l[1]), "v"(x.l[2]), "v"(x.l[3]), "v"(x.l[4]), "v"(x.l[5]), "v"(x.l[6]), "v"(x.l[7])
        : "vcc");
}

// ---- variant 3: multiplication by a FIXED scalar r through a table R_i = r * 2^(32 i) * 2^64 mod p
// (canonical, 8 x 8 limbs, wave-uniform -> SGPR operands).  d * r * 2^-64... : S = sum_i d_i R_i
// (64 mads, < 2^291), then two 32-bit Montgomery steps (16 mads) bring it under 2p.
struct RTable { uint32_t w[8][8]; };
__device__ __forceinline__ Fr mul_fixed_v3(const Fr& d, const RTable& T) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    uint32_t s[11];
    uint64_t acc = 0; uint32_t ex = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) mac96ns(acc, ex, d.l[i], T.w[i][c]);
        s[c] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)ex << 32); ex = 0;
    }
    s[8] = (uint32_t)acc; s[9] = (uint32_t)(acc >> 32); s[10] = 0;
    // two Montgomery steps on the 10-limb value
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t m = s[k] * GKR_INV32;
        uint64_t a2 = s[k]; uint32_t e2 = 0;
        mac96ns(a2, e2, m, p[0]);
        a2 = (a2 >> 32) | ((uint64_t)e2 << 32); e2 = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            a2 += s[k + j];                 // cannot overflow: a2 < 2^33 here
            mac96ns(a2, e2, m, p[j]);
            s[k + j] = (uint32_t)a2;
            a2 = (a2 >> 32) | ((uint64_t)e2 << 32); e2 = 0;
        }
        // propagate into the remaining limbs
#pragma unroll
        for (int j = k + 8; j < 11; ++j) {
            a2 += s[j];
            s[j] = (uint32_t)a2;
            a2 >>= 32;
        }
    }
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = s[i + 2];
    // value < 2p (s[10] == 0)
    return fr_cond_sub_v2(t);
}

template <int V>
__global__ void __launch_bounds__(256) k_fold(const Fr* __restrict__ src, Fr* __restrict__ dst, uint32_t q, Fr r, RTable T,
                                              Acc<9>* __restrict__ sums) {
    Acc<9> a0 = acc_zero<9>(), a1 = acc_zero<9>();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < q; i += gridDim.x * blockDim.x) {
        Fr x0 = ld(src + i), x1 = ld(src + i + 2 * (size_t)q), x2 = ld(src + i + q), x3 = ld(src + i + 3 * (size_t)q);
        Fr y0, y1;
        if (V == 0) { y0 = fr_add(x0, mont_mul_portable(fr_sub(x1, x0), r)); y1 = fr_add(x2, mont_mul_portable(fr_sub(x3, x2), r)); }
        else if (V == 1) { y0 = fr_add(x0, mont_mul_v1(fr_sub(x1, x0), r)); y1 = fr_add(x2, mont_mul_v1(fr_sub(x3, x2), r)); }
        else if (V == 2) { y0 = fr_add_v2(x0, mont_mul_v2(fr_sub_v2(x1, x0), r)); y1 = fr_add_v2(x2, mont_mul_v2(fr_sub_v2(x3, x2), r)); }
        else if (V == 3) { y0 = fr_add_v2(x0, mul_fixed_v3(fr_sub_v2(x1, x0), T)); y1 = fr_add_v2(x2, mul_fixed_v3(fr_sub_v2(x3, x2), T)); }
        else { y0 = x0; y1 = x2; y0.l[0] ^= x1.l[0]; y1.l[0] ^= x3.l[0]; }   // V == 9: memory pattern only
        st(dst + i, y0); st(dst + i + q, y1);
        if (V == 2 || V == 3) { acc_add_v2(a0, y0); acc_add_v2(a1, y1); }
        else if (V != 9) { acc_add_fr(a0, y0); acc_add_fr(a1, y1); }
    }
    if (V != 9) {
        // crude per-thread dump (not the product's reduction): keeps the sums live
        Acc<9> t = a0; acc_add_acc(t, a1);
        if ((threadIdx.x & 63) == 0 && t.l[8] == 0xdeadbeef) sums[blockIdx.x] = t;
    }
}

// memory-pattern study on the V3 arithmetic (or none): NT loads/stores, 2 pair-pairs per iteration
template <bool ARITH, bool NT, int UNROLL>
__global__ void __launch_bounds__(256) k_fold_opt(const Fr* __restrict__ src, Fr* __restrict__ dst, uint32_t q, RTable T,
                                                  Acc<9>* __restrict__ sums) {
    Acc<9> a0 = acc_zero<9>(), a1 = acc_zero<9>();
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < q; i0 += stride * UNROLL) {
        Fr x[UNROLL][4];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t i = i0 + u * stride;
            if (i < q) {
                if (NT) { x[u][0] = ld_nt(src + i); x[u][1] = ld_nt(src + i + 2 * (size_t)q); x[u][2] = ld_nt(src + i + q); x[u][3] = ld_nt(src + i + 3 * (size_t)q); }
                else { x[u][0] = ld(src + i); x[u][1] = ld(src + i + 2 * (size_t)q); x[u][2] = ld(src + i + q); x[u][3] = ld(src + i + 3 * (size_t)q); }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint32_t i = i0 + u * stride;
            if (i < q) {
                Fr y0, y1;
                if (ARITH) { y0 = fr_add_v2(x[u][0], mul_fixed_v3(fr_sub_v2(x[u][1], x[u][0]), T)); y1 = fr_add_v2(x[u][2], mul_fixed_v3(fr_sub_v2(x[u][3], x[u][2]), T)); }
                else { y0 = x[u][0]; y1 = x[u][2]; y0.l[0] ^= x[u][1].l[0]; y1.l[0] ^= x[u][3].l[0]; }
                if (NT) { st_nt(dst + i, y0); st_nt(dst + i + q, y1); } else { st(dst + i, y0); st(dst + i + q, y1); }
                if (ARITH) { acc_add_v2(a0, y0); acc_add_v2(a1, y1); }
            }
        }
    }
    if (ARITH) { Acc<9> t = a0; acc_add_acc(t, a1); if ((threadIdx.x & 63) == 0 && t.l[8] == 0xdeadbeef) sums[blockIdx.x] = t; }
}

__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

static uint64_t mix64(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 26;          // source table 2^n elements (2 GiB at n = 26)
    const size_t len = (size_t)1 << n, q = len / 4;
    Fr *src, *dst[4]; Acc<9>* sums;
    CK(hipMalloc(&src, len * 32));
    for (int v = 0; v < 4; ++v) CK(hipMalloc(&dst[v], len / 2 * 32));
    CK(hipMalloc(&sums, 4096 * sizeof(Acc<9>)));
    std::vector<Fr> h(len);
    for (size_t i = 0; i < len; ++i) for (int j = 0; j < 4; ++j) { uint64_t w = mix64(1234 + 4 * i + j); if (j == 3) w &= 0x0FFFFFFFFFFFFFFFULL; h[i].l[2*j] = (uint32_t)w; h[i].l[2*j+1] = (uint32_t)(w >> 32); }
    CK(hipMemcpy(src, h.data(), len * 32, hipMemcpyHostToDevice));
    // challenge r (canonical), its Montgomery form, and the fixed-scalar table
    Fr rc; for (int j = 0; j < 4; ++j) { uint64_t w = mix64(99 + j); if (j == 3) w &= 0x0FFFFFFFFFFFFFFFULL; rc.l[2*j] = (uint32_t)w; rc.l[2*j+1] = (uint32_t)(w >> 32); }
    Fr rm = to_mont(rc);
    RTable T;
    {   // R_i = r * 2^(32 i) * 2^64 mod p, canonical: start from r * 2^64 and multiply by 2^32 each step
        Fr two32 = fr_zero(); two32.l[1] = 1; Fr two64 = fr_zero(); two64.l[2] = 1;
        Fr cur = fr_mul(rc, two64);
        for (int i = 0; i < 8; ++i) { for (int c = 0; c < 8; ++c) T.w[i][c] = cur.l[c]; cur = fr_mul(cur, two32); }
    }
    const int grid = 2048;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](int v, Fr* out) {
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            switch (v) {
                case 0: hipLaunchKernelGGL(k_fold<0>, dim3(grid), dim3(256), 0, 0, src, out, (uint32_t)q, rm, T, sums); break;
                case 1: hipLaunchKernelGGL(k_fold<1>, dim3(grid), dim3(256), 0, 0, src, out, (uint32_t)q, rm, T, sums); break;
                case 2: hipLaunchKernelGGL(k_fold<2>, dim3(grid), dim3(256), 0, 0, src, out, (uint32_t)q, rm, T, sums); break;
                case 3: hipLaunchKernelGGL(k_fold<3>, dim3(grid), dim3(256), 0, 0, src, out, (uint32_t)q, rm, T, sums); break;
                default: hipLaunchKernelGGL(k_fold<9>, dim3(grid), dim3(256), 0, 0, src, out, (uint32_t)q, rm, T, sums); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        double bytes = (double)q * 6 * 32;
        printf("variant %d: %.3f ms  %.1f GB/s  (%.2f Gfold/s)\n", v, best, bytes / best / 1e6, 2.0 * q / best / 1e6);
    };
    run(9, dst[0]);
    for (int v = 0; v < 4; ++v) run(v, dst[v]);
    {
        float best = 1e9;
        for (int it = 0; it < 4; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst[0], len / 2 * 2);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("uint4 copy (read %zu MiB, write same): %.3f ms  %.1f GB/s\n", len / 2 * 32 >> 20, best, 2.0 * (len / 2 * 32) / best / 1e6);
    }
#define RUNOPT(ARITH, NT, UN, GRID) { float best = 1e9; for (int it = 0; it < 4; ++it) { CK(hipEventRecord(e0)); \
        hipLaunchKernelGGL((k_fold_opt<ARITH, NT, UN>), dim3(GRID), dim3(256), 0, 0, src, dst[1], (uint32_t)q, T, sums); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
        CK(hipGetLastError()); printf("opt arith=%d nt=%d unroll=%d grid=%d: %.3f ms %.1f GB/s\n", ARITH, NT, UN, GRID, best, (double)q * 6 * 32 / best / 1e6); }
    RUNOPT(false, false, 1, 2048) RUNOPT(false, false, 1, 4096) RUNOPT(false, false, 1, 8192) RUNOPT(false, false, 1, 1024)
    RUNOPT(false, true, 1, 2048) RUNOPT(false, false, 2, 2048) RUNOPT(false, true, 2, 2048) RUNOPT(false, true, 2, 1024)
    RUNOPT(true, false, 1, 1280) RUNOPT(true, false, 1, 2560) RUNOPT(true, false, 1, 2048) RUNOPT(true, false, 1, 5120)
    RUNOPT(true, true, 1, 1280) RUNOPT(true, true, 1, 2560) RUNOPT(true, false, 2, 1280) RUNOPT(true, true, 2, 1280) RUNOPT(true, true, 2, 1024)
    std::vector<Fr> o0(len / 2), ov(len / 2);
    CK(hipMemcpy(o0.data(), dst[0], len / 2 * 32, hipMemcpyDeviceToHost));
    for (int v = 1; v < 4; ++v) {
        CK(hipMemcpy(ov.data(), dst[v], len / 2 * 32, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < len / 2; ++i) if (memcmp(&o0[i], &ov[i], 32)) { if (!bad) printf("  first mismatch v%d at %zu\n", v, i); ++bad; }
        printf("variant %d vs 0: %zu mismatches\n", v, bad);
    }
    // host check of variant 0 on a few entries
    size_t badh = 0;
    for (size_t i = 0; i < 1000; ++i) { Fr e = fr_add(h[i], mont_mul_portable(fr_sub(h[i + 2 * q], h[i]), rm)); if (memcmp(&e, &o0[i], 32)) ++badh; }
    printf("variant 0 vs host: %zu mismatches\n", badh);
    return 0;
}
