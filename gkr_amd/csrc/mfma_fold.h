// Multi-round fold pass with the multiplications on the matrix cores (device code, gfx950).
//
//     T'[i] = sum_{b < 2^J} w_b * T[b * S + i]  (mod p)
//
// is a sum of 2^J products of a 256-bit table entry with a per-sumcheck constant.  Written over the BYTES
// a_{b,j} of the entries (T[b*S+i] = sum_j a_{b,j} 2^(8j)) it is
//     T'[i] = sum_{b,j} a_{b,j} * R_{b,j},      R_{b,j} = w_b * 2^(8j) mod p   (32 * 2^J constants per sumcheck)
// and with every R in signed radix-256 digits d_m(R) in [-128, 127] (32 digits: R < p < 2^254)
//     T'[i] = sum_m 2^(8m) * C_m(i),            C_m(i) = sum_{b,j} d_m(R_{b,j}) * a_{b,j}(i)
// i.e. C = D^T X: a (32 x K) by (K x entries) int8 product with int32 sums, K = 32 * 2^J -- what
// v_mfma_i32_32x32x32_i8 computes, 32 entries per instruction and k-step (one k-step = the 32 bytes of ONE
// source entry, so the B operand of a lane is a 16-byte half of the entry it loaded: no data movement).
// The VALU is left with: flipping the sign bit of the loaded bytes (a = a_signed + 128; the "+128" part is a
// per-sumcheck constant folded into the accumulators' start value), joining the 32 column sums into limbs,
// and ONE reduction of a < 2^274 value per output (quotient estimate from the top 50 bits, one product with p).
// That is ~250 VALU instructions per output instead of ~2300 for eight 256-bit products on v_mad_u64_u32,
// which takes the pass from instruction-issue-bound to HBM-bound -- and makes J = 4, 5 affordable (fewer
// passes, less traffic).  Same field elements: bit-exact.
#pragma once
#include "fr32.h"

namespace gkr {

constexpr int kMfmaMaxJ = 5;   // variables one matrix-core pass can bind

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef int mf_v16i __attribute__((ext_vector_type(16)));
typedef uint32_t mf_v2u __attribute__((ext_vector_type(2)));
typedef uint32_t mf_v4u __attribute__((ext_vector_type(4)));

// Column start values.  |sum over K rows of digit * signed byte| < 2^(20+J) =: Bias, so Bias makes every
// column sum positive; the bytes z_m of (-Bias * (2^256 - 1) / 255) mod p take the bias out again modulo p:
// sum_m (Bias + z_m) 2^(8m) = 0 (mod p).  Row J-1 holds Bias + z_m for J = 1 .. 5.
__constant__ const uint32_t kMfmaColumnBias[kMfmaMaxJ][32] = {
    {0x2000f3, 0x2000a9, 0x2000e0, 0x2000af, 0x2000bc, 0x2000a4, 0x20007a, 0x20007a, 0x200093, 0x20009f, 0x200035,
     0x2000d6, 0x200006, 0x200035, 0x20007d, 0x20004d, 0x2000d7, 0x20003f, 0x2000fc, 0x200030, 0x20009b, 0x200087,
     0x200017, 0x2000d2, 0x200026, 0x200072, 0x2000b6, 0x200064, 0x200095, 0x20002d, 0x2000e0, 0x200006},
    {0x4000e6, 0x400053, 0x4000c1, 0x40005f, 0x400079, 0x400049, 0x4000f5, 0x4000f4, 0x400026, 0x40003f, 0x40006b,
     0x4000ac, 0x40000d, 0x40006a, 0x4000fa, 0x40009a, 0x4000ae, 0x40007f, 0x4000f8, 0x400061, 0x400036, 0x40000f,
     0x40002f, 0x4000a4, 0x40004d, 0x4000e4, 0x40006c, 0x4000c9, 0x40002a, 0x40005b, 0x4000c0, 0x40000d},
    {0x8000cc, 0x8000a7, 0x800082, 0x8000bf, 0x8000f2, 0x800092, 0x8000ea, 0x8000e9, 0x80004d, 0x80007e, 0x8000d6,
     0x800058, 0x80001b, 0x8000d4, 0x8000f4, 0x800035, 0x80005d, 0x8000ff, 0x8000f0, 0x8000c3, 0x80006c, 0x80001e,
     0x80005e, 0x800048, 0x80009b, 0x8000c8, 0x8000d9, 0x800092, 0x800055, 0x8000b6, 0x800080, 0x80001b},
    {0x1000097, 0x100004f, 0x1000005, 0x100008f, 0x1000051, 0x1000030, 0x10000f3, 0x100008f, 0x100000a, 0x100008c, 0x10000f3,
     0x1000037, 0x10000ee, 0x10000bf, 0x10000b5, 0x1000043, 0x100005d, 0x10000a6, 0x1000060, 0x1000006, 0x1000023, 0x10000f7,
     0x100006b, 0x10000d8, 0x100000c, 0x10000f1, 0x1000081, 0x1000044, 0x1000038, 0x100001e, 0x100009d, 0x1000006},
    {0x200002e, 0x200009f, 0x200000a, 0x200001e, 0x20000a3, 0x2000060, 0x20000e6, 0x200001f, 0x2000015, 0x2000018, 0x20000e7,
     0x200006f, 0x20000dc, 0x200007f, 0x200006b, 0x2000087, 0x20000ba, 0x200004c, 0x20000c1, 0x200000c, 0x2000046, 0x20000ee,
     0x20000d7, 0x20000b0, 0x2000019, 0x20000e2, 0x2000003, 0x2000089, 0x2000070, 0x200003c, 0x200003a, 0x200000d}};

// x = sum l[i] 2^(32 i) < 2^274  ->  x mod p, canonical.  With t = x >> 224 (50 bits), D = (p >> 224) + 1 and
// mu = floor(2^61 / D), q = floor(t * mu / 2^61) never exceeds x / p and falls short of it by less than
// 2^21 * 2e-9 + 1 (floor), so x - q p is in [0, 2p): one conditional subtraction.
__device__ __forceinline__ Fr mf_reduce_274(const uint32_t (&l)[9]) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    constexpr uint32_t mu = 0xa948e8c4u;   // floor(2^61 / 811880051)
    const uint64_t t1 = (uint64_t)l[7] * mu, t2 = (uint64_t)l[8] * mu;
    const uint32_t q = (uint32_t)((t2 + (t1 >> 32)) >> 29);
    // r = (x - q p) mod 2^256
    uint32_t r[8];
    uint64_t prod = 0;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        prod += (uint64_t)q * p[i];
        const uint32_t s = (uint32_t)prod;
        prod >>= 32;
        const uint64_t d = (uint64_t)l[i] - s - borrow;
        r[i] = (uint32_t)d;
        borrow = (uint32_t)(d >> 63);
    }
    return cond_sub_mod(r);
}

// What a pass needs per sumcheck besides the table: the digit matrix in A-fragment order,
//     digits[((b * 2 + hh) * 32 + m) * 16 + t] = d_m(R_{b, 16 hh + t}),
// and the accumulators' start values, start[h][r] for lane half h, accumulator register r:
//     128 * (column sum of the digits) + Bias + z   for the column that register holds.
// Built once per (sumcheck, pass) by mfma_plan_block, read by every block of the pass.
struct alignas(16) MfmaFoldPlan {
    unsigned char digits[32 * 32 * (1 << kMfmaMaxJ)];
    int32_t start[2][16];
};

// One 256-thread block builds the plan of one sumcheck from its 2^JIN Montgomery-form weights.
template <int JIN>
__device__ __forceinline__ void mfma_plan_block(const Fr* __restrict__ w_mont, MfmaFoldPlan* __restrict__ plan,
                                                unsigned char* lds /* 32 * K bytes, 16-aligned */) {
    constexpr int NB = 1 << JIN;      // source entries per output = k-steps
    constexpr int K = 32 * NB;        // rows of the digit matrix
    const uint32_t tid = threadIdx.x, lane = tid & 63u, c = lane & 31u, h = lane >> 5;
    // the weights may live in pinned host memory: 4-byte reads by as few lanes as it takes, not 32 bytes per thread
    __shared__ uint32_t w_lds[8 * NB];
    for (uint32_t i = tid; i < (uint32_t)(8 * NB); i += blockDim.x) w_lds[i] = reinterpret_cast<const uint32_t*>(w_mont)[i];
    __syncthreads();
    for (uint32_t row = tid; row < (uint32_t)K; row += blockDim.x) {
        const uint32_t b = row >> 5, j = row & 31u;
        Fr sh = fr_zero(), wb;
        sh.l[j >> 2] = 1u << (8u * (j & 3u));                    // 2^(8j), canonical (< 2^248)
#pragma unroll
        for (int i = 0; i < 8; ++i) wb.l[i] = w_lds[b * 8u + (uint32_t)i];
        const Fr R = mont_mul(wb, sh);                            // Montgomery * canonical = canonical product
        uint32_t carry = 0;
        const uint32_t base = ((b * 2u + (j >> 4)) * 32u) * 16u + (j & 15u);
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            uint32_t v = ((R.l[m >> 2] >> (8 * (m & 3))) & 0xffu) + carry;
            carry = (v + 128u) >> 8;                              // v >= 128 -> digit v - 256, carry 1
            lds[base + (uint32_t)m * 16u] = (unsigned char)v;     // low byte is the two's-complement digit
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < (uint32_t)(32 * K / 16); i += blockDim.x)
        reinterpret_cast<mf_v4u*>(plan->digits)[i] = reinterpret_cast<const mf_v4u*>(lds)[i];
    if (tid < 64u) {
        // 128 * (column sums) = 2 * (digits x bytes of 64), on the matrix cores like the pass itself
        mf_v16i t = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const mf_v4i x64 = {0x40404040, 0x40404040, 0x40404040, 0x40404040};
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const mf_v4i wf = *reinterpret_cast<const mf_v4i*>(lds + ((b * 2 + h) * 32 + c) * 16);
            t = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, x64, t, 0, 0, 0);
        }
        if (c == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                plan->start[h][r] = 2 * t[r] + (int)kMfmaColumnBias[JIN - 1][(r & 3) + 8 * (r >> 2) + 4 * h];
        }
    }
}

// two finished accumulator tiles -> this lane's output entry (lanes 0-31: tile 0, lanes 32-63: tile 1).
// Registers 4g .. 4g+3 of a lane are digit columns 8g + 4h + (0..3): limb 2g + h of the entry in the lane's
// column.  Join them (all positive, < 2^27, so a limb's four columns stay below 2^52), then trade halves so
// that every lane holds one whole entry.
__device__ __forceinline__ Fr mf_finish(const mf_v16i& a0, const mf_v16i& a1) {
    uint64_t ev[4], od[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint64_t v0 = (uint64_t)(uint32_t)a0[4 * g] + ((uint64_t)(uint32_t)a0[4 * g + 1] << 8) +
                            ((uint64_t)(uint32_t)a0[4 * g + 2] << 16) + ((uint64_t)(uint32_t)a0[4 * g + 3] << 24);
        const uint64_t v1 = (uint64_t)(uint32_t)a1[4 * g] + ((uint64_t)(uint32_t)a1[4 * g + 1] << 8) +
                            ((uint64_t)(uint32_t)a1[4 * g + 2] << 16) + ((uint64_t)(uint32_t)a1[4 * g + 3] << 24);
        const mf_v2u lo = __builtin_amdgcn_permlane32_swap((uint32_t)v0, (uint32_t)v1, false, false);
        const mf_v2u hi = __builtin_amdgcn_permlane32_swap((uint32_t)(v0 >> 32), (uint32_t)(v1 >> 32), false, false);
        ev[g] = (uint64_t)lo.x | ((uint64_t)hi.x << 32);   // limb 2g
        od[g] = (uint64_t)lo.y | ((uint64_t)hi.y << 32);   // limb 2g + 1
    }
    uint32_t l[9];
    uint64_t run = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        run += ev[g];
        l[2 * g] = (uint32_t)run;
        run >>= 32;
        run += od[g];
        l[2 * g + 1] = (uint32_t)run;
        run >>= 32;
    }
    l[8] = (uint32_t)run;
    return mf_reduce_274(l);
}

// One pass for a (sumcheck, chunk) block: same contract as k_mle_multifold (kernels.hip).  256 threads = 4 waves;
// a wave turns 64 consecutive output entries per iteration (two 32-entry MFMA tiles); (end - begin) % 64 == 0.
// The 2^JIN source entries of an output are taken in stages of (at most) eight k-steps: the loads of the next
// stage (of this or the next iteration) are issued before the current stage's arithmetic, so a wave keeps
// 2 x 16 KB in flight whatever JIN is.  lds: 32 * 32 * 2^JIN bytes for JIN > 2 (the digit matrix; it stays
// in registers for JIN <= 2), unused otherwise.
// `rot`: each block starts at its own iteration and wraps around, so the blocks of a launch do not walk
// their chunks in step.
template <int JIN>
__device__ __forceinline__ void mfma_multifold_block(const Fr* __restrict__ s, Fr* __restrict__ d, uint32_t S,
                                                     const MfmaFoldPlan* __restrict__ plan, uint32_t begin, uint32_t end,
                                                     uint32_t rot, Acc<9>& acc_out, unsigned char* lds) {
    constexpr int NB = 1 << JIN;
    constexpr int KS = NB < 8 ? NB : 8;     // k-steps per stage
    constexpr int NST = NB / KS;            // stages per iteration: 1, 2 or 4
    constexpr bool kRegs = NB <= 4;         // digit matrix in registers (from eight k-steps on, LDS: the registers
                                            // are needed for the two load buffers)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, c = lane & 31u, h = lane >> 5;
    mf_v4i wf[kRegs ? NB : 1];
    if constexpr (kRegs) {
#pragma unroll
        for (int b = 0; b < NB; ++b) wf[b] = *reinterpret_cast<const mf_v4i*>(plan->digits + ((b * 2 + h) * 32 + c) * 16);
    } else {
        for (uint32_t i = tid; i < (uint32_t)(32 * 32 * NB / 16); i += blockDim.x)
            reinterpret_cast<mf_v4u*>(lds)[i] = reinterpret_cast<const mf_v4u*>(plan->digits)[i];
        __syncthreads();
    }
    mf_v16i init;
#pragma unroll
    for (int r = 0; r < 16; ++r) init[r] = plan->start[h][r];

    const uint32_t wave = tid >> 6;
    // addresses as (wave-uniform stream base) + (one 32-bit lane offset) + (immediate): the 2^JIN streams cost
    // scalar registers, not a 64-bit vector address each
    auto load_stage = [&](uint32_t e0, int stage, mf_v4u (&x)[2][KS]) {
        const uint32_t voff = (e0 + c) * 32u + 16u * h;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
#if defined(MF_STREAM_PAD)   // experiment (tools/ubench_mfma_fold.hip): streams MF_STREAM_PAD entries further apart
            const char* base = reinterpret_cast<const char*>(s + (size_t)(stage * KS + k) * (S + MF_STREAM_PAD));
#else
            const char* base = reinterpret_cast<const char*>(s + (size_t)(stage * KS + k) * S);
#endif
#pragma unroll
            for (int t = 0; t < 2; ++t)
                x[t][k] = __builtin_nontemporal_load(reinterpret_cast<const mf_v4u*>(base + voff + 1024u * t));
        }
    };
    auto mac_stage = [&](int stage, const mf_v4u (&x)[2][KS], mf_v16i& a0, mf_v16i& a1) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            mf_v4i w;
            if constexpr (kRegs)
                w = wf[stage * KS + k];
            else
                w = *reinterpret_cast<const mf_v4i*>(lds + (((stage * KS + k) * 2 + h) * 32 + c) * 16);
            const mf_v4u f0 = x[0][k] ^ 0x80808080u, f1 = x[1][k] ^ 0x80808080u;
            a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, (mf_v4i)f0, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, (mf_v4i)f1, a1, 0, 0, 0);
        }
    };
    auto store_entry = [&](uint32_t e0, const mf_v16i& a0, const mf_v16i& a1) {
        const Fr y = mf_finish(a0, a1);
        mf_v4u* o = reinterpret_cast<mf_v4u*>(d + e0 + 32u * h + c);
        const mf_v4u y0 = {y.l[0], y.l[1], y.l[2], y.l[3]}, y1 = {y.l[4], y.l[5], y.l[6], y.l[7]};
        __builtin_nontemporal_store(y0, o);
        __builtin_nontemporal_store(y1, o + 1);
        acc_add_fr(acc_out, y);
    };

    const uint32_t span = end - begin;
    const uint32_t iters = (span + 255u) / 256u;
    const uint32_t woff = wave * 64u;
    if (woff >= span) return;                             // chunks shorter than 256 leave waves idle
    uint32_t cur = rot % iters;
    auto next_pos = [&]() {   // entry offset of the iteration after the current one (wraps around the chunk)
        cur = cur + 1u == iters ? 0u : cur + 1u;
        return begin + cur * 256u + woff;
    };
    // (the scheduling barriers keep the compiler from hoisting a buffer's next loads above its last use, which
    // would cost a third and fourth buffer in registers)
    mf_v4u xa[2][KS], xb[2][KS];
    uint32_t e0 = begin + cur * 256u + woff;
    load_stage(e0, 0, xa);
    if constexpr (NST == 1) {
        // one stage per iteration: iterations alternate between the two buffers
        for (uint32_t it = 0; it < iters; it += 2u) {
            const bool more1 = it + 1u < iters;
            const uint32_t e1 = next_pos();
            if (more1) load_stage(e1, 0, xb);
            {
                mf_v16i a0 = init, a1 = init;
                mac_stage(0, xa, a0, a1);
                store_entry(e0, a0, a1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!more1) break;
            const bool more2 = it + 2u < iters;
            e0 = next_pos();
            if (more2) load_stage(e0, 0, xa);
            {
                mf_v16i a0 = init, a1 = init;
                mac_stage(0, xb, a0, a1);
                store_entry(e1, a0, a1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // stages alternate between the two buffers; NST is even, so every iteration starts in xa
        for (uint32_t it = 0; it < iters; ++it) {
            const bool more = it + 1u < iters;
            const uint32_t e1 = next_pos();
            mf_v16i a0 = init, a1 = init;
#pragma unroll
            for (int st = 0; st < NST; st += 2) {
                load_stage(e0, st + 1, xb);
                mac_stage(st, xa, a0, a1);
                __builtin_amdgcn_sched_barrier(0);
                if (st + 2 < NST)
                    load_stage(e0, st + 2, xa);
                else if (more)
                    load_stage(e1, 0, xa);
                mac_stage(st + 1, xb, a0, a1);
                __builtin_amdgcn_sched_barrier(0);
            }
            store_entry(e0, a0, a1);
            e0 = e1;
        }
    }
}

#endif

}  // namespace gkr
