// The cross sums of a wide layer's product passes on the matrix cores (device code, gfx950): a pass with nothing pending -- a
// phase's first, or a later one whose pending fold ran through mfma_fold.h first (kernels.hip, launch_prod_pass).
//
// A product pass (kernels.hip, k_prod_cross) hands the host, for the next J = 3 rounds, the cross sums
//     m[a][b] = sum_{i < S} W[a S + i] * X[b S + i]      (a, b < 8;  S = 2^(m - 3) entries per sub-block)
// of two tables of 2^m entries -- 64 products of two 254-bit numbers per index i, 8.4 * 10^6 products for tables of 2^20
// entries: on v_mad_u64_u32 that pass runs at the chip's product rate (86 us) where its 96 MiB are 12 us of HBM.  Both
// operands vary with i, so the fold pass's trick (mfma_fold.h: one operand a per-sumcheck constant, its digit matrix the
// shared MFMA operand) does not apply as it stands; what does is the same exactness argument one level up.  Over the
// BYTES w_d, x_e of the entries (W = sum_d w_d 256^d, X = sum_e x_e 256^e)
//     m[a][b] = sum_{d, e < 32} 256^(d + e) * C_ab[d][e],      C_ab[d][e] = sum_i w_d(a S + i) * x_e(b S + i)
// and C_ab = A_a^T B_b is a (32 x S) by (S x 32) int8 product with int32 sums -- v_mfma_i32_32x32x32_i8 with the ENTRY
// INDEX as the K dimension: a k-step is 32 entries.  The operands need the bytes of one digit position of 16 consecutive
// entries side by side -- the transpose of how entries lie in memory -- so a block stages every k-step through LDS:
// 16-byte half entries in, 4 x 4 byte transposes across quads of lanes (v_perm_b32 + DPP), dwords out to
// [table][sub-block][digit][entry], 16-byte fragments back.  (Single-byte stores instead of the quad transposes: 64 us
// for tables of 2^20 entries, the LDS busy with 256 conflicted ds_write_b8 per k-step.)
// Bytes are unsigned, the instruction signed: w = w' + 128 (the top bit flipped), so
//     C[d][e] = C'[d][e] + 128 (A_d + B_e) + 128^2 n,      A_d = sum_i w'_d,  B_e = sum_i x'_e,
// the digit sums accumulated beside the MFMAs with v_dot4_i32_i8.  What the 512-bit sum needs of C are its anti-diagonal
// sums D_s = sum_{d + e = s} C[d][e] <= 32 n 255^2 < 2^32 (n <= 2048): every term is added modulo 2^32 -- the signed C',
// the window sums of A and B -- and the result is the exact D_s.  Every quantity is an exact integer; the sum
// sum_s D_s 256^s is reduced mod r once per block and pair (cross_reduce: the Montgomery form of W makes that the product's
// value) -- the same field elements as the VALU form, bit for bit.
//
// A block takes KC = 128 .. 2048 entries of every sub-block (|C'| <= KC * 2^14 < 2^31) and leaves one partial record
// (72 values) like a block of k_prod_cross; 512 threads = 8 waves, wave w: sub-blocks a in {2 (w & 3), +1} x b in
// {4 (w >> 2) .. +3} -- eight 32 x 32 int32 tiles, 128 accumulator registers.
#pragma once
#include "dev_util.h"
#include "fr32.h"
#include "kernels.h"
#include "mfma_fold.h"

namespace gkr {

constexpr uint32_t kCrossMinM = 15;          // tables of 2^15 entries and more (32 blocks of 128: 14 us where k_prod_cross<32> takes 15-33)
// entries of every sub-block per block: the fewest (of 128 .. 2048) that still give every block of the launch a CU of its own --
// a block's epilogue costs what ~10 k-steps do, so blocks should be long, but a second round of blocks costs a whole block's
// time (a table of 2^16 entries for seven proofs: 448 blocks of 128 took 37 us, 224 of 256 take 17)
inline uint32_t cross_pass_kc(uint32_t S, uint32_t batch) {
    for (uint32_t kc = 128u; kc < 2048u; kc <<= 1)
        if ((size_t)(S / kc) * batch <= 256u) return kc;
    return 2048u;
}

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

// bytes M[r][b] of four lanes' dwords (lane r of a quad) -> lane r holds M[0 .. 3][r]
__device__ __forceinline__ uint32_t cross_tr4(uint32_t own, uint32_t sel1, uint32_t sel2) {
    uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
    own = __builtin_amdgcn_perm(other, own, sel1);   // even lane: (o0, p0, o2, p2); odd: (p1, o1, p3, o3)
    other = (uint32_t)__builtin_amdgcn_mov_dpp((int)own, 0x4E, 0xF, 0xF, true);            // quad_perm [2, 3, 0, 1]
    return __builtin_amdgcn_perm(other, own, sel2);  // lanes 0, 1: (o0, o1, p0, p1); lanes 2, 3: (p2, p3, o2, o3)
}

// (lo, hi) += the same 96-bit number of the lane a DPP control names
template <int CTRL>
__device__ __forceinline__ void cross_add96(uint64_t& lo, uint32_t& hi) {
    const uint32_t o0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)lo, CTRL, 0xF, 0xF, true);
    const uint32_t o1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(lo >> 32), CTRL, 0xF, 0xF, true);
    const uint32_t o2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)hi, CTRL, 0xF, 0xF, true);
    const uint64_t n = lo + ((uint64_t)o0 | ((uint64_t)o1 << 32));
    hi = hi + o2 + (n < lo ? 1u : 0u);
    lo = n;
}

template <int CTRL>
__device__ __forceinline__ void cross_add64(uint64_t& v) {
    const uint32_t o0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)v, CTRL, 0xF, 0xF, true);
    const uint32_t o1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), CTRL, 0xF, 0xF, true);
    v += (uint64_t)o0 | ((uint64_t)o1 << 32);
}

// A block's sum of KC <= 2^11 products of values below r, as 17 limbs -> its value over 2^256 mod r, canonical: lazy_reduce's
// eight Montgomery steps leave (x + M r) / 2^256 < 2^263 + r, which the fold pass's short reduction finishes (a quotient
// estimate and one product with r: mfma_fold.h) where acc_reduce spends a whole product on the top limbs.
__device__ __forceinline__ Fr cross_reduce(const Lazy17& x) {
    constexpr uint32_t p[8] = GKR_MOD_LIMBS;
    uint32_t t[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) t[i] = x.l[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t m = t[k] * GKR_INV32;
        uint64_t a2 = t[k];
        uint32_t e2 = 0;
        mac96_s(a2, e2, m, p[0]);
        a2 = (a2 >> 32) | ((uint64_t)e2 << 32);
        e2 = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            a2 += t[k + j];
            mac96_s(a2, e2, m, p[j]);
            t[k + j] = (uint32_t)a2;
            a2 = (a2 >> 32) | ((uint64_t)e2 << 32);
            e2 = 0;
        }
#pragma unroll
        for (int j = k + 8; j < 17; ++j) {   // (no carry out of limb 16: the sum stays below 2^520 + 2^256 r)
            a2 += t[j];
            t[j] = (uint32_t)a2;
            a2 >>= 32;
        }
    }
    uint32_t hi[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) hi[i] = t[8 + i];
    return mf_reduce_274(hi);
}

// inclusive prefix sums over the wave's lanes (lane i: the sum of lanes 0 .. i)
__device__ __forceinline__ int32_t cross_wave_scan(int32_t v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int32_t o = __shfl_up(v, off, 64);
        v += lane >= (uint32_t)off ? o : 0;
    }
    return v;
}

// grid = (S / KC, batch), block = 512.  Wt (Montgomery), Xt, Yt: tables of 2^m entries per proof (stride wstride).
// partials: [proof][block][72] canonical values (m[a * 8 + b], then the eight sub-block sums of Y).
template <uint32_t KC>
__global__ void __launch_bounds__(512) k_prod_cross_mfma(const Fr* __restrict__ Wt, const Fr* __restrict__ Xt, const Fr* __restrict__ Yt,
                                                         uint32_t m, Fr* __restrict__ partials, uint32_t wstride) {
    static_assert(KC % 128u == 0 && KC <= 2048u, "a block's anti-diagonal sums must stay below 2^32: 32 * 2048 * 255^2 < 2^32");
    // [buffer][table][sub-block][digit][entry]: 2 x 2 x 8 x 32 x 32 bytes = 32 KB
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * 8 * 32 * 32];
    __shared__ uint32_t s_part[64][8][3];    // [pair][group of eight anti-diagonals]: 96-bit partial totals
    __shared__ Acc<9> s_y[8];                // the waves' totals of Y
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, c = lane & 31u, h = lane >> 5;
    const uint32_t proof = blockIdx.y, S = 1u << (m - 3u), i0 = blockIdx.x * KC;
    const Fr* W = Wt + (size_t)proof * wstride;
    const Fr* X = Xt + (size_t)proof * wstride;
    const Fr* Y = Yt + (size_t)proof * wstride;
    Fr* out = partials + ((size_t)proof * gridDim.x + blockIdx.x) * kProdRecValues;
    const uint32_t a0 = 2u * (wave & 3u), b0 = 4u * (wave >> 2);
    const uint32_t r4 = lane & 3u, q4 = c >> 2;
    const uint32_t sel1 = (r4 & 1u) ? 0x03070105u : 0x06020400u, sel2 = (r4 & 2u) ? 0x03020706u : 0x05040100u;
    const uint32_t frag = c * 32u + 16u * (h ^ ((c >> 3) & 1u));   // row c, the 16 entries of half h (where the stores put them)
    mf_v16i acc[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0;
    int32_t sumA[2] = {0, 0}, sumB[4] = {0, 0, 0, 0};
    uint64_t ycol[4] = {0, 0, 0, 0};   // Y: plain sums, the four 32-bit columns of this lane's 16-byte pieces added in 64 bits
    // this wave stages sub-block `wave` of W and of X: lane (c, h) takes the 16-byte half h of entry c of the k-step (and the
    // same piece of Y's sub-block `wave`)
    const char* gw = reinterpret_cast<const char*>(W + (size_t)wave * S + i0 + c) + 16u * h;
    const char* gx = reinterpret_cast<const char*>(X + (size_t)wave * S + i0 + c) + 16u * h;
    const char* gy = reinterpret_cast<const char*>(Y + (size_t)wave * S + i0 + c) + 16u * h;
    // PF k-steps' half entries in flight per lane; trips of kTrip steps, unrolled in full -- across a loop's back edge the
    // compiler waits for every load in flight (vmcnt(0)), which left a rolled loop without its prefetch
    constexpr uint32_t kSteps = KC / 32u, PF = 4, kTrip = kSteps < 32u ? kSteps : 32u;
    mf_v4u rw[PF], rx[PF], ry[PF];
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u) {
        rw[u] = __builtin_nontemporal_load(reinterpret_cast<const mf_v4u*>(gw + (size_t)u * 1024u));
        rx[u] = __builtin_nontemporal_load(reinterpret_cast<const mf_v4u*>(gx + (size_t)u * 1024u));
        ry[u] = __builtin_nontemporal_load(reinterpret_cast<const mf_v4u*>(gy + (size_t)u * 1024u));
    }
    for (uint32_t trip = 0; trip < kSteps / kTrip; ++trip)
#pragma unroll
    for (uint32_t t0 = 0; t0 < kTrip; t0 += PF)
#pragma unroll
    for (uint32_t u = 0; u < PF; ++u) {
        const uint32_t t = t0 + u;   // (the step within the trip)
        const mf_v4u cw = rw[u] ^ 0x80808080u, cx = rx[u] ^ 0x80808080u, cy = ry[u];
        if (kSteps == kTrip ? t + PF < kTrip : true) {   // (compile-time.  One trip: nothing past its end.  Several: the last
            // steps of the last trip load its last step again -- no branch, the counts of loads in flight stay static)
            const uint32_t tn = kSteps == kTrip ? t + PF : (trip * kTrip + t + PF < kSteps ? trip * kTrip + t + PF : kSteps - 1u);
            rw[u] = __builtin_nontemporal_load(reinterpret_cast<const mf_v4u*>(gw + (size_t)tn * 1024u));
            rx[u] = __builtin_nontemporal_load(reinterpret_cast<const mf_v4u*>(gx + (size_t)tn * 1024u));
            ry[u] = __builtin_nontemporal_load(reinterpret_cast<const mf_v4u*>(gy + (size_t)tn * 1024u));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) ycol[q] += cy[q];
        unsigned char* buf = lds + (t & 1u) * (2u * 8u * 1024u);
        // 4 x 4 byte transposes across the four lanes of a group (entries 4 q .. 4 q + 3): dword j of lane r becomes digit
        // 16 h + 4 j + r of those four entries -- two quad exchanges and two v_perm each, instead of sixteen byte stores
        uint32_t* tw = reinterpret_cast<uint32_t*>(buf + (0u * 8u + wave) * 1024u + (16u * h + r4) * 32u);
        uint32_t* tx = reinterpret_cast<uint32_t*>(buf + (1u * 8u + wave) * 1024u + (16u * h + r4) * 32u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // row 4 j of the half, dword q of the row at q ^ 4 [(row >> 3) & 1]: the fragment reads below are conflict-free
            const uint32_t col = q4 ^ (4u * ((uint32_t)(j >> 1) & 1u));
            tw[j * 32 + col] = cross_tr4(cw[j], sel1, sel2);
            tx[j * 32 + col] = cross_tr4(cx[j], sel1, sel2);
        }
        __syncthreads();
        mf_v4i fa[2], fb[4];
#pragma unroll
        for (int x = 0; x < 2; ++x) fa[x] = *reinterpret_cast<const mf_v4i*>(buf + (0u * 8u + a0 + x) * 1024u + frag);
#pragma unroll
        for (int y = 0; y < 4; ++y) fb[y] = *reinterpret_cast<const mf_v4i*>(buf + (1u * 8u + b0 + y) * 1024u + frag);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[x], fb[y], acc[x][y], 0, 0, 0);
        // the digit sums of the signed bytes of this wave's sub-blocks (lane (c, h): digit c, sixteen of the k-step's entries)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int q = 0; q < 4; ++q) sumA[x] = __builtin_amdgcn_sdot4(fa[x][q], 0x01010101, sumA[x], false);
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int q = 0; q < 4; ++q) sumB[y] = __builtin_amdgcn_sdot4(fb[y][q], 0x01010101, sumB[y], false);
        // (pinned here: left alone, the compiler sinks the whole chain of dot products into the epilogue's `h == 0` branch and
        // keeps every k-step's fragments alive for it -- in scratch memory)
#pragma unroll
        for (int x = 0; x < 2; ++x) asm volatile("" : "+v"(sumA[x]));
#pragma unroll
        for (int y = 0; y < 4; ++y) asm volatile("" : "+v"(sumB[y]));
        // (no second barrier: the next k-step writes the other buffer, and a wave passes that step's barrier only after it
        // has read this one)
        __builtin_amdgcn_sched_barrier(0);   // (the steps stay apart: moved across them, the loads' registers overflow the file)
    }
    // ---- epilogue ----
    {   // Y: the wave's total of its sub-block's pieces.  The columns first, within the wave's halves (half h of an entry is
        // limbs 4 h .. 4 h + 3): four DPP stages and one lane permute each, then lane 0 puts the two halves' columns together.
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cross_add64<0xB1>(ycol[q]);    // quad_perm [1, 0, 3, 2]
            cross_add64<0x4E>(ycol[q]);    // quad_perm [2, 3, 0, 1]
            cross_add64<0x141>(ycol[q]);   // row_half_mirror
            cross_add64<0x140>(ycol[q]);   // row_mirror: the sixteen lanes' total in all of them
            ycol[q] += __shfl_xor(ycol[q], 16, 64);
        }
        uint64_t upper[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) upper[q] = __shfl(ycol[q], 32, 64);
        if (lane == 0) {
            Acc<9> ysum;
            uint64_t carry = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                carry += q < 4 ? ycol[q] : upper[q - 4];
                ysum.l[q] = (uint32_t)carry;
                carry >>= 32;
            }
            ysum.l[8] = (uint32_t)carry;
            s_y[wave] = ysum;
        }
    }
    // The corrections' share of the anti-diagonal sums, modulo 2^32: lane s gets 128 sum_{d + e = s} A_d -- the sum of A over
    // the window max(0, s - 31) <= d <= min(31, s), a difference of two prefix sums -- likewise for B; the 128^2 n of every
    // product goes with A's.
    const uint32_t terms = lane < 32u ? lane + 1u : 63u - lane;   // products on anti-diagonal s = lane (none on 63)
    uint32_t corrA[2], corrB[4];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int32_t both = sumA[x] + __shfl_xor(sumA[x], 32, 64);   // (the halves hold sixteen entries each of every k-step)
        const int32_t digit = h ? 0 : both;
        const int32_t pre = cross_wave_scan(digit, lane);
        const int32_t below = __shfl(pre, (int)((lane - 32u) & 63u), 64);
        corrA[x] = 128u * (uint32_t)(pre - (lane >= 32u ? below : 0)) + terms * (KC << 14);
    }
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        const int32_t both = sumB[y] + __shfl_xor(sumB[y], 32, 64);
        const int32_t digit = h ? 0 : both;
        const int32_t pre = cross_wave_scan(digit, lane);
        const int32_t below = __shfl(pre, (int)((lane - 32u) & 63u), 64);
        corrB[y] = 128u * (uint32_t)(pre - (lane >= 32u ? below : 0));
    }
    // The 64 sums sum_s D_s 256^s, a pair at a time.  Accumulator register r of lane (c, h) is C'[d][e = c] with
    // d = 8 (r >> 2) + (r & 3) + 4 h, a term of anti-diagonal d + c.  One lane permute per register moves all 64 of them:
    // lane (l, h) takes the term of row d whose diagonal is l modulo 32 -- column (l - d) mod 32, diagonal l if l >= d, else
    // l + 32 -- so a pair costs 16 permutes, every one of them full; the two halves then exchange what they hold of each
    // other's diagonal, and lane s has D_s.  No LDS round trip.
    const int32_t lh = (int32_t)c - 4 * (int32_t)h;   // l - 4 h: row d = d0 + 4 h has l >= d where lh >= d0
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            uint32_t low = 0, all = 0;   // this lane's terms of diagonal l, and of both l and l + 32
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int32_t d0 = 8 * (r >> 2) + (r & 3);
                const uint32_t src = ((uint32_t)(lh - d0) & 31u) | (32u * h);
                const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), acc[x][y][r]);
                low += lh >= d0 ? v : 0u;
                all += v;
            }
            const uint32_t give = h ? low : all - low;                  // the other half's diagonal
            const uint32_t D = (h ? all - low : low) + __shfl_xor(give, 32, 64) + corrA[x] + corrB[y];
            const uint32_t sh = 8u * (lane & 7u);
            uint64_t plo = (uint64_t)D << sh;
            uint32_t phi = sh > 32u ? D >> (64u - sh) : 0u;
            // (the eight lanes' total in all of them: quad exchanges and a mirror of the eight, on the VALU -- the LDS pipe
            // is what this loop is bound by, 16 permutes per pair and wave)
            cross_add96<0xB1>(plo, phi);    // quad_perm [1, 0, 3, 2]
            cross_add96<0x4E>(plo, phi);    // quad_perm [2, 3, 0, 1]
            cross_add96<0x141>(plo, phi);   // row_half_mirror: lane i of the eight <-> lane 7 - i (the other quad's total)
            if ((lane & 7u) == 0u) {
                uint32_t* pp = s_part[(a0 + x) * 8u + b0 + y][lane >> 3];
                pp[0] = (uint32_t)plo;
                pp[1] = (uint32_t)(plo >> 32);
                pp[2] = phi;
            }
        }
    }
    __syncthreads();
    if (tid < 64u) {   // the pair's 17 limbs: 64-bit word q = group q's low word + group q - 1's high part + the carry so far
        Lazy17 v;
        uint64_t carry = 0, prev_hi = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint64_t qlo = (uint64_t)s_part[tid][q][0] | ((uint64_t)s_part[tid][q][1] << 32);
            const uint64_t w1 = qlo + prev_hi;
            const uint64_t c1 = w1 < qlo ? 1ull : 0ull;
            const uint64_t w2 = w1 + carry;
            const uint64_t c2 = w2 < w1 ? 1ull : 0ull;
            v.l[2 * q] = (uint32_t)w2;
            v.l[2 * q + 1] = (uint32_t)(w2 >> 32);
            carry = c1 + c2;
            prev_hi = s_part[tid][q][2];
        }
        v.l[16] = (uint32_t)(prev_hi + carry);   // (the sum is below KC 2^512)
        store_fr(out + tid, cross_reduce(v));
    } else if (tid < 128u && lane < 8u) {   // (a second wave, beside the first)
        store_fr(out + 64 + lane, mf_reduce_274(s_y[lane].l));   // (a sum of KC values below r: below 2^265)
    }
}

#endif

}  // namespace gkr
