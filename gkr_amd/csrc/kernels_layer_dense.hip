// The DENSE-TABLE forms of the GKR layer sumcheck (prove_sumcheck_opt, rust/src/gkr/sumcheck.rs:36-156): rounds over the
// 2^{2k}-entry predicate tables A = add(z, ., .), M = mult(z, ., .).  Two documented paths stand on them: the device
// transcript (gkr_ctx_set_transcript(GKR_TRANSCRIPT_DEVICE), k_next <= 14: k_layer_round_b, k_layer_round, k_layer_fold,
// k_layer_round_hash) and the step-wise dense sessions of the trailing-variable split (gkr_layer_session_*: k_layer_round,
// k_layer_fold, k_layer_round_reduce, k_fold_small).  The default path of a layer sumcheck -- gate lists, segment passes,
// product passes -- is in kernels.hip and kernels_wide.hip; nothing here is on it.  (Round 5 retired what only a switch
// reached: the per-round kernels over U, V and the row, the dense tables with linear-time rounds, the resident kernel.)
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "kernels.h"
#include "dev_util.h"
#include "gate_seg.h"
#include "mimc7.h"

namespace gkr {

// ---------------------------------------------------------------------------
// GKR layer sumcheck (reference: prove_sumcheck_opt, sumcheck.rs:36-156) on
//   f(b, c) = A(b,c) (W(b) + W(c)) + M(b,c) W(b) W(c),   index = b * 2^k + c.
// A, M are streamed (canonical); the W copies are tiny, folded separately and
// kept in Montgomery form:
//   phase 0 (binding a b variable): p0 = Wb[row], p1 = Wb[row + hb], q0 = q1 = Wc[col]
//   phase 1 (binding a c variable): p0 = p1 = Wb[0],   q0 = Wc[col], q1 = Wc[col + h]
// Per pair (entry i, entry i + h):
//   c0   += a0 (p0 + q0) + m0 p0 q0
//   g(1) += a1 (p1 + q1) + m1 p1 q1
//   c2   += (a1 - a0)((p1 + q1) - (p0 + q0)) + (m1 - m0)(p1 q1 - p0 q0)
// (p q is linear in the bound variable because only one of p, q depends on it;
//  c1 = g(1) - c0 - c2.)
// FOLD: also fold A, M with the previous challenge while reading (fused pass).
// ---------------------------------------------------------------------------

struct PairTerms {
    Fr c0, g1, c2;
};

__device__ __forceinline__ PairTerms layer_pair(const Fr& a0, const Fr& a1, const Fr& m0, const Fr& m1, const Fr& p0,
                                                const Fr& p1, const Fr& q0, const Fr& q1) {
    // p*, q* are Montgomery: s = (p+q) R, pq = mont_mul(pR, qR) = pq R
    Fr s0 = fr_add(p0, q0), s1 = fr_add(p1, q1);
    Fr pq0 = mont_mul(p0, q0), pq1 = mont_mul(p1, q1);
    PairTerms t;
    t.c0 = fr_add(mont_mul(a0, s0), mont_mul(m0, pq0));
    t.g1 = fr_add(mont_mul(a1, s1), mont_mul(m1, pq1));
    t.c2 = fr_add(mont_mul(fr_sub(a1, a0), fr_sub(s1, s0)), mont_mul(fr_sub(m1, m0), fr_sub(pq1, pq0)));
    return t;
}

// Sums of one round over the current A, M (each 2h entries).
//   phase 0: h = hb * 2^k entries per half; row = i >> k, col = i & (2^k - 1)
//   phase 1: h entries per half, col = i
// grid = (blocks), partial per block
__global__ void __launch_bounds__(256) k_layer_round(const Fr* __restrict__ A, const Fr* __restrict__ M, uint32_t h,
                                                     uint32_t k, uint32_t phase, uint32_t hb,
                                                     const Fr* __restrict__ Wb, const Fr* __restrict__ Wc,
                                                     LayerPartial* __restrict__ partials, LayerBatch lb) {
    __shared__ Acc<9> smem[4 * 3];
    Acc<9> acc[3] = {acc_zero<9>(), acc_zero<9>(), acc_zero<9>()};
    A += blockIdx.y * lb.tstride;   // grid.y = proof of a batch
    M += blockIdx.y * lb.tstride;
    Wb += blockIdx.y * lb.wstride;
    Wc += blockIdx.y * lb.wstride;
    partials += blockIdx.y * lb.pstride;
    const uint32_t cmask = (1u << k) - 1;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < h; i += gridDim.x * blockDim.x) {
        Fr a0 = load_fr(A + i), a1 = load_fr(A + i + h);
        Fr m0 = load_fr(M + i), m1 = load_fr(M + i + h);
        Fr p0, p1, q0, q1;
        if (phase == 0) {
            const uint32_t row = i >> k, col = i & cmask;
            p0 = load_fr(Wb + row);
            p1 = load_fr(Wb + row + hb);
            q0 = load_fr(Wc + col);
            q1 = q0;
        } else {
            p0 = load_fr(Wb);
            p1 = p0;
            q0 = load_fr(Wc + i);
            q1 = load_fr(Wc + i + h);
        }
        PairTerms t = layer_pair(a0, a1, m0, m1, p0, p1, q0, q1);
        acc_add_fr(acc[0], t.c0);
        acc_add_fr(acc[1], t.g1);
        acc_add_fr(acc[2], t.c2);
    }
    block_sum<9, 3>(acc, smem);
    if (threadIdx.x == 0) {
        LayerPartial* p = partials + blockIdx.x;
        p->c0 = acc[0];
        p->g1 = acc[1];
        p->c2 = acc[2];
    }
}

// Fused b-phase round (the bandwidth-bound part of the layer sumcheck): fold the table with the
// previous challenge while reading it, write the folded table, and accumulate this round's sums
// in the same pass -- with the row-uniform factors pulled out so that every product has a
// wave-uniform multiplier and no modular reduction:
//     thread = (table T in {A, M}, column c, chunk of row pairs);  per row pair (b, b + hb):
//         U0 += y0 * p0,  U1 += y1 * p1,  D += (y1 - y0) * (p1 - p0)      (unreduced 544-bit sums)
//         S0 += y0,       S1 += y1                                        (T = A only)
//     at the end, with q = W(c):
//         T = A:  c0 += U0 + q S0,   g(1) += U1 + q S1,   c2 += D
//         T = M:  c0 += q U0,        g(1) += q U1,        c2 += q D
// because  a (p + q) + m p q  summed over rows = sum(a p) + q (sum(a) + sum(m p))  for a fixed
// column.  y0, y1 are the entries of rows b, b + hb of the folded table; p0 = Wb[b], p1 = Wb[b+hb]
// (Montgomery) are the same for the whole wave.  3 x 64 partial products per entry pair instead of
// 8 reduced products (1024) for both tables together.
// grid = (column blocks, row chunks, 2 tables), block = 256 columns
template <bool FOLD>
__global__ void __launch_bounds__(256) k_layer_round_b(const Fr* __restrict__ A_src, const Fr* __restrict__ M_src,
                                                       Fr* __restrict__ A_dst, Fr* __restrict__ M_dst, uint32_t hb,
                                                       uint32_t kc, uint32_t rows_per_chunk,
                                                       const FixedMul* __restrict__ rtab, const Fr* __restrict__ Wb,
                                                       const Fr* __restrict__ Wc, LayerPartial* __restrict__ partials,
                                                       LayerBatch lb, uint32_t chunks) {
    __shared__ Acc<9> smem[4 * 3];
    const bool is_m = blockIdx.z != 0;
    const uint32_t proof = blockIdx.y / chunks, chunk_id = blockIdx.y % chunks;   // grid.y = (proof, row chunk)
    const Fr* src = (is_m ? M_src : A_src) + proof * lb.tstride;
    Fr* dst = (is_m ? M_dst : A_dst) + proof * lb.tstride;
    Wb += proof * lb.wstride;
    Wc += proof * lb.wstride;
    rtab += proof;
    partials += proof * lb.pstride;
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ncols = 1u << kc;
    const size_t h = (size_t)hb << kc;   // entries per half of the folded table
    const uint32_t r0 = chunk_id * rows_per_chunk;
    uint32_t r1 = r0 + rows_per_chunk;
    if (r1 > hb) r1 = hb;
    FixedMul T;
    if (FOLD) T = *rtab;
    Lazy17 U0 = lazy_zero(), U1 = lazy_zero(), D = lazy_zero();
    Acc<9> S0 = acc_zero<9>(), S1 = acc_zero<9>();
    const bool active = col < ncols;
    for (uint32_t row = r0; row < r1; ++row) {
        // row-uniform multipliers; readfirstlane makes the uniformity explicit for the SGPR operands
        Fr p0 = Wb[row], p1 = Wb[row + hb];
        Fr dp = fr_sub(p1, p0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            p0.l[i] = __builtin_amdgcn_readfirstlane(p0.l[i]);
            p1.l[i] = __builtin_amdgcn_readfirstlane(p1.l[i]);
            dp.l[i] = __builtin_amdgcn_readfirstlane(dp.l[i]);
        }
        if (active) {
            const size_t idx = ((size_t)row << kc) | col;
            Fr y0, y1;
            if (FOLD) {
                Fr x0 = load_fr(src + idx), x1 = load_fr(src + idx + 2 * h);
                Fr x2 = load_fr(src + idx + h), x3 = load_fr(src + idx + 3 * h);
                fr_fold_fixed2(x0, x1, x2, x3, T, y0, y1);
                store_fr(dst + idx, y0);
                store_fr(dst + idx + h, y1);
            } else {
                y0 = load_fr(src + idx);
                y1 = load_fr(src + idx + h);
            }
            lazy_mac3_s(U0, y0, p0, U1, y1, p1, D, fr_sub(y1, y0), dp);
            if (!is_m) {
                acc_add_fr(S0, y0);
                acc_add_fr(S1, y1);
            }
        }
    }
    Acc<9> acc[3] = {acc_zero<9>(), acc_zero<9>(), acc_zero<9>()};
    if (active) {
        const Fr q = load_fr(Wc + col);
        const Fr u0 = lazy_reduce(U0), u1 = lazy_reduce(U1), d = lazy_reduce(D);
        Fr c0, g1, c2;
        if (is_m) {
            c0 = mont_mul(u0, q);
            g1 = mont_mul(u1, q);
            c2 = mont_mul(d, q);
        } else {
            c0 = fr_add(u0, mont_mul(acc_reduce(S0), q));
            g1 = fr_add(u1, mont_mul(acc_reduce(S1), q));
            c2 = d;
        }
        acc_add_fr(acc[0], c0);
        acc_add_fr(acc[1], g1);
        acc_add_fr(acc[2], c2);
    }
    block_sum<9, 3>(acc, smem);
    if (threadIdx.x == 0) {
        LayerPartial* p = partials + ((size_t)blockIdx.z * chunks + chunk_id) * gridDim.x + blockIdx.x;
        p->c0 = acc[0];
        p->g1 = acc[1];
        p->c2 = acc[2];
    }
}

// fold A and M in place with the challenge of the round just hashed: T[i] += r (T[i+h] - T[i])
__global__ void __launch_bounds__(256) k_layer_fold(Fr* __restrict__ A, Fr* __restrict__ M, uint32_t h,
                                                    const FixedMul* __restrict__ rtab, LayerBatch lb) {
    A += blockIdx.y * lb.tstride;   // grid.y = proof of a batch
    M += blockIdx.y * lb.tstride;
    const FixedMul T = rtab[blockIdx.y];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < h; i += gridDim.x * blockDim.x) {
        store_fr(A + i, fr_fold_fixed(load_fr(A + i), load_fr(A + i + h), T));
        store_fr(M + i, fr_fold_fixed(load_fr(M + i), load_fr(M + i + h), T));
    }
}

// One wave: total the partials, build the round vector with the reference's
// length (2 + dep of the variable), hash it, publish r, and fold the small W
// copy that depends on the bound variable (Montgomery fold: both operands
// Montgomery gives a Montgomery result).
__global__ void __launch_bounds__(64) k_layer_round_hash(const LayerPartial* __restrict__ partials, uint32_t nblk,
                                                         uint32_t round, uint32_t k, const uint32_t* __restrict__ dep,
                                                         const Fr* __restrict__ cts, Fr* __restrict__ out_coeffs,
                                                         uint32_t* __restrict__ out_len, Fr* __restrict__ out_r,
                                                         FixedMul* __restrict__ rtab, Fr* __restrict__ Wb,
                                                         Fr* __restrict__ Wc) {
    __shared__ Fr s_r;
    Acc<10> c0 = acc_zero<10>(), g1 = acc_zero<10>(), c2 = acc_zero<10>();
    for (uint32_t i = threadIdx.x; i < nblk; i += 64) {
        acc_add_acc(c0, partials[i].c0);
        acc_add_acc(g1, partials[i].g1);
        acc_add_acc(c2, partials[i].c2);
    }
    c0 = wave_sum(c0);
    g1 = wave_sum(g1);
    c2 = wave_sum(c2);
    if (threadIdx.x == 0) {
        Fr f0 = acc_reduce(c0), f1 = acc_reduce(g1), f2 = acc_reduce(c2);
        Fr lin = fr_sub(fr_sub(f1, f0), f2);
        const uint32_t len = 2u + (dep[round % k] ? 1u : 0u);
        Fr vec[3] = {f2, lin, f0};
        Fr r = mimc7_multi_hash(vec + (3 - len), (int)len, cts);
        Fr* oc = out_coeffs + (size_t)round * 3;
        oc[0] = (len == 3) ? f2 : fr_zero();
        oc[1] = lin;
        oc[2] = f0;
        out_len[round] = len;
        out_r[round] = r;
        Fr rm = to_mont(r);
        store_fixed_mul(rtab + round, r);
        s_r = rm;
    }
    __syncthreads();
    const Fr rm = s_r;
    // fold the W copy bound in this round: rounds 0..k-1 bind b (Wb), k..2k-1 bind c (Wc)
    Fr* W = (round < k) ? Wb : Wc;
    const uint32_t hw = 1u << (k - 1 - (round % k));
    // in place: lane i reads i and i + hw, writes i; a grid-stride loop inside one wave
    // would let a later iteration read a slot an earlier one wrote only if
    // i + hw < hw, which cannot happen
    for (uint32_t i = threadIdx.x; i < hw; i += 64) {
        Fr lo = load_fr(W + i), hi = load_fr(W + i + hw);
        store_fr(W + i, fr_fold(lo, hi, rm));
    }
}

// Host-transcript tail of a layer round: totals -> pinned host record.
__global__ void __launch_bounds__(64) k_layer_round_reduce(const LayerPartial* __restrict__ partials, uint32_t nblk,
                                                           LayerHostRec* __restrict__ host_rec, uint32_t ticket,
                                                           uint32_t pstride) {
    partials += (size_t)blockIdx.x * pstride;   // grid.x = proof of a batch
    host_rec += blockIdx.x;
    Acc<10> c0 = acc_zero<10>(), g1 = acc_zero<10>(), c2 = acc_zero<10>();
    for (uint32_t i = threadIdx.x; i < nblk; i += 64) {
        acc_add_acc(c0, partials[i].c0);
        acc_add_acc(g1, partials[i].g1);
        acc_add_acc(c2, partials[i].c2);
    }
    c0 = wave_sum(c0);
    g1 = wave_sum(g1);
    c2 = wave_sum(c2);
    if (threadIdx.x == 0) {
        host_rec->c0 = acc_reduce(c0);
        host_rec->g1 = acc_reduce(g1);
        host_rec->c2 = acc_reduce(c2);
        __hip_atomic_store(&host_rec->seq, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------------------
// b-phase in linear time (host transcript).  Summed over c, the layer polynomial is
//     h(b) = sum_c f(b, c) = W(b) U(b) + V(b),   U(b) = sum_c [a(b,c) + m(b,c) W(c)],   V(b) = sum_c a(b,c) W(c)
// with U, V multilinear in b -- so the k rounds that bind b are a sumcheck over three tables of 2^k entries
// (W, U, V) instead of k passes over the 2^{2k}-entry predicate tables: ONE pass over A, M builds U, V
// (k_layer_uv), the rounds run in one small block per proof (k_uv_round, which also publishes the host record:
// no reduce launch), and ONE more pass collapses the rows at the bound point u = (r_1..r_k),
//     a_u(c) = sum_b eq(u, b) a(b, c)   (k_layer_collapse_*),
// which is the single remaining row the c-phase kernels expect.  Same round polynomials: bit-exact.
// ---------------------------------------------------------------------------

// The three round sums of one proof (thread 0 holds them after block_sum) -> canonical values in the pinned host
// record.  The three reductions run side by side in lanes 0..2 of the first wave; lane 0's release store of the
// sequence number follows the wave's record stores in program order.
__device__ __forceinline__ void publish_round(const Acc<9> (&acc)[3], Acc<9>* tot /* shared, 3 */, LayerHostRec* r, uint32_t ticket) {
    if (threadIdx.x == 0) {
        tot[0] = acc[0];
        tot[1] = acc[1];
        tot[2] = acc[2];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const Fr v = acc_reduce(tot[threadIdx.x]);
        Fr* dst = threadIdx.x == 0 ? &r->c0 : (threadIdx.x == 1 ? &r->g1 : &r->c2);
        *dst = v;
        // lanes 1 and 2 order their own stores to host memory ahead of the barrier below; lane 0's release store of
        // the sequence number then follows all three in every memory model, not only because they share a wave
        __threadfence_system();
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&r->seq, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// In-place fold of a small Montgomery table (the W copy bound in this round):
// W[i] += r (W[i + hw] - W[i]).  One block; see k_layer_round_hash for why the
// strided in-place loop is safe.
__global__ void __launch_bounds__(256) k_fold_small(Fr* __restrict__ W, uint32_t hw, const FixedMul* __restrict__ rtab,
                                                    uint32_t wstride) {
    W += (size_t)blockIdx.x * wstride;   // grid.x = proof of a batch
    const FixedMul T = rtab[blockIdx.x];
    for (uint32_t base = 0; base < hw; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        Fr v = fr_zero();
        if (i < hw) v = fr_fold_fixed(load_fr(W + i), load_fr(W + i + hw), T);
        __syncthreads();   // all reads of this stripe done before any write lands in [0, hw)
        if (i < hw) store_fr(W + i, v);
    }
}


void launch_layer_round_reduce(const LayerPartial* partials, uint32_t nblk, LayerHostRec* host_rec, uint32_t ticket,
                               LayerBatch lb, hipStream_t s) {
    hipLaunchKernelGGL(k_layer_round_reduce, dim3(lb.batch), dim3(64), 0, s, partials, nblk, host_rec, ticket, lb.pstride);
}

void launch_fold_small(Fr* W, uint32_t hw, const FixedMul* rtab, LayerBatch lb, hipStream_t s) {
    hipLaunchKernelGGL(k_fold_small, dim3(lb.batch), dim3(256), 0, s, W, hw, rtab, (uint32_t)lb.wstride);
}








uint32_t layer_blocks(uint32_t h) { return blocks_for(h, kMaxLayerBlocks); }

void launch_layer_round(const Fr* A, const Fr* M, uint32_t h, uint32_t k, uint32_t phase, uint32_t hb, const Fr* Wb,
                        const Fr* Wc, uint32_t nblk, LayerPartial* partials, LayerBatch lb, hipStream_t s) {
    hipLaunchKernelGGL(k_layer_round, dim3(nblk, lb.batch), dim3(256), 0, s, A, M, h, k, phase, hb, Wb, Wc, partials, lb);
}

// returns the number of partials written (= blocks)
uint32_t launch_layer_round_b(bool fold, const Fr* A_src, const Fr* M_src, Fr* A_dst, Fr* M_dst, uint32_t hb, uint32_t kc,
                              const FixedMul* rtab, const Fr* Wb, const Fr* Wc, LayerPartial* partials, LayerBatch lb,
                              hipStream_t s) {
    const uint32_t col_blocks = ((1u << kc) + 255u) / 256u;
    // ~1024 blocks per table over the whole batch; every block gets at least one row pair
    uint32_t chunks = 1024u / (col_blocks * lb.batch);
    if (chunks < 1) chunks = 1;
    if (chunks > hb) chunks = hb;
    const uint32_t rows_per_chunk = (hb + chunks - 1) / chunks;
    chunks = (hb + rows_per_chunk - 1) / rows_per_chunk;
    dim3 grid(col_blocks, chunks * lb.batch, 2);
    if (fold)
        hipLaunchKernelGGL(k_layer_round_b<true>, grid, dim3(256), 0, s, A_src, M_src, A_dst, M_dst, hb, kc, rows_per_chunk,
                           rtab, Wb, Wc, partials, lb, chunks);
    else
        hipLaunchKernelGGL(k_layer_round_b<false>, grid, dim3(256), 0, s, A_src, M_src, A_dst, M_dst, hb, kc, rows_per_chunk,
                           rtab, Wb, Wc, partials, lb, chunks);
    return col_blocks * chunks * 2;   // partials per proof
}

void launch_layer_fold(Fr* A, Fr* M, uint32_t h, const FixedMul* rtab, LayerBatch lb, hipStream_t s) {
    hipLaunchKernelGGL(k_layer_fold, dim3(blocks_for(h, 4096 / lb.batch + 1), lb.batch), dim3(256), 0, s, A, M, h, rtab, lb);
}

void launch_layer_round_hash(const LayerPartial* partials, uint32_t nblk, uint32_t round, uint32_t k,
                             const uint32_t* dep, const Fr* cts, Fr* out_coeffs, uint32_t* out_len, Fr* out_r,
                             FixedMul* rtab, Fr* Wb, Fr* Wc, hipStream_t s) {
    hipLaunchKernelGGL(k_layer_round_hash, dim3(1), dim3(64), 0, s, partials, nblk, round, k, dep, cts, out_coeffs,
                       out_len, out_r, rtab, Wb, Wc);
}


}  // namespace gkr
