// The EARLIER forms of the GKR layer sumcheck (prove_sumcheck_opt, rust/src/gkr/sumcheck.rs:36-156), kept because paths still
// stand on them -- the device transcript (rounds over the dense 2^{2k}-entry predicate tables: k_layer_round, k_layer_round_b,
// k_layer_fold, k_layer_round_hash), the dense step-wise sessions of the trailing-variable split, layers too dense for gate
// lists (k_layer_uv, k_layer_collapse_*), the per-round schedule (GKR_LAYER_PER_ROUND: k_uv_round*, k_c_round*), the resident
// kernel (GKR_LAYER_PERSISTENT: k_layer_persistent) -- and because every one of them is a parity-tested second derivation of
// the same transcript.  The DEFAULT path of a layer sumcheck is in kernels.hip (gate lists, segment passes, product passes) and
// kernels_wide.hip (wide layers); nothing here is on it.
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

// grid = (2^k rows, batch), block = 256: U[row], V[row] of one proof.  W: the Montgomery copy over columns.
__global__ void __launch_bounds__(256) k_layer_uv(const Fr* __restrict__ A, const Fr* __restrict__ M, const Fr* __restrict__ W,
                                                  Fr* __restrict__ U, Fr* __restrict__ V, uint32_t k, LayerBatch lb) {
    __shared__ Acc<9> smem[4 * 2];
    const uint32_t row = blockIdx.x, cols = 1u << k;
    const Fr* a = A + blockIdx.y * lb.tstride + (size_t)row * cols;
    const Fr* m = M + blockIdx.y * lb.tstride + (size_t)row * cols;
    const Fr* w = W + blockIdx.y * lb.wstride;
    Acc<9> acc[2] = {acc_zero<9>(), acc_zero<9>()};
    if (cols >= 4u * blockDim.x) {
        // several entries per thread: unreduced dot products, one reduction each at the end
        Lazy17 mw = lazy_zero(), aw = lazy_zero();
        for (uint32_t c = threadIdx.x; c < cols; c += blockDim.x) {
            const Fr av = load_fr(a + c), mv = load_fr(m + c), q = load_fr(w + c);
            acc_add_fr(acc[0], av);
            lazy_mac_v(mw, mv, q);
            lazy_mac_v(aw, av, q);
        }
        acc_add_fr(acc[0], lazy_reduce(mw));
        acc_add_fr(acc[1], lazy_reduce(aw));
    } else {
        // short rows: a reduction per thread would cost more than the products it saves
        for (uint32_t c = threadIdx.x; c < cols; c += blockDim.x) {
            const Fr av = load_fr(a + c), mv = load_fr(m + c), q = load_fr(w + c);
            acc_add_fr(acc[0], av);
            acc_add_fr(acc[0], mont_mul(mv, q));
            acc_add_fr(acc[1], mont_mul(av, q));
        }
    }
    block_sum<9, 2>(acc, smem);
    if (threadIdx.x == 0) {
        store_fr(U + blockIdx.y * lb.wstride + row, acc_reduce(acc[0]));
        store_fr(V + blockIdx.y * lb.wstride + row, acc_reduce(acc[1]));
    }
}

// One b-round of one proof: (FOLD) bind the previous variable in U, V with that round's challenge, then
//     c0 = sum W_lo U_lo + V_lo,   g(1) = sum W_hi U_hi + V_hi,   c2 = sum (W_hi - W_lo)(U_hi - U_lo)
// over the h pairs (i, i + h); W (Montgomery) is folded along with U, V.  grid = (batch), block = 256.
template <bool FOLD>
__global__ void __launch_bounds__(256) k_uv_round(Fr* __restrict__ W, Fr* __restrict__ U, Fr* __restrict__ V, uint32_t h,
                                                  const FixedMul* __restrict__ rtab, LayerHostRec* __restrict__ host_rec,
                                                  uint32_t ticket, uint32_t wstride) {
    __shared__ Acc<9> smem[4 * 3];
    W += (size_t)blockIdx.x * wstride;
    U += (size_t)blockIdx.x * wstride;
    V += (size_t)blockIdx.x * wstride;
    if (FOLD) {
        const FixedMul T = rtab[blockIdx.x];
        const uint32_t hw = 2u * h;   // size of the tables after this fold
        for (uint32_t base = 0; base < hw; base += blockDim.x) {
            const uint32_t i = base + threadIdx.x;
            Fr u = fr_zero(), v = fr_zero(), w = fr_zero();
            if (i < hw) {
                u = fr_fold_fixed(load_fr(U + i), load_fr(U + i + hw), T);
                v = fr_fold_fixed(load_fr(V + i), load_fr(V + i + hw), T);
                w = fr_fold_fixed(load_fr(W + i), load_fr(W + i + hw), T);
            }
            __syncthreads();   // all reads of this stripe done before any write lands in [0, hw)
            if (i < hw) {
                store_fr(U + i, u);
                store_fr(V + i, v);
                store_fr(W + i, w);
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    Acc<9> acc[3] = {acc_zero<9>(), acc_zero<9>(), acc_zero<9>()};
    if (h * 4u <= blockDim.x) {
        // few pairs: one product per thread
        const uint32_t i = threadIdx.x >> 2, part = threadIdx.x & 3u;
        if (i < h && part < 3u) {
            if (part == 0) {
                acc_add_fr(acc[0], mont_mul(load_fr(U + i), load_fr(W + i)));
                acc_add_fr(acc[0], load_fr(V + i));
            } else if (part == 1) {
                acc_add_fr(acc[1], mont_mul(load_fr(U + i + h), load_fr(W + i + h)));
                acc_add_fr(acc[1], load_fr(V + i + h));
            } else {
                acc_add_fr(acc[2], mont_mul(fr_sub(load_fr(U + i + h), load_fr(U + i)), fr_sub(load_fr(W + i + h), load_fr(W + i))));
            }
        }
    } else {
        for (uint32_t i = threadIdx.x; i < h; i += blockDim.x) {
            const Fr wl = load_fr(W + i), wh = load_fr(W + i + h);
            const Fr ul = load_fr(U + i), uh = load_fr(U + i + h);
            acc_add_fr(acc[0], mont_mul(ul, wl));
            acc_add_fr(acc[0], load_fr(V + i));
            acc_add_fr(acc[1], mont_mul(uh, wh));
            acc_add_fr(acc[1], load_fr(V + i + h));
            acc_add_fr(acc[2], mont_mul(fr_sub(uh, ul), fr_sub(wh, wl)));
        }
    }
    block_sum<9, 3>(acc, smem);
    __syncthreads();   // smem is reused for the totals
    publish_round(acc, smem, host_rec + blockIdx.x, ticket);
}

// One c-round of one proof on the single remaining row (2h entries of A, M; W over c in Montgomery form;
// p = W(u), Montgomery): (FOLD) bind the previous variable in A, M, Wc, then the round's sums, published to
// the host record -- one launch per round instead of fold + fold + round + reduce.  grid = (batch), block = 256.
template <bool FOLD>
__global__ void __launch_bounds__(256) k_c_round(Fr* __restrict__ A, Fr* __restrict__ M, Fr* __restrict__ Wc,
                                                 const Fr* __restrict__ Wb, uint32_t h, const FixedMul* __restrict__ rtab,
                                                 LayerHostRec* __restrict__ host_rec, uint32_t ticket, LayerBatch lb) {
    __shared__ Acc<9> smem[4 * 3];
    A += blockIdx.x * lb.tstride;
    M += blockIdx.x * lb.tstride;
    Wc += blockIdx.x * lb.wstride;
    const Fr p = load_fr(Wb + blockIdx.x * lb.wstride);
    if (FOLD) {
        const FixedMul T = rtab[blockIdx.x];
        const uint32_t hw = 2u * h;
        for (uint32_t base = 0; base < hw; base += blockDim.x) {
            const uint32_t i = base + threadIdx.x;
            Fr a = fr_zero(), m = fr_zero(), w = fr_zero();
            if (i < hw) {
                a = fr_fold_fixed(load_fr(A + i), load_fr(A + i + hw), T);
                m = fr_fold_fixed(load_fr(M + i), load_fr(M + i + hw), T);
                w = fr_fold_fixed(load_fr(Wc + i), load_fr(Wc + i + hw), T);
            }
            __syncthreads();   // all reads of this stripe done before any write lands in [0, hw)
            if (i < hw) {
                store_fr(A + i, a);
                store_fr(M + i, m);
                store_fr(Wc + i, w);
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    Acc<9> acc[3] = {acc_zero<9>(), acc_zero<9>(), acc_zero<9>()};
    if (h * 4u <= blockDim.x) {
        // few pairs: the three sums of a pair go to three different threads (a pair is ~8 dependent-free products,
        // and with one thread per pair most of the block would idle through them)
        const uint32_t i = threadIdx.x >> 2, part = threadIdx.x & 3u;
        if (i < h && part < 3u) {
            const Fr q0 = load_fr(Wc + i), q1 = load_fr(Wc + i + h);
            const Fr s0 = fr_add(p, q0), s1 = fr_add(p, q1);
            const Fr pq0 = mont_mul(p, q0), pq1 = mont_mul(p, q1);
            if (part == 0) {
                acc_add_fr(acc[0], fr_add(mont_mul(load_fr(A + i), s0), mont_mul(load_fr(M + i), pq0)));
            } else if (part == 1) {
                acc_add_fr(acc[1], fr_add(mont_mul(load_fr(A + i + h), s1), mont_mul(load_fr(M + i + h), pq1)));
            } else {
                const Fr da = fr_sub(load_fr(A + i + h), load_fr(A + i)), dm = fr_sub(load_fr(M + i + h), load_fr(M + i));
                acc_add_fr(acc[2], fr_add(mont_mul(da, fr_sub(s1, s0)), mont_mul(dm, fr_sub(pq1, pq0))));
            }
        }
    } else {
        for (uint32_t i = threadIdx.x; i < h; i += blockDim.x) {
            const PairTerms t = layer_pair(load_fr(A + i), load_fr(A + i + h), load_fr(M + i), load_fr(M + i + h), p, p,
                                           load_fr(Wc + i), load_fr(Wc + i + h));
            acc_add_fr(acc[0], t.c0);
            acc_add_fr(acc[1], t.g1);
            acc_add_fr(acc[2], t.c2);
        }
    }
    block_sum<9, 3>(acc, smem);
    __syncthreads();   // smem is reused for the totals
    publish_round(acc, smem, host_rec + blockIdx.x, ticket);
}

// ---------------------------------------------------------------------------
// The same two rounds for SMALL tables (h <= 64 pairs: every layer of a circom-sized circuit).  A round of such a
// layer is pure latency -- a chain of load, fold, product, reduction, reduction mod r, store to the host -- and the
// general kernels above spend it serially: the folded tables go to global memory and come back, and every lane
// reduces all three round sums one after the other.  Here the folded tables stay in LDS (and are written to global
// memory on the side, for the next round's launch), and the three sums belong to three different waves: wave 0
// sums c0, wave 1 g(1), wave 2 c2, each one wave-level reduction; their lane 0 reduces mod r and stores its field
// of the host record; one barrier later thread 0 releases the sequence number.  grid = (batch), block = 256.
// ---------------------------------------------------------------------------
constexpr uint32_t kSmallRoundPairs = 64;

__device__ __forceinline__ void publish_round_waves(const Acc<9>& a, uint32_t wave, uint32_t lane, LayerHostRec* r, uint32_t ticket) {
    if (wave < 3 && lane == 0) {
        const Fr v = acc_reduce(a);
        Fr* dst = wave == 0 ? &r->c0 : (wave == 1 ? &r->g1 : &r->c2);
        *dst = v;
        __threadfence_system();   // this lane's record stores are ordered ahead of the barrier and the release below
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&r->seq, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <bool FOLD>
__global__ void __launch_bounds__(256) k_uv_round_small(Fr* __restrict__ W, Fr* __restrict__ U, Fr* __restrict__ V, uint32_t h,
                                                        const FixedMul* __restrict__ rtab, LayerHostRec* __restrict__ host_rec,
                                                        uint32_t ticket, uint32_t wstride) {
    __shared__ Fr sT[3][2 * kSmallRoundPairs];   // U, V, W of this round: 2h entries each
    Fr* T[3] = {U + (size_t)blockIdx.x * wstride, V + (size_t)blockIdx.x * wstride, W + (size_t)blockIdx.x * wstride};
    const uint32_t hw = 2u * h;
    if (FOLD) {
        const FixedMul F = rtab[blockIdx.x];
        // entry i of the folded table is read (slots i, i + hw) and written (slot i) by one thread only
        for (uint32_t idx = threadIdx.x; idx < 3u * hw; idx += blockDim.x) {
            const uint32_t t = idx / hw, i = idx - t * hw;
            const Fr x = fr_fold_fixed(load_fr(T[t] + i), load_fr(T[t] + i + hw), F);
            sT[t][i] = x;
            store_fr(T[t] + i, x);
        }
    } else {
        for (uint32_t idx = threadIdx.x; idx < 3u * hw; idx += blockDim.x) {
            const uint32_t t = idx / hw, i = idx - t * hw;
            sT[t][i] = load_fr(T[t] + i);
        }
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    Acc<9> a = acc_zero<9>();
    if (wave < 3 && lane < h) {
        if (wave == 0) {
            acc_add_fr(a, mont_mul(sT[0][lane], sT[2][lane]));
            acc_add_fr(a, sT[1][lane]);
        } else if (wave == 1) {
            acc_add_fr(a, mont_mul(sT[0][lane + h], sT[2][lane + h]));
            acc_add_fr(a, sT[1][lane + h]);
        } else {
            acc_add_fr(a, mont_mul(fr_sub(sT[0][lane + h], sT[0][lane]), fr_sub(sT[2][lane + h], sT[2][lane])));
        }
    }
    if (wave < 3) a = wave_sum(a);
    publish_round_waves(a, wave, lane, host_rec + blockIdx.x, ticket);
}

template <bool FOLD>
__global__ void __launch_bounds__(256) k_c_round_small(Fr* __restrict__ A, Fr* __restrict__ M, Fr* __restrict__ Wc,
                                                       const Fr* __restrict__ Wb, uint32_t h, const FixedMul* __restrict__ rtab,
                                                       LayerHostRec* __restrict__ host_rec, uint32_t ticket, LayerBatch lb) {
    __shared__ Fr sT[3][2 * kSmallRoundPairs];   // A, M, then Wc; after the set-up below slot 2 holds S = p + Wc
    __shared__ Fr sPQ[2 * kSmallRoundPairs];     // p * Wc
    Fr* T[3] = {A + blockIdx.x * lb.tstride, M + blockIdx.x * lb.tstride, Wc + blockIdx.x * lb.wstride};
    const Fr p = load_fr(Wb + blockIdx.x * lb.wstride);
    const uint32_t hw = 2u * h;
    FixedMul F;
    if (FOLD) F = rtab[blockIdx.x];
    for (uint32_t idx = threadIdx.x; idx < 3u * hw; idx += blockDim.x) {
        const uint32_t t = idx / hw, i = idx - t * hw;
        Fr x;
        if (FOLD) {
            x = fr_fold_fixed(load_fr(T[t] + i), load_fr(T[t] + i + hw), F);
            store_fr(T[t] + i, x);
        } else {
            x = load_fr(T[t] + i);
        }
        if (t == 2) {   // the thread that holds W(c) also prepares the two factors every pair needs of it
            sPQ[i] = mont_mul(p, x);
            x = fr_add(p, x);
        }
        sT[t][i] = x;
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    Acc<9> a = acc_zero<9>();
    if (wave < 3 && lane < h) {
        const uint32_t i = lane, j = lane + h;
        if (wave == 0) {
            acc_add_fr(a, fr_add(mont_mul(sT[0][i], sT[2][i]), mont_mul(sT[1][i], sPQ[i])));
        } else if (wave == 1) {
            acc_add_fr(a, fr_add(mont_mul(sT[0][j], sT[2][j]), mont_mul(sT[1][j], sPQ[j])));
        } else {
            acc_add_fr(a, fr_add(mont_mul(fr_sub(sT[0][j], sT[0][i]), fr_sub(sT[2][j], sT[2][i])),
                                 mont_mul(fr_sub(sT[1][j], sT[1][i]), fr_sub(sPQ[j], sPQ[i]))));
        }
    }
    if (wave < 3) a = wave_sum(a);
    publish_round_waves(a, wave, lane, host_rec + blockIdx.x, ticket);
}

// ---------------------------------------------------------------------------
// A whole layer sumcheck as ONE resident kernel (small layers, host transcript): one block per proof; W, U, V and
// later the row a_u, m_u live in LDS for all 2k rounds.  A round is: the three sums (one wave each) -> the pinned
// host record -> thread 0 polls the proof's challenge slot in pinned host memory until the host has hashed the
// round -> every table is bound to the challenge in LDS.  No launch, no global-memory traffic and no stream
// operation per round: what is left of a round's latency is the PCIe hop each way and the host's hash.
// Proofs advance independently (the host answers each record as it lands), so blocks that are not resident yet
// hold nobody up; `abort_flag` (pinned) ends the waiting if the host gives up.
// ---------------------------------------------------------------------------
#ifdef GKR_PERSIST_DEBUG
__device__ unsigned long long g_persist_dbg[4096];
#define PDBG(slot_)                                                                  \
    do {                                                                             \
        if (blockIdx.x == 0 && threadIdx.x == 0 && dbg_n < 2040) {                   \
            g_persist_dbg[2 * dbg_n] = wall_clock64();                               \
            g_persist_dbg[2 * dbg_n + 1] = ((unsigned long long)(slot_) << 56) | (clock64() & 0xFFFFFFFFFFFFFFull); \
            ++dbg_n;                                                                 \
        }                                                                            \
    } while (0)
#else
#define PDBG(slot_) do { } while (0)
#endif

__device__ __forceinline__ bool wait_challenge(const LayerChallenge* slot, const uint32_t* abort_flag, uint32_t ticket, Fr* s_r,
                                               uint32_t* s_abort) {
    if (threadIdx.x == 0) {
        uint32_t aborted = 0;
        for (uint32_t spins = 0;; ++spins) {
            // relaxed polls (an acquire load would invalidate the caches on every try); one acquire fence at the end
            if (__hip_atomic_load(&slot->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == ticket) break;
            if ((spins & 15u) == 15u && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
                aborted = 1;
                break;
            }
            // every poll is a PCIe read that paces itself; a short sleep in between, longer once the host is clearly
            // busy (a hash call takes 24 - 30 us), so that hundreds of waiting blocks do not flood the link
            if (spins < 64u)
                __builtin_amdgcn_s_sleep(2);
            else
                __builtin_amdgcn_s_sleep(16);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        Fr r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = __hip_atomic_load(&slot->r_mont.l[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        *s_r = r;
        *s_abort = aborted;
    }
    __syncthreads();
    return *s_abort == 0;
}

__global__ void __launch_bounds__(256) k_layer_persistent(GateSpan span, uint32_t k, const uint32_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ cursor, const uint32_t* __restrict__ list,
                                                          const uint8_t* __restrict__ gate_type, const uint32_t* __restrict__ left,
                                                          const uint32_t* __restrict__ right, const Fr* __restrict__ e_hi,
                                                          const Fr* __restrict__ e_lo_mont, uint32_t kl, uint32_t kh,
                                                          const Fr* __restrict__ W, LayerHostRec* __restrict__ host_rec,
                                                          const LayerChallenge* __restrict__ challenges,
                                                          const uint32_t* __restrict__ abort_flag, uint32_t ticket_base,
                                                          uint32_t wstride) {
    constexpr uint32_t kMax = 1u << kPersistentMaxK;
    __shared__ Fr sA[kMax], sB[kMax], sW[kMax], sWc[kMax], sPQ[kMax];   // b-phase: U, V, W(b); c-phase: a_u, m_u, p + W(c), W(c), p W(c)
    __shared__ Fr s_r;
    __shared__ uint32_t s_abort;
    const uint32_t n = 1u << k, lmask = (1u << kl) - 1u, gate_base = (uint32_t)span.base;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    e_hi += (size_t)blockIdx.x << kh;
    e_lo_mont += (size_t)blockIdx.x << kl;
    W += (size_t)blockIdx.x * wstride;
    LayerHostRec* rec = host_rec + blockIdx.x;
    const LayerChallenge* slot = challenges + blockIdx.x;
    // W in Montgomery form, once for b and once for c
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const Fr w = to_mont(load_fr(W + i));
        sW[i] = w;
        sWc[i] = w;
    }
    __syncthreads();
    // U[b], V[b]: thread b sums the gates whose left operand is b (k_gate_uv's sums, one thread per bucket)
    for (uint32_t b = threadIdx.x; b < n; b += blockDim.x) {
        Fr u = fr_zero(), v = fr_zero();
        for (uint32_t i = offsets[b]; i < cursor[b]; ++i) {
            const uint32_t g = list[i], gg = g + gate_base;
            const Fr e = mont_mul(load_fr(e_hi + (gg >> kl)), load_fr(e_lo_mont + (gg & lmask)));
            const Fr ew = mont_mul(e, sWc[right[g]]);
            if (gate_type[g]) {
                u = fr_add(u, ew);
            } else {
                u = fr_add(u, e);
                v = fr_add(v, ew);
            }
        }
        sA[b] = u;
        sB[b] = v;
    }
    __syncthreads();
    uint32_t ticket = ticket_base;
#ifdef GKR_PERSIST_DEBUG
    uint32_t dbg_n = 0;
#endif
    PDBG(0);
    // ---- the k rounds that bind b: g(x) = sum_i W_i(x) U_i(x) + V_i(x)
    for (uint32_t h = n >> 1; h >= 1; h >>= 1, ++ticket) {
        PDBG(1);
        Acc<9> a = acc_zero<9>();
        if (wave < 3) {
            for (uint32_t i = lane; i < h; i += 64u) {
                if (wave == 0) {
                    acc_add_fr(a, mont_mul(sA[i], sW[i]));
                    acc_add_fr(a, sB[i]);
                } else if (wave == 1) {
                    acc_add_fr(a, mont_mul(sA[i + h], sW[i + h]));
                    acc_add_fr(a, sB[i + h]);
                } else {
                    acc_add_fr(a, mont_mul(fr_sub(sA[i + h], sA[i]), fr_sub(sW[i + h], sW[i])));
                }
            }
            a = wave_sum(a);
        }
        PDBG(2);
        publish_round_waves(a, wave, lane, rec, ticket);
        PDBG(3);
        if (!wait_challenge(slot, abort_flag, ticket, &s_r, &s_abort)) return;
        PDBG(4);
        const Fr r = s_r;
        // bind the variable in U, V, W and grow eq(u, .) by it -- one product per thread: thread t < 3h folds entry
        // t % h of table t / h (h <= 64: all at once; h = 128: two turns), the next `cur` threads make the two
        // children of an eq entry (kept in sPQ, Montgomery, first variable most significant as host_eq_table has it)
        const uint32_t cur = n / (2u * h);   // eq entries before this round's variable
        auto tabs = [&](uint32_t tb) -> Fr* { return tb == 0u ? sA : (tb == 1u ? sB : sW); };
        Fr folded[2], lo_v, hi_v;
        uint32_t nf = 0;
        for (uint32_t idx = threadIdx.x; idx < 3u * h; idx += blockDim.x, ++nf) {
            const uint32_t tb = idx / h, i = idx - tb * h;
            folded[nf] = fr_fold(tabs(tb)[i], tabs(tb)[i + h], r);
        }
        const uint32_t et = threadIdx.x >= blockDim.x - cur ? threadIdx.x - (blockDim.x - cur) : 0xFFFFFFFFu;   // the last `cur` threads
        if (et != 0xFFFFFFFFu) {
            Fr one = fr_zero();
            one.l[0] = 1u;
            const Fr e = cur == 1u ? to_mont(one) : sPQ[et];
            hi_v = mont_mul(e, r);
            lo_v = fr_sub(e, hi_v);
        }
        __syncthreads();
        nf = 0;
        for (uint32_t idx = threadIdx.x; idx < 3u * h; idx += blockDim.x, ++nf) {
            const uint32_t tb = idx / h, i = idx - tb * h;
            tabs(tb)[i] = folded[nf];
        }
        if (et != 0xFFFFFFFFu) {
            sPQ[2u * et] = lo_v;
            sPQ[2u * et + 1u] = hi_v;
        }
        __syncthreads();
    }
    PDBG(5);
    const Fr p = sW[0];   // W(u), Montgomery
    // ---- the row at b = u: thread c sums the gates whose right operand is c (k_gate_rows)
    for (uint32_t c = threadIdx.x; c < n; c += blockDim.x) {
        Fr am[2] = {fr_zero(), fr_zero()};
        const uint32_t bucket = n + c;
        for (uint32_t i = offsets[bucket]; i < cursor[bucket]; ++i) {
            const uint32_t g = list[i], gg = g + gate_base;
            const Fr e = mont_mul(load_fr(e_hi + (gg >> kl)), load_fr(e_lo_mont + (gg & lmask)));
            const Fr tt = mont_mul(e, sPQ[left[g]]);
            const uint32_t w = gate_type[g] ? 1u : 0u;
            am[w] = fr_add(am[w], tt);
        }
        sA[c] = am[0];
        sB[c] = am[1];
    }
    __syncthreads();
    // ---- the k rounds that bind c, on the row: a (p + W) + m p W
    for (uint32_t h = n >> 1; h >= 1; h >>= 1, ++ticket) {
        for (uint32_t i = threadIdx.x; i < 2u * h; i += blockDim.x) {
            const Fr w = sWc[i];
            sPQ[i] = mont_mul(p, w);
            sW[i] = fr_add(p, w);
        }
        __syncthreads();
        Acc<9> a = acc_zero<9>();
        if (wave < 3) {
            for (uint32_t i = lane; i < h; i += 64u) {
                const uint32_t j = i + h;
                if (wave == 0) {
                    acc_add_fr(a, fr_add(mont_mul(sA[i], sW[i]), mont_mul(sB[i], sPQ[i])));
                } else if (wave == 1) {
                    acc_add_fr(a, fr_add(mont_mul(sA[j], sW[j]), mont_mul(sB[j], sPQ[j])));
                } else {
                    acc_add_fr(a, fr_add(mont_mul(fr_sub(sA[j], sA[i]), fr_sub(sW[j], sW[i])),
                                         mont_mul(fr_sub(sB[j], sB[i]), fr_sub(sPQ[j], sPQ[i]))));
                }
            }
            a = wave_sum(a);
        }
        publish_round_waves(a, wave, lane, rec, ticket);
        if (h == 1) break;   // the last challenge binds nothing the device still needs
        if (!wait_challenge(slot, abort_flag, ticket, &s_r, &s_abort)) return;
        const Fr r = s_r;
        auto tabs = [&](uint32_t tb) -> Fr* { return tb == 0u ? sA : (tb == 1u ? sB : sWc); };
        Fr folded[2];
        uint32_t nf = 0;
        for (uint32_t idx = threadIdx.x; idx < 3u * h; idx += blockDim.x, ++nf) {
            const uint32_t tb = idx / h, i = idx - tb * h;
            folded[nf] = fr_fold(tabs(tb)[i], tabs(tb)[i + h], r);
        }
        __syncthreads();
        nf = 0;
        for (uint32_t idx = threadIdx.x; idx < 3u * h; idx += blockDim.x, ++nf) {
            const uint32_t tb = idx / h, i = idx - tb * h;
            tabs(tb)[i] = folded[nf];
        }
        __syncthreads();
    }
}

// Row collapse, stage 1: partial[chunk][c] = sum over the chunk's rows of eq[row] * T[row][c] for T = A (z = 0)
// and M (z = 1).  eq: Montgomery, the same value for the whole wave.  grid = (column blocks, chunks, 2 * batch).
__global__ void __launch_bounds__(256) k_layer_collapse_rows(const Fr* __restrict__ A, const Fr* __restrict__ M,
                                                             const Fr* __restrict__ eq, Fr* __restrict__ partial, uint32_t k,
                                                             uint32_t rows_per_chunk, LayerBatch lb) {
    const uint32_t cols = 1u << k, c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t proof = blockIdx.z >> 1, table = blockIdx.z & 1u, chunks = gridDim.y;
    if (c >= cols) return;
    const Fr* t = (table ? M : A) + proof * lb.tstride;
    const Fr* e = eq + (size_t)proof * cols;
    const uint32_t r0 = blockIdx.y * rows_per_chunk, r1 = min(r0 + rows_per_chunk, cols);
    Lazy17 acc = lazy_zero();
    for (uint32_t r = r0; r < r1; ++r) lazy_mac_s(acc, load_fr(t + (size_t)r * cols + c), e[r]);
    store_fr(partial + (((size_t)blockIdx.z * chunks + blockIdx.y) << k) + c, lazy_reduce(acc));
}

// stage 2: row 0 of A / M = sum of the chunk partials.  grid = (column blocks, 2 * batch)
__global__ void __launch_bounds__(256) k_layer_collapse_sum(const Fr* __restrict__ partial, Fr* __restrict__ A, Fr* __restrict__ M,
                                                            uint32_t k, uint32_t chunks, LayerBatch lb) {
    const uint32_t cols = 1u << k, c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t proof = blockIdx.y >> 1, table = blockIdx.y & 1u;
    if (c >= cols) return;
    Acc<9> acc = acc_zero<9>();
    for (uint32_t ch = 0; ch < chunks; ++ch) acc_add_fr(acc, load_fr(partial + (((size_t)blockIdx.y * chunks + ch) << k) + c));
    store_fr((table ? M : A) + proof * lb.tstride + c, acc_reduce(acc));
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

void launch_layer_uv(const Fr* A, const Fr* M, const Fr* W, Fr* U, Fr* V, uint32_t k, LayerBatch lb, hipStream_t s) {
    hipLaunchKernelGGL(k_layer_uv, dim3(1u << k, lb.batch), dim3(256), 0, s, A, M, W, U, V, k, lb);
}

void launch_c_round(bool fold, Fr* A, Fr* M, Fr* Wc, const Fr* Wb, uint32_t h, const FixedMul* rtab, LayerHostRec* host_rec,
                    uint32_t ticket, LayerBatch lb, hipStream_t s) {
    static const bool general = getenv("GKR_NO_SMALL_ROUNDS") != nullptr;
    if (h <= kSmallRoundPairs && !general) {
        if (fold)
            hipLaunchKernelGGL(k_c_round_small<true>, dim3(lb.batch), dim3(256), 0, s, A, M, Wc, Wb, h, rtab, host_rec, ticket, lb);
        else
            hipLaunchKernelGGL(k_c_round_small<false>, dim3(lb.batch), dim3(256), 0, s, A, M, Wc, Wb, h, rtab, host_rec, ticket, lb);
        return;
    }
    if (fold)
        hipLaunchKernelGGL(k_c_round<true>, dim3(lb.batch), dim3(256), 0, s, A, M, Wc, Wb, h, rtab, host_rec, ticket, lb);
    else
        hipLaunchKernelGGL(k_c_round<false>, dim3(lb.batch), dim3(256), 0, s, A, M, Wc, Wb, h, rtab, host_rec, ticket, lb);
}

#ifdef GKR_PERSIST_DEBUG
extern "C" void gkr_debug_read_persist(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_persist_dbg), sizeof(unsigned long long) * 4096);
}
#endif

void launch_layer_persistent(GateSpan span, uint32_t k_i, uint32_t k, const uint32_t* offsets, const uint32_t* cursor,
                             const uint32_t* list, const uint8_t* gate_type, const uint32_t* left, const uint32_t* right,
                             const Fr* e_hi, const Fr* e_lo_mont, uint32_t kl, const Fr* W, LayerHostRec* host_rec,
                             const LayerChallenge* challenges, const uint32_t* abort_flag, uint32_t ticket_base, LayerBatch lb,
                             hipStream_t s) {
    hipLaunchKernelGGL(k_layer_persistent, dim3(lb.batch), dim3(256), 0, s, span, k, offsets, cursor, list, gate_type, left, right, e_hi,
                       e_lo_mont, kl, k_i - kl, W, host_rec, challenges, abort_flag, ticket_base, (uint32_t)lb.wstride);
}

void launch_uv_round(bool fold, Fr* W, Fr* U, Fr* V, uint32_t h, const FixedMul* rtab, LayerHostRec* host_rec, uint32_t ticket,
                     LayerBatch lb, hipStream_t s) {
    static const bool general = getenv("GKR_NO_SMALL_ROUNDS") != nullptr;
    if (h <= kSmallRoundPairs && !general) {
        if (fold)
            hipLaunchKernelGGL(k_uv_round_small<true>, dim3(lb.batch), dim3(256), 0, s, W, U, V, h, rtab, host_rec, ticket, (uint32_t)lb.wstride);
        else
            hipLaunchKernelGGL(k_uv_round_small<false>, dim3(lb.batch), dim3(256), 0, s, W, U, V, h, rtab, host_rec, ticket, (uint32_t)lb.wstride);
        return;
    }
    if (fold)
        hipLaunchKernelGGL(k_uv_round<true>, dim3(lb.batch), dim3(256), 0, s, W, U, V, h, rtab, host_rec, ticket, (uint32_t)lb.wstride);
    else
        hipLaunchKernelGGL(k_uv_round<false>, dim3(lb.batch), dim3(256), 0, s, W, U, V, h, rtab, host_rec, ticket, (uint32_t)lb.wstride);
}

// chunks of rows the collapse is split into (enough blocks to fill the chip); scratch = 2 * batch * chunks * 2^k elements
uint32_t layer_collapse_chunks(uint32_t k, uint32_t batch) {
    const uint32_t rows = 1u << k, col_blocks = (rows + 255u) / 256u;
    uint32_t chunks = 2048u / (col_blocks * 2u * batch);
    if (chunks < 1) chunks = 1;
    if (chunks > rows) chunks = rows;
    return chunks;
}

void launch_layer_collapse(Fr* A, Fr* M, const Fr* eq, Fr* scratch, uint32_t k, LayerBatch lb, hipStream_t s) {
    const uint32_t rows = 1u << k, col_blocks = (rows + 255u) / 256u;
    uint32_t chunks = layer_collapse_chunks(k, lb.batch);
    const uint32_t rows_per_chunk = (rows + chunks - 1) / chunks;
    chunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
    hipLaunchKernelGGL(k_layer_collapse_rows, dim3(col_blocks, chunks, 2 * lb.batch), dim3(256), 0, s, A, M, eq, scratch, k,
                       rows_per_chunk, lb);
    hipLaunchKernelGGL(k_layer_collapse_sum, dim3(col_blocks, 2 * lb.batch), dim3(256), 0, s, scratch, A, M, k, chunks, lb);
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
