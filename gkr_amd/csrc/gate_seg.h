// The gate passes of the linear-time layer sumcheck over SEGMENTS of the sorted gate lists.
//
// The reference sums, for every assignment, over the gate list (rust/src/gkr/sumcheck.rs:50-63, 97-124).  In the
// linear-time form every table the rounds work on is such a sum:
//     U[b] = sum_{gates g with left = b}  eq(z, g) * (mult ? W[right] : 1)       V[b] = sum_{add g, left = b} eq(z, g) * W[right]
//     a_u[c] = sum_{add g with right = c} eq(z, g) * eq(u, left)                 m_u[c] likewise over the mult gates
// with eq(z, g) = E_hi[g >> s] * E_lo[g & (2^s - 1)].  The first form of these passes (k_gate_uv / k_gate_rows) formed
// eq(z, g) per gate -- one reduced product (128 multiply-adds) -- before the product that matters (64 more).  Here the
// bucket's list is cut where g >> s changes (the counting sort already leaves a bucket's gates in ascending blocks of
// the gate index), so E_hi is a common factor of a whole SEGMENT:
//     sum_g E_hi[g >> s] E_lo[..] T[..]  =  sum_segments E_hi[run] * ( sum_{g in segment} E_lo[g & mask] * T[other operand] )
// One unreduced 512-bit multiply-add per gate (64 multiply-adds), one short reduction and one more product per segment
// of ~16 gates.  Segments longer than kSegCap are cut into items of at most kSegCap gates; items are processed one per
// lane in order of decreasing length, so the lanes of a wave run the same number of iterations.
#pragma once
#include "fr32.h"

namespace gkr {

constexpr uint32_t kSegCap = 32;       // gates per item at most (lazy_reduce_partial32's bound)
constexpr uint32_t kSegMeanLog2 = 4;   // segments of 2^4 gates on average: s = k + 4

// one gate of an item.  e = E_lo[g & mask] and t = W[right] resp. eq(u, left), both in Montgomery form.
//   ROWS == false (U, V):  mult gate: L0 += e t;  add gate: L1 += e t and L0 += e (no second factor)
//   ROWS == true  (a_u, m_u):  add gate: L0 += e t;  mult gate: L1 += e t
template <bool ROWS>
GKR_HD void seg_gate(Lazy17& L0, Lazy17& L1, const Fr& e, const Fr& t, bool is_mult) {
    lazy_mac_sel(L0, L1, ROWS ? !is_mult : is_mult, e, t);
    if (!ROWS) lazy_add_hi(L0, e, !is_mult);
}

// The form k_seg_pass runs since the items keep their add gates first (k_seg_pack): no select per gate.  R0 is the
// accumulator the products go to, R1 the other one; at the item's first mult gate the two are exchanged (`sw`), once.
//   one gate:   seg_gate_ordered<ROWS>(R0, R1, sw, e, t, is_mult)      (an add gate after a mult gate is a caller's error)
//   at the end: seg_item_sums<ROWS>(R0, R1, sw, L0, L1)                 L0, L1 as seg_gate leaves them
template <bool ROWS>
GKR_HD void seg_gate_ordered(Lazy17& R0, Lazy17& R1, bool& sw, const Fr& e, const Fr& t, bool is_mult) {
    if (is_mult && !sw) {
        const Lazy17 x = R0;
        R0 = R1;
        R1 = x;
        sw = true;
    }
    lazy_mac_v(R0, e, t);
    if (!ROWS && !sw) lazy_add_hi(R1, e, true);   // U: the add gate's term without a second factor joins the mult gates' sum
}
template <bool ROWS>
GKR_HD void seg_item_sums(const Lazy17& R0, const Lazy17& R1, bool sw, Lazy17& L0, Lazy17& L1) {
    const bool r0_first = ROWS ? !sw : sw;   // ROWS: add gates -> L0, mult -> L1;  U, V: mult (+ plain add terms) -> L0, add -> L1
    L0 = r0_first ? R0 : R1;
    L1 = r0_first ? R1 : R0;
}

}  // namespace gkr
