#!/usr/bin/env python3
"""Writes tests/golden/convert_hand_derived.json: three small R1CS systems and the layered circuits that
rust/src/convert.rs turns them into, DERIVED BY HAND from the reference's source (no implementation -- neither
oracle/convert.py nor gkr_amd/csrc/r1cs.cpp -- was run to obtain them; tests/test_convert.py then holds both against
this file).  The reference cannot be run here (Rust nightly, no toolchain), so this is the strongest pin of row f1
the image allows: every number below follows from the cited lines.

Rules used (all in /root/reference/rust/src/convert.rs):
  count_mult / neg          :363-379, :476-485   neg <=> (#A != 1) + (#B != 1) + (#C != -1)  >  (#A != -1) + (#B != -1) + (#C != 1)
  term -> leaf or Mult      A :513-541 (negated when neg), B :556-567, C :571-604 (negated when NOT neg)
  merge_nodes               :108-138             pairs (2i, 2i+1) -> Add; an odd last node is added on top
  root                      :606-610             Add(Mult(merge A, merge B), merge C)
  compile                   :154-358             stable sort by depth :164-168; pairwise merge while > 20 :171-185;
                                                 per layer: pad to 2^k with Value(0) :205-212; op node: operands found by deep
                                                 equality in next_nodes or pushed :274-299; Value node: relay Add(x, zero), zero
                                                 slot allocated at the first relay of the layer :316-320, `used` map per layer
                                                 :311-315, Value(0) relays as (zero, zero) :325-329; last gate layer :222-276
  get_k                     :140-152
"""
import json
import os

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
M1 = P - 1
V = lambda i: ["var", i]
C = lambda v: ["val", v]
Z = C(0)
cases = []

# ---- 1. neg = true: (-w1 - w2) * w3 = w4
cases.append({
    "name": "neg_flag_true_with_relays_and_a_lazy_zero_slot",
    "why": "count_mult (convert.rs:363-379): A = (2,0), B = (0,1), C = (0,1) -> mult_cnt = 2+0+1 = 3 > m_mult_cnt = 0+1+0 = 1 -> neg (:476-485). "
           "A terms with coefficient -1 become bare variables (:513-516), C terms with coefficient 1 bare variables (:571-574): root = Add(Mult(Add(V1,V2),V3),V4), depth 4. "
           "compile (:188-352): V4 is a Value at layer 1 and 2 -> relay Add(x, zero) gates with the zero slot allocated where the first relay meets it (:316-320): index 2, not 0; "
           "the zero node itself relays as (zero, zero) (:325-329); padding zeros of layer 3 find Value(0) in `used` (:311-315).",
    "n_wires": 5, "n_pub_out": 1, "n_pub_in": 1, "n_prv_in": 2,
    "constraints": [[[[M1, 1], [M1, 2]], [[1, 3]], [[1, 4]]]],
    "circuits": [{
        "k": [0, 1, 2, 3, 3],
        "layers": [[[0], [0], [1]],
                   [[1, 0], [0, 3], [1, 2]],
                   [[0, 0, 0, 0], [0, 3, 2, 4], [1, 2, 2, 2]],
                   [[0] * 8, [1, 2, 0, 3, 4, 0, 0, 0], [0] * 8]],
        "inputs": [Z, V(1), V(2), V(3), V(4), Z, Z, Z]}]})

# ---- 2. odd merge_nodes, a coefficient that is neither 1 nor -1: (w1 + 2 w2 + w3) * w4 = -w5
cases.append({
    "name": "odd_merge_nodes_and_a_general_coefficient",
    "why": "count_mult: A = (1,3), B = (0,1), C = (1,0) -> mult_cnt = 1+0+0 = 1, m_mult_cnt = 3+1+1 = 5 -> not neg. A = [V1, Mult(2,V2), V3]; merge_nodes of three (convert.rs:108-138): "
           "Add(Add(n0,n1), n2); C coefficient -1 -> bare V5 (:583-586). root = Add(Mult(Add(Add(V1,Mult(2,V2)),V3),V4),V5), depth 6. "
           "At layer 4 the Value V1 comes BEFORE the Mult node: the zero slot is allocated at index 0 there, at index 2 in layers 1-3.",
    "n_wires": 6, "n_pub_out": 1, "n_pub_in": 1, "n_prv_in": 3,
    "constraints": [[[[1, 1], [2, 2], [1, 3]], [[1, 4]], [[M1, 5]]]],
    "circuits": [{
        "k": [0, 1, 2, 3, 3, 3, 3],
        "layers": [[[0], [0], [1]],
                   [[1, 0], [0, 3], [1, 2]],
                   [[0, 0, 0, 0], [0, 3, 2, 4], [1, 2, 2, 2]],
                   [[0] * 8, [0, 3, 2, 4, 5, 2, 2, 2], [1, 2, 2, 2, 2, 2, 2, 2]],
                   [[0, 1, 0, 0, 0, 0, 0, 0], [1, 2, 0, 4, 5, 6, 0, 0], [0, 3, 0, 0, 0, 0, 0, 0]],
                   [[0] * 8, [0, 1, 2, 3, 4, 5, 6, 0], [0] * 8]],
        "inputs": [Z, V(1), C(2), V(2), V(3), V(4), V(5), Z]}]})

# ---- 3. 21 constraints: the stable depth sort moves the deeper constraint 0 to the end, one pairwise merge -> 11 circuits
cons = [[[[1, 1], [1, 1]], [[1, 1]], [[1, 2]]]]                       # (w1 + w1) * w1 = w2, depth 4
for i in range(1, 21):
    cons.append([[[1, i]], [[1, i]], [[1, i + 1]]])                   # w_i * w_i = w_{i+1}, depth 3
circuits = []
for j in range(10):
    a1, c1, c2 = 2 * j + 1, 2 * j + 2, 2 * j + 3
    circuits.append({"k": [1, 2, 2, 3],
                     "layers": [[[0, 0], [0, 2], [1, 3]],
                                [[1, 1, 1, 1], [0, 1, 2, 1], [0, 2, 2, 3]],
                                [[0, 0, 0, 0], [1, 2, 3, 4], [0, 0, 0, 0]]],
                     "inputs": [Z, V(a1), C(M1), V(c1), V(c2), Z, Z, Z]})
circuits.append({"k": [0, 1, 2, 3, 2],
                 "layers": [[[0], [0], [1]],
                            [[1, 1], [0, 2], [1, 3]],
                            [[0, 0, 0, 0], [0, 2, 3, 4], [0, 1, 1, 1]],
                            [[0] * 8, [1, 0, 1, 2, 3, 0, 0, 0], [0] * 8]],
                 "inputs": [Z, V(1), C(M1), V(2)]})
cases.append({
    "name": "width_limit_merge_depth_sort_dedupe_across_trees",
    "why": "21 constraints > WIDTH_LIMIT = 20 (convert.rs:10): the stable sort by depth (:164-168) moves constraint 0 (depth 4) behind the twenty of depth 3, one pairwise merge "
           "(:171-185) leaves 10 circuits of two trees and the odd one alone.  In a two-tree circuit the second tree's Mult(V_c1, V_c1) finds V_c1 and Value(-1) already in next_nodes "
           "(`contains`, deep equality, :282-297): operands (2,2) and (1,3).  In the last circuit the relay of V1 at layer 2 does NOT see the copy of V1 that Add(V1,V1) pushed at index 0 "
           "(relays consult `used`, op nodes `contains`): V1 is pushed a second time (index 2) and the zero slot sits at index 1; at the bottom layer the second V1 is found in `used` "
           "(:311-315) -> (1, 0).",
    "n_wires": 22, "n_pub_out": 1, "n_pub_in": 1, "n_prv_in": 19,
    "constraints": cons, "circuits": circuits})

out = {"what": "Layered circuits derived BY HAND from rust/src/convert.rs (convert_constraints_to_nodes :360-632, merge_nodes :108-138, compile :154-358, get_k :140-152); "
               "written out literally by tests/golden/make_convert_hand_derived.py -- neither the oracle's restatement nor the product's compiler produced them. "
               "gate type 0 = Add, 1 = Mult; layers = [types, left, right]; inputs: [\"val\", v] | [\"var\", wire].",
       "cases": cases}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "convert_hand_derived.json"), "w") as f:
    json.dump(out, f, indent=1)
