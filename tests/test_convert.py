"""R1CS / witness ingest and the R1CS -> layered-circuit compiler: the
product's host C++ (gkr_amd/csrc/r1cs.cpp through the C ABI) against the oracle's restatement of
rust/src/convert.rs (oracle/convert.py), structure for structure, plus one unit test per rule of appendix B.
CPU only: nothing here needs a GPU."""

import random
import struct

import pytest

from gkr_amd import GkrError
from gkr_amd import _native as N
from gkr_amd import convert as product
from oracle import convert as oracle
from oracle.field import P


def _build(r):
    return product.R1cs.build(r["n_wires"], r["n_pub_out"], r["n_pub_in"], r["n_prv_in"], r["constraints"])


def _compiled(r1cs):
    lay = r1cs.compile()
    out = []
    for j in range(len(lay)):
        c = lay.circuit(j)
        out.append(dict(k=c.get_k_list(), layers=[(list(map(int, l.gate_type)), list(map(int, l.left)), list(map(int, l.right)))
                                                  for l in c.layer], inputs=lay.input_layer(j)))
    lay.close()
    return out


def _oracle_compiled(r):
    circuits, inputs = oracle.compile_groups(oracle.convert_constraints_to_nodes(r["constraints"]))
    return [dict(k=[oracle.get_k(len(t[0])) for t in layers] + [oracle.get_k(len(inp))],
                 layers=[(list(t), list(l), list(rr)) for t, l, rr in layers], inputs=list(inp))
            for layers, inp in zip(circuits, inputs)]


def random_r1cs(rng, n_constraints, n_wires=24, max_terms=5):
    """A random satisfiable system: random witness, random A and B, C solved in its last coefficient."""
    w = [1] + [rng.randrange(1, P) for _ in range(n_wires - 1)]
    special = [1, P - 1, 0, 2, P - 2]

    def coeff():
        return rng.choice(special) if rng.random() < 0.6 else rng.randrange(P)

    def lc(n):
        return [(coeff(), rng.randrange(n_wires)) for _ in range(n)]
    ev = lambda v: sum(c * w[i] for c, i in v) % P
    cons = []
    for _ in range(n_constraints):
        a, b = lc(rng.randint(1, max_terms)), lc(rng.randint(1, max_terms))
        c = lc(rng.randint(0, max_terms - 1))
        wire = rng.randrange(n_wires)
        c.append(((ev(a) * ev(b) - ev(c)) * pow(w[wire], -1, P) % P, wire))
        cons.append((a, b, c))
    return dict(n_wires=n_wires, n_pub_out=1, n_pub_in=2, n_prv_in=n_wires - 4, constraints=cons), w


# ---------------------------------------------------------------------------------------------- containers

def test_r1cs_container_round_trip_and_byte_equality_with_the_oracle_writer():
    for style in ("plain", "negated"):
        r = oracle.mimc7_r1cs(style=style)
        mine = _build(r)
        image = mine.serialize()
        assert image == oracle.write_r1cs(r["n_wires"], r["n_pub_out"], r["n_pub_in"], r["n_prv_in"], r["constraints"])
        again = product.R1cs.parse(image)
        info = again.info()
        assert (info["n_wires"], info["n_pub_out"], info["n_pub_in"], info["n_prv_in"]) == (r["n_wires"], 1, 1, 1)
        assert info["n_constraints"] == 364 and info["n_labels"] == r["n_wires"]
        assert again.constraints() == [tuple(c) for c in r["constraints"]] == oracle.read_r1cs(image)["constraints"]


def test_r1cs_sections_in_any_order_and_unknown_sections_are_skipped():
    r = oracle.mimc7_r1cs(nrounds=3)
    image = oracle.write_r1cs(r["n_wires"], 1, 1, 1, r["constraints"])
    secs, off = [], 12
    for _ in range(3):
        ty, size = struct.unpack_from("<IQ", image, off)
        secs.append(image[off:off + 12 + size])
        off += 12 + size
    custom = struct.pack("<IQ", 4, 5) + b"hello"
    shuffled = image[:8] + struct.pack("<I", 4) + secs[2] + custom + secs[1] + secs[0]
    assert product.R1cs.parse(shuffled).constraints() == [tuple(c) for c in r["constraints"]]


def test_r1cs_malformed_files_are_statuses_not_crashes():
    r = oracle.mimc7_r1cs(nrounds=2)
    image = oracle.write_r1cs(r["n_wires"], 1, 1, 1, r["constraints"])
    for bad in (b"", b"r1cs", b"wtns" + image[4:], image[:40], image[:-7], image[:4] + struct.pack("<I", 2) + image[8:]):
        with pytest.raises(GkrError):
            product.R1cs.parse(bad)
    wrong_prime = bytearray(image)
    wrong_prime[12 + 12 + 4] ^= 1          # first byte of the prime in the header section
    with pytest.raises(GkrError):
        product.R1cs.parse(bytes(wrong_prime))
    noncanon = oracle.write_r1cs(4, 0, 0, 3, [([(1, 1)], [(1, 2)], [(1, 3)])]).replace((1).to_bytes(32, "little"), b"\xff" * 32, 1)
    with pytest.raises(GkrError) as e:
        product.R1cs.parse(noncanon)
    assert e.value.status == N.GKR_ERR_NON_CANONICAL
    with pytest.raises(GkrError):       # wire index beyond nWires
        product.R1cs.build(3, 0, 0, 2, [([(1, 1)], [(1, 2)], [(1, 3)])])


def test_wtns_container_round_trip_and_byte_equality_with_the_oracle_writer():
    w = oracle.mimc7_witness(2, 3)
    image = product.write_wtns(w)
    assert image == oracle.write_wtns(w)
    assert product.read_wtns(image) == w == oracle.read_wtns(image)
    assert product.read_wtns(product.write_wtns([])) == []
    for bad in (b"", image[:30], b"r1cs" + image[4:], image[:-1]):
        with pytest.raises(GkrError):
            product.read_wtns(bad)


# ---------------------------------------------------------------------------------------------- both compilers vs circuits derived by hand

def test_compilers_match_the_circuits_derived_by_hand_from_convert_rs():
    """tests/golden/convert_hand_derived.json: three systems whose layered circuits were worked out by hand from
    rust/src/convert.rs (neg flag, odd merge_nodes, relay gates with the lazily allocated zero slot, the relay / op-node
    dedupe asymmetry, the width-limit merge after the stable depth sort).  Both the oracle's restatement and the
    product's C++ compiler must reproduce them -- until now the two only checked each other."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convert_hand_derived.json")))
    assert len(gold["cases"]) == 3
    for case in gold["cases"]:
        cons = [tuple([(int(c), int(w)) for c, w in part] for part in con) for con in case["constraints"]]
        r = dict(n_wires=case["n_wires"], n_pub_out=case["n_pub_out"], n_pub_in=case["n_pub_in"], n_prv_in=case["n_prv_in"], constraints=cons)
        want = [dict(k=c["k"], layers=[tuple(map(list, l)) for l in c["layers"]], inputs=[tuple(x) for x in c["inputs"]]) for c in case["circuits"]]
        got_oracle = _oracle_compiled(r)
        got_oracle = [dict(k=c["k"], layers=[tuple(map(list, l)) for l in c["layers"]], inputs=[tuple(x) for x in c["inputs"]]) for c in got_oracle]
        assert got_oracle == want, case["name"]
        got_product = _compiled(_build(r))
        got_product = [dict(k=c["k"], layers=[tuple(map(list, l)) for l in c["layers"]], inputs=[tuple(x) for x in c["inputs"]]) for c in got_product]
        assert got_product == want, case["name"]


# ---------------------------------------------------------------------------------------------- compiler vs oracle

@pytest.mark.parametrize("style", ["plain", "negated"])
def test_mimc7_demo_circuit_compiles_like_the_oracle(style):
    """The hand-written R1CS equivalent of rust/t.circom (BASELINE configs[0]): 364 constraints -> 12 circuits."""
    r = oracle.mimc7_r1cs(style=style)
    got, want = _compiled(_build(r)), _oracle_compiled(r)
    assert got == want
    assert len(got) == 12 and all(len(c["k"]) >= 3 for c in got)


@pytest.mark.parametrize("seed", range(12))
def test_random_systems_compile_like_the_oracle(seed):
    rng = random.Random(100 + seed)
    r, _ = random_r1cs(rng, n_constraints=rng.choice([1, 2, 7, 20, 21, 41, 90]), n_wires=rng.choice([5, 24, 200]),
                       max_terms=rng.choice([1, 2, 5, 9]))
    assert _compiled(_build(r)) == _oracle_compiled(r)


def test_compiled_circuits_vanish_on_a_satisfying_witness_only():
    """Every tree is <A,w><B,w> - <C,w>: all outputs of every sub-circuit are 0 on a satisfying witness
    (calculate_input asserts output 0, convert.rs:838), and some output is not once the witness is spoiled."""
    rng = random.Random(7)
    r, w = random_r1cs(rng, 33)
    r1cs = _build(r)
    lay = r1cs.compile()

    def outputs(witness):
        outs = []
        for j in range(len(lay)):
            c = lay.circuit(j)
            layers = [(list(l.gate_type), list(l.left), list(l.right)) for l in c.layer]
            outs.append(oracle.forward_values(layers, lay.input_values(j, witness))[0])
        return outs
    assert all(v == 0 for out in outputs(w) for v in out)
    spoiled = list(w)
    spoiled[3] = (spoiled[3] + 1) % P
    assert any(v != 0 for out in outputs(spoiled) for v in out)
    ours, theirs = product.convert_r1cs_wtns_gkr(r1cs, w), oracle.convert_r1cs_wtns_gkr(r, w)
    assert [c.get_k_list() for c in ours[0]] == [t["k"] for t in theirs]
    assert ours[1] == [t["input_values"] for t in theirs]


# ---------------------------------------------------------------------------------------------- appendix B, rule by rule

def _one(a, b, c, n_wires=8):
    return dict(n_wires=n_wires, n_pub_out=0, n_pub_in=0, n_prv_in=n_wires - 1, constraints=[(a, b, c)])


def test_get_k():
    # convert.rs:140-152 through the width of layer 0: n constraints in one group -> k = ceil(log2 n), get_k(1) = 0
    for n, k in ((1, 0), (2, 1), (3, 2), (4, 2), (5, 3), (8, 3), (9, 4), (16, 4), (17, 5), (20, 5)):
        r = dict(n_wires=4, n_pub_out=0, n_pub_in=0, n_prv_in=3, constraints=[([(1, 1)], [(1, 2)], [(P - 1, 3)])] * n)
        got = _compiled(_build(r))
        assert len(got) == (n if n <= 20 else 1)
        assert oracle.get_k(n) == k
    r = dict(n_wires=4, n_pub_out=0, n_pub_in=0, n_prv_in=3, constraints=[([(1, 1)], [(1, 2)], [(P - 1, 3)])] * 21)
    assert [c["k"][0] for c in _compiled(_build(r))] == [1] * 10 + [0]   # 21 groups -> 10 pairs + the odd last one


def test_coefficient_one_is_a_bare_variable_and_others_cost_a_product():
    # B.1 steps 2-4 (!neg): A/B coeff 1 -> Variable; C coeff -1 -> Variable; anything else Mult(Value, Variable)
    plain = _compiled(_build(_one([(1, 1)], [(1, 2)], [(P - 1, 3)])))[0]
    # Add(Mult(v1, v2), v3): height 3 -> gate layers of 1, 2, 4 gates (the last one relays the leaves) + 4 inputs
    assert plain["k"] == [0, 1, 2, 2] and plain["layers"][0] == ([0], [0], [1])
    assert plain["layers"][1][0][0] == 1                                                # the product gate
    assert sorted(plain["inputs"]) == sorted([("val", 0), ("var", 1), ("var", 2), ("var", 3)])
    scaled = _compiled(_build(_one([(5, 1)], [(1, 2)], [(P - 1, 3)])))[0]
    assert len(scaled["k"]) == 5 and ("val", 5) in scaled["inputs"]                     # one level deeper, constant 5 an input


def test_neg_flag_picks_the_reading_with_fewer_coefficient_products():
    # convert.rs:476-485: (-1) a * b = (+1) c  -> neg: A's -1 and C's +1 become bare variables, no constants at all
    neg = _compiled(_build(_one([(P - 1, 1)], [(1, 2)], [(1, 3)])))[0]
    assert all(kind == "var" or val == 0 for kind, val in neg["inputs"])
    # a tie is not "greater": (+1) a * b = (+1) c stays un-negated, so C's +1 is multiplied by -1
    tie = _compiled(_build(_one([(1, 1)], [(1, 2)], [(1, 3)])))[0]
    assert ("val", P - 1) in tie["inputs"]
    # B is never negated (:544-567): a -1 in B stays a constant under either reading
    b_neg = _compiled(_build(_one([(P - 1, 1)], [(P - 1, 2)], [(1, 3)])))[0]
    assert ("val", P - 1) in b_neg["inputs"]
    for r in (_one([(P - 1, 1)], [(1, 2)], [(1, 3)]), _one([(1, 1)], [(1, 2)], [(1, 3)]), _one([(P - 1, 1)], [(P - 1, 2)], [(1, 3)])):
        assert _compiled(_build(r)) == _oracle_compiled(r)


@pytest.mark.parametrize("n_terms", [1, 2, 3, 4, 5, 6, 7, 9])
def test_merge_nodes_pairs_neighbours_and_adds_an_odd_last_on_the_right(n_terms):
    # convert.rs:108-138 seen through the depth of the compiled circuit and the oracle's tree
    a = [(1, 1 + i) for i in range(n_terms)]
    r = _one(a, [(1, 1)], [(P - 1, 2)], n_wires=n_terms + 2)
    tree = oracle.merge_nodes([("var", 1 + i) for i in range(n_terms)])
    expect_depth = {1: 1, 2: 2, 3: 3, 4: 3, 5: 4, 6: 4, 7: 5, 9: 5}[n_terms]
    assert oracle.depth(tree) == expect_depth
    if n_terms % 2 == 1 and n_terms > 1:
        assert tree[0] == "add" and tree[2] == ("var", n_terms)          # the odd last element sits on the right
    got = _compiled(_build(r))
    assert got == _oracle_compiled(r) and len(got[0]["k"]) == expect_depth + 2 + 1   # + Mult + root Add, + input layer


def test_groups_are_sorted_by_depth_and_merged_pairwise_down_to_twenty():
    # convert.rs:164-186: 45 constraints of two depths -> stable sort, 45 -> 23 -> 12 circuits
    shallow = ([(1, 1)], [(1, 2)], [(P - 1, 3)])
    deep = ([(3, 1), (1, 2), (1, 3)], [(1, 2)], [(P - 1, 3)])
    cons = [deep if i % 3 == 0 else shallow for i in range(45)]
    r = dict(n_wires=4, n_pub_out=0, n_pub_in=0, n_prv_in=3, constraints=cons)
    got = _compiled(_build(r))
    assert len(got) == 12 and got == _oracle_compiled(r)
    depths = [len(c["k"]) for c in got]
    assert depths == sorted(depths)       # shallow groups first


def test_equal_subtrees_share_a_slot_and_leaves_are_relayed_through_add_gates():
    # convert.rs:288-303 (dedupe by structure) and :307-342 (relay gates Add(x, zero)); zero maps to (zero, zero)
    r = _one([(1, 1), (1, 2)], [(1, 1), (1, 2)], [(P - 1, 3)])       # A and B are the same tree v1 + v2
    c = _compiled(_build(r))[0]
    assert c == _oracle_compiled(r)[0]
    types, left, right = c["layers"][1]                              # layer under the root: Mult(A, B) and the leaf v3
    assert types[0] == 1 and left[0] == right[0]                     # both operands are the one shared slot
    assert types[1] == 0                                             # v3 relayed by an Add gate ...
    zero_slot = right[1]
    assert c["layers"][2][0][zero_slot] == 0                         # ... whose right operand is a relayed zero
    assert c["layers"][2][1][zero_slot] == c["layers"][2][2][zero_slot]   # zero = Add(zero', zero') one layer down
    pad = _compiled(_build(dict(n_wires=4, n_pub_out=0, n_pub_in=0, n_prv_in=3,
                                constraints=[([(1, 1)], [(1, 2)], [(P - 1, 3)])] * 3)))
    assert all(len(layer[0]) == 1 << k for circ in pad for layer, k in zip(circ["layers"], circ["k"]))   # padded to 2^k


def test_constraints_the_reference_cannot_compile_are_reported():
    # convert.rs:619-622 (empty A or B) and :612 (empty C): merge_nodes(vec![]) never returns
    for bad in (([], [(1, 2)], [(1, 3)]), ([(1, 1)], [], [(1, 3)]), ([(1, 1)], [(1, 2)], [])):
        good = ([(1, 1)], [(1, 2)], [(P - 1, 3)])
        r1cs = _build(dict(n_wires=4, n_pub_out=0, n_pub_in=0, n_prv_in=3, constraints=[good, good, bad]))
        with pytest.raises(GkrError) as e:
            r1cs.compile()
        assert e.value.status == N.GKR_ERR_UNSUPPORTED and "constraint 2" in str(e.value)
        with pytest.raises(RecursionError):
            oracle.convert_constraints_to_nodes([good, bad])


def test_input_values_gather_constants_and_witness_entries():
    # convert.rs:796-810: Value(c) -> c, Variable(w) -> witness[w]; a witness that is too short is an error, not a crash
    r = _one([(5, 1)], [(1, 2)], [(P - 1, 3)], n_wires=4)
    lay = _build(r).compile()
    spec = lay.input_layer(0)
    w = [1, 11, 13, 5 * 11 * 13 % P]
    vals = lay.input_values(0, w)
    assert vals == [val if kind == "val" else w[val] for kind, val in spec]
    with pytest.raises(GkrError):
        lay.input_values(0, w[:2])


def test_demo_circuit_builder_matches_the_committed_fixture_and_the_oracle():
    """gkr_amd.synth builds the t.circom-equivalent R1CS and its witnesses for bench.py without the oracle: they
    must be the committed fixture (written by the oracle's writers, tests/golden/make_mimc7_fixture.py) byte for
    byte."""
    import os
    from gkr_amd import synth
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for style, name in (("plain", "t_mimc7.r1cs"), ("negated", "t_mimc7_negated.r1cs")):
        assert synth.mimc7_demo_r1cs(style=style).serialize() == open(os.path.join(golden, name), "rb").read()
    for i, (in1, in2) in enumerate(synth.EXAMPLE_INPUTS, 1):
        w = synth.mimc7_demo_witness(in1, in2)
        assert w == oracle.mimc7_witness(in1, in2)
        assert product.write_wtns(w) == open(os.path.join(golden, "t_mimc7_input%d.wtns" % i), "rb").read()
        r = oracle.mimc7_r1cs()
        ev = lambda v: sum(c * w[j] for c, j in v) % P
        assert all(ev(a) * ev(b) % P == ev(c) for a, b, c in r["constraints"])
