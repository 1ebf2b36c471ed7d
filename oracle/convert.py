"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- never imported by the product.

CPU restatement of the step in front of the hot path: R1CS + witness -> the <= 20 layered GKR circuits
(rust/src/convert.rs), SURVEY.md section 8f row f1 / appendix B.  Function by function:

    get_k                          convert.rs:140-152
    merge_nodes                    convert.rs:108-138
    convert_constraints_to_nodes   convert.rs:360-632   (count_mult :363-379, neg flag :476-485)
    compile                        convert.rs:154-358
    calculate_input (values only)  convert.rs:787-838
    gate lists of a layer          convert.rs:703-777   (the 0/1 wire strings are a function of them)

and the two iden3 binary containers the reference reads through the third-party crates `r1cs-file` / `wtns-file`
(git jeong0982/zeropool-utils, no rev pinned, absent from /root/reference: the published iden3 formats are
restated -- `.r1cs` magic "r1cs" version 1, sections 1 header / 2 constraints / 3 wire2label; `.wtns` magic "wtns"
version 2, sections 1 header / 2 values; all integers little-endian, field elements 32 bytes little-endian).

Parity status: the reference cannot be run here (no Rust toolchain, no circom) and holds no fixture for this
step, so this restatement is "parity unpinned": it is checked for internal consistency (every constraint tree
evaluates to 0 on a satisfying witness, every compiled sub-circuit's output 0 is zero, convert.rs:838) and is the
twin the product's C++ compiler (gkr_amd/csrc/r1cs.cpp) is compared with, structure for structure.

Trees are nested tuples, so Python's == is the deep structural equality of convert.rs:33-57 on well-formed trees
(a gate has both children, a leaf none; the reference's comparison of `self.right.is_some()` with itself, :37, can
only matter for trees that are not well formed, which convert_constraints_to_nodes never builds):

    ("val", c)          NodeType::Value(Expression::Value(c))         c: int, canonical
    ("var", w)          NodeType::Value(Expression::Variable(w))
    ("mul", l, r)       NodeType::Mult
    ("add", l, r)       NodeType::Add
"""

import struct

from .field import P

WIDTH_LIMIT = 20   # convert.rs:11
ZERO = ("val", 0)  # zero_node, convert.rs:93-100


# ------------------------------------------------------------------------------------------------ iden3 containers

def write_r1cs(n_wires, n_pub_out, n_pub_in, n_prv_in, constraints, n_labels=None, wire2label=None):
    """constraints: list of (A, B, C), each a list of (coeff, wire) -- the tuple order of the reference's
    `Constraint` (convert.rs:368: `for (coeff, x_i) in v`); on disk a term is wire id, then coefficient."""
    hdr = struct.pack("<I", 32) + P.to_bytes(32, "little")
    hdr += struct.pack("<IIIIQI", n_wires, n_pub_out, n_pub_in, n_prv_in, n_wires if n_labels is None else n_labels,
                       len(constraints))
    body = b""
    for lcs in constraints:
        for lc in lcs:
            body += struct.pack("<I", len(lc))
            for coeff, wire in lc:
                body += struct.pack("<I", wire) + (coeff % P).to_bytes(32, "little")
    labels = wire2label if wire2label is not None else list(range(n_wires))
    w2l = b"".join(struct.pack("<Q", x) for x in labels)
    out = b"r1cs" + struct.pack("<II", 1, 3)
    for ty, data in ((1, hdr), (2, body), (3, w2l)):
        out += struct.pack("<IQ", ty, len(data)) + data
    return out


def _sections(data, magic, version):
    if data[:4] != magic:
        raise ValueError("bad magic")
    ver, n_sec = struct.unpack_from("<II", data, 4)
    if ver != version:
        raise ValueError("unsupported version %d" % ver)
    off, out = 12, {}
    for _ in range(n_sec):
        ty, size = struct.unpack_from("<IQ", data, off)
        off += 12
        if off + size > len(data):
            raise ValueError("truncated section")
        out.setdefault(ty, data[off:off + size])   # sections may come in any order
        off += size
    return out


def read_r1cs(data):
    sec = _sections(data, b"r1cs", 1)
    hdr = sec[1]
    (fs,) = struct.unpack_from("<I", hdr, 0)
    if fs != 32 or int.from_bytes(hdr[4:36], "little") != P:
        raise ValueError("not a BN254 R1CS")
    n_wires, n_pub_out, n_pub_in, n_prv_in, n_labels, n_cons = struct.unpack_from("<IIIIQI", hdr, 36)
    body, off, cons = sec[2], 0, []
    for _ in range(n_cons):
        lcs = []
        for _ in range(3):
            (n,) = struct.unpack_from("<I", body, off)
            off += 4
            lc = []
            for _ in range(n):
                (wire,) = struct.unpack_from("<I", body, off)
                lc.append((int.from_bytes(body[off + 4:off + 36], "little"), wire))
                off += 36
            lcs.append(lc)
        cons.append(tuple(lcs))
    return dict(n_wires=n_wires, n_pub_out=n_pub_out, n_pub_in=n_pub_in, n_prv_in=n_prv_in, n_labels=n_labels,
                constraints=cons)


def write_wtns(values):
    hdr = struct.pack("<I", 32) + P.to_bytes(32, "little") + struct.pack("<I", len(values))
    body = b"".join((v % P).to_bytes(32, "little") for v in values)
    out = b"wtns" + struct.pack("<II", 2, 2)
    for ty, data in ((1, hdr), (2, body)):
        out += struct.pack("<IQ", ty, len(data)) + data
    return out


def read_wtns(data):
    sec = _sections(data, b"wtns", 2)
    hdr = sec[1]
    (fs,) = struct.unpack_from("<I", hdr, 0)
    if fs != 32 or int.from_bytes(hdr[4:36], "little") != P:
        raise ValueError("not a BN254 witness")
    (n,) = struct.unpack_from("<I", hdr, 36)
    return [int.from_bytes(sec[2][32 * i:32 * i + 32], "little") for i in range(n)]


# ------------------------------------------------------------------------------------------------ trees

def depth(node):
    """IntermediateNode::depth, convert.rs:86-90: a leaf has depth 1."""
    return 1 if node[0] in ("val", "var") else 1 + max(depth(node[1]), depth(node[2]))


def evaluate(node, witness):
    if node[0] == "val":
        return node[1] % P
    if node[0] == "var":
        return witness[node[1]] % P
    l, r = evaluate(node[1], witness), evaluate(node[2], witness)
    return (l * r if node[0] == "mul" else l + r) % P


def merge_nodes(nodes):
    """convert.rs:108-138: pair neighbours with Add; with an odd count the pairs are merged first and the last
    element is added on the right.  An empty list recurses without end in the reference (:116-136)."""
    if not nodes:
        raise RecursionError("merge_nodes([]): unbounded recursion in the reference (convert.rs:108-138)")
    if len(nodes) == 1:
        return nodes[0]
    new = [("add", nodes[2 * i], nodes[2 * i + 1]) for i in range(len(nodes) // 2)]
    if len(nodes) % 2 == 1:
        return ("add", merge_nodes(new), nodes[-1])
    return merge_nodes(new)


def get_k(n):
    """convert.rs:140-152."""
    k, m = 0, n
    while m > 1:
        m >>= 1
        k += 1
    return k if n & (n - 1) == 0 else k + 1


def count_mult(lc):
    """convert.rs:363-379 -> (a, b)."""
    a = b = 0
    for coeff, _ in lc:
        if coeff % P == 1:
            b += 1
        elif coeff % P == P - 1:
            a += 1
        else:
            a += 1
            b += 1
    return a, b


def _term(coeff, wire, unit):
    """Variable(wire) if coeff == unit, else Mult(Value(coeff'), Variable(wire)); coeff' is coeff, or -coeff when
    the unit is -1 (the negated reading), convert.rs:512-542, 554-566, 578-610."""
    coeff %= P
    if coeff == unit:
        return ("var", wire)
    return ("mul", ("val", coeff if unit == 1 else (P - coeff) % P), ("var", wire))


def convert_constraints_to_nodes(constraints):
    """convert.rs:360-632.  The symbol-table shortcut (:487-511, :545-553) never fires: its only insertion site is
    commented out (:576), so `used` stays empty and every constraint yields one single-tree group (:625-631)."""
    groups = []
    for a, b, c in constraints:
        ca, cb, cc = count_mult(a), count_mult(b), count_mult(c)
        neg = (ca[0] + cb[0] + cc[1]) > (ca[1] + cb[1] + cc[0])   # :480-485
        node_a = [_term(co, w, P - 1 if neg else 1) for co, w in a]     # :486-543
        node_b = [_term(co, w, 1) for co, w in b]                        # :544-567 (never negated)
        if node_a and node_b:                                            # :568
            a_times_b = ("mul", merge_nodes(node_a), merge_nodes(node_b))
            node_c = [_term(co, w, 1 if neg else P - 1) for co, w in c]  # :578-611
            groups.append([("add", a_times_b, merge_nodes(node_c))])     # :612-618
        else:
            groups.append([merge_nodes([])])                             # :619-622 -> unbounded recursion
    return groups


# ------------------------------------------------------------------------------------------------ layering

def compile_groups(groups):
    """compile, convert.rs:154-358 -> (circuits, inputs): circuits[j] = list of layers (gate_type, left, right) with
    gate_type 0 = Add / 1 = Mult, layer 0 first; inputs[j] = list of ("val", c) / ("var", w) of the input layer."""
    ordered = sorted(groups, key=lambda g: max((depth(n) for n in g), default=0))   # stable, :164-169
    while len(ordered) > WIDTH_LIMIT:                                                # :171-186
        merged = [ordered[2 * i] + ordered[2 * i + 1] for i in range(len(ordered) // 2)]
        if len(ordered) % 2 == 1:
            merged.append(ordered[-1])
        ordered = merged
    circuits, all_inputs = [], []
    for one in ordered:
        layers = []
        height = max((depth(n) for n in one), default=0)
        if height == 0:
            return [layers], []                                                      # :197-199
        current, inputs = list(one), None
        for d in range(height + 1):                                                  # :206
            current = current + [ZERO] * ((1 << get_k(len(current))) - len(current))  # :209-214
            if d == height:                                                          # :215-221
                inputs = list(current)
                assert all(n[0] in ("val", "var") for n in inputs)
                break
            types, ops, nxt, used, zero_index = [], [], [], {}, None
            for node in current:
                if node[0] in ("mul", "add"):                                        # :280-306
                    if d == height - 1:
                        raise ValueError("Unsupported")                              # :225-227
                    types.append(1 if node[0] == "mul" else 0)
                    idx = []
                    for child in (node[1], node[2]):
                        if child in nxt:                                             # deep equality, first match
                            idx.append(nxt.index(child))
                        else:
                            nxt.append(child)
                            idx.append(len(nxt) - 1)
                    ops.append(tuple(idx))
                else:                                                                # leaf: relay gate, :307-342 / :228-264
                    types.append(0)
                    if node in used:
                        ops.append((used[node], zero_index))
                        continue
                    if zero_index is None:
                        zero_index = len(nxt)
                        nxt.append(ZERO)
                    if node == ZERO:
                        used[node] = zero_index
                        ops.append((zero_index, zero_index))
                    else:
                        used[node] = len(nxt)
                        ops.append((len(nxt), zero_index))
                        nxt.append(node)
            layers.append((types, [o[0] for o in ops], [o[1] for o in ops]))
            current = nxt
        circuits.append(layers)
        all_inputs.append(inputs)
    return circuits, all_inputs


def input_values(inputs, witness):
    """convert.rs:796-810."""
    return [n[1] % P if n[0] == "val" else witness[n[1]] % P for n in inputs]


def forward_values(layers, values):
    """convert.rs:812-831 -> value vectors, output layer first."""
    out = [list(values)]
    for types, left, right in reversed(layers):
        prev = out[-1]
        out.append([(prev[l] * prev[r] if t else prev[l] + prev[r]) % P for t, l, r in zip(types, left, right)])
    out.reverse()
    return out


def convert_r1cs_wtns_gkr(r1cs, witness):
    """convert.rs:667-785 without the term-list forms: per sub-circuit the k list, the gate lists and the input
    layer's values; asserts output 0 == 0 (:838)."""
    circuits, inputs = compile_groups(convert_constraints_to_nodes(r1cs["constraints"]))
    out = []
    for layers, inp in zip(circuits, inputs):
        vals = input_values(inp, witness)
        w = forward_values(layers, vals)
        assert w[0][0] == 0, "d_values[0] != 0 (convert.rs:838)"
        ks = [get_k(len(t[0])) for t in layers] + [get_k(len(inp))]
        out.append(dict(k=ks, layers=layers, inputs=inp, input_values=vals))
    return out


def wire_strings(layers, input_k):
    """The 0/1 strings of convert.rs:715-767 for every layer: (add strings, mult strings), gate order."""
    out = []
    for i, (types, left, right) in enumerate(layers):
        k_i = get_k(len(types))
        k_next = input_k if i == len(layers) - 1 else get_k(len(layers[i + 1][0]))
        adds, mults = [], []
        for g, t in enumerate(types):
            s = (format(g, "0%db" % k_i) if k_i else "") + format(left[g], "0%db" % k_next) + format(right[g], "0%db" % k_next)
            (mults if t else adds).append(s)
        out.append((adds, mults))
    return out


# ------------------------------------------------------------------------------------------------ the demo circuit

def mimc7_r1cs(nrounds=91, style="plain"):
    """A hand-written R1CS equivalent to rust/t.circom (circomlib MiMC7(91) with k = 0 on the public input in1, a
    private input in2 that is not used, public output out) after circom's linear simplification: four quadratic
    constraints per round (t^2, t^4, t^6, t^7).  Wires: 0 one, 1 out, 2 in1, 3 in2, then t2, t4, t6, t7 of every
    round.  circom itself is not available here, so its coefficient signs and wire order are NOT reproduced;
    style = "negated" writes every constraint as (-A) * B = -C, the shape circom tends to emit, to exercise the
    reference's neg heuristic (convert.rs:476-485)."""
    from .mimc7 import CTS as cts
    wire = 4
    cons = []
    prev_t7 = None

    def emit(a, b, c):
        if style == "negated":
            a = [((P - co) % P, w) for co, w in a]
            c = [((P - co) % P, w) for co, w in c]
        cons.append((a, b, c))
    for i in range(nrounds):
        t = [(1, 2)] if i == 0 else [(cts[i] % P, 0), (1, prev_t7)]
        if i > 0 and cts[i] % P == 0:
            t = [(1, prev_t7)]
        t2, t4, t6 = wire, wire + 1, wire + 2
        wire += 3
        emit(list(t), list(t), [(1, t2)])
        emit([(1, t2)], [(1, t2)], [(1, t4)])
        emit([(1, t4)], [(1, t2)], [(1, t6)])
        if i < nrounds - 1:
            t7 = wire
            wire += 1
            emit([(1, t6)], list(t), [(1, t7)])
            prev_t7 = t7
        else:
            emit([(1, t6)], list(t), [(1, 1)])
    return dict(n_wires=wire, n_pub_out=1, n_pub_in=1, n_prv_in=1, n_labels=wire, constraints=cons)


def mimc7_witness(in1, in2, nrounds=91):
    """The witness circom's generated calculator would produce for mimc7_r1cs: wire order as above."""
    from .mimc7 import CTS as cts
    w = [1, 0, in1 % P, in2 % P]
    t7 = None
    for i in range(nrounds):
        t = in1 % P if i == 0 else (t7 + cts[i]) % P
        t2 = t * t % P
        t4 = t2 * t2 % P
        t6 = t4 * t2 % P
        w += [t2, t4, t6]
        t7 = t6 * t % P
        if i < nrounds - 1:
            w.append(t7)
    w[1] = t7   # out = t6 * t + k, k = 0
    return w
