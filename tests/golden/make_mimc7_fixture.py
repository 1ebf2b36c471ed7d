#!/usr/bin/env python3
"""Fixture of BASELINE configs[0]: a hand-written R1CS equivalent to /root/reference/rust/t.circom and the
witnesses of /root/reference/rust/example/input{1,2,3}.json, as iden3 `.r1cs` / `.wtns` files.

    python tests/golden/make_mimc7_fixture.py     # rewrites tests/golden/t_mimc7*.{r1cs,wtns}

Written with the ORACLE's container writers and witness calculator (oracle/convert.py) -- circom and its witness
generator are not available offline, so circom's own coefficient signs and wire order are not reproduced (see
oracle.convert.mimc7_r1cs).  The product's writers must produce the same bytes (tests/test_convert.py)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import convert, mimc7  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
EXAMPLES = "/root/reference/rust/example/input%d.json"


def main():
    for style in ("plain", "negated"):
        r = convert.mimc7_r1cs(style=style)
        name = "t_mimc7.r1cs" if style == "plain" else "t_mimc7_negated.r1cs"
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(convert.write_r1cs(r["n_wires"], r["n_pub_out"], r["n_pub_in"], r["n_prv_in"], r["constraints"]))
    for i in (1, 2, 3):
        try:
            inp = json.load(open(EXAMPLES % i))
            in1, in2 = int(inp["in1"]), int(inp["in2"])
        except OSError:   # outside the build container: the three example inputs, restated
            in1, in2 = [(2, 3), (3, 3), (3, 4)][i - 1]
        w = convert.mimc7_witness(in1, in2)
        assert w[1] == mimc7.mimc7_hash(in1, 0)
        with open(os.path.join(HERE, "t_mimc7_input%d.wtns" % i), "wb") as f:
            f.write(convert.write_wtns(w))
    print("wrote", sorted(n for n in os.listdir(HERE) if n.startswith("t_mimc7")))


if __name__ == "__main__":
    main()
