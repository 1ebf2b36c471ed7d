"""The `gkr-aggregator prove -c CIRCUIT -i INPUTS...` surface (rust/src/bin.rs:8-27, aggregator.rs:385-435) on the
CPU: file discovery, the first proving step, <input>_output.json, aggregated.json and aggregated.circom.  The prover
is injected (the oracle's dense prover wrapped as gkr_amd.Proof objects) because there is no GPU here; the GPU test
runs the same command with the real one (tests/test_gpu_circom_pipeline.py)."""

import json
import os

import pytest

from gkr_amd import Proof, cli, synth
from gkr_amd.convert import write_wtns
from oracle import cdense, dense
from oracle import convert as oconv

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def oracle_prover(r1cs, witnesses):
    parsed = oconv.read_r1cs(r1cs.serialize())
    out = []
    for w in witnesses:
        proofs = []
        for sub in oconv.convert_r1cs_wtns_gkr(parsed, w):
            ref = cdense.prove(sub["layers"], sub["input_values"])
            ks = ref["k"]
            proofs.append(Proof(sumcheck_proofs=ref["sumcheck_proofs"], sumcheck_r=ref["sumcheck_r"],
                                d=dense.monomial_terms(ref["values"][0], ks[0]), q=ref["q"], z=ref["z"], r=ref["r"],
                                depth=ref["depth"], input_func=dense.monomial_terms(sub["input_values"], ks[-1]), k=ks))
        out.append(proofs)
    return out


def _inputs(tmp_path):
    paths = []
    for i, (a, b) in enumerate(synth.EXAMPLE_INPUTS, 1):
        p = tmp_path / ("input%d.json" % i)
        p.write_text(json.dumps({"in1": str(a), "in2": str(b)}))
        paths.append(str(p))
    return paths


def test_prove_demo_three_inputs_writes_the_reference_artefacts(tmp_path):
    inputs = _inputs(tmp_path)
    circuit = tmp_path / "t.circom"
    circuit.write_text("pragma circom 2.0.0;\ntemplate A(){\n    signal input in1;\n    signal input in2;\n    signal output out;\n}\n\n"
                       "component main {public [in1]}= A();\n")
    rc = cli.main(["prove", "-c", str(circuit), "-i"] + inputs + ["--demo", "--out-dir", str(tmp_path)], prover=oracle_prover)
    assert rc == 0
    out = json.load(open(tmp_path / "input1_output.json"))
    w = synth.mimc7_demo_witness(2, 3)
    assert out == {"out": str(w[1]), "in1": "2"}                       # MiMC7(2, key 0) and the public input
    agg = json.load(open(tmp_path / "aggregated.json"))
    assert agg["in1"] == "3" and agg["in2"] == "3"                      # the SECOND input's own signals ...
    assert all(("sumcheckProof%d" % j) in agg and ("inputFunc%d" % j) in agg for j in range(12)) and len(agg) == 2 + 7 * 12
    text = open(tmp_path / "aggregated.circom").read()
    assert text.startswith("pragma circom 2.0.0;\ninclude \"../gkr-verifier-circuits/circom/circom/verifier.circom\";\n")
    assert "component verifier[12];" in text and text.count("= VerifyGKR([") == 12
    assert text.index("verifier[11].inputFunc") < text.index("component main")   # injected before the template's closing brace


def test_prove_reads_circoms_files_and_sym_names(tmp_path):
    inputs = _inputs(tmp_path)[:1]
    circuit = tmp_path / "c.circom"
    circuit.write_text("pragma circom 2.0.0;\n")
    open(tmp_path / "c.r1cs", "wb").write(open(os.path.join(GOLDEN, "t_mimc7.r1cs"), "rb").read())
    open(tmp_path / "input1.wtns", "wb").write(write_wtns(synth.mimc7_demo_witness(2, 3)))
    (tmp_path / "c.sym").write_text("1,1,0,main.out\n2,2,0,main.in1\n3,3,0,main.in2\n")
    assert cli.main(["prove", "-c", str(circuit), "-i"] + inputs + ["--out-dir", str(tmp_path)], prover=oracle_prover) == 0
    out = json.load(open(tmp_path / "input1_output.json"))
    assert set(out) == {"out", "in1"} and out["in1"] == "2"
    assert not os.path.exists(tmp_path / "aggregated.json")            # a single input: one step, nothing to aggregate


def test_prove_without_circoms_products_says_what_is_missing(tmp_path):
    inputs = _inputs(tmp_path)[:1]
    with pytest.raises(SystemExit) as e:
        cli.main(["prove", "-c", str(tmp_path / "nothing.circom"), "-i"] + inputs, prover=oracle_prover)
    assert "circom" in str(e.value)
    assert cli.main(["mock-groth", "-z", "k.zkey"]) == 2
