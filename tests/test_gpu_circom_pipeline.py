"""BASELINE configs[0] and configs[3] end to end on the GPU: the R1CS of rust/t.circom (hand-written equivalent,
tests/golden/t_mimc7*.r1cs) and the witnesses of rust/example/input{1,2,3}.json -> R1CS compiler -> <= 20 layered
circuits -> prover::prove on every (circuit, input) pair -> verifier.circom input signals; then the same circuit
with 64 inputs, all proofs of a sub-circuit advancing together.

Checked: the product's compiler output equals the oracle's restatement of convert.rs; every proof is bit-equal to
the oracle prover's on the same circuit and input; output 0 of every sub-circuit is zero (convert.rs:838); every
proof passes the verifier; the circom input signals follow aggregator.rs:92-213.
What cannot be checked here: byte-equality with circom's own R1CS for t.circom (no circom offline), and the
recursion steps for inputs 2 and 3 (they need circom to compile the aggregated circuit, aggregator.rs:316-363):
each of the three inputs is proven as a first step."""

import os

import pytest

from gkr_amd import Context, synth, verify
from gkr_amd import convert as product
from gkr_amd.aggregate import aggregated_input, circom_input, circom_meta, prove_step
from helpers import canon_circom, expected_circom_input, expected_circom_meta
from oracle import cdense
from oracle import convert as oracle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _same_proof(pr, ref):
    return (pr.sumcheck_proofs == ref["sumcheck_proofs"] and pr.sumcheck_r == ref["sumcheck_r"] and pr.q == ref["q"]
            and pr.z == ref["z"] and pr.r == ref["r"] and pr.k == ref["k"])


@pytest.mark.parametrize("name", ["t_mimc7.r1cs", "t_mimc7_negated.r1cs"])
def test_config0_three_example_inputs_end_to_end(ctx, name):
    image = open(os.path.join(GOLDEN, name), "rb").read()
    r1cs = product.R1cs.parse(image)
    witnesses = [product.read_wtns(open(os.path.join(GOLDEN, "t_mimc7_input%d.wtns" % i), "rb").read()) for i in (1, 2, 3)]
    assert [w[2:4] for w in witnesses] == [list(p) for p in synth.EXAMPLE_INPUTS]
    proofs = prove_step(ctx, r1cs, witnesses)                     # require_zero_output: convert.rs:838
    circuits, _ = product.convert_r1cs_wtns_gkr(r1cs, witnesses[0])
    assert 1 <= len(circuits) <= 20 and all(len(p) == len(circuits) for p in proofs)
    for w, witness in enumerate(witnesses):
        want = oracle.convert_r1cs_wtns_gkr(oracle.read_r1cs(image), witness)
        assert len(want) == len(circuits)
        for j, (pr, sub) in enumerate(zip(proofs[w], want)):
            ref = cdense.prove(sub["layers"], sub["input_values"])
            assert ref["values"][0][0] == 0
            assert _same_proof(pr, ref), (w, j)
            assert pr.d == []                                      # every output of a satisfied sub-circuit is zero
            assert verify(pr, circuits[j]), (w, j)
            assert circom_meta(pr) == expected_circom_meta(pr)
            assert canon_circom(circom_input(pr, j)) == canon_circom(expected_circom_input(pr, j))
        merged = aggregated_input({"in1": str(witness[2]), "in2": str(witness[3])}, proofs[w])
        assert len(merged) == 2 + 7 * len(circuits)


def test_config0_unsatisfying_witness_is_refused(ctx):
    from gkr_amd import GkrError
    r1cs = synth.mimc7_demo_r1cs()
    w = synth.mimc7_demo_witness(2, 3)
    w[2] += 1      # in1 no longer matches t2[0] = in1 * in1: the first tree of the first sub-circuit is not zero
    with pytest.raises(GkrError):
        prove_step(ctx, r1cs, [w])


def test_config3_sixty_four_inputs_of_the_demo_circuit(ctx):
    """configs[3] on one GPU: 64 inputs of the t.circom-equivalent R1CS, 12 sub-circuits x 64 proofs, each
    sub-circuit's 64 proofs advancing together.  Every proof against the oracle for four of the inputs, the
    verifier on eight, and the built R1CS must be the committed fixture byte for byte."""
    r1cs = synth.mimc7_demo_r1cs()
    image = r1cs.serialize()
    assert image == open(os.path.join(GOLDEN, "t_mimc7.r1cs"), "rb").read()
    witnesses = [synth.mimc7_demo_witness(2 + i, 3 + (i % 5)) for i in range(64)]
    assert witnesses[0] == oracle.mimc7_witness(2, 3) and witnesses[63] == oracle.mimc7_witness(65, 6)
    proofs = prove_step(ctx, r1cs, witnesses)
    circuits, _ = product.convert_r1cs_wtns_gkr(r1cs, witnesses[0])
    assert len(proofs) == 64 and all(len(p) == len(circuits) == 12 for p in proofs)
    parsed = oracle.read_r1cs(image)
    for w in (0, 1, 31, 63):
        for j, sub in enumerate(oracle.convert_r1cs_wtns_gkr(parsed, witnesses[w])):
            assert _same_proof(proofs[w][j], cdense.prove(sub["layers"], sub["input_values"])), (w, j)
    for w in range(0, 64, 8):
        assert all(verify(proofs[w][j], circuits[j]) for j in range(12)), w


def test_concurrent_contexts_prove_the_same_transcripts(ctx):
    """The sub-circuits proven from several contexts at once (the reference's par_iter over the pairs,
    aggregator.rs:350-355), twice (second time from the cached circuits): same challenges as one context in turn."""
    import numpy as np
    from gkr_amd.aggregate import ProvingStep
    from gkr_amd.field import as_limbs
    step = ProvingStep(synth.mimc7_demo_r1cs())
    inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(5 + i, i)) for i in range(20)]))
    want = step.prove_raw(ctx, inputs)
    ctxs = ProvingStep.contexts_for(0, 8)
    try:
        for _ in range(2):
            got = step.prove_raw_concurrent(ctxs, inputs)
            assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))
    finally:
        for c in ctxs:
            c.close()
        step.close()


def test_prove_many_matches_proofs_proven_one_by_one(ctx):
    """gkr_prove_many (the whole proving step in one call, the library's own threads and child contexts): every output
    array of every item equals the same item proven alone with gkr_prove_batch -- for 1, 5 and all-CPUs threads, twice
    (the second call finds the circuits in the children's caches), with ragged batch sizes; a bad item fails alone."""
    import ctypes
    import numpy as np
    from gkr_amd import _native as N
    from gkr_amd.aggregate import ProvingStep
    from gkr_amd.field import as_limbs
    from gkr_amd.prover import GkrError
    step = ProvingStep(synth.mimc7_demo_r1cs())
    inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(7 + i, 2 * i)) for i in range(18)]))
    inputs = [x[:3 + (5 * j) % 16].copy() for j, x in enumerate(inputs)]       # 3 .. 18 witnesses per sub-circuit
    work = list(zip(step.circuits, inputs))
    with Context(0) as one:
        alone = one.prepare_many(work)
        for item, (circuit, x) in zip(alone["keep"], work):
            bufs = item[4]
            desc, _ = one._circuit_desc(circuit)
            rc = N.lib().gkr_prove_batch(one._h, ctypes.byref(desc), x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(x.shape[0]),
                                                       ctypes.c_int(1), bufs)
            assert rc == 0
    want = alone["outs"]
    with Context(0) as many:
        for threads in (1, 5, 0, 5):
            prepared = many.prepare_many(work)
            got = many.prove_many_raw(prepared, threads)
            for j in range(len(work)):
                for a, b in zip(got[j], want[j]):
                    assert np.array_equal(a, b), (threads, j)
        # one host process over several devices (gkr_ctx_create_multi; the one GPU listed four times: four logical devices,
        # the items dealt over child contexts created round-robin on them) -- the same proofs
        with Context(devices=[0, 0, 0, 0]) as multi:
            assert N.lib().gkr_ctx_device_count(multi._h) == 4 and N.lib().gkr_ctx_device_count(many._h) == 1
            for threads in (0, 3, 8):
                got = multi.prove_many_raw(multi.prepare_many(work), threads)
                for j in range(len(work)):
                    for a, b in zip(got[j], want[j]):
                        assert np.array_equal(a, b), ("multi", threads, j)
        with pytest.raises(GkrError):
            Context(devices=[0, 99])
        # an item whose witness does not satisfy the circuit: that item's status, the others still proven
        bad = many.prepare_many(work)
        bad["items"][0].require_zero_output = 1
        bad["keep"][0][2][0, :, 0] ^= 1                                    # every input value of one witness changed
        with pytest.raises(GkrError):
            many.prove_many_raw(bad, 4)
        assert [it.status for it in bad["items"]].count(0) == len(work) - 1 and bad["items"][0].status != 0
        for j in range(len(work)):
            if j != 0:
                assert np.array_equal(bad["outs"][j][2], want[j][2]), j
    step.close()


@pytest.mark.parametrize("pieces", ["0", "14", "19"])
def test_prove_many_with_items_cut_in_two(pieces):
    """GKR_PROVE_MANY_PIECES: gkr_prove_many cuts its costliest items (>= 32 witnesses) in two until there are that many --
    the halves write into the caller's proof buffers at their offsets.  40 inputs x 12 sub-circuits, every proof against
    the committed digests of the CPU checker's proofs (tests/prove_many_pieces_worker.py)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "prove_many_pieces_worker.py")], env=dict(os.environ, GKR_PROVE_MANY_PIECES=pieces),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_cli_prove_demo_on_the_gpu(tmp_path):
    """`gkr-aggregator prove -c t.circom -i input1.json input2.json input3.json` (bin.rs:17-22) with the real prover:
    the artefacts of the first step and the inputs of the second (tests/test_cli.py checks their content on the CPU
    with an injected prover; here the proofs come from the GPU and must satisfy the verifier)."""
    import json
    from gkr_amd import cli
    paths = []
    for i, (a, b) in enumerate(synth.EXAMPLE_INPUTS, 1):
        p = tmp_path / ("input%d.json" % i)
        p.write_text(json.dumps({"in1": str(a), "in2": str(b)}))
        paths.append(str(p))
    circuit = tmp_path / "t.circom"
    circuit.write_text("pragma circom 2.0.0;\ntemplate A(){\n    signal input in1;\n}\ncomponent main = A();\n")
    args = cli.argparse.Namespace(circuit=str(circuit), inputs=paths, r1cs=None, sym=None, wtns=None, demo=True, out_dir=str(tmp_path), device=0)
    proofs, written = cli.prove_all(args, log=lambda *_: None)
    circuits, _ = product.convert_r1cs_wtns_gkr(synth.mimc7_demo_r1cs(), synth.mimc7_demo_witness(2, 3))
    assert len(proofs) == 12 and all(verify(p, c) for p, c in zip(proofs, circuits))
    assert [os.path.basename(w) for w in written] == ["input1_output.json", "aggregated.json", "aggregated.circom"]
    agg = json.load(open(tmp_path / "aggregated.json"))
    assert agg["in1"] == "3" and len(agg) == 2 + 7 * 12


def test_large_r1cs_end_to_end_through_wide_layers():
    """The reference's use case at size: an R1CS of 262 144 constraints (the demo's constraint shapes, 65 536 rounds) -> the
    product's compiler -> 16 layered circuits with layers of 2^14 .. 2^16 values -> gkr_prove_many, output 0 of every
    sub-circuit required to be zero (convert.rs:838) -> every proof's arrays against the CPU checker's linear-time prover on
    the same circuits, and -- when the committed file is there -- against the checker's END-TO-END digests, its own
    restatement of convert.rs having compiled the R1CS (tests/golden/large_r1cs_digests.json, made offline)."""
    import numpy as np
    from gkr_amd.aggregate import ProvingStep
    from gkr_amd.field import as_limbs
    nrounds, pair = 65536, (2, 3)
    step = ProvingStep(synth.mimc7_demo_r1cs(nrounds=nrounds))
    ks_all = [c.get_k_list() for c in step.circuits]
    assert max(max(ks) for ks in ks_all) >= 15
    inputs = step.inputs_for(np.stack([as_limbs(synth.mimc7_demo_witness(pair[0], pair[1], nrounds=nrounds))]))
    gold = synth.large_r1cs_digests()
    with Context(0) as ctx:
        prepared = ctx.prepare_many(list(zip(step.circuits, inputs)), require_zero_output=True)
        outs = ctx.prove_many_raw(prepared, 0)
        for j, (circuit, arrs) in enumerate(zip(step.circuits, outs)):
            ks = ks_all[j]
            if j % 5 == 0:      # (the checker's prover on four of the sixteen: ~2 s each)
                ref = cdense.prove_raw([lay.arrays() for lay in circuit.layer], inputs[j][0])
                want = synth.proof_arrays_from_checker(ref, ks)
                assert all(np.array_equal(np.asarray(a[0]).reshape(w.shape), w) for a, w in zip(arrs[:7], want)), j
            if gold is not None:
                assert ks == gold["k"][j] and synth.proof_arrays_digest(ks, *[a[0] for a in arrs[:7]]) == gold["digests"][j], j
    step.close()
