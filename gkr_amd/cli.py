"""Command-line surface of the reference's `gkr-aggregator` binary (rust/src/bin.rs:8-27) over this library.

    python -m gkr_amd prove -c CIRCUIT.circom -i INPUT.json [INPUT.json ...]
                            [--r1cs CIRCUIT.r1cs] [--sym CIRCUIT.sym] [--wtns INPUT.wtns ...] [--demo] [--out-dir DIR]
    python -m gkr_amd mock-groth -z KEY.zkey

`prove` mirrors prove_all (aggregator.rs:385-435) as far as it can without circom:

  * the reference shells out to `circom` and to the generated witness calculator (file_utils.rs:74-114) for
    CIRCUIT.r1cs / CIRCUIT.sym / witness.wtns; this build never spawns them -- it reads those files (next to the
    circuit / the input, or where --r1cs / --sym / --wtns point).  `--demo` synthesises them for rust/t.circom
    (hand-written equivalent R1CS, gkr_amd.synth) from the inputs' in1 / in2;
  * first input: convert_r1cs_wtns_gkr + prover::prove on every (circuit, input) pair on the GPU, then
    `<input>_output.json` with the public signals (write_output, file_utils.rs:42-47; names from the .sym file);
  * every further input needs the PREVIOUS step's proofs verified inside the circuit: the reference writes
    aggregated.json (input + proof signals, file_utils.rs:49-67) and aggregated.circom (modify_circom_file,
    aggregator.rs:215-314) and runs circom on them.  Both files are written here; compiling aggregated.circom is the
    user's step, after which `prove -c aggregated.circom --r1cs ... --wtns ...` continues the chain.
"""

import argparse
import json
import os
import sys


def _stem(path):
    """get_name, file_utils.rs:69-74: file name up to its first dot."""
    return os.path.basename(path).split(".")[0]


def parse_sym(path, num_public):
    """parse_sym, convert.rs:851-871: 4th CSV column of the first num_public lines, the part after the first dot."""
    names = []
    if num_public == 0:
        return names
    with open(path) as f:
        for line in f.read().splitlines():
            names.append(line.split(",")[3].split(".")[1])
            if len(names) == num_public:
                break
    return names


def write_output(path, witness, names):
    """make_output + write_output (convert.rs:652-665, file_utils.rs:30-47): {name: decimal} of witness[1..n_pub]."""
    with open(path, "w") as f:
        json.dump({name: str(witness[i + 1]) for i, name in enumerate(names)}, f)


def gpu_prover(device=0):
    from .aggregate import prove_step
    from .prover import Context
    ctx = Context(device)
    return lambda r1cs, witnesses: prove_step(ctx, r1cs, witnesses)


def prove_all(args, prover=None, log=print):
    from . import synth
    from .aggregate import aggregated_input, circom_meta, modify_circom_file
    from .convert import R1cs, read_wtns
    out_dir = args.out_dir or os.getcwd()
    if not args.inputs:
        raise SystemExit("prove: at least one --inputs file")
    inputs = [json.load(open(p)) for p in args.inputs]
    if args.demo:
        r1cs = synth.mimc7_demo_r1cs()
        witnesses = [synth.mimc7_demo_witness(int(i["in1"]), int(i["in2"])) for i in inputs]
        names = ["out", "in1"]
    else:
        base = os.path.splitext(args.circuit)[0]
        r1cs_path = args.r1cs or (base + ".r1cs" if os.path.exists(base + ".r1cs") else os.path.join(os.getcwd(), _stem(args.circuit) + ".r1cs"))
        if not os.path.exists(r1cs_path):
            raise SystemExit("prove: %s not found -- compile the circuit with `circom %s --r1cs --sym --wasm` first (this build does "
                             "not spawn circom) or pass --r1cs / --demo" % (r1cs_path, args.circuit))
        r1cs = R1cs.read(r1cs_path)
        wtns_paths = args.wtns or [os.path.splitext(p)[0] + ".wtns" for p in args.inputs[:1]]
        for p in wtns_paths:
            if not os.path.exists(p):
                raise SystemExit("prove: witness %s not found (generate it with the circuit's witness calculator) or pass --wtns" % p)
        witnesses = [read_wtns(open(p, "rb").read()) for p in wtns_paths]
        info = r1cs.info()
        n_pub = info["n_pub_in"] + info["n_pub_out"]
        sym = args.sym or os.path.splitext(r1cs_path)[0] + ".sym"
        names = parse_sym(sym, n_pub) if os.path.exists(sym) else ["w%d" % (i + 1) for i in range(n_pub)]
    log("r1cs is converted to GKR intermediate layers")
    log("Proving starts..")
    prover = prover or gpu_prover(args.device)
    proofs = prover(r1cs, witnesses[:1])[0]          # first step: the (circuit, input) pairs of inputs[0]
    log("Proving done: %d proofs" % len(proofs))
    out_path = os.path.join(out_dir, "%s_output.json" % _stem(args.inputs[0]))
    write_output(out_path, witnesses[0], names)
    log("%s written" % out_path)
    written = [out_path]
    if len(args.inputs) > 1:
        # the next step's circuit input (prove_recursively_circom / prove_groth, aggregator.rs:316-383)
        agg_json = os.path.join(out_dir, "aggregated.json")
        with open(agg_json, "w") as f:
            json.dump(aggregated_input(inputs[1], proofs), f, indent=2)
        written.append(agg_json)
        if args.circuit and os.path.exists(args.circuit):
            agg_circom = os.path.join(out_dir, "aggregated.circom")
            with open(agg_circom, "w") as f:
                f.write(modify_circom_file(open(args.circuit).read(), [circom_meta(p) for p in proofs]))
            written.append(agg_circom)
            log("%s generated" % agg_circom)
        log("next: compile aggregated.circom with circom, compute its witness on aggregated.json, then "
            "`prove -c aggregated.circom --r1cs aggregated.r1cs --wtns witness.wtns -i %s`" % " ".join(args.inputs[1:]))
    return proofs, written


def main(argv=None, prover=None):
    ap = argparse.ArgumentParser(prog="gkr-aggregator", description="GKR proof aggregation on MI355X (CLI of jeong0982/gkr's rust/src/bin.rs)")
    sub = ap.add_subparsers(dest="command")
    p = sub.add_parser("prove", help="prove the circuit for the inputs (bin.rs:17-22)")
    p.add_argument("-c", "--circuit", default="", help="the .circom file")
    p.add_argument("-i", "--inputs", nargs="*", default=[], help="input JSON files")
    p.add_argument("--r1cs", help="circom's CIRCUIT.r1cs (default: next to the circuit / in the working directory)")
    p.add_argument("--sym", help="circom's CIRCUIT.sym (names of the public signals)")
    p.add_argument("--wtns", nargs="*", help="witness of the first input (default: INPUT.wtns)")
    p.add_argument("--demo", action="store_true", help="rust/t.circom: synthesise the R1CS and the witnesses from in1 / in2")
    p.add_argument("--out-dir", help="where the output files go (default: the working directory)")
    p.add_argument("--device", type=int, default=0)
    g = sub.add_parser("mock-groth", help="bin.rs:23-26: needs snarkjs, which this build does not spawn")
    g.add_argument("-z", "--zkey", required=True)
    args = ap.parse_args(argv)
    if args.command == "prove":
        prove_all(args, prover)
        return 0
    if args.command == "mock-groth":
        print("mock-groth runs `snarkjs zkey verify` / `snarkjs groth16 prove` on aggregated.r1cs in the reference "
              "(bin.rs:27-61); this build does not spawn external tools -- run them on the files `prove` wrote.", file=sys.stderr)
        return 2
    ap.print_help()
    return 0


if __name__ == "__main__":
    sys.exit(main())
