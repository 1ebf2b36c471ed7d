"""bench.py's output contract, on a small instance of every leg: stdout is exactly ONE JSON line with the driver's
keys, the roofline and cpu_baseline objects, the check of the timed outputs, and all five BASELINE configs."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_LIMIT = 6000      # bytes; the driver did not parse round 5's 20 KB line
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config")


def _run(*args, env=None, timeout=900):
    """-> (the ONE stdout line, parsed; the run's bench_detail.json).  Checks on every run: exit code 0, exactly one line,
    the line under LINE_LIMIT bytes with every contract key."""
    import tempfile
    detail = tempfile.NamedTemporaryFile(suffix=".json", delete=False).name
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout,
                         env=dict(os.environ, GKR_BENCH_DETAIL=detail, **(env or {})))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line, got %d:\n%s" % (len(lines), out.stdout[:2000])
    assert len(lines[0].encode()) < LINE_LIMIT, "the stdout line is %d bytes" % len(lines[0])
    line = json.loads(lines[0])
    for key in CONTRACT_KEYS:
        assert key in line, key
    with open(detail) as f:
        full = json.load(f)
    os.unlink(detail)
    for key in ("value", "ms_per_step", "n_gpus", "steps", "warmup"):
        assert line[key] == pytest.approx(full[key], rel=1e-6), key
    return line, full


def test_the_drivers_exact_command_gives_a_short_line():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` -- the driver's command, full sizes, every leg: the line parses, is under
    6 000 bytes and carries roofline + cpu_baseline + verified + the flat legs summary."""
    line, full = _run("--gpus", "1", "--steps", "20", "--warmup", "5", timeout=1500)
    assert line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1
    r = line["roofline"]
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "algorithmic_bytes_per_launch"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0.5 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert abs(r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9 - r["achieved"]) < 1e-3 * r["achieved"]
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert line["verified"]["ok"] is True and line["exit"] == {"code": 0, "why": "ok"}
    legs = line["legs"]
    for name in ("n16", "layer24", "wide20", "wide_prove", "config0", "config3", "multi_device", "large_r1cs"):
        assert name in legs and legs[name]["ok"] is True and legs[name]["ms"] > 0, (name, legs.get(name))
    # one device: the one-process-over-all-devices path may not cost more than 3 % over the direct one
    assert full["aggregated_proofs"]["multi_device"]["device_ids"] == list(range(full["aggregated_proofs"]["multi_device"]["devices_seen"]))


def test_default_mode_line():
    line, d = _run("--batch", "64", "--steps", "2", "--warmup", "1", "--proofs", "4", "--cpu-seconds", "0.5", "--ref-algo-seconds", "1",
                   "--layer-k-i", "20", "--layer-k", "10")
    for key in ("roofline", "cpu_baseline", "verified", "legs", "exit"):
        assert key in line, key
    assert line["roofline"]["frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-6) and line["cpu_baseline"]["kind"] == "port"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and d["unit"] == "field-ops/s" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["copy_GBps_measured"] > 1000 and r["alu_products_per_sec_measured"] > 1e10
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # what was timed was checked: digests of rank 0's tables (the seeds are the full run's) and the verifier's relations
    v = d["verified"]
    assert v["ok"] is True and v["tables_matching_their_digest"] == 64 and v["table0_golden_digest"] is True and v["verifier_relations"]["ok"] is True
    # the other configs on the line
    assert d["n16"]["value"] > 0 and d["n16"]["roofline"]["floor_ms"] > 0
    assert d["layer24"]["matches_golden_digest"] is True and d["layer24"]["roofline"]["bound"] == "alu"
    assert d["aggregated_proofs"]["config0_three_inputs"]["proofs"] == 36 and d["aggregated_proofs"]["config3"]["proofs"] == 4 * 12
    assert d["host_transcript"]["floor_ms"] > 0


def test_layer_split_mode_with_a_process_group_of_one_rank():
    """--mode layer-split under GKR_BENCH_FORCE_GROUP=1: backend nccl (RCCL) initialised before the first GPU call, both
    exchanges of every sumcheck as all-reduces on the library's stream; golden transcript."""
    line, d = _run("--mode", "layer-split", "--k-i", "20", "--k", "10", "--steps", "3", "--warmup", "1",
             env={"GKR_BENCH_FORCE_GROUP": "1", "MASTER_PORT": "29633", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["matches_golden_digest"] is True and d["exchange"]["calls_per_step"] == 2.0 and d["exchange"]["us_per_call"] > 0
    assert d["roofline"]["bound"] == "alu" and d["scaling"] == "strong"
    assert line["verified"]["ok"] is True and line["exchange"]["calls_per_step"] == 2.0


def test_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it: the process starts torch.distributed.run itself (a child,
    before anything touched the GPU), two ranks come up -- here both on the one GPU over gloo, the single-device test
    hook -- and the one JSON line says n_gpus 2 with every leg verified on every rank."""
    env = {"GKR_BENCH_BACKEND": "gloo", "GKR_BENCH_DEVICE": "0", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    line, d = _run("--gpus", "2", "--batch", "32", "--steps", "2", "--warmup", "1", "--proofs", "4", "--layer-k-i", "20", "--layer-k", "10", env=env)
    assert d["n_gpus"] == 2 and d["collective"]["world_size_seen"] == 2 and d["collective"]["backend"] == "gloo"
    assert d["verified"]["ok"] is True and d["verified"]["all_ranks_ok"] is True
    assert d["layer24_split"]["matches_golden_digest"] is True and d["mle_split"]["matches_golden_digest"] is True
    assert d["aggregated_proofs"]["config3"]["verified"]["all_ranks_ok"] is True
    assert "multi_device" not in d["aggregated_proofs"]       # (one process over all devices is the N = 1 line's leg)
    assert line["n_gpus"] == 2 and line["legs"]["layer24_split"]["ok"] is True and line["legs"]["mle_split"]["ok"] is True
    assert line["exit"]["code"] == 0


def test_one_process_over_all_devices_leg():
    """aggregated_proofs.multi_device on the N = 1 line: configs[3] through ONE process and gkr_ctx_create_multi over every
    visible device, the proofs checked against the committed digests.  A failure of the leg fails this test, however many
    devices the box has."""
    line, d = _run("--batch", "16", "--steps", "1", "--warmup", "1", "--proofs", "4", "--no-extras", "--no-cpu-baseline")
    m = d["aggregated_proofs"]["multi_device"]
    assert "error" not in m, m
    assert m["devices_seen"] >= 1 and m["proofs"] == 48 and m["ms"] > 0 and (m["verified"] is None or m["verified"]["ok"] is True)
    assert line["legs"]["multi_device"]["ok"] is not False and "error" not in line["legs"]["multi_device"]


def test_a_failed_split_leg_reaches_the_exit_code():
    """A split leg that raises (here: forced through GKR_BENCH_FAIL_SPLIT=1) leaves the line on stdout with exit.code 2 and the
    process exits non-zero -- bench.py's own launcher hands that code on (such failures used to end in rc 0)."""
    import tempfile
    env = {"GKR_BENCH_BACKEND": "gloo", "GKR_BENCH_DEVICE": "0", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "GKR_BENCH_FAIL_SPLIT": "1",
           "GKR_BENCH_DETAIL": tempfile.NamedTemporaryFile(suffix=".json", delete=False).name}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        os.environ.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--batch", "16", "--steps", "1", "--warmup", "1", "--proofs", "0",
                          "--layer-k-i", "20", "--layer-k", "10", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, **env))
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and out.returncode != 0, (out.returncode, out.stdout[-500:], out.stderr[-2000:])
    line = json.loads(lines[0])
    assert line["exit"]["code"] == 2 and "layer24_split" in line["exit"]["why"] and line["legs"]["layer24_split"]["ok"] is False
    assert line["verified"]["ok"] is True          # the headline itself stood
