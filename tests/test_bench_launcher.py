"""bench.py --gpus N without an outer launcher starts its own ranks (`--gpus` was parsed and never read).
Here, without N devices, it must refuse -- before importing torch in the parent or touching a GPU -- with a non-zero
exit and no JSON line; the N-rank run itself is tests/test_gpu_bench_contract.py (two ranks over gloo on one GPU)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_refuses_when_the_devices_are_not_there():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GKR_BENCH_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    assert out.stdout.strip() == "" and "--gpus 64 but only" in out.stderr


def test_launcher_builds_the_documented_command(monkeypatch):
    """launch_ranks starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <the same arguments>` as a child and relays exactly one JSON line."""
    sys.path.insert(0, REPO)
    import bench
    seen = {}

    class FakeChild:
        pid = 0
        stdout = [b"noise from a library\n", b'{"metric": "m", "n_gpus": 2}\n', b"{second json is not relayed}\n"]

        def wait(self):
            return 0

    def fake_popen(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        return FakeChild()
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setenv("GKR_BENCH_DEVICE", "0")          # the single-device test hook: no device count
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    r, w = os.pipe()
    monkeypatch.setattr(bench, "_REAL_STDOUT", w)
    assert bench.launch_ranks(2) == 0
    os.close(w)
    relayed = os.read(r, 4096)
    os.close(r)
    assert relayed == b'{"metric": "m", "n_gpus": 2}\n'
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    assert seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_the_stdout_line_of_a_full_record_stays_short():
    """(CPU) compact_line on the full records committed under profiles/ (every leg of the default line, N = 1 and the two-rank
    gloo run): under 6 000 bytes, every contract key, roofline / cpu_baseline / verified / legs with their scalar fields --
    the bound BENCH_r05 broke (a 20 KB line the driver did not parse) cannot be broken again without a GPU noticing."""
    import glob
    import json
    sys.path.insert(0, REPO)
    import bench
    records = sorted(glob.glob(os.path.join(REPO, "profiles", "r06", "z_bench_*detail.json")))
    assert records, "no committed detail records"
    for path in records:
        full = json.load(open(path))
        line = bench.compact_line(full)
        text = json.dumps(line)
        assert len(text.encode()) < bench.LINE_LIMIT, (path, len(text))
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                    "data", "config"):
            assert key in line, (path, key)
        assert line["value"] == float("%.9g" % full["value"]) and "workload" in line["config"]
        if "roofline" in full and "achieved" in full["roofline"]:
            for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
                assert key in line["roofline"], (path, key)
        if os.path.basename(path) == "z_bench_default_detail.json":
            assert set(line["legs"]) >= {"n16", "layer24", "wide20", "wide_prove", "config0", "config3", "multi_device", "large_r1cs"}
            assert line["cpu_baseline"]["kind"] == "port" and line["verified"]["ok"] is True and line["exit"]["code"] == 0
