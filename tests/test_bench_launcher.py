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
