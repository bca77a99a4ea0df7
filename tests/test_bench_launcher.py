"""CPU: `python bench.py --gpus N` starts its own N ranks -- the parent builds the
torch.distributed.run command before anything touches a device, refuses a WORLD_SIZE that disagrees with --gpus, and
stays a plain single process for --gpus 1."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_dry_launch_prints_the_torchrun_command_and_touches_no_device():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""           # a device call would fail loudly
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "2",
                          "--dry-launch"], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])
    assert cmd["n_ranks"] == 4
    launch = cmd["launch"]
    assert launch[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in launch and "--nnodes=1" in launch
    assert launch[launch.index("--master-addr") + 1] == "127.0.0.1"
    tail = launch[launch.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "7", "--warmup", "2"]       # the ranks get the same arguments, no --dry-launch


def test_launcher_decisions(monkeypatch):
    b = _bench()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    a1 = b.parse(["--gpus", "1"])
    assert b.maybe_launch(a1, ["--gpus", "1"]) is None                     # single process: unchanged behaviour
    monkeypatch.setenv("WORLD_SIZE", "2")
    a2 = b.parse(["--gpus", "2"])
    assert b.maybe_launch(a2, ["--gpus", "2"]) is None                     # already a rank of a 2-rank job
    a8 = b.parse(["--gpus", "8"])
    assert b.maybe_launch(a8, ["--gpus", "8"]) == 2                        # launched with the wrong world size: refuse
