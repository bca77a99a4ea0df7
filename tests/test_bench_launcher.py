"""CPU: `python bench.py --gpus N` starts its own N ranks -- the parent builds the
torch.distributed.run command before anything touches a device, refuses a WORLD_SIZE that disagrees with --gpus, and
stays a plain single process for --gpus 1."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

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


@pytest.mark.gpu
def test_bench_json_contract_on_the_gpu():
    """`python bench.py` end to end at a reduced size (17 synthetic instances = the throughput regime, 3 timed steps):
    exactly one JSON line with the contract's keys, a roofline object for the streamed SpMM with frac = achieved / peak,
    parity figures inside it, and a cpu_baseline object (one epoch's worth is enough here)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
                        "--synthetic-instances", "17", "--synthetic-steps", "2", "--spmm-reps", "3"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and "spmm_stream_kernel" in rf["kernel"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.05 < rf["frac"] < 1.0
    k0 = rf["kernels"][0]
    assert abs(k0["GBps"] - k0["alg_bytes"] / k0["ms"] / 1e6) < 1e-6 * k0["GBps"] and k0["max_rel_diff_vs_generic"] < 2e-6
    assert d["synthetic"]["logits_max_rel_diff_streamed_vs_generic"] < 5e-6
    assert d["synthetic"]["stream_build_s"] > 0 and d["synthetic"]["parity_gates"]["failed"] == [] and d["parity_gate_failures"] == []
    # both orientations of the attention conv, forward and backward, generic / LDS-tiled / streamed, with their parity figures
    names = [k["kernel"] for k in rf["kernels"]]
    for label in ("dst=constraints", "dst=variables"):
        for what in ("tconv_fwd16 generic", "tconv_bwd16 generic", "tconv_fwd16 LDS-tiled", "tconv_bwd16 LDS-tiled",
                     "tconv_fwd16 streamed", "tconv_bwd16 streamed"):
            assert any(n.startswith(what) and n.endswith(label) for n in names), (what, label)
        for what in ("tconv_fwd1 generic", "tconv_fwd1 LDS-tiled", "tconv_fwd1 streamed", "tconv_bwd1 streamed"):
            assert any(n.startswith(what) and n.endswith(label) for n in names), (what, label)
    for k in rf["kernels"]:
        if k["kernel"].startswith("tconv_fwd1 streamed"):
            assert k["max_rel_diff_vs_generic"] < 2e-6 and 0.05 < k["frac"] < 1.0
        if k["kernel"].startswith("tconv_bwd1 streamed"):
            assert k["max_rel_diff_vs_generic"] < 5e-6
        if k["kernel"].startswith("tconv_fwd16 streamed"):
            assert k["max_rel_diff_vs_generic"] < 2e-6 and 0.05 < k["frac"] < 1.0
        if k["kernel"].startswith("tconv_bwd16 streamed"):
            assert max(k["max_rel_diff_vs_generic"].values()) < 5e-6
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"]
    assert d["value"] > 100 * cb["value"]


@pytest.mark.gpu
def test_bench_exits_nonzero_when_a_parity_gate_fails():
    """the full-size parity figures are gates.  A deliberately perturbed comparison (MLLP_BENCH_PERTURB adds
    to the measured SpMM difference) still prints the JSON line, lists the failed gate and exits 1."""
    env = dict(os.environ, MLLP_BENCH_PERTURB="1e-3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--synthetic-instances", "17", "--synthetic-steps", "1", "--spmm-reps", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 1, (r.returncode, r.stderr[-2000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert any("spmm_stream A*H" in f for f in d["parity_gate_failures"])
    assert "parity gate" in r.stderr
