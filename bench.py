#!/usr/bin/env python3
"""bench.py -- LP instances/s (forward + backward + Adam) on the Netlib batch, and achieved HBM GB/s
of the CSR SpMM on the synthetic roofline batch (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

One JSON line on rank 0:
  value        whole-job LP instances/s on the full Netlib batch (BASELINE.json configs[2], 97 instances as
               block-diagonal batches, fp32), inputs resident in HBM, K steps timed between barrier +
               synchronize pairs, max over ranks.  `--scaling strong` (default): the 97 instances are SHARDED over
               the ranks by nnz (trainer.shard_instances, as BASELINE.json's north_star words it), one Adam step per
               global batch, gradients all-reduced over RCCL each step.  `--scaling weak`: every rank owns all 97.
               With N > 1 the other mode is measured too and reported under "netlib_weak" / "netlib_strong".
  roofline     plain CSR SpMM  Y = A H  on the synthetic batch (configs[3]: m=10k, n=20k, ~1% dense,
               256 instances per GPU): algorithmic bytes / average launch duration, HIP events on the
               launch stream, against the 8 TB/s HBM3E peak.  `kernels` lists the other sweeps likewise.
  synthetic    LP instances/s of the full training step on that synthetic batch (configs[3]; with N
               ranks 256 instances per GPU, weak scaling).
  synthetic_strong  configs[4] as strong scaling: a FIXED global batch of 2048 instances per Adam step, as 8 / N
               micro-batches of 256 per rank whose gradients are accumulated before the all-reduce.
  cpu_baseline the oracle's literal restatement of the reference loop (per-instance graph rebuild,
               forward, BCE, autograd backward, Adam; reference linear_program_experiment.py:120-144)
               timed on this host for one epoch of the same 97 instances (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

# Large batches run their step on two HIP streams (the latency-regime step is one stream); with RCCL's own streams in
# the process the default of 4 hardware queues made the pair share a queue (measured 1.17 vs 0.98 ms/step in round 1).
# The HIP runtime reads this when it loads, i.e. at `import torch`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np   # noqa: E402
import torch         # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--synthetic-instances", type=int, default=256, help="instances per GPU of the roofline batch")
    ap.add_argument("--synthetic-steps", type=int, default=10)
    ap.add_argument("--spmm-reps", type=int, default=30)
    ap.add_argument("--no-synthetic", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="what `value` measures with N > 1 ranks: the 97 Netlib instances sharded over the ranks "
                         "(strong) or replicated per rank (weak)")
    ap.add_argument("--cpu-threads", type=int, default=8, help="torch threads of the cpu_baseline leg")
    ap.add_argument("--no-hip-graph", action="store_true")
    ap.add_argument("--hip-graph", default="auto", help="auto (capture only launch-bound batches) | True")
    ap.add_argument("--dry-launch", action="store_true",
                    help="with --gpus N > 1 and no WORLD_SIZE: print the launch command as JSON and exit (no device touched)")
    ap.add_argument("--master-port", type=int, default=29571)
    return ap.parse_args(argv)


def launch_command(args, argv):
    """`python bench.py --gpus N` without a launcher around it: the command that starts the N ranks (one process per
    GPU, RCCL over xGMI; the parent never touches a device -- a process that has initialised the GPU must not be
    replaced or forked on this pool).  `argv` is passed through unchanged."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(args.master_port),
            os.path.abspath(__file__)] + [a for a in argv if a != "--dry-launch"]


def maybe_launch(args, argv):
    """Returns an exit code if this process was only the launcher, None if it is a rank (or the single process)."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            print(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={ws}", file=sys.stderr)
            return 2
        return None
    if args.gpus <= 1:
        return None
    cmd = launch_command(args, argv)
    if args.dry_launch:
        print(json.dumps({"launch": cmd, "n_ranks": args.gpus}))
        return 0
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    return subprocess.call(cmd, env=env)


def timed(fn, reps, warm=2):
    """average ms per call of fn(), HIP events on the current (launch) stream"""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def barrier_sync(dist_on):
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, dist_on):
    if not dist_on:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def spmm_bytes(nnz, rows, cols, C=16):
    # nnz (4 B index + 4 B value) + row pointers + source rows once + output rows once
    return nnz * 8 + 4 * (rows + 1) + cols * C * 4 + rows * C * 4


def load_traffic(kernel_key):
    """HBM bytes per launch from the committed PMC pass (profiles/hbm_traffic.json), or None.  Counters need
    rocprofv3 around the process, so this figure is NOT measured by this run: the JSON line says where it is from."""
    p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(p) as fh:
            return json.load(fh).get(kernel_key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def cpu_baseline(instances, threads):
    from oracle import pyg_restatement as o1   # the ONLY use of the oracle in this file: the CPU leg
    torch.set_num_threads(max(1, min(int(threads), os.cpu_count() or 1)))
    sd = o1.init_state(42, torch.float32)
    tr = o1.ReferenceTrainer(sd, lr=1e-3, dtype=torch.float32, rebuild_graph=True)
    small = sorted(instances, key=lambda i: i.nnz)[:3]
    for i in small:            # warm-up (allocator, thread pool)
        tr.step(i)
    # whole epochs until about 12 s of CPU work have been timed (the contract asks for a 10-30 s sample)
    t0 = time.perf_counter()
    n_done = epochs = 0
    while True:
        for i in instances:
            tr.step(i)
        n_done += len(instances)
        epochs += 1
        dt = time.perf_counter() - t0
        if dt >= 12.0 or epochs >= 64:
            break
    # the same loop with the per-instance graphs cached (a fairer lower bound than the reference's per-step rebuild,
    # SURVEY.md section 8d); the first pass fills the cache and is not timed
    trc = o1.ReferenceTrainer(sd, lr=1e-3, dtype=torch.float32, rebuild_graph=False)
    for i in instances:
        trc._cache[i.name] = o1.instance_graph(i, torch.float32)
    t1 = time.perf_counter()
    n_c = 0
    while True:
        for i in instances:
            trc.step(i)
        n_c += len(instances)
        dtc = time.perf_counter() - t1
        if dtc >= 6.0 or n_c >= 64 * len(instances):
            break
    return dict(value=n_done / dt, unit="instances/s", cores=int(torch.get_num_threads()),
                host_cpus=os.cpu_count(), kind="port", seconds=dt,
                graph_cached={"value": n_c / dtc, "unit": "instances/s", "seconds": dtc,
                              "sample": f"{n_c // len(instances)} epochs of the same instances, graphs prebuilt"},
                sample=f"{epochs} epochs of the same {len(instances)} Netlib instances ({dt:.1f} s), one Adam step per "
                       "instance, graph rebuilt per step (reference experiment.py:123-144), fp32 torch CPU; "
                       "PyG itself is not installable here, so this is the oracle's restatement")


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    rc = maybe_launch(args, argv)          # before anything touches a device
    if rc is not None:
        raise SystemExit(rc)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or os.environ.get("MLLP_BENCH_FORCE_DIST") == "1"   # force: exercise the RCCL path with 1 rank
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from mllp_amd import _lib
    _lib.lib()   # fail loudly if the HIP library is missing
    from mllp_amd.data import load_packed
    from mllp_amd.graph import LPBatch, synthetic_batch
    from mllp_amd.trainer import LPTrainer
    from mllp_amd.model import GNNModel, set_seed
    from mllp_amd.trainer import shard_instances

    set_seed(42)                                    # reference linear_program_experiment.py:19
    params0 = GNNModel().flat_parameters().detach().float().cuda()   # random init of the reference architecture
    instances = load_packed()
    n_inst = len(instances)
    use_graph = False if args.no_hip_graph else args.hip_graph

    def run_netlib(mode):
        """K timed steps on the Netlib batch.  strong: this rank's nnz-balanced shard of the 97 instances, global
        batch = 97; weak: all 97 on every rank, global batch = 97 * world.  One Adam step per global batch."""
        if mode == "strong" and world > 1:
            mine = shard_instances([i.nnz for i in instances], world)[rank]
            total = n_inst
        else:
            mine = list(range(n_inst))
            total = n_inst * world
        batch = LPBatch.from_instances([instances[i] for i in mine]) if mine else None
        trainer = LPTrainer(params0, lr=1e-3, use_hip_graph=use_graph, global_instances=total)

        def one():
            trainer.step(batch) if batch is not None else trainer.step_empty()
        for _ in range(max(args.warmup, 2)):          # >= 2: eager pass + graph capture
            one()
        barrier_sync(dist_on)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one()
        barrier_sync(dist_on)
        dt = max_over_ranks(time.perf_counter() - t0, dist_on)
        return dict(value=total * args.steps / dt, ms_per_step=1e3 * dt / args.steps, scaling=mode,
                    global_batch=total, instances_this_rank=len(mine), nnz_this_rank=batch.nnz if batch else 0,
                    hip_graph=trainer.uses_graph(batch) if batch else False,
                    final_loss_rank0=float(trainer.last_loss(batch)[0]) if batch else None)

    # ---------------- headline: Netlib batch, instances/s --------------------------------------
    head = run_netlib(args.scaling)
    value = head["value"]
    out = {
        "metric": "LP instances/sec (fwd+bwd) on Netlib batch; achieved HBM GB/s on CSR SpMM",
        "value": value, "unit": "instances/s", "n_gpus": world,
        "rccl_ranks": (__import__("torch.distributed").distributed.get_world_size() if dist_on else 1),
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": head["scaling"], "vs_baseline": None,
        "dtype": "f32", "data": "Netlib LP (97 normalized instances packed in data/netlib_norm.npz); "
                                "synthetic sparse LPs for the roofline section",
        "config": {"workload": "netlib_full: BASELINE.json configs[2], 97 instances / 1,074,147 nnz per global batch, "
                               + ("sharded over the ranks by nnz" if head["scaling"] == "strong" and world > 1
                                  else "one block-diagonal batch per GPU")
                               + ", one Adam step per global batch: fwd + BCE + bwd + gradient all-reduce + Adam, fp32",
                   "global_batch": head["global_batch"], "instances_rank0": head["instances_this_rank"],
                   "nnz_rank0": head["nnz_this_rank"], "hip_graph": head["hip_graph"],
                   "parallelism": f"dp{world}", "final_loss_rank0": head["final_loss_rank0"]},
    }
    if world > 1:      # the other scaling mode beside the headline
        other = "weak" if args.scaling == "strong" else "strong"
        out["netlib_" + other] = run_netlib(other)

    # reference semantics beside the batched headline: ONE Adam step per instance (experiment.py:123-144),
    # every instance its own HBM-resident batch; like-for-like with cpu_baseline (rank 0, single-GPU runs)
    if world == 1 and not args.no_cpu_baseline:
        singles = [LPBatch.from_instances([i]) for i in instances]
        tr1 = LPTrainer(params0, lr=1e-3, use_hip_graph=use_graph, global_instances=1)
        for b in singles:
            tr1.step(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            for b in singles:
                tr1.step(b)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t0
        out["per_instance_steps"] = {"value": 3 * n_inst / dt1, "unit": "instances/s",
                                     "what": "batch_size 1: one Adam step per instance, 3 epochs of the 97 instances"}
        del tr1, singles

    # ---------------- roofline + synthetic throughput ------------------------------------------
    if not args.no_synthetic:
        t_gen = time.perf_counter()
        sb = synthetic_batch(args.synthetic_instances, seed=1234 + 100000 * rank)
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t_gen
        Hn = torch.randn(sb.N, 16, device="cuda")
        Hm = torch.randn(sb.M, 16, device="cuda")
        Ym = torch.empty(sb.M, 16, device="cuda")
        Yn = torch.empty(sb.N, 16, device="cuda")
        b_a, b_at = spmm_bytes(sb.nnz, sb.M, sb.N), spmm_bytes(sb.nnz, sb.N, sb.M)
        # generic sweep (gathers from L2) first, then the streamed copy the library uses for large batches (built by HIP
        # kernels, library-owned: mllp_graph_build_spmm_copy), then round 2's LDS-tiled kernel for comparison
        ms_ga = timed(lambda: sb.spmm(Hn, out=Ym), args.spmm_reps)
        ms_gat = timed(lambda: sb.spmm(Hm, transpose=True, out=Yn), args.spmm_reps)
        ref_a, ref_at = Ym.clone(), Yn.clone()
        info_a, info_at = sb.build_spmm_copy(False), sb.build_spmm_copy(True)
        ms_a = timed(lambda: sb.spmm(Hn, out=Ym), args.spmm_reps, warm=5)
        ms_at = timed(lambda: sb.spmm(Hm, transpose=True, out=Yn), args.spmm_reps, warm=5)
        err_a = float((Ym - ref_a).abs().max() / ref_a.abs().max())
        err_at = float((Yn - ref_at).abs().max() / ref_at.abs().max())
        gbs_a, gbs_at = b_a / ms_a / 1e6, b_at / ms_at / 1e6
        sb.drop_spmm_copy(False)
        sb.drop_spmm_copy(True)
        # round 2's kernel (LDS-tiled copy, variant 0) and its opt-in bf16 feature image (NOT the
        # parity path; printed beside the fp32 lines, never as them)
        t_v0 = time.perf_counter()
        tiled_a = sb.enable_tiled(False)
        torch.cuda.synchronize()
        t_v0 = time.perf_counter() - t_v0
        r02_line = bf16_line = None
        if tiled_a:
            ms_r02 = timed(lambda: sb.spmm(Hn, out=Ym), args.spmm_reps, warm=5)
            r02_line = {"kernel": "spmm_tiled_ws_kernel A*H (round 2: 512 x 1024 LDS-tiled copy, entries staged in LDS)",
                        "ms": ms_r02, "alg_bytes": b_a, "GBps": b_a / ms_r02 / 1e6, "frac": b_a / ms_r02 / 1e6 / HBM_PEAK_GBS,
                        "copy_build_s": t_v0}
            Hb = Hn.to(torch.bfloat16).contiguous()
            ms_b = timed(lambda: sb.spmm_bf16(Hb, out=Ym), args.spmm_reps, warm=5)
            b_b = b_a - sb.N * 32                                   # source rows are 32 bytes instead of 64
            bf16_line = {"kernel": "spmm_tiled A*H, bf16 feature image (opt-in, fp32 accumulate)", "ms": ms_b,
                         "alg_bytes": b_b, "GBps": b_b / ms_b / 1e6, "frac": b_b / ms_b / 1e6 / HBM_PEAK_GBS,
                         "max_rel_diff_vs_fp32": float((Ym - ref_a).abs().max() / ref_a.abs().max())}
            del Hb
            sb.disable_tiled(False)
        kernels = [
            {"kernel": "spmm_stream_kernel A*H", "ms": ms_a, "alg_bytes": b_a, "GBps": gbs_a, "frac": gbs_a / HBM_PEAK_GBS,
             "max_rel_diff_vs_generic": err_a, "copy_bytes": info_a["bytes"], "entry_slots_per_nnz": info_a["entry_slots"] / sb.nnz,
             "copy_build_s": info_a["build_us"] / 1e6},
            {"kernel": "spmm_stream_kernel At*H", "ms": ms_at, "alg_bytes": b_at, "GBps": gbs_at, "frac": gbs_at / HBM_PEAK_GBS,
             "max_rel_diff_vs_generic": err_at, "copy_bytes": info_at["bytes"],
             "entry_slots_per_nnz": info_at["entry_slots"] / sb.nnz, "copy_build_s": info_at["build_us"] / 1e6},
        ] + ([r02_line] if r02_line else []) + ([bf16_line] if bf16_line else []) + [
            {"kernel": "sweep_kernel<SpmmOp> A*H (generic, L2 gathers)", "ms": ms_ga, "alg_bytes": b_a,
             "GBps": b_a / ms_ga / 1e6, "frac": b_a / ms_ga / 1e6 / HBM_PEAK_GBS},
            {"kernel": "sweep_kernel<SpmmOp> At*H (generic, L2 gathers)", "ms": ms_gat, "alg_bytes": b_at,
             "GBps": b_at / ms_gat / 1e6, "frac": b_at / ms_gat / 1e6 / HBM_PEAK_GBS},
        ]
        del Ym, Yn, ref_a, ref_at
        # the full-size parity figures are gates, not prints: a regression ends the run non-zero
        gate_fail = []
        if os.environ.get("MLLP_BENCH_PERTURB"):      # (tests: a deliberately perturbed comparison must end the run non-zero)
            err_a += float(os.environ["MLLP_BENCH_PERTURB"])
        for name_, err_, lim_ in (("spmm_stream A*H", err_a, 2e-6), ("spmm_stream At*H", err_at, 2e-6)):
            if not err_ <= lim_:
                gate_fail.append(f"{name_}: max_rel_diff_vs_generic {err_:.3e} > {lim_:.0e}")
        sb.tiled_build_s = 0.0                       # from here on: the copies of the training step
        lg_generic = sb.loss_step(params0, 1.0 / sb.n_inst)[1].clone()      # whole model on the generic sweeps, full size
        # One 16-channel attention conv, forward and backward (all its kernels, as the step runs them), in BOTH orientations
        # (dst = constraints walks A, dst = variables walks A^T: twice the rows, half the row length): generic sweeps, LDS-tiled
        # sweeps (rounds 2-3), streamed sweeps (round 4, stream_attn.hip).  Bytes: forward = pattern + source rows once + per
        # destination node_qp (x read, q' and t written) and the sweep (q', t, x read; h, Z, aux written) = 408 B; backward =
        # two pattern sweeps + node tensors.
        conv_diffs = {}
        for dst_is_var, off_, label in ((False, 1392, "dst=constraints"), (True, 288, "dst=variables")):
            nd_, ns_ = (sb.N, sb.M) if dst_is_var else (sb.M, sb.N)
            x_s, x_d = (Hm, Hn) if dst_is_var else (Hn, Hm)
            cp = params0[off_:off_ + 1104].contiguous()
            ws = sb.tconv_workspace(dst_is_var, 16)
            dh = torch.randn(nd_, 16, device="cuda")
            h = [None]

            def conv_f():
                h[0] = sb.tconv_fwd(dst_is_var, 16, cp, x_s, x_d, ws)

            def conv_b():
                return sb.tconv_bwd(dst_is_var, 16, cp, x_s, x_d, h_ref, ws, dh.clone())
            b_f = sb.nnz * 8 + 4 * (nd_ + 1) + ns_ * 64 + nd_ * 408
            b_b = 16 * sb.nnz + 192 * ns_ + 1216 * nd_

            def line(kind, what, ms_, by_):
                return {"kernel": f"{kind} {what}, {label}", "ms": ms_, "alg_bytes": by_, "GBps": by_ / ms_ / 1e6,
                        "frac": by_ / ms_ / 1e6 / HBM_PEAK_GBS}
            ms_fg = timed(conv_f, 6, warm=2)
            h_ref = h[0].clone()                      # (every backward below masks with THIS forward's ReLU pattern)
            ms_bg = timed(conv_b, 4, warm=1)
            ref_b = [t_.clone() for t_ in conv_b()[:3]]
            kernels.append(line("tconv_fwd16", "generic (prep + node_qp + attention sweep from L2)", ms_fg, b_f))
            kernels.append(line("tconv_bwd16", "generic (bwd_pre + dst sweep + src gather sweep + stats)", ms_bg, b_b))
            for kind in ("LDS-tiled", "streamed"):
                if kind == "LDS-tiled":
                    ok = (sb.enable_tiled(dst_is_var, variant=1) and sb.enable_tiled(not dst_is_var, variant=2)
                          and sb.enable_tiled(dst_is_var, variant=4))
                    names = ("fwd16_tiled_kernel", "bwdsrc16_tiled_kernel + bwddst16_lane_kernel")
                else:
                    infos_ = {g_: sb.build_stream_copy(dst_is_var if g_ != 2 else not dst_is_var, g_) for g_ in (1, 2, 3)}
                    ok = all(i_["n_tiles"] > 0 for i_ in infos_.values())
                    names = ("fwd16_stream_kernel", "bwdsrc16_stream_kernel + bwddst16_stream_kernel")
                if not ok:
                    continue
                ms_f = timed(conv_f, 10, warm=3)
                d_h = float((h[0] - h_ref).abs().max() / h_ref.abs().max())
                ms_b = timed(conv_b, 6, warm=2)
                got_b = conv_b()[:3]
                d_b = [float((g_ - r_).abs().max() / r_.abs().max()) for g_, r_ in zip(got_b, ref_b)]
                lf = line("tconv_fwd16", f"{kind} (prep + node_qp + {names[0]})", ms_f, b_f)
                lb = line("tconv_bwd16", f"{kind} (bwd_pre + {names[1]} + stats)", ms_b, b_b)
                lf["max_rel_diff_vs_generic"] = d_h
                lb["max_rel_diff_vs_generic"] = {"param_grads": d_b[0], "dx_dst": d_b[1], "dx_src": d_b[2]}
                if kind == "streamed":
                    lf["copy_build_s"] = infos_[1]["build_us"] / 1e6
                    lb["copy_build_s"] = (infos_[2]["build_us"] + infos_[3]["build_us"]) / 1e6
                    lf["entry_slots_per_nnz"] = infos_[1]["entry_slots"] / sb.nnz
                    conv_diffs[label] = (d_h, max(d_b))
                    if not d_h <= 2e-6:
                        gate_fail.append(f"tconv_fwd16 streamed {label}: {d_h:.3e} > 2e-6")
                    if not max(d_b) <= 5e-6:
                        gate_fail.append(f"tconv_bwd16 streamed {label}: {max(d_b):.3e} > 5e-6")
                kernels.append(lf)
                kernels.append(lb)
                if kind == "LDS-tiled":       # (the streamed copies take precedence anyway; drop these: 34 GB at 256 instances)
                    sb.disable_tiled(dst_is_var, variant=1); sb.disable_tiled(not dst_is_var, variant=2)
                    sb.disable_tiled(dst_is_var, variant=4)
                else:
                    for g_ in (1, 2, 3):
                        sb.drop_stream_copy(dst_is_var if g_ != 2 else not dst_is_var, g_)
            del ws, dh, h_ref, ref_b
        del Hm
        # layer-1 (one channel) conv, both orientations: generic sweeps, the LDS-tiled lane-per-row kernel (variant 3), the
        # lane-per-row STREAMED copy (geometry 4, lane_stream.hip: the one the training step uses)
        for dst_is_var, off1 in ((False, 144), (True, 0)):
            label = "dst=variables" if dst_is_var else "dst=constraints"
            nd, ns = (sb.N, sb.M) if dst_is_var else (sb.M, sb.N)
            cp1 = params0[off1:off1 + 144].contiguous()
            x1s, x1d = torch.randn(ns, device="cuda"), torch.randn(nd, device="cuda")
            dh1 = torch.randn(nd, 16, device="cuda")
            ws1 = sb.tconv_workspace(dst_is_var, 1)
            b_f1 = sb.nnz * 8 + 4 * (nd + 1) + ns * 4 + nd * (4 + 64 + 4 + 16)       # CSR bytes: what the reference's layout costs
            b_b1 = sb.nnz * 8 + 4 * (nd + 1) + ns * 4 + nd * (32 + 12 + 64 + 64)
            h1_ref = pg1_ref = None
            keep1 = torch.ones(144, dtype=torch.bool, device="cuda")
            keep1[16:32] = False                                                      # lin_key.bias cancels in the softmax
            for kind in ("generic", "LDS-tiled", "streamed"):
                if kind == "LDS-tiled" and not sb.enable_tiled(dst_is_var, variant=3):
                    continue
                if kind == "streamed":
                    sb.build_stream_copy(dst_is_var, 4)
                ms_f1 = timed(lambda: sb.tconv_fwd(dst_is_var, 1, cp1, x1s, x1d, ws1), 10, warm=3)
                h1 = sb.tconv_fwd(dst_is_var, 1, cp1, x1s, x1d, ws1)
                hh = h1 if h1_ref is None else h1_ref          # every backward is masked with the generic forward's output
                ms_b1 = timed(lambda: sb.tconv_bwd(dst_is_var, 1, cp1, x1s, x1d, hh, ws1, dh1), 10, warm=3)
                pg1 = sb.tconv_bwd(dst_is_var, 1, cp1, x1s, x1d, hh, ws1, dh1)[0]
                how = {"generic": "lane per nonzero, 4-byte gathers from L2",
                       "LDS-tiled": "scalar_tiled_kernel, lane per row, 1024-column blocks",
                       "streamed": "lane1_kernel, lane per row, entries HBM -> registers at 6 B / nonzero, the instance's x in LDS"}[kind]
                lf = {"kernel": f"tconv_fwd1 {kind} ({how}), {label}", "ms": ms_f1, "alg_bytes": b_f1,
                      "GBps": b_f1 / ms_f1 / 1e6, "frac": b_f1 / ms_f1 / 1e6 / HBM_PEAK_GBS}
                lb = {"kernel": f"tconv_bwd1 {kind} (bwd_pre + destination-major sweep + parameter statistics + finalize), {label}",
                      "ms": ms_b1, "alg_bytes": b_b1, "GBps": b_b1 / ms_b1 / 1e6, "frac": b_b1 / ms_b1 / 1e6 / HBM_PEAK_GBS}
                if h1_ref is None:
                    h1_ref, pg1_ref = h1.clone(), pg1.clone()
                else:
                    d_h = float((h1 - h1_ref).abs().max() / h1_ref.abs().max())
                    d_g = float((pg1[keep1] - pg1_ref[keep1]).abs().max() / pg1_ref[keep1].abs().max())
                    lf["max_rel_diff_vs_generic"] = d_h
                    lb["max_rel_diff_vs_generic"] = d_g
                    if kind == "streamed" and not d_h <= 2e-6:
                        gate_fail.append(f"tconv_fwd1 streamed {label}: {d_h:.3e} > 2e-6")
                    if kind == "streamed" and not d_g <= 5e-6:
                        gate_fail.append(f"tconv_bwd1 streamed {label}: {d_g:.3e} > 5e-6")
                kernels += [lf, lb]
                if kind == "LDS-tiled":
                    sb.disable_tiled(dst_is_var, variant=3)
                if kind == "streamed":
                    sb.drop_stream_copy(dst_is_var, 4)
            del ws1, x1s, x1d, dh1, h1_ref, pg1_ref
        out["roofline"] = {"bound": "hbm", "kernel": "spmm_stream_kernel (plain CSR SpMM, Y = A*H, C=16, fp32, streamed copy: "
                                                     "entries HBM -> registers, H double-buffered in LDS by LDS-DMA)",
                           "workload": f"synthetic BASELINE.json configs[3]: {sb.n_inst} x (m=10000, n=20000), "
                                       f"nnz={sb.nnz}; geometric-gap generator (graph.py::synthetic_batch), seed 1234 + "
                                       "first instance of each 16-instance chunk", "achieved": gbs_a, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s",
                           "frac": gbs_a / HBM_PEAK_GBS, "alg_bytes_per_launch": b_a, "ms_per_launch": ms_a,
                           "traffic": load_traffic("spmm_stream"),
                           "traffic_source": "profiles/hbm_traffic.json (separate rocprofv3 --pmc passes, gfx950-"
                                             "corrected; precomputed, not measured by this run)",
                           "kernels": kernels}
        del Hn
        # full training step on the synthetic batch with the copies LPTrainer attaches by itself (streamed copies of the
        # 16-channel sweeps, lane-per-row copies for layer 1); its logits against those of the generic sweeps at the same full
        # size (taken above, before any copy was attached)
        sb.tiled_build_s = 0.0
        sb.stream_build_s = 0.0
        tr = LPTrainer(params0, lr=1e-3, use_hip_graph=False, global_instances=sb.n_inst * world)
        tr._plan(sb)
        lg_tiled = sb.loss_step(params0, 1.0 / sb.n_inst)[1]
        logits_diff = float((lg_tiled - lg_generic).abs().max() / lg_generic.abs().max())
        if not logits_diff <= 5e-6:
            gate_fail.append(f"training step logits, streamed vs generic: {logits_diff:.3e} > 5e-6")
        del lg_generic, lg_tiled
        for _ in range(2):
            tr.step(sb)
        barrier_sync(dist_on)
        t0 = time.perf_counter()
        for _ in range(args.synthetic_steps):
            tr.step(sb)
        barrier_sync(dist_on)
        dts = max_over_ranks(time.perf_counter() - t0, dist_on)
        out["synthetic"] = {"workload": "configs[3] per GPU (configs[4] at 8 GPUs): fwd + BCE + bwd + Adam",
                            "instances_per_gpu": sb.n_inst, "nnz_per_gpu": sb.nnz, "steps": args.synthetic_steps,
                            "ms_per_step": 1e3 * dts / args.synthetic_steps,
                            "value": sb.n_inst * world * args.synthetic_steps / dts, "unit": "instances/s",
                            "generate_s": t_gen - getattr(sb, "graph_build_s", 0.0),
                            "graph_build_s": getattr(sb, "graph_build_s", None),
                            "graph_build": "CSR -> CSC (one stable device sort) + row tiers, mllp_graph_create_device",
                            "tiled_build_s": getattr(sb, "tiled_build_s", None),
                            "stream_build_s": getattr(sb, "stream_build_s", None),
                            "copies": "8 streamed copies (mllp_graph_build_stream_copy, device builders): geometries 1-3 of the "
                                      "16-channel attention sweeps + the lane-per-row copy (4) of the layer-1 sweeps, x 2 orientations",
                            "logits_max_rel_diff_streamed_vs_generic": logits_diff,
                            "logits_max_rel_diff_tiled_vs_generic": logits_diff,
                            "parity_gates": {"failed": list(gate_fail), "limits": "SpMM / conv forward 2e-6, conv backward and "
                                             "step logits 5e-6 (max-norm relative, streamed vs generic sweeps, full size)"}}
        out["parity_gate_failures"] = list(gate_fail)
        # configs[4] as STRONG scaling: one Adam step per 2048 instances = 8 / N micro-batches of 256 per rank, gradients
        # accumulated, one all-reduce.  The micro-batches reuse the resident synthetic batch (same work per launch;
        # 8 distinct 256-instance batches with their tiled copies do not fit in 288 GB).
        if sb.n_inst == 256 and 8 % world == 0:
            from mllp_amd.trainer import allreduce_sum_
            micro = 8 // world
            acc = torch.zeros_like(tr.params)
            g_ = torch.zeros_like(tr.params)
            lg_, ls_ = torch.empty(sb.N, device="cuda"), torch.empty(1, device="cuda")

            def strong_step():
                acc.zero_()
                for _ in range(micro):
                    sb.loss_step(tr.params, 1.0 / 2048.0, lg_, ls_, g_)
                    acc.add_(g_)
                if dist_on:
                    allreduce_sum_(acc)
                tr.opt.step(acc)
            strong_step()
            barrier_sync(dist_on)
            t0 = time.perf_counter()
            n_strong = max(2, args.synthetic_steps // 4)
            for _ in range(n_strong):
                strong_step()
            barrier_sync(dist_on)
            dtg = max_over_ranks(time.perf_counter() - t0, dist_on)
            out["synthetic_strong"] = {"workload": "configs[4]: global batch 2048 instances per Adam step, sharded "
                                                   f"{256 * micro} per GPU as {micro} micro-batches of 256",
                                       "scaling": "strong", "global_batch": 2048, "steps": n_strong,
                                       "ms_per_step": 1e3 * dtg / n_strong, "value": 2048 * n_strong / dtg,
                                       "unit": "instances/s"}
        del tr, sb
        torch.cuda.empty_cache()

    # ---------------- CPU baseline (rank 0, single-GPU runs only) ------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(instances, args.cpu_threads)
        out["speedup_vs_cpu_baseline"] = {"batched_step": value / out["cpu_baseline"]["value"],
                                          "per_instance_steps (like for like)":
                                              out["per_instance_steps"]["value"] / out["cpu_baseline"]["value"]}
    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if out.get("parity_gate_failures"):
        print("bench.py: parity gate(s) failed: " + "; ".join(out["parity_gate_failures"]), file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
