"""Batched training step of the learned-LP path: forward + BCE + backward in ONE library call,
gradient all-reduce across data-parallel ranks (RCCL over xGMI through torch.distributed), flat Adam.

Replaces the inner loop of reference linear_program_experiment.py:120-157 (graph rebuild, forward,
BCEWithLogitsLoss, autograd backward, Adam step, top-m metrics -- all per instance on the CPU).

Data parallelism: LP instances are independent blocks of the block-diagonal
batch, so a rank owns a subset of instances and builds its own LPBatch; the loss is the sum over
instances of the per-instance mean BCE divided by the GLOBAL instance count, hence summed per-rank
gradients equal the single-GPU batch gradient.  One all-reduce of the flat 4721-float gradient buffer
(18.9 KB, latency bound) per step; weights stay replicated because every rank applies the same Adam.
"""
import os
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .graph import LPBatch, adam_step

NUM_PARAMS = _lib.NUM_PARAMS


def shard_instances(sizes: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy longest-processing-time assignment of instances (weights = nnz) to ranks.
    Returns per-rank lists of instance indices (each sorted ascending); deterministic."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    loads = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda q: (loads[q], q))
        out[r].append(i)
        loads[r] += int(sizes[i])
    return [sorted(v) for v in out]


class FlatAdam:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8) on flat buffers, backend-agnostic:
    `backend='hip'` uses mllp_adam_step, `backend='torch'` is the same arithmetic in torch ops
    (used by the CPU/gloo tests of the data-parallel logic)."""

    def __init__(self, params: torch.Tensor, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, backend="hip"):
        self.params = params
        self.m = torch.zeros_like(params)
        self.v = torch.zeros_like(params)
        self.eps = eps
        self.backend = backend
        self.state = torch.tensor([0.0, lr, betas[0], betas[1]], dtype=torch.float32, device=params.device)

    def step(self, grads, grad_scale=1.0):
        if self.backend == "hip":
            adam_step(self.params, grads, self.m, self.v, self.state, self.eps, grad_scale)
            return
        step = float(self.state[0]) + 1.0
        lr, b1, b2 = float(self.state[1]), float(self.state[2]), float(self.state[3])
        g = grads * grad_scale
        self.m.mul_(b1).add_(g, alpha=1 - b1)
        self.v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = self.v.sqrt() / (bc2 ** 0.5) + self.eps
        self.params.addcdiv_(self.m, denom, value=-lr / bc1)
        self.state[0] = step

    def state_dict(self):
        return dict(m=self.m.clone(), v=self.v.clone(), state=self.state.clone())

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.state.copy_(sd["state"])


def allreduce_sum_(t: torch.Tensor):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (
            dist.get_world_size() > 1 or os.environ.get("MLLP_BENCH_FORCE_DIST") == "1"):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class DataParallelStep:
    """grad function -> all-reduce -> identical Adam on every rank.  `grad_fn(params) -> (loss, grads)`
    must already scale by 1 / global instance count."""

    def __init__(self, params, grad_fn: Callable, lr=1e-3, adam_backend="hip"):
        self.params = params
        self.grad_fn = grad_fn
        self.opt = FlatAdam(params, lr=lr, backend=adam_backend)

    def step(self):
        loss, grads = self.grad_fn(self.params)
        allreduce_sum_(grads)
        allreduce_sum_(loss)
        self.opt.step(grads)
        return loss


class LPTrainer:
    """HIP fast path: one `mllp_gnn_loss_step` + all-reduce + `mllp_adam_step` per batch, launched eagerly on one
    stream (the latency-regime step of the Netlib batch is 13 back-to-back launches), optionally captured in hipGraphs."""

    # "auto": capture batches up to this many nonzeros.  0 = never: the fused step's launches already run back to back,
    # a single-stream capture replays within 1.5 % of eager (0.4755 vs 0.4687 ms), and the
    # per-instance loop is slower replayed (6.0 k vs 7.6 k instances/s).  `use_hip_graph=True` still forces capture.
    GRAPH_NNZ_LIMIT = 0

    # "auto": batches of at least this many nonzeros get the re-blocked copies of the attention sweeps, both orientations
    # (the throughput regime of the library's row tiers starts at the same size): the STREAMED copies for the 16-channel
    # sweeps and the lane-per-row copies for the 1-channel sweeps of layer 1 (round 4, LPBatch.enable_stream_step); with
    # `stream_copies=False` all sweeps run on LDS-tiled copies (round 2-3, LPBatch.enable_tiled_step); smaller batches run
    # the fused latency-regime kernels
    TILED_NNZ_MIN = 32 << 20

    def __init__(self, params_flat: torch.Tensor, lr=1e-3, use_hip_graph="auto",
                 global_instances: Optional[int] = None, with_metrics=False, tiled_copies="auto", stream_copies=True):
        assert params_flat.is_cuda and params_flat.numel() == NUM_PARAMS
        self.params = params_flat.detach().clone().float().contiguous()
        self.opt = FlatAdam(self.params, lr=lr)
        self.use_graph = use_hip_graph
        self.global_instances = global_instances
        self.with_metrics = with_metrics
        self.tiled_copies = tiled_copies
        self.stream_copies = stream_copies
        self._plans = {}
        self._gen = 0                  # bumped at every library-side write of the parameters (LPBatch.train_step)

    def _plan(self, batch: LPBatch):
        key = batch.token          # unique per LPBatch for the life of the process (id() can be reused after a free)
        p = self._plans.get(key)
        if p is None:
            dev = self.params.device
            graph = (batch.nnz <= self.GRAPH_NNZ_LIMIT) if self.use_graph == "auto" else bool(self.use_graph)
            want_tiled = (batch.nnz >= self.TILED_NNZ_MIN) if self.tiled_copies == "auto" else bool(self.tiled_copies)
            if want_tiled and self.stream_copies:
                if not getattr(batch, "_streams", None):
                    batch._streams = batch.enable_stream_step()
            elif want_tiled and not getattr(batch, "_tiled", None):
                batch.enable_tiled_step()
            p = dict(batch=batch, logits=torch.empty(batch.N, device=dev), loss=torch.zeros(1, device=dev),
                     grads=torch.zeros(NUM_PARAMS, device=dev), metrics=torch.zeros(batch.n_inst, 2, device=dev),
                     g_fwd=None, g_opt=None, warm=0, graph=graph)
            self._plans[key] = p
        return p

    def metrics_of(self, batch: LPBatch) -> torch.Tensor:
        """[n_inst, 2] (correct_num, f1) of the last step on `batch` (with_metrics=True); device tensor."""
        return self._plans[batch.token]["metrics"]

    def last_loss(self, batch: LPBatch) -> torch.Tensor:
        return self._plans[batch.token]["loss"]

    def uses_graph(self, batch: LPBatch) -> bool:
        return bool(self._plans[batch.token]["graph"])

    def release(self, batch: LPBatch):
        """Drop the buffers (and the reference to the batch) kept for `batch`."""
        self._plans.pop(batch.token, None)

    def _whole_step(self, p):
        """single rank: forward + loss + backward + Adam in one library call (the tail of the fused path is one launch
        that leaves the folded weights of the NEXT step in the batch's workspace: valid for the next step when it is on
        the same batch and nobody else wrote the parameters -- this trainer owns them)"""
        b = p["batch"]
        inv = 1.0 / float(self.global_instances or b.n_inst)
        b.train_step(self.params, self.opt.m, self.opt.v, self.opt.state, self.opt.eps, inv, p["logits"], p["loss"],
                     p["grads"], param_gen=self._gen)
        self._gen += 1
        if self.with_metrics:
            b.topm_metrics(p["logits"], p["metrics"])

    def _fwd_bwd(self, p):
        b = p["batch"]
        inv = 1.0 / float(self.global_instances or b.n_inst)
        b.loss_step(self.params, inv, p["logits"], p["loss"], p["grads"])
        if self.with_metrics:
            b.topm_metrics(p["logits"], p["metrics"])

    def _opt(self, p):
        self.opt.step(p["grads"])

    def step_empty(self):
        """A rank that owns no instance of the current batch: contribute a zero gradient and apply the same Adam."""
        if not hasattr(self, "_zero"):
            self._zero = torch.zeros(NUM_PARAMS, device=self.params.device)
        self._zero.zero_()
        allreduce_sum_(self._zero)
        self.opt.step(self._zero)
        self._gen += 1

    def step(self, batch: LPBatch):
        """One optimizer step on `batch`; returns (loss, logits) device tensors (valid until the next step)."""
        import torch.distributed as dist
        p = self._plan(batch)
        multi = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size() > 1 or os.environ.get("MLLP_BENCH_FORCE_DIST") == "1")
        if not p["graph"] or p["warm"] < 1:
            if multi:
                self._fwd_bwd(p)
                allreduce_sum_(p["grads"])
                self._opt(p)
                self._gen += 1
            else:
                self._whole_step(p)
            p["warm"] += 1
            return p["loss"], p["logits"]
        if p["g_fwd"] is None:
            torch.cuda.synchronize()
            if multi:
                p["g_fwd"], p["g_opt"] = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(p["g_fwd"]):
                    self._fwd_bwd(p)
                with torch.cuda.graph(p["g_opt"]):
                    self._opt(p)
            else:
                p["g_fwd"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(p["g_fwd"]):
                    self._fwd_bwd(p)
                    self._opt(p)
            # capture does not execute: the captured step runs below
        self._gen += 1
        p["g_fwd"].replay()
        if multi:
            allreduce_sum_(p["grads"])
            p["g_opt"].replay()
        return p["loss"], p["logits"]
