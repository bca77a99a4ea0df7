#!/usr/bin/env python3
"""Step time of LPTrainer on single-instance batches: eager launches vs hipGraph replay."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from mllp_amd.data import load_packed
from mllp_amd.graph import LPBatch
from mllp_amd.trainer import LPTrainer
from mllp_amd.model import GNNModel, set_seed

params = (set_seed(42), GNNModel().flat_parameters().detach().float().cuda())[1]
insts = sorted(load_packed(), key=lambda i: i.nnz)
for inst in (insts[0], insts[len(insts) // 2], insts[-1]):
    b = LPBatch.from_instances([inst])
    for mode in (False, True):
        tr = LPTrainer(params, use_hip_graph=mode)
        for _ in range(5):
            tr.step(b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300):
            tr.step(b)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
        print(f"{inst.name:12s} nnz={inst.nnz:7d} {'graph' if mode else 'eager'} {dt * 1e6:8.1f} us/step", flush=True)
