#!/usr/bin/env python3
"""Netlib step time for different row-tier thresholds (rows longer than tier_wave leave the 16-lane group tier,
rows longer than tier_block the wave tier)."""
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
insts = load_packed()
for tw, tb in ((0, 0), (8, 256), (16, 256), (32, 256), (64, 256), (16, 128), (32, 128), (32, 512), (128, 512)):
    b = LPBatch.from_instances(insts, tier_wave=tw, tier_block=tb)
    tr = LPTrainer(params, use_hip_graph=False)
    for _ in range(10):
        tr.step(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        tr.step(b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    d = b.dims()
    print(f"tier_wave={tw:4d} tier_block={tb:4d}  {dt * 1e3:.4f} ms/step   A: group {d['A_group']} wave {d['A_wave']} chunk {d['A_block']} | At: {d['At_group']} {d['At_wave']} {d['At_block']}", flush=True)
