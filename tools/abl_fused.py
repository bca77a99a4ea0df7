#!/usr/bin/env python3
"""Timing-only ablations of fused_bwd16_kernel (needs `make -C mllp_amd/csrc timing`; results are WRONG by design).
usage: MLLP_FUSED_ABL=<mask> python3 tools/abl_fused.py   -> ms per Netlib training step"""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from mllp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mllp_amd", "csrc", "libmllp_hip_timing.so")
from mllp_amd.data import load_packed
from mllp_amd.graph import LPBatch
from mllp_amd.model import GNNModel, set_seed
set_seed(42)
params = GNNModel().flat_parameters().detach().float().cuda()
b = LPBatch.from_instances(load_packed())
logits = torch.empty(b.N, device="cuda"); loss = torch.empty(1, device="cuda"); grads = torch.empty(4721, device="cuda")
for _ in range(5):
    b.loss_step(params, None, logits, loss, grads)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    b.loss_step(params, None, logits, loss, grads)
torch.cuda.synchronize()
print("MLLP_FUSED_ABL", os.environ.get("MLLP_FUSED_ABL", "0"), "ms/step (no Adam)", 1e3 * (time.perf_counter() - t0) / 200)
