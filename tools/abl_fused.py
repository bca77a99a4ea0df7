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
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(200):
    b.forward(params, logits)
torch.cuda.synchronize()
print("MLLP_FUSED_ABL", os.environ.get("MLLP_FUSED_ABL", "0"), "ms/forward", 1e3 * (time.perf_counter() - t1) / 200)
print("MLLP_FUSED_ABL", os.environ.get("MLLP_FUSED_ABL", "0"), "ms/step (no Adam)", 1e3 * (t1 - t0) / 200)

import ctypes, numpy as np
L = _lib.lib()
buf = (ctypes.c_ulonglong * (3072 * 8))()
torch.cuda.synchronize()
b.forward(params, logits)          # last fwd16 launch = layer 3 (one job, rows = variables)
torch.cuda.synchronize()
assert ctypes.CDLL(_lib.LIB_PATH).mllp_timing_read_stamps(buf, 3072 * 8) == 0
a = np.array(buf, dtype=np.float64).reshape(3072, 8)
items = a[:, 7].sum()
names = ["between items", "gathers issued + prologue GEMM", "sweep", "prefetch issue", "merge + epilogue GEMM", "stores + head"]
print(f"fwd16 layer 3: {int(items)} items, {a[:, 6].mean():.0f} cycles per wavefront, {a[:, 6].max():.0f} max; per item:")
for k, nm in enumerate(names):
    print(f"   {nm:34s} {a[:, k].sum() / items:8.0f} cycles")
print(f"   {'sum':34s} {a[:, :6].sum() / items:8.0f} cycles;  items per wavefront mean {a[:,7].mean():.2f} max {a[:,7].max():.0f}")
