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

# where the slowest wavefronts are: per workgroup (blockIdx = bi * 8 + partition), max and mean over its 12 wavefronts
w = a[:, 6].reshape(-1, 12)
n_wg = w.shape[0]
print("per-partition workgroup totals (k cycles): columns = rank bi of the workgroup inside its partition")
for px in range(8):
    mx = w[px::8].max(1) / 1e3
    print(f"  part {px}: max-wave per WG:", " ".join(f"{v:4.0f}" for v in mx[:40]))
it = a[:, 7].reshape(-1, 12)
print("items per wavefront, partition 0, first 6 WGs:", it[0:48:8].astype(int).tolist())
print("cycles per wavefront, partition 0, first 6 WGs (k):", (w[0:48:8] / 1e3).round(0).astype(int).tolist())

# bwd1 (layer-1 backward, both jobs) of the last loss_step
b.loss_step(params, None, logits, loss, grads)
torch.cuda.synchronize()
assert ctypes.CDLL(_lib.LIB_PATH).mllp_timing_read_stamps1(buf, 3072 * 8) == 0
a = np.array(buf, dtype=np.float64).reshape(3072, 8)
items = a[:, 7].sum()
names = ["between items", "row data, g tile", "sweep", "merges, scalar sums", "statistics on the MFMA"]
print(f"bwd1 (both jobs): {int(items)} items, {a[:, 6].mean():.0f} cycles per wavefront, {a[:, 6].max():.0f} max; per item:")
for k, nm in enumerate(names):
    print(f"   {nm:34s} {a[:, k].sum() / items:8.0f} cycles")
print(f"   {'sum':34s} {a[:, :5].sum() / items:8.0f} cycles;  items per wavefront mean {a[:,7].mean():.2f} max {a[:,7].max():.0f}")
w = a[:, 6].reshape(-1, 12)
for px in range(8):
    print(f"  part {px}: max-wave per WG (k):", " ".join(f"{v:4.0f}" for v in (w[px::8].max(1) / 1e3)[:32]))
