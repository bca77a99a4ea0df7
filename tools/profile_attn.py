#!/usr/bin/env python3
"""Workload for rocprofv3: the three 16-channel attention sweeps of the synthetic training step alone (one conv forward +
backward on the LDS-tiled copies: fwd16_tiled_kernel, bwdsrc16_tiled_kernel, bwddst16_lane_kernel), dst = constraints.
usage: python3 tools/profile_attn.py [instances] [reps]"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from mllp_amd.graph import synthetic_batch
from mllp_amd.model import GNNModel, set_seed

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
b = synthetic_batch(n_inst)
set_seed(42)
params = GNNModel().flat_parameters().detach().float().cuda()
dst_is_var, off = False, 1392
nd, ns = b.M, b.N
cp = params[off:off + 1104].contiguous()
xs = torch.randn(ns, 16, device="cuda"); xd = torch.randn(nd, 16, device="cuda")
ws = b.tconv_workspace(dst_is_var, 16)
assert b.enable_tiled(dst_is_var, variant=1) and b.enable_tiled(not dst_is_var, variant=2) and b.enable_tiled(dst_is_var, variant=4)
dh = torch.randn(nd, 16, device="cuda")
for _ in range(reps):
    h = b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
    b.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh)
torch.cuda.synchronize()
print("done", b.dims())
