#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc: a few launches of each 16-wide sweep on the synthetic batch.
usage: python3 tools/profile_spmm.py [instances] [reps] [which=spmm|conv|all]"""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from mllp_amd.graph import synthetic_batch  # noqa: E402
from mllp_amd.model import GNNModel, set_seed  # noqa: E402

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
which = sys.argv[3] if len(sys.argv) > 3 else "spmm"
b = synthetic_batch(n_inst)
H = torch.randn(b.N, 16, device="cuda")
Ht = torch.randn(b.M, 16, device="cuda")
Y = torch.empty(b.M, 16, device="cuda")
Yt = torch.empty(b.N, 16, device="cuda")
for _ in range(reps):
    b.spmm(H, out=Y)
    b.spmm(Ht, transpose=True, out=Yt)
if which in ("conv", "all"):
    params = (set_seed(42), GNNModel().flat_parameters().detach().float().cuda())[1]
    cp = params[1392:1392 + 1104].contiguous()
    ws = b.tconv_workspace(False, 16)
    for _ in range(reps):
        h = b.tconv_fwd(False, 16, cp, H, Ht, ws)
        b.tconv_bwd(False, 16, cp, H, Ht, h, ws, torch.randn_like(h))
torch.cuda.synchronize()
print("done", b.dims())
