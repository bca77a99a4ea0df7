#!/usr/bin/env python3
"""Workload for rocprofv3: a few launches of the streamed SpMM (both orientations) on the synthetic batch.
usage: python3 tools/profile_stream.py [instances] [reps]"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from mllp_amd.graph import synthetic_batch

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
b = synthetic_batch(n_inst)
H = torch.randn(b.N, 16, device="cuda"); Ht = torch.randn(b.M, 16, device="cuda")
Y = torch.empty(b.M, 16, device="cuda"); Yt = torch.empty(b.N, 16, device="cuda")
b.build_spmm_copy(False); b.build_spmm_copy(True)
for _ in range(reps):
    b.spmm(H, out=Y)
    b.spmm(Ht, transpose=True, out=Yt)
torch.cuda.synchronize()
print("done", b.dims(), b.spmm_copy_info(False))
