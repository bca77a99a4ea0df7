#!/usr/bin/env python3
"""Workload for rocprofv3: the layer-1 (one input channel) conv forward + backward on the synthetic batch with the
lane-per-row streamed copies (lane1_kernel<L1Fwd>, lane1_kernel<L1Bwd>), both orientations.
usage: python3 tools/profile_lane.py [instances] [reps]"""
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
for dst_is_var, off in ((False, 144), (True, 0)):
    nd, ns = (b.N, b.M) if dst_is_var else (b.M, b.N)
    cp = params[off:off + 144].contiguous()
    g = torch.Generator(device="cuda").manual_seed(1)
    xs = torch.randn(ns, device="cuda", generator=g); xd = torch.randn(nd, device="cuda", generator=g)
    dh = torch.randn(nd, 16, device="cuda", generator=g)
    ws = b.tconv_workspace(dst_is_var, 1)
    print(dst_is_var, b.build_stream_copy(dst_is_var, 4))
    for _ in range(reps):
        h = b.tconv_fwd(dst_is_var, 1, cp, xs, xd, ws)
        b.tconv_bwd(dst_is_var, 1, cp, xs, xd, h, ws, dh)
    b.drop_stream_copy(dst_is_var, 4)
torch.cuda.synchronize()
print("done", b.dims())
