#!/usr/bin/env python3
"""Workload for rocprofv3: one 16-channel TransformerConv forward + backward on the synthetic batch with the STREAMED copies
(fwd16_stream_kernel, bwddst16_stream_kernel, bwdsrc16_stream_kernel) -- or, with `tiled`, on the LDS-tiled copies.
usage: python3 tools/profile_attn_stream.py [instances] [reps] [stream|tiled] [dst_is_var 0|1]"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
if os.environ.get("MLLP_LIB"):
    from mllp_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "mllp_amd", "csrc", os.environ["MLLP_LIB"])
from mllp_amd.graph import synthetic_batch
from mllp_amd.model import GNNModel, set_seed

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = sys.argv[3] if len(sys.argv) > 3 else "stream"
dst_is_var = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
b = synthetic_batch(n_inst)
set_seed(42)
params = GNNModel().flat_parameters().detach().float().cuda()
off = 288 if dst_is_var else 1392
nd, ns = (b.N, b.M) if dst_is_var else (b.M, b.N)
cp = params[off:off + 1104].contiguous()
g = torch.Generator(device="cuda").manual_seed(1)
xs = torch.randn(ns, 16, device="cuda", generator=g); xd = torch.randn(nd, 16, device="cuda", generator=g)
dh = torch.randn(nd, 16, device="cuda", generator=g)
ws = b.tconv_workspace(dst_is_var, 16)
if kind == "tiled":
    assert b.enable_tiled(dst_is_var, variant=1) and b.enable_tiled(not dst_is_var, variant=2) and b.enable_tiled(dst_is_var, variant=4)
else:
    for gm in (1, 2, 3):
        b.build_stream_copy(dst_is_var if gm != 2 else not dst_is_var, gm)
for _ in range(reps):
    h = b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
    b.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh.clone())
torch.cuda.synchronize()
print("done", b.dims())
