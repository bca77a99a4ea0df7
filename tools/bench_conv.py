#!/usr/bin/env python3
"""Attention conv forward microbenchmark on the synthetic batch: generic sweep vs LDS-tiled (variant 1).
usage: python3 tools/bench_conv.py [instances] [reps]"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
if os.environ.get("MLLP_LIB"):              # experiments: a variant build of the library
    from mllp_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "mllp_amd", "csrc", os.environ["MLLP_LIB"])
from mllp_amd.graph import synthetic_batch
from mllp_amd.model import GNNModel, set_seed

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
b = synthetic_batch(n_inst)
set_seed(42)
params = GNNModel().flat_parameters().detach().float().cuda()
def timed(fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for dst_is_var, off in ((False, 1392), (True, 288)):
    nd, ns = (b.N, b.M) if dst_is_var else (b.M, b.N)
    cp = params[off:off + 1104].contiguous()
    xs = torch.randn(ns, 16, device="cuda"); xd = torch.randn(nd, 16, device="cuda")
    ws = b.tconv_workspace(dst_is_var, 16)
    byt = b.nnz * 8 + 4 * (nd + 1) + ns * 64 + nd * 408
    ms = timed(lambda: b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)); ref = b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws).clone()
    print(f"dst_is_var={dst_is_var} generic fwd {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s")
    info = b.enable_tiled(dst_is_var, variant=1)
    ms = timed(lambda: b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)); got = b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print(f"dst_is_var={dst_is_var} tiled   fwd {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s  ({byt/ms/1e6/8000:.3f} of 8 TB/s) maxrel={err:.2e}  {info}")
    # backward: generic source-major gather sweep vs LDS-tiled (variant 2, attached to the orientation whose rows are the sources)
    h = b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
    dh = torch.randn(nd, 16, device="cuda")
    ms_g = timed(lambda: b.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh)); ref = b.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh)[2].clone()
    info = b.enable_tiled(not dst_is_var, variant=2)
    ms_t = timed(lambda: b.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh)); got = b.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh)[2]
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print(f"dst_is_var={dst_is_var} conv bwd (all kernels) generic {ms_g:.3f} ms, with tiled source sweep {ms_t:.3f} ms  maxrel={err:.2e}  {info}")
    info = b.enable_tiled(dst_is_var, variant=4)
    ms_d = timed(lambda: b.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh))
    print(f"dst_is_var={dst_is_var} conv bwd with tiled source AND destination sweeps {ms_d:.3f} ms  {info}")
    b.disable_tiled(not dst_is_var, variant=2); b.disable_tiled(dst_is_var, variant=1); b.disable_tiled(dst_is_var, variant=4)
# layer-1 (one channel) convs: generic sweeps vs LDS-tiled (variant 3)
for dst_is_var, off in ((False, 144), (True, 0)):
    nd, ns = (b.N, b.M) if dst_is_var else (b.M, b.N)
    cp = params[off:off + 144].contiguous()
    xs = torch.randn(ns, device="cuda"); xd = torch.randn(nd, device="cuda"); dh = torch.randn(nd, 16, device="cuda")
    ws = b.tconv_workspace(dst_is_var, 1)
    h = b.tconv_fwd(dst_is_var, 1, cp, xs, xd, ws).clone()
    ms_f = timed(lambda: b.tconv_fwd(dst_is_var, 1, cp, xs, xd, ws))
    ms_b = timed(lambda: b.tconv_bwd(dst_is_var, 1, cp, xs, xd, h, ws, dh))
    info = b.enable_tiled(dst_is_var, variant=3)
    h2 = b.tconv_fwd(dst_is_var, 1, cp, xs, xd, ws)
    err = (h2 - h).abs().max().item() / h.abs().max().item()
    ms_ft = timed(lambda: b.tconv_fwd(dst_is_var, 1, cp, xs, xd, ws))
    ms_bt = timed(lambda: b.tconv_bwd(dst_is_var, 1, cp, xs, xd, h, ws, dh))
    print(f"dst_is_var={dst_is_var} layer-1 conv fwd generic {ms_f:.3f} ms tiled {ms_ft:.3f} ms | bwd generic {ms_b:.3f} ms tiled {ms_bt:.3f} ms  maxrel={err:.2e} {info}")
    b.disable_tiled(dst_is_var, variant=3)
