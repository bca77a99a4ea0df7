#!/usr/bin/env python3
"""Attention conv microbenchmark on the synthetic batch: LDS-tiled sweeps vs the STREAMED sweeps (stream_attn.hip), both
orientations, forward and backward of one 16-channel TransformerConv (all kernels of the conv, as the step runs them).
usage: python3 tools/bench_attn_stream.py [instances] [reps] [fwd|all]"""
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
what = sys.argv[3] if len(sys.argv) > 3 else "all"
GEOMS = tuple(int(x) for x in os.environ.get("MLLP_GEOMS", "1,2,3").split(","))
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
def rel(a, r): return (a - r).abs().max().item() / r.abs().max().item()
for dst_is_var, off in ((False, 1392), (True, 288)):
    nd, ns = (b.N, b.M) if dst_is_var else (b.M, b.N)
    cp = params[off:off + 1104].contiguous()
    g = torch.Generator(device="cuda").manual_seed(1)
    xs = torch.randn(ns, 16, device="cuda", generator=g); xd = torch.randn(nd, 16, device="cuda", generator=g)
    dh = torch.randn(nd, 16, device="cuda", generator=g)
    ws = b.tconv_workspace(dst_is_var, 16)
    byt_f = b.nnz * 8 + 4 * (nd + 1) + ns * 64 + nd * 408
    byt_b = 2 * b.nnz * 8 + 64 * ns + nd * 360 + 160 * nd + 128 * ns
    ref = b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws).clone()
    refb = [t.clone() for t in b.tconv_bwd(dst_is_var, 16, cp, xs, xd, ref, ws, dh.clone())[:3]]
    b.enable_tiled(dst_is_var, variant=1); b.enable_tiled(dst_is_var, variant=4); b.enable_tiled(not dst_is_var, variant=2)
    ms = timed(lambda: b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws))
    print(f"dst_is_var={dst_is_var} tiled    fwd {ms:.3f} ms  {byt_f/ms/1e6:.0f} GB/s ({byt_f/ms/1e6/8000:.3f})", flush=True)
    if what == "all":
        h = b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
        ms = timed(lambda: b.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh.clone()))
        print(f"dst_is_var={dst_is_var} tiled    bwd {ms:.3f} ms  {byt_b/ms/1e6:.0f} GB/s ({byt_b/ms/1e6/8000:.3f})", flush=True)
    infos = {gm: b.build_stream_copy(dst_is_var if gm != 2 else not dst_is_var, gm) for gm in (GEOMS if what == "all" else (1,))}
    ms = timed(lambda: b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)); got = b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
    print(f"dst_is_var={dst_is_var} streamed fwd {ms:.3f} ms  {byt_f/ms/1e6:.0f} GB/s ({byt_f/ms/1e6/8000:.3f}) maxrel={rel(got, ref):.2e}", flush=True)
    if what == "all":
        ms = timed(lambda: b.tconv_bwd(dst_is_var, 16, cp, xs, xd, got, ws, dh.clone()))
        gb = b.tconv_bwd(dst_is_var, 16, cp, xs, xd, got, ws, dh.clone())
        print(f"dst_is_var={dst_is_var} streamed bwd {ms:.3f} ms  {byt_b/ms/1e6:.0f} GB/s ({byt_b/ms/1e6/8000:.3f}) maxrel pg={rel(gb[0], refb[0]):.2e} dxd={rel(gb[1], refb[1]):.2e} dxs={rel(gb[2], refb[2]):.2e}", flush=True)
    for k, i in infos.items():
        print(f"   copy geom {k}: {i}", flush=True)
    b.drop_stream_copy(dst_is_var, 1); b.drop_stream_copy(not dst_is_var, 2); b.drop_stream_copy(dst_is_var, 3)
    b.disable_tiled(not dst_is_var, variant=2); b.disable_tiled(dst_is_var, variant=1); b.disable_tiled(dst_is_var, variant=4)
