#!/usr/bin/env python3
"""Layer-1 (one input channel) conv microbenchmark on the synthetic batch: generic sweeps, LDS-tiled (variant 3) and the
lane-per-row streamed copy (lane_stream.hip, geometry 4), both orientations, forward and backward of one
TransformerConv(1, 16) (all kernels of the conv, as the step runs them).
usage: python3 tools/bench_lane.py [instances] [reps]"""
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
def rel(a, r): return (a - r).abs().max().item() / r.abs().max().item()
for dst_is_var, off in ((False, 144), (True, 0)):
    nd, ns = (b.N, b.M) if dst_is_var else (b.M, b.N)
    cp = params[off:off + 144].contiguous()
    g = torch.Generator(device="cuda").manual_seed(1)
    xs = torch.randn(ns, device="cuda", generator=g); xd = torch.randn(nd, device="cuda", generator=g)
    dh = torch.randn(nd, 16, device="cuda", generator=g)
    ws = b.tconv_workspace(dst_is_var, 1)
    byt = b.nnz * 6                                       # the streamed entries; everything else is per node
    ref = refb = None
    for kind in ("generic", "tiled", "lane"):
        if kind == "tiled": b.enable_tiled(dst_is_var, variant=3)
        if kind == "lane": info = b.build_stream_copy(dst_is_var, 4)
        msf = timed(lambda: b.tconv_fwd(dst_is_var, 1, cp, xs, xd, ws))
        h = b.tconv_fwd(dst_is_var, 1, cp, xs, xd, ws)
        hh = ref if ref is not None else h                # every backward gets the generic forward's output (same ReLU mask)
        msb = timed(lambda: b.tconv_bwd(dst_is_var, 1, cp, xs, xd, hh, ws, dh))
        pg = b.tconv_bwd(dst_is_var, 1, cp, xs, xd, hh, ws, dh)[0]
        if ref is None: ref, refb = h.clone(), pg.clone()
        keep = torch.ones(144, dtype=torch.bool, device="cuda"); keep[16:32] = False
        print(f"dst_is_var={dst_is_var} {kind:8s} fwd {msf:.3f} ms ({byt/msf/1e6/8000:.3f} of 8 TB/s at 6 B/nnz)  bwd (all kernels) {msb:.3f} ms"
              f"  maxrel h={rel(h, ref):.2e} pg={rel(pg[keep], refb[keep]):.2e}", flush=True)
        if kind == "tiled": b.disable_tiled(dst_is_var, variant=3)
        if kind == "lane":
            print("   copy:", info, f"padding {info['entry_slots'] / b.nnz - 1:.3%}", flush=True)
            b.drop_stream_copy(dst_is_var, 4)
