#!/usr/bin/env python3
"""Streamed SpMM (stream_spmm.hip) on the synthetic batch: host-built vs device-built copy (bytes compared), result vs
the generic sweep, time vs the round-2 tiled kernel.  usage: python3 tools/bench_stream.py [instances] [reps] [host|nohost]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
if os.environ.get("MLLP_LIB"):              # experiments: a variant build of the library
    from mllp_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "mllp_amd", "csrc", os.environ["MLLP_LIB"])
if os.environ.get("MLLP_TIMING_LIB"):
    from mllp_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "mllp_amd", "csrc", "libmllp_hip_timing.so")
from mllp_amd.graph import synthetic_batch

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
with_host = (sys.argv[3] if len(sys.argv) > 3 else "host") == "host"
with_tiled = os.environ.get("BENCH_TILED", "1") == "1"
b = synthetic_batch(n_inst)
print("dims", b.dims(), flush=True)

def timed(fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for tr in (False, True):
    n_in, n_out = (b.M, b.N) if tr else (b.N, b.M)
    H = torch.randn(n_in, 16, device="cuda"); Y = torch.empty(n_out, 16, device="cuda")
    byt = b.nnz * 8 + 4 * (n_out + 1) + n_in * 64 + n_out * 64
    ms = timed(lambda: b.spmm(H, transpose=tr, out=Y)); ref = Y.clone()
    print(f"transpose={tr} generic  {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s", flush=True)
    host = None
    if with_host:
        t0 = time.time(); info = b.build_spmm_copy(tr, "host"); torch.cuda.synchronize()
        print(f"  host build {time.time()-t0:.2f}s", info, f"slots/nnz={info['entry_slots']/b.nnz:.4f}", flush=True)
        Y.zero_(); b.spmm(H, transpose=tr, out=Y); torch.cuda.synchronize()
        err = (Y - ref).abs().max().item() / ref.abs().max().item()
        ms = timed(lambda: b.spmm(H, transpose=tr, out=Y))
        print(f"transpose={tr} stream(host copy) {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s ({byt/ms/1e6/8000:.3f})  maxrel={err:.2e}", flush=True)
        host = b.export_spmm_copy(tr)
    t0 = time.time(); info = b.build_spmm_copy(tr, "device"); torch.cuda.synchronize()
    print(f"  device build {time.time()-t0:.3f}s", info, f"slots/nnz={info['entry_slots']/b.nnz:.4f}", flush=True)
    if host is not None:
        dev = b.export_spmm_copy(tr)
        for name, a, d in zip(("tile_blk", "blk_id", "rows", "ent", "tile_row", "hdr"), host, dev):
            same = a.shape == d.shape and np.array_equal(a, d)
            print(f"    {name}: host == device: {same}", "" if same else f"(first diff at {np.flatnonzero(a.ravel() != d.ravel())[:4] if a.shape == d.shape else (a.shape, d.shape)})")
        del host, dev
    Y.zero_(); b.spmm(H, transpose=tr, out=Y); torch.cuda.synchronize()
    err = (Y - ref).abs().max().item() / ref.abs().max().item()
    ms = timed(lambda: b.spmm(H, transpose=tr, out=Y))
    print(f"transpose={tr} stream   {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s  ({byt/ms/1e6/8000:.3f} of 8 TB/s)  maxrel={err:.2e}", flush=True)
    Y2 = torch.empty_like(Y); b.spmm(H, transpose=tr, out=Y2)
    print("    bitwise run-to-run:", bool(torch.equal(Y, Y2)))
    b.drop_spmm_copy(tr)
    if with_tiled:
        info = b.enable_tiled(tr); torch.cuda.synchronize()
        ms = timed(lambda: b.spmm(H, transpose=tr, out=Y))
        err = (Y - ref).abs().max().item() / ref.abs().max().item()
        print(f"transpose={tr} tiled(r02) {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s  ({byt/ms/1e6/8000:.3f})  maxrel={err:.2e}", flush=True)
        b.disable_tiled(tr)
