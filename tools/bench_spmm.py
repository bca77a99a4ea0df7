#!/usr/bin/env python3
"""SpMM microbenchmark on the synthetic batch: generic sweep vs LDS-tiled, both orientations.
usage: python3 tools/bench_spmm.py [instances] [reps]"""
import os, sys, time
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
if os.environ.get("MLLP_TIMING_LIB"):       # experiments: the timing library honours MLLP_TILED_ABLATION
    from mllp_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "mllp_amd", "csrc", "libmllp_hip_timing.so")
from mllp_amd.graph import synthetic_batch

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
b = synthetic_batch(n_inst)
print("dims", b.dims())
def timed(fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for tr in ((False,) if os.environ.get("MLLP_TIMING_LIB") else (False, True)):
    n_in, n_out = (b.M, b.N) if tr else (b.N, b.M)
    H = torch.randn(n_in, 16, device="cuda"); Y = torch.empty(n_out, 16, device="cuda")
    byt = b.nnz * 8 + 4 * (n_out + 1) + n_in * 64 + n_out * 64
    ms = timed(lambda: b.spmm(H, transpose=tr, out=Y)); ref = Y.clone()
    print(f"transpose={tr} generic {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s")
    t0 = time.time(); info = b.enable_tiled(tr); torch.cuda.synchronize()
    print("  tiled build", f"{time.time()-t0:.2f}s", info)
    if info:
        ms = timed(lambda: b.spmm(H, transpose=tr, out=Y))
        err = (Y - ref).abs().max().item() / ref.abs().max().item()
        print(f"transpose={tr} tiled   {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s  ({byt/ms/1e6/8000:.3f} of 8 TB/s)  maxrel={err:.2e}")
