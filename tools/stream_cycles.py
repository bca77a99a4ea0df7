#!/usr/bin/env python3
"""Per-phase cycle breakdown and ablation timings of the streamed SpMM kernel.  Needs the TIMING library
(`make -C mllp_amd/csrc timing`); the product library has no such switch.
usage: python3 tools/stream_cycles.py [instances] [both]"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from mllp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mllp_amd", "csrc", "libmllp_hip_timing.so")
from mllp_amd.graph import synthetic_batch

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 64
both = len(sys.argv) > 2 and sys.argv[2] == "both"
b = synthetic_batch(n_inst)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ABL = ((0, "full"), (1, "no walk (loads + staging + barriers)"), (2, "no staging"))
ROLES = (("8 walking wavefronts", 0, 8, ["wait for the prefetch to land", "wait at the barrier", "walk", "top of the block"]),)
for tr in ((False, True) if both else (False,)):
    n_in, n_out = (b.M, b.N) if tr else (b.N, b.M)
    H = torch.randn(n_in, 16, device="cuda")
    Y = torch.zeros(n_out, 16, device="cuda")
    info = b.build_spmm_copy(tr, "device")
    n_tiles, n_tb = info["n_tiles"], info["n_tb"]
    print(f"transpose={tr} tiles={n_tiles} blocks={n_tb} nnz/block={b.nnz / n_tb:.0f} slots/nnz={info['entry_slots'] / b.nnz:.4f}")
    for abl, name in ABL:
        os.environ["MLLP_STREAM_ABLATION"] = str(abl)
        print(f"   {name:40s} {timed(lambda: b.spmm(H, transpose=tr, out=Y)):.3f} ms")
    for stamp in (16, 18, 20, 24, 28):
        os.environ["MLLP_STREAM_ABLATION"] = str(stamp)
        print(f"  -- stamps, ablation bits {stamp - 16} (2 = no staging, 4 = no LDS reads, 8 = no FMAs, 64 = no entry reloads, 256 = only wavefront 0 walks: its time = 16 x the walk figure)")
        for _ in range(2):
            b.spmm(H, transpose=tr, out=Y)
        torch.cuda.synchronize()
        c = Y[0:2 * n_tiles:2, :].double().cpu()   # [tiles, 16]
        for title, o, nw, names in ROLES:
            tot = c[:, o + 4].sum()
            print(f"  {title}: {tot / nw / n_tiles:10.0f} cycles per tile, {tot / nw / n_tb:8.0f} per block")
            for k in range(4):
                if names[k] != "-":
                    print(f"     {names[k]:34s} {c[:, o + k].sum() / tot * 100:6.1f} %   {c[:, o + k].sum() / nw / n_tb:8.0f} cycles per block")
    os.environ["MLLP_STREAM_ABLATION"] = "0"
    b.drop_spmm_copy(tr)
