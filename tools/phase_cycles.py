#!/usr/bin/env python3
"""Per-phase cycle breakdown of the tiled SpMM kernel.  Needs the TIMING library (`make -C mllp_amd/csrc timing`,
libmllp_hip_timing.so: cycle counters instead of results under MLLP_TILED_ABLATION=16); the product library has no
such switch.
usage: python3 tools/phase_cycles.py [instances]"""
import os, sys
os.environ["MLLP_TILED_ABLATION"] = "16"
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from mllp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mllp_amd", "csrc", "libmllp_hip_timing.so")
from mllp_amd.graph import synthetic_batch

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 64
v1 = False
b = synthetic_batch(n_inst)
for tr in (False, True):
    n_in, n_out = (b.M, b.N) if tr else (b.N, b.M)
    H = torch.randn(n_in, 16, device="cuda"); Y = torch.zeros(n_out, 16, device="cuda")
    info = b.enable_tiled(tr)
    n_tiles, n_tb = info["n_tiles"], info["n_tb"]
    for _ in range(2):
        b.spmm(H, transpose=tr, out=Y)
    torch.cuda.synchronize()
    print(f"transpose={tr} tiles={n_tiles} blocks={n_tb} nnz/block={b.nnz / n_tb:.0f}")
    if v1:
        c = Y[:n_tiles, :16].double().cpu()
        groups = [("all 16 waves", 0, 16, ["barrier1 (walk imbalance)", "vmcnt wait + ds_write", "barrier2", "prefetch issue", "walk"])]
    else:
        c = Y[0:2 * n_tiles:2, :16].double().cpu()
        pw = Y[1:2 * n_tiles:2, :16].double().cpu()
        groups = [("walkers (8 waves)", 0, 8, ["wait barrier A", "idle A..B (stage)", "-", "-", "walk"]),
                  ("loaders (4 H + 4 entry waves)", 8, 8, ["wait barrier B", "vmcnt wait + ds_write", "wait barrier A", "load issue", "-"])]
    for title, o, nw, names in groups:
        tot = c[:, o + 5].sum()
        print(f"  {title}: {c[:, o + 5].mean() / nw:10.0f} cycles per tile, {c[:, o + 5].sum() / nw / n_tb:8.0f} per block")
        for k in range(5):
            if names[k] != "-":
                print(f"     {names[k]:26s} {c[:, o + k].sum() / tot * 100:6.1f} %   {c[:, o + k].sum() / nw / n_tb:8.0f} cycles per block")
    if not v1:
        print("   per walker wave, cycles per block: walk  ", " ".join(f"{v:5.0f}" for v in (pw[:, :8].sum(0) / n_tb).tolist()))
        print("                                      wait A", " ".join(f"{v:5.0f}" for v in (pw[:, 8:].sum(0) / n_tb).tolist()))
