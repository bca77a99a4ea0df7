#!/usr/bin/env python3
"""Per-phase cycle breakdown of the streamed attention sweeps (stream_attn.hip).  Needs the TIMING library
(`make -C mllp_amd/csrc timing`); the product library has no such switch.
usage: python3 tools/attn_stream_cycles.py [instances] [dst_is_var 0|1]"""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from mllp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mllp_amd", "csrc", os.environ.get("MLLP_LIB", "libmllp_hip_timing.so"))
from mllp_amd.graph import synthetic_batch
from mllp_amd.model import GNNModel, set_seed

n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dst_is_var = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
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
infos = {gm: b.build_stream_copy(dst_is_var if gm != 2 else not dst_is_var, gm) for gm in (1, 2, 3)}
h = b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws).clone()
WALK = ["init", "top of the block", "pass 0", "pass 1", "wait for the prefetch", "barrier"]
STAGE = ["init", "issue", "wait for the pieces", "barrier", "-", "-"]


def report(name, out, info, nw=8, ns_=4):
    n_tiles, n_tb = info["n_tiles"], info["n_tb"]
    c = out[0:2 * n_tiles:2, :].double().cpu()
    c2 = out[1:2 * n_tiles:2, :].double().cpu()
    steps = info["n_groups"] * 4
    print(f"{name}: tiles={n_tiles} blocks={n_tb} nnz/block={b.nnz / n_tb:.0f} slots/nnz={info['entry_slots'] / b.nnz:.3f} wave-steps/block={steps / n_tb / nw:.1f}")
    for title, cc, n, names in (("walkers", c, nw, WALK), ("stagers", c2, ns_, STAGE)):
        tot = cc[:, 6].sum()
        print(f"  {title}: {tot / n / n_tiles:10.0f} cycles per tile, {tot / n / n_tb:8.0f} per block")
        for k in range(6):
            if names[k] != "-":
                print(f"     {names[k]:26s} {cc[:, k].sum() / tot * 100:6.1f} %   {cc[:, k].sum() / n / n_tb:8.0f} cycles per block")


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"lib {os.path.basename(_lib.LIB_PATH)}: conv forward {timed(lambda: b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)):.3f} ms")
os.environ["MLLP_ATTN_STAMPS"] = "1"
for _ in range(2):
    hs = b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
torch.cuda.synchronize()
report("forward", hs, infos[1])
if len(sys.argv) > 3 and sys.argv[3] == "fwd":
    sys.exit(0)
# backward: the destination-major sweep writes dq' (inside the workspace), the source-major one dx_src
os.environ["MLLP_ATTN_STAMPS"] = "0"
b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
os.environ["MLLP_ATTN_STAMPS"] = "1"
for _ in range(2):
    pg, dxd, dxs, gg = b.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh.clone())
torch.cuda.synchronize()
report("source-major backward", dxs, infos[2])
os.environ["MLLP_ATTN_STAMPS"] = "0"
