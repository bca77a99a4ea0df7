#!/usr/bin/env python3
"""AngleModel ('angleNet') training step on one Netlib instance: ms per step (forward + BCE + backward + Adam).
usage: python3 tools/bench_angle.py [instance] [feat_dim] [steps]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
if os.environ.get("MLLP_LIB"):              # experiments: a variant build of the library (tools/variant_lib.sh)
    from mllp_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "mllp_amd", "csrc", os.environ["MLLP_LIB"])
from mllp_amd.angle import AngleModel, AngleStepper, build_graph_from_Q_sets, dense_instance_tensors
from mllp_amd.data import load_packed
from mllp_amd.model import set_seed

name = sys.argv[1] if len(sys.argv) > 1 else "25fv47"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
inst = load_packed([name])[0]
t0 = time.perf_counter()
Q, coefs, basis = dense_instance_tensors(inst)
g = build_graph_from_Q_sets(Q, coefs, torch.device("cuda"), inst.name, basis)
torch.cuda.synchronize()
print(f"{inst.name}: m={inst.m} n={inst.n} -> N={g.num_nodes} nodes, {g.num_nodes * (g.num_nodes - 1)} edges; "
      f"QR + cosine matrix {time.perf_counter() - t0:.2f} s")
def run(kind):
    set_seed(42)
    model = AngleModel(feat_dim=F).to("cuda")
    y = torch.tensor(basis, dtype=torch.float, device="cuda")
    losses = []
    if kind == "module":          # the reference's loop shape: nn.Module + autograd + torch.optim.Adam
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        crit = torch.nn.BCEWithLogitsLoss()
        def one():
            opt.zero_grad()
            loss = crit(model(g), y)
            loss.backward()
            opt.step()
            return loss
    else:                         # flat parameters, no autograd, the library's Adam kernel
        st = AngleStepper(model, lr=1e-3)
        one = lambda: st.step(g, y)[0]
    for it in range(steps + 2):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        losses.append(one().detach())
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    return ms, [float(l) for l in losses]


N = g.num_nodes
# MFMA work of one step: attention 9 N^2 F multiply-adds per layer (forward 2, dQ 3, dK / dV 4), projections and their
# gradients 12 N F C per layer (layer 1: C = 2, no input gradient)
flops = 2 * (3 * 9 * N * N * F + 2 * 12 * N * F * F)
for kind in ("module", "flat"):
    ms, losses = run(kind)
    print(f"feat_dim={F} [{kind}]: {ms:.3f} ms per step, loss {losses[0]:.5f} -> {losses[-1]:.5f}, "
          f"{flops / ms / 1e9:.1f} TFLOP/s of MFMA work = {flops / ms / 1e9 / 157.3:.3f} of the 157.3 TFLOP/s fp32 matrix peak")
