#!/usr/bin/env python3
"""AngleModel ('angleNet') training step on one Netlib instance: ms per step (forward + BCE + backward + Adam).
usage: python3 tools/bench_angle.py [instance] [feat_dim] [steps]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from mllp_amd.angle import AngleModel, build_graph_from_Q_sets, dense_instance_tensors
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
set_seed(42)
model = AngleModel(feat_dim=F).to("cuda")
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
crit = torch.nn.BCEWithLogitsLoss()
y = torch.tensor(basis, dtype=torch.float, device="cuda")
losses = []
for it in range(steps + 2):
    if it == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad()
    loss = crit(model(g), y)
    loss.backward()
    opt.step()
    losses.append(float(loss.detach()))
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / steps
N = g.num_nodes
flops = 3 * (2 * 2 * N * N * F) * 3 + 3 * 2 * 4 * N * F * F * 3      # fwd 2 + bwd 4 N^2 F GEMMs per layer (x2 flops), projections
print(f"feat_dim={F}: {ms:.3f} ms per step, loss {losses[0]:.5f} -> {losses[-1]:.5f}, ~{flops / ms / 1e9:.1f} TFLOP/s of GEMM work")
