#!/usr/bin/env python3
"""Per-parameter-group comparison of the fused and the generic whole-model paths (debugging aid)."""
import os, sys
import numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import scipy.sparse as sp
from mllp_amd.data import LPInstance, load_packed
from mllp_amd.graph import LPBatch
from oracle import pyg_restatement as o1

def ramp():
    n, m = 700, 3
    rng = np.random.default_rng(5)
    dense = np.zeros((m, n))
    dense[0, :] = np.linspace(-1.0, 1.0, n)
    dense[1, ::7] = rng.standard_normal(len(range(0, n, 7)))
    dense[2, :40] = -np.linspace(0.1, 1.0, 40)
    A = sp.csr_matrix(dense); A.sort_indices()
    return [LPInstance("ramp", A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data,
                       np.linspace(-1, 1, n), np.array([3.0, 0.0, 1.0]), (rng.random(n) < 0.4).astype(np.int32))]

which = sys.argv[1] if len(sys.argv) > 1 else "ramp"
inst = ramp() if which == "ramp" else load_packed()
sd = o1.init_state(9, torch.float64)
if which == "ramp":
    for k in sd:
        if "lin_query" in k or "lin_edge" in k:
            sd[k] = sd[k] * 6.0
flat = o1.flatten_state(sd).float().cuda()
bg, bf = LPBatch.from_instances(inst).set_path(1), LPBatch.from_instances(inst).set_path(2)
lg, zg, gg = [t.clone() for t in bg.loss_step(flat)]
lf, zf, gf = [t.clone() for t in bf.loss_step(flat)]
print("loss", float(lg), float(lf), "logits maxdiff", float((zg - zf).abs().max()))
off = 0
for k, shp in o1.state_dict_spec():
    c = int(np.prod(shp))
    a, b = gg[off:off + c], gf[off:off + c]
    d = float((a - b).abs().max()); sc = float(a.abs().max())
    flag = "  <<<<" if d > 1e-4 * max(sc, 1e-6) and "lin_key.bias" not in k else ""
    print(f"{k:32s} ref max {sc:.3e}  diff {d:.3e}{flag}")
    off += c
