import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mllp_amd.data import load_packed, SUBSET5
from mllp_amd.graph import LPBatch
from mllp_amd.trainer import LPTrainer
from oracle import pyg_restatement as o1
gold = np.load("tests/golden/subset5.npz")
flat = torch.tensor(gold["weights_flat"], dtype=torch.float32, device="cuda")
names = [k for k, s in o1.state_dict_spec() for _ in range(int(np.prod(s)))]
for which in (["afiro.mps"], SUBSET5, None):
    inst = load_packed(which)
    b = LPBatch.from_instances(inst)
    ref = None
    for rep in range(6):
        use_graph = rep >= 3
        tr = LPTrainer(flat, lr=1e-3, use_hip_graph=use_graph)
        for _ in range(3):
            tr.step(b)
        torch.cuda.synchronize()
        p = tr.params.cpu().numpy()
        l, z, g = b.loss_step(flat)
        g = g.cpu().numpy()
        if ref is None:
            ref, gref = p, g
        d = np.abs(p - ref); dg = np.abs(g - gref)
        i = int(np.argmax(d))
        print(len(inst), "rep", rep, "graph" if use_graph else "eager", "max param diff vs rep0", d.max(), names[i], "| grad diff", dg.max(), names[int(np.argmax(dg))])
