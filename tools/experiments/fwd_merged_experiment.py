"""Experiment script (NOT a test; run by hand on a GPU box): three forward launches merged into one persistent launch."""
import os, sys, time, numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
if os.environ.get("MLLP_LIB"):
    from mllp_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "mllp_amd", "csrc", os.environ["MLLP_LIB"])
from mllp_amd.data import load_packed
from mllp_amd.graph import LPBatch
from mllp_amd.trainer import LPTrainer
gold = np.load(os.path.join(ROOT, "tests", "golden", "subset5.npz"), allow_pickle=False)
params = torch.tensor(gold["weights_flat"], dtype=torch.float32, device="cuda")
b5 = LPBatch.from_instances(load_packed(["adlittle.mps","afiro.mps","blend.mps","kb2.mps","sc50a.mps"]))
names=[str(n) for n in gold["names"]]
b5 = LPBatch.from_instances(load_packed(names))
loss, logits, grads = b5.loss_step(params)
print("subset5 logits maxrel", float(np.abs(logits.cpu().numpy()-gold["batch_logits"]).max()/np.abs(gold["batch_logits"]).max()), "loss", float(loss[0]), float(gold["batch_loss"]))
full = LPBatch.from_instances(load_packed())
tr = LPTrainer(params, lr=1e-3, use_hip_graph=False)
for _ in range(10): tr.step(full)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(300): tr.step(full)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/300
print("netlib step ms", dt*1e3, "final loss", float(tr.last_loss(full)[0]))
