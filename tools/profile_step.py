#!/usr/bin/env python3
"""Workload for rocprofv3: eager (no hipGraph) training steps so every kernel shows up by name.
usage: python3 tools/profile_step.py netlib|synthetic [steps] [synthetic_instances]"""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
if os.environ.get("MLLP_LIB"):              # experiments: a variant build of the library (tools/variant_lib.sh)
    from mllp_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "mllp_amd", "csrc", os.environ["MLLP_LIB"])
from mllp_amd.data import load_packed  # noqa: E402
from mllp_amd.graph import LPBatch, synthetic_batch  # noqa: E402
from mllp_amd.trainer import LPTrainer  # noqa: E402
from mllp_amd.model import GNNModel, set_seed  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "netlib"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n_syn = int(sys.argv[3]) if len(sys.argv) > 3 else 32
params = (set_seed(42), GNNModel().flat_parameters().detach().float().cuda())[1]
if which == "netlib":
    batch = LPBatch.from_instances(load_packed())
else:
    batch = synthetic_batch(n_syn)
    H = torch.randn(batch.N, 16, device="cuda")
    Y = torch.empty(batch.M, 16, device="cuda")
    for _ in range(steps):
        batch.spmm(H, out=Y)
    Ht = torch.randn(batch.M, 16, device="cuda")
    Yt = torch.empty(batch.N, 16, device="cuda")
    for _ in range(steps):
        batch.spmm(Ht, transpose=True, out=Yt)
    del H, Y, Ht, Yt
tr = LPTrainer(params, use_hip_graph=False, with_metrics=(which == "netlib"))
for _ in range(steps):
    tr.step(batch)
torch.cuda.synchronize()
print("done", which, batch.dims())
