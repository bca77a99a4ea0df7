#!/usr/bin/env python3
"""Per-phase cycle breakdown of the AngleModel attention kernels (angle.hip::attn_kernel).  Needs the TIMING library
(`make -C mllp_amd/csrc timing`: s_memtime stamps summed over the wavefronts); the product library has no such code.
usage: python3 tools/angle_cycles.py [instance] [feat_dim] [steps]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from mllp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mllp_amd", "csrc", os.environ.get("MLLP_LIB", "libmllp_hip_timing.so"))
from mllp_amd.angle import AngleModel, AngleStepper, build_graph_from_Q_sets, dense_instance_tensors
from mllp_amd.data import load_packed
from mllp_amd.model import set_seed

name = sys.argv[1] if len(sys.argv) > 1 else "25fv47"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
inst = load_packed([name])[0]
Q, coefs, basis = dense_instance_tensors(inst)
g = build_graph_from_Q_sets(Q, coefs, torch.device("cuda"), inst.name, basis)
set_seed(42)
st = AngleStepper(AngleModel(feat_dim=F).to("cuda"), lr=1e-3)
y = torch.tensor(basis, dtype=torch.float, device="cuda")
L = _lib.lib()
L.mllp_debug_angle_stamps.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 30)()
st.step(g, y)
assert L.mllp_debug_angle_stamps(buf) == 0           # (clears the warm-up)
for _ in range(steps):
    st.step(g, y)
assert L.mllp_debug_angle_stamps(buf) == 0
a = np.array(list(buf), dtype=np.float64).reshape(3, 10)
names = ["prologue", "dot products", "arithmetic", "accumulation", "wait for the next tiles", "barrier", "epilogue", "issue of the next loads", "first read of the tile"]
N = g.num_nodes
print(f"{inst.name}: N = {N}, F = {F}; cycles per wavefront and launch (s_memtime, summed over {steps} steps x 3 layers)")
for m, title in enumerate(("FWD", "BQ", "BKV")):
    waves = a[m, 0]
    tot = a[m, 1:].sum()
    print(f"  {title}: {int(waves)} wavefronts, {tot / waves:9.0f} cycles each")
    for k, nm in enumerate(names):
        print(f"     {nm:26s} {100 * a[m, 1 + k] / tot:5.1f} %  {a[m, 1 + k] / waves:9.0f}")
