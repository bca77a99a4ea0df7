#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the build container only).

Inputs are captured THROUGH THE IMPORTED REFERENCE LOADER
(`/root/reference/linear_program_data.py:58-80`, importable here: SURVEY.md §8c), so they pin the
loader contract (tuple order, dtypes, CSR edge order).  Expected outputs come from the fp64
literal oracle (`oracle/pyg_restatement.py`) with seeded weights stored alongside.  The model
code of the reference itself cannot be run (torch_geometric absent, unpinned): the outputs are
"parity unpinned" with respect to PyG, as stated in the oracle's header.

Writes  tests/golden/subset5.npz  (afiro, adlittle, blend, kb2, sc50a: inputs, weights, fp64
logits / loss / grads for the 5-instance batch and for afiro alone) and
tests/golden/loader_digest.json  (per-instance shape + checksum of every one of the 97 packed
instances as returned by the reference loader).
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from mllp_amd.data import SUBSET5, LPInstance, load_packed  # noqa: E402
from oracle import pyg_restatement as o1  # noqa: E402


def digest(inst):
    h = hashlib.sha256()
    for a in (inst.indptr.astype(np.int64), inst.indices.astype(np.int32), inst.values.astype(np.float64),
              inst.coefs.astype(np.float64), inst.rhs.astype(np.float64), inst.basis.astype(np.int32)):
        h.update(np.ascontiguousarray(a).tobytes())
    return dict(m=inst.m, n=inst.n, nnz=inst.nnz, sha256=h.hexdigest())


def main():
    ref = "/root/reference"
    sys.path.insert(0, ref)
    cwd = os.getcwd()
    os.chdir(ref)
    try:
        import linear_program_data as ref_data
        dataset, train_dict = ref_data.get_netlib_dataset(normalize=True)
    finally:
        os.chdir(cwd)
    assert list(train_dict.keys())[0] == "obj"
    by_name = {t[0]: LPInstance.from_reference_tuple(t) for t in dataset}

    # 1. loader digest for all 97 instances (pins the packed fixture to the reference loader)
    packed = {i.name: i for i in load_packed()}
    assert sorted(packed) == sorted(by_name)
    dig = {}
    for name in sorted(by_name):
        d_ref, d_pack = digest(by_name[name]), digest(packed[name])
        assert d_ref == d_pack, name
        dig[name] = d_ref
    with open(os.path.join(HERE, "loader_digest.json"), "w") as fh:
        json.dump(dig, fh, indent=0, sort_keys=True)

    # 2. subset inputs + fp64 oracle outputs
    inst = [by_name[n] for n in SUBSET5]
    sd = o1.init_state(42, torch.float64)
    out = {"names": np.array(SUBSET5), "weights_flat": o1.flatten_state(sd).numpy()}
    for i in inst:
        k = i.name.replace(".mps", "")
        out[f"{k}_indptr"], out[f"{k}_indices"], out[f"{k}_values"] = i.indptr, i.indices, i.values
        out[f"{k}_coefs"], out[f"{k}_rhs"], out[f"{k}_basis"] = i.coefs, i.rhs, i.basis
    loss, logits, grads = o1.batch_loss_and_grads(sd, inst, torch.float64)
    out["batch_loss"] = np.array(float(loss))
    out["batch_logits"] = torch.cat(logits).numpy()
    out["batch_grads"] = grads.numpy()
    afiro = [by_name["afiro.mps"]]
    loss, logits, grads = o1.batch_loss_and_grads(sd, afiro, torch.float64)
    out["afiro_loss"] = np.array(float(loss))
    out["afiro_logits"] = logits[0].numpy()
    out["afiro_grads"] = grads.numpy()
    # three reference-style Adam steps on afiro (one step per instance, experiment.py:123-144), fp64
    tr = o1.ReferenceTrainer(sd, lr=1e-3, dtype=torch.float64, rebuild_graph=True)
    losses = [tr.step(afiro[0])[0] for _ in range(3)]
    out["afiro_adam3_losses"] = np.array(losses)
    out["afiro_adam3_weights"] = o1.flatten_state(tr.sd).detach().numpy()
    np.savez_compressed(os.path.join(HERE, "subset5.npz"), **out)
    print("golden written:", {k: getattr(v, "shape", None) for k, v in out.items() if not k.split("_")[0] in
                              ("adlittle", "blend", "kb2", "sc50a")})


if __name__ == "__main__":
    main()
