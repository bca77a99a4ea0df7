#!/usr/bin/env python3
"""Pack the reference's normalized Netlib tensors into one compressed .npz.

The reference loader (`/root/reference/linear_program_data.py:58-80`) lists
`netlib_mps/` for instance names and reads four files per instance from
`dataset/netlib_mps_norm/`.  The GPU box only sees this repository, so the 97
reachable instances are packed here once (data, not code): concatenated CSR
arrays plus per-instance offsets, fp64/int32 preserved exactly as on disk.

This script IMPORTS the reference loader (allowed: SURVEY.md §8c) so that the
pack is by construction what `get_netlib_dataset(normalize=True)` returns.  It
runs in the build container only; nothing on the GPU box needs it.

usage: python tools/pack_netlib.py [/root/reference] [data/netlib_norm.npz]
"""
import os
import sys

import numpy as np


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = os.path.abspath(sys.argv[2] if len(sys.argv) > 2 else
                          os.path.join(os.path.dirname(__file__), "..", "data", "netlib_norm.npz"))
    sys.path.insert(0, ref)
    cwd = os.getcwd()
    os.chdir(ref)  # the reference loader uses cwd-relative paths (data.py:59,66)
    try:
        import linear_program_data as ref_data
        dataset, _ = ref_data.get_netlib_dataset(normalize=True)
        listdir_order = [t[0] for t in dataset]
    finally:
        os.chdir(cwd)

    dataset.sort(key=lambda t: t[0])  # sorted-name order is the build's canonical order
    names, ms, ns, nnzs = [], [], [], []
    indptr, indices, values, coefs, rhs, basis = [], [], [], [], [], []
    for name, constrs, weights, c, b, y in dataset:
        m, n = len(b), len(c)
        assert len(constrs) == m and len(y) == n
        ptr = np.zeros(m + 1, dtype=np.int64)
        ptr[1:] = np.cumsum([len(r) for r in constrs])
        assert ptr[-1] == len(weights)
        names.append(name)
        ms.append(m)
        ns.append(n)
        nnzs.append(int(ptr[-1]))
        indptr.append(ptr)
        indices.append(np.concatenate(constrs).astype(np.int32) if m else np.zeros(0, np.int32))
        values.append(np.asarray(weights, dtype=np.float64))
        coefs.append(np.asarray(c, dtype=np.float64))
        rhs.append(np.asarray(b, dtype=np.float64))
        basis.append(np.asarray(y, dtype=np.int32))
    np.savez_compressed(
        out,
        names=np.array(names),
        listdir_order=np.array(listdir_order),
        m=np.array(ms, dtype=np.int64),
        n=np.array(ns, dtype=np.int64),
        nnz=np.array(nnzs, dtype=np.int64),
        indptr=np.concatenate(indptr),          # per instance: m+1 local offsets
        indices=np.concatenate(indices),        # local column ids
        values=np.concatenate(values),
        coefs=np.concatenate(coefs),
        rhs=np.concatenate(rhs),
        basis=np.concatenate(basis),
    )
    print(f"packed {len(names)} instances, nnz={sum(nnzs)}, "
          f"sum m={sum(ms)}, sum n={sum(ns)} -> {out} ({os.path.getsize(out)/1e6:.1f} MB)")


if __name__ == "__main__":
    main()
