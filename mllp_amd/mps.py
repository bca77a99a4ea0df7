"""MPS -> the tensors of the reference's loader.

The reference reads `dataset/netlib_mps_norm/<name>_{constrs.npz,coefs.npy,rhs.npy,basis.npy}`
(linear_program_data.py:58-80) but does not contain the script that made them from `netlib_mps/<name>.mps`.
`read_mps` / `convert_directory` are that step: the parser and the normalisation run in the library
(mllp_amd/csrc/mps_reader.cpp through `mllp_mps_read`, host code), and reproduce the reference's tensors for all 97
Netlib instances (raw stage exactly, normalized stage to 1e-12; tests/test_mps.py).  Labels need an LP solver, which is
not part of this build: `basis` is assembled from a solver's statuses when they are given, else left empty.
"""
import ctypes
import os
from ctypes import c_int64, c_void_p
from typing import Optional

import numpy as np

from . import _lib
from .data import LPInstance


def read_mps(path: str, normalize: bool = True, name: Optional[str] = None, var_status=None, constr_status=None):
    """Parse one MPS file.  Returns (LPInstance, info); info = dict(n_struct, n_range, n_slack, slack_rows).
    `var_status` (per column incl. range columns) / `constr_status` (per row): 0/1 basis statuses of a solver
    (the reference's `_v.npy`, `_c.npy`): with both, instance.basis = [v ; c[slack_rows]] as in the reference's files."""
    L = _lib.lib()
    h = c_void_p()
    _lib.check(L.mllp_mps_read(os.fsencode(path), int(bool(normalize)), ctypes.byref(h)))
    try:
        d = (c_int64 * 6)()
        _lib.check(L.mllp_lp_dims(h, d))
        m, n, nnz, n_struct, n_range, n_slack = [int(v) for v in d]
        indptr = np.zeros(m + 1, np.int64)
        indices = np.zeros(nnz, np.int32)
        values = np.zeros(nnz, np.float64)
        coefs = np.zeros(n, np.float64)
        rhs = np.zeros(m, np.float64)
        slack_rows = np.zeros(n_slack, np.int32)
        p = lambda a: a.ctypes.data_as(c_void_p)
        _lib.check(L.mllp_lp_export(h, p(indptr), p(indices), p(values), p(coefs), p(rhs), p(slack_rows)))
    finally:
        L.mllp_lp_free(h)
    basis = np.zeros(0, np.int32)
    if var_status is not None and constr_status is not None:
        basis = np.concatenate([np.asarray(var_status), np.asarray(constr_status)[slack_rows]]).astype(np.int32)
        if basis.shape[0] != n:
            raise ValueError(f"{path}: {basis.shape[0]} statuses for {n} columns")
    inst = LPInstance(name or os.path.basename(path), indptr, indices, values, coefs, rhs, basis)
    return inst, dict(n_struct=n_struct, n_range=n_range, n_slack=n_slack, slack_rows=slack_rows)


def convert_directory(mps_dir: str, out_dir: str, normalize: bool = True, status_dir: Optional[str] = None):
    """Every `<name>.mps` of `mps_dir` -> `<out_dir>/<name>.mps_{constrs.npz,coefs.npy,rhs.npy}` in the reference's
    layout (and `_basis.npy` when `<status_dir>/<name>.mps_{v,c}.npy` exist).  Returns the list of names written."""
    import scipy.sparse as sp
    os.makedirs(out_dir, exist_ok=True)
    done = []
    for f in sorted(os.listdir(mps_dir)):
        if not f.lower().endswith(".mps"):
            continue
        v = c = None
        if status_dir and os.path.exists(os.path.join(status_dir, f + "_v.npy")):
            v = np.load(os.path.join(status_dir, f + "_v.npy"))
            c = np.load(os.path.join(status_dir, f + "_c.npy"))
        inst, _ = read_mps(os.path.join(mps_dir, f), normalize, name=f, var_status=v, constr_status=c)
        A = sp.csr_matrix((inst.values, inst.indices, inst.indptr.astype(np.int32)), shape=(inst.m, inst.n))
        sp.save_npz(os.path.join(out_dir, f + "_constrs.npz"), A)
        np.save(os.path.join(out_dir, f + "_coefs.npy"), inst.coefs)
        np.save(os.path.join(out_dir, f + "_rhs.npy"), inst.rhs)
        if inst.basis.size:
            np.save(os.path.join(out_dir, f + "_basis.npy"), inst.basis)
        done.append(f)
    return done
