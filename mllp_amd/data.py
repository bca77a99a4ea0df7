"""Netlib LP instance loading (hot-path row a1 of SURVEY.md §8).

`get_netlib_dataset(normalize=True)` keeps the reference contract
(reference `linear_program_data.py:58-80`): it returns ``(dataset, train_dict)``
where each dataset entry is the 6-tuple
``(file, constrs, constrs_weights, coefs, rhs, basis_opt)`` --
``constrs`` = list of m int32 arrays (column ids of each CSR row),
``constrs_weights`` = CSR ``data`` in row-major order (float64),
``coefs`` (n,), ``rhs`` (m,), ``basis_opt`` (n,) int32 in {0,1}; and
``train_dict = {"obj": [], file: [], ...}``.

Two sources, tried in this order:
  1. the reference's cwd-relative layout, when present: names from listing
     ``netlib_mps/`` and tensors from ``dataset/netlib_mps_norm/<name>_{basis,coefs,rhs}.npy``,
     ``<name>_constrs.npz`` (exactly the reference's file naming);
  2. the packed fixture ``data/netlib_norm.npz`` shipped with this repository
     (made by ``tools/pack_netlib.py`` from those same files; 97 instances).
The reference iterates in ``os.listdir`` order, which is filesystem dependent
(SURVEY.md §7 "os.listdir order"); this build always sorts names.
"""
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

_PACK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data", "netlib_norm.npz")

SUBSET5 = ["adlittle.mps", "afiro.mps", "blend.mps", "kb2.mps", "sc50a.mps"]  # BASELINE.json configs[1]


@dataclass
class LPInstance:
    """One LP in the layout the kernels consume: CSR of A (m x n), c, b, labels."""
    name: str
    indptr: np.ndarray    # (m+1,) int64, local
    indices: np.ndarray   # (nnz,) int32 local column ids, sorted within a row
    values: np.ndarray    # (nnz,) float64
    coefs: np.ndarray     # (n,) float64
    rhs: np.ndarray       # (m,) float64
    basis: np.ndarray     # (n,) int32 in {0,1}

    @property
    def m(self):
        return int(self.rhs.shape[0])

    @property
    def n(self):
        return int(self.coefs.shape[0])

    @property
    def nnz(self):
        return int(self.values.shape[0])

    def as_reference_tuple(self):
        """The 6-tuple of reference linear_program_data.py:78."""
        constrs = np.split(self.indices, self.indptr)[1:-1]
        return (self.name, constrs, self.values, self.coefs, self.rhs, self.basis)

    @staticmethod
    def from_reference_tuple(t):
        name, constrs, weights, coefs, rhs, basis = t
        m = len(constrs)
        indptr = np.zeros(m + 1, dtype=np.int64)
        if m:
            indptr[1:] = np.cumsum([len(r) for r in constrs])
        indices = (np.concatenate([np.asarray(r, dtype=np.int32) for r in constrs])
                   if m and indptr[-1] > 0 else np.zeros(0, np.int32))
        return LPInstance(str(name), indptr, indices.astype(np.int32),
                          np.asarray(weights, dtype=np.float64),
                          np.asarray(coefs, dtype=np.float64),
                          np.asarray(rhs, dtype=np.float64),
                          np.asarray(basis, dtype=np.int32))


def pack_path():
    return os.path.abspath(_PACK)


def load_packed(names: Optional[Sequence[str]] = None, path: Optional[str] = None) -> List[LPInstance]:
    """Instances from the packed fixture, in sorted-name order (or the order of `names`)."""
    z = np.load(path or _PACK, allow_pickle=False)
    all_names = [str(s) for s in z["names"]]
    m, n, nnz = z["m"], z["n"], z["nnz"]
    ptr_off = np.concatenate([[0], np.cumsum(m + 1)])
    nnz_off = np.concatenate([[0], np.cumsum(nnz)])
    n_off = np.concatenate([[0], np.cumsum(n)])
    m_off = np.concatenate([[0], np.cumsum(m)])
    want = list(all_names) if names is None else [s if s.endswith(".mps") else s + ".mps" for s in names]
    out = []
    for s in want:
        if s not in all_names:
            raise KeyError(f"instance {s!r} not in packed Netlib fixture")
        i = all_names.index(s)
        out.append(LPInstance(
            s,
            z["indptr"][ptr_off[i]:ptr_off[i + 1]].astype(np.int64),
            z["indices"][nnz_off[i]:nnz_off[i + 1]].astype(np.int32),
            z["values"][nnz_off[i]:nnz_off[i + 1]].astype(np.float64),
            z["coefs"][n_off[i]:n_off[i + 1]].astype(np.float64),
            z["rhs"][m_off[i]:m_off[i + 1]].astype(np.float64),
            z["basis"][n_off[i]:n_off[i + 1]].astype(np.int32)))
    return out


def _load_reference_layout(normalize=True) -> List[LPInstance]:
    import scipy.sparse
    files = sorted(os.listdir("netlib_mps"))
    folder = "dataset/netlib_mps_norm/" if normalize else "dataset/netlib_mps/"
    out = []
    for f in files:
        basis = np.load(folder + f + "_basis.npy")
        coefs = np.load(folder + f + "_coefs.npy")
        rhs = np.load(folder + f + "_rhs.npy")
        sp = scipy.sparse.load_npz(folder + f + "_constrs.npz").tocsr()
        sp.sort_indices()
        out.append(LPInstance(f, sp.indptr.astype(np.int64), sp.indices.astype(np.int32),
                              sp.data.astype(np.float64), coefs.astype(np.float64),
                              rhs.astype(np.float64), basis.astype(np.int32)))
    return out


def load_instances(names: Optional[Sequence[str]] = None, normalize=True) -> List[LPInstance]:
    if os.path.isdir("netlib_mps") and os.path.isdir("dataset/netlib_mps_norm" if normalize else "dataset/netlib_mps"):
        inst = _load_reference_layout(normalize)
        if names is not None:
            by = {i.name: i for i in inst}
            inst = [by[s if s.endswith(".mps") else s + ".mps"] for s in names]
        return inst
    if not normalize:
        raise FileNotFoundError("un-normalized Netlib tensors are only available in the reference layout "
                                "(dataset/netlib_mps/); the packed fixture holds the normalized set")
    return load_packed(names)


def get_netlib_dataset(normalize=True, names: Optional[Sequence[str]] = None):
    """Drop-in for reference linear_program_data.py:58-80."""
    dataset, train_dict = [], {"obj": []}
    for inst in load_instances(names, normalize):
        dataset.append(inst.as_reference_tuple())
        train_dict[inst.name] = []
    return dataset, train_dict


def get_netlib_dataset_dense(normalize=True, names: Optional[Sequence[str]] = None):
    """Drop-in for reference linear_program_data.py:22-55 (the `angleNet` method): ONE instance -- the reference stops
    after the first entry of `os.listdir`, whose order is arbitrary; here the first instance of the (sorted, or named)
    dataset -- as (name, Q of [A | b]^T, coefs with a 0 appended, basis)."""
    from .angle import dense_instance_tensors
    dataset, train_dict = [], {"obj": []}
    for inst in load_instances(names, normalize)[:1]:
        Q, coefs, basis = dense_instance_tensors(inst)
        print("Instance {} size: {}".format(1, (inst.m, inst.n)))
        print(Q.shape)
        dataset.append((inst.name, Q, coefs, basis))
        train_dict[inst.name] = []
    return dataset, train_dict


# ----------------------------------------------------------------------------------------------
# synthetic LPs (BASELINE.json configs[3]/[4]; generator spec in SURVEY.md §8d)
# ----------------------------------------------------------------------------------------------
def synthetic_instance(seed: int, m: int = 10000, n: int = 20000, mean_row_nnz: float = 200.0,
                       max_row_nnz: int = 2000) -> LPInstance:
    """Host (numpy) generator for small synthetic LPs; statistics follow normalized Netlib:
    row nnz ~ Poisson(mean) clipped to [1, max], distinct sorted uniform column ids, values N(0,1)
    scaled to unit row 2-norm, coefs N(0,1) with 45% zeros then unit norm, rhs 0 w.p. 0.73 else
    U(0,5), labels Bernoulli(0.37)."""
    rng = np.random.default_rng(seed)
    cnt = np.clip(rng.poisson(mean_row_nnz, size=m), 1, min(max_row_nnz, n)).astype(np.int64)
    indptr = np.zeros(m + 1, dtype=np.int64)
    indptr[1:] = np.cumsum(cnt)
    indices = np.empty(indptr[-1], dtype=np.int32)
    values = rng.standard_normal(indptr[-1])
    for r in range(m):
        cols = rng.choice(n, size=cnt[r], replace=False)
        cols.sort()
        indices[indptr[r]:indptr[r + 1]] = cols
        seg = values[indptr[r]:indptr[r + 1]]
        seg /= max(np.linalg.norm(seg), 1e-12)
    coefs = rng.standard_normal(n)
    coefs[rng.random(n) < 0.45] = 0.0
    coefs /= max(np.linalg.norm(coefs), 1e-12)
    rhs = np.where(rng.random(m) < 0.73, 0.0, rng.random(m) * 5.0)
    basis = (rng.random(n) < 0.37).astype(np.int32)
    return LPInstance(f"synth{seed}", indptr, indices, values, coefs, rhs, basis)
