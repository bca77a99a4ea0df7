"""MPS -> normalized tensors.  CPU only (the parser and the normalisation are host code of the
library).  Three layers: (1) the oracle (oracle/mps_norm.py) against every tensor the reference holds, when
/root/reference is present (it is in the build container; the GPU box has no reference); (2) the oracle and the
library against the five committed fixtures and the packed reference tensors (data/netlib_norm.npz, byte-identical to
the reference's files); (3) library == oracle on a synthetic file that exercises RANGES on L / G / E rows, a G row,
negative right-hand sides, blank set names, a right-hand side on the objective, Fortran exponents and a MARKER line."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from mllp_amd.data import SUBSET5, load_packed
from oracle import mps_norm as mn

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "mps")
REF = "/root/reference"

SYNTH = """NAME          SYNTH
ROWS
 N  COST
 L  R1
 G  R2
 E  R3
 G  R4
 L  R5
 E  R6
 N  FREE
COLUMNS
    MARKER                 'MARKER'                 'INTORG'
    X1        COST         1.0   R1           2.0
    X1        R2          -1.5   R4           1.0
    X2        COST        -2.0   R3           4.0
    X2        R5           0.5   R6          -0.25
    X3        R1           1.0D0 R4          -3.0
    X3        R5           8.0   FREE         9.0
    X4        R6           3.0
RHS
    RHS       R1          -4.0   R2           7.0
              R4          12.0   COST        -9.0
    RHS       R5        -900.0   R6         -60.0
RANGES
    RNG       R1           2.0   R2           3.0
    RNG       R3          -1.0
BOUNDS
 UP BND       X1           4.0
 FR BND       X2
ENDATA
"""


def _csr(inst):
    return sp.csr_matrix((inst.values, inst.indices, inst.indptr), shape=(inst.m, inst.n))


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "netlib_mps")), reason="the reference is not on this machine")
def test_oracle_reproduces_every_reference_tensor():
    """Pins the oracle: all 97 instances, raw stage exact, normalized stage to 1e-12, labels from the solver outputs."""
    from mllp_amd.mps import read_mps
    label_mismatch = []
    names = sorted(os.listdir(os.path.join(REF, "netlib_mps")))
    assert len(names) == 97
    for nm in names:
        p = mn.parse_mps(os.path.join(REF, "netlib_mps", nm))
        A, c, b = mn.raw_tensors(p)
        raw = os.path.join(REF, "dataset", "netlib_mps", nm)
        R = sp.load_npz(raw + "_constrs.npz")
        assert A.shape == R.shape and abs(A - R).max() == 0 and A.nnz == R.nnz, nm
        assert np.array_equal(c, np.load(raw + "_coefs.npy")) and np.array_equal(b, np.load(raw + "_rhs.npy")), nm
        B, cn, bn = mn.normalize(p, A, c, b)
        nrm = os.path.join(REF, "dataset", "netlib_mps_norm", nm)
        N = sp.load_npz(nrm + "_constrs.npz")
        assert B.shape == N.shape and B.nnz == N.nnz and abs(B - N).max() < 1e-12, nm
        assert np.abs(cn - np.load(nrm + "_coefs.npy")).max() < 1e-12, nm
        assert np.abs(bn - np.load(nrm + "_rhs.npy")).max() < 1e-12, nm
        v, cs = np.load(raw + "_v.npy"), np.load(raw + "_c.npy")
        if not np.array_equal(mn.basis_labels(p, v, cs), np.load(nrm + "_basis.npy")):
            label_mismatch.append(nm)
        # the library on the same file
        inst, info = read_mps(os.path.join(REF, "netlib_mps", nm), True, var_status=v, constr_status=cs)
        assert abs(_csr(inst) - N).max() < 1e-12 and inst.nnz == N.nnz, nm
        assert np.abs(inst.coefs - np.load(nrm + "_coefs.npy")).max() < 1e-12, nm
        assert np.abs(inst.rhs - np.load(nrm + "_rhs.npy")).max() < 1e-12, nm
        raw_inst, _ = read_mps(os.path.join(REF, "netlib_mps", nm), False)
        assert abs(_csr(raw_inst) - R).max() == 0 and np.array_equal(raw_inst.coefs, np.load(raw + "_coefs.npy")), nm
    # the reference's wood1p labels do not follow from its own _v / _c files (every other instance does)
    assert label_mismatch in ([], ["wood1p.mps"]), label_mismatch


@pytest.mark.parametrize("name", SUBSET5)
def test_fixture_mps_files_give_the_packed_reference_tensors(name):
    """tests/golden/mps/<name> (copies of the reference's netlib_mps data files) -> oracle and library -> the
    reference's normalized tensors as packed in data/netlib_norm.npz."""
    from mllp_amd.mps import read_mps
    want = load_packed([name])[0]
    W = _csr(want)
    B, cn, bn, p = mn.mps_to_normalized(os.path.join(FIX, name))
    assert B.shape == W.shape and B.nnz == W.nnz and abs(B - W).max() < 1e-12
    assert np.abs(cn - want.coefs).max() < 1e-12 and np.abs(bn - want.rhs).max() < 1e-12
    inst, info = read_mps(os.path.join(FIX, name), True)
    assert inst.name == name and inst.m == want.m and inst.n == want.n
    assert np.array_equal(inst.indptr, want.indptr) and np.array_equal(inst.indices, want.indices)
    assert np.abs(inst.values - want.values).max() < 1e-12
    assert np.abs(inst.coefs - want.coefs).max() < 1e-12 and np.abs(inst.rhs - want.rhs).max() < 1e-12
    assert info["n_struct"] + info["n_range"] + info["n_slack"] == want.n
    assert [int(i) for i in info["slack_rows"]] == [i for i, _ in mn.slack_rows(p)]


def test_library_equals_oracle_on_every_branch(tmp_path):
    from mllp_amd import _lib
    from mllp_amd.mps import convert_directory, read_mps
    f = tmp_path / "synth.mps"
    f.write_text(SYNTH.replace("1.0D0", "1.0"))           # (python's float() has no Fortran exponent)
    p = mn.parse_mps(str(f))
    A, c, b = mn.raw_tensors(p)
    B, cn, bn = mn.normalize(p, A, c, b)
    f.write_text(SYNTH)
    raw, info = read_mps(str(f), False)
    assert (raw.m, raw.n) == (6, 7) and info["n_struct"] == 4 and info["n_range"] == 3 and info["n_slack"] == 0
    assert abs(_csr(raw) - A).max() == 0 and np.array_equal(raw.coefs, c) and np.array_equal(raw.rhs, b)
    assert raw.rhs.tolist() == [-4.0, 7.0, 0.0, 12.0, -900.0, -60.0]        # the objective's RHS entry is dropped
    assert _csr(raw).toarray()[:3, 4:].tolist() == [[1, 0, 0], [0, -1, 0], [0, 0, -1]]   # L: +1, G: -1, E with R < 0: -1
    nrm, info = read_mps(str(f), True, var_status=[1, 0, 1, 0, 0, 1, 0], constr_status=[0, 1, 0, 1, 1, 0])
    assert info["n_slack"] == 2 and info["slack_rows"].tolist() == [3, 4]     # ranged rows get no slack
    assert abs(_csr(nrm) - B).max() < 1e-15 and np.abs(nrm.coefs - cn).max() < 1e-15 and np.abs(nrm.rhs - bn).max() < 1e-15
    assert nrm.rhs[4] == 5.0 and nrm.rhs[5] == 5.0 and _csr(nrm)[4, 1] < 0   # capped rows flip with a negative rhs
    assert abs(np.linalg.norm(_csr(nrm)[0].toarray()) - 1.0) < 1e-15
    assert nrm.basis.tolist() == [1, 0, 1, 0, 0, 1, 0, 1, 1]
    with pytest.raises(ValueError):
        read_mps(str(f), True, var_status=[1, 0], constr_status=[0] * 6)
    with pytest.raises(_lib.MllpError, match="cannot open"):
        read_mps(str(tmp_path / "missing.mps"))
    (tmp_path / "bad.mps").write_text("NAME X\nROWS\n N  COST\n L  R1\nCOLUMNS\n    X1        R1           abc\nENDATA\n")
    with pytest.raises(_lib.MllpError, match="bad number"):
        read_mps(str(tmp_path / "bad.mps"))
    # the reference's on-disk layout (linear_program_data.py:65-75) from a directory of MPS files
    d = tmp_path / "mps"; d.mkdir()
    (d / "synth.mps").write_text(SYNTH)
    names = convert_directory(str(d), str(tmp_path / "out"))
    assert names == ["synth.mps"]
    got = sp.load_npz(tmp_path / "out" / "synth.mps_constrs.npz")
    assert got.format == "csr" and abs(got - B).max() < 1e-15
    assert np.abs(np.load(tmp_path / "out" / "synth.mps_coefs.npy") - cn).max() < 1e-15
