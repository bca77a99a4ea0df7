"""ORACLE for SURVEY.md section 8f-2 (test infrastructure -- never imported by the product path).

MPS file -> the tensors `get_netlib_dataset` loads (reference linear_program_data.py:58-80 reads
`dataset/netlib_mps_norm/<name>_{constrs.npz,coefs.npy,rhs.npy,basis.npy}`; the script that produced them from
`netlib_mps/*.mps` is NOT in the reference).  The conversion below was recovered from the data and is PINNED by the
reference's own files: for all 97 instances listed by `netlib_mps/` it reproduces
  * `dataset/netlib_mps/<name>_{constrs,coefs,rhs}` (the raw stage) exactly,
  * `dataset/netlib_mps_norm/<name>_{constrs,coefs,rhs}` to 1e-12, and
  * `dataset/netlib_mps_norm/<name>_basis` exactly from the raw stage's solver outputs `_v.npy` / `_c.npy`,
checked by tests/test_mps.py::test_oracle_reproduces_every_reference_tensor whenever /root/reference is present
(it is in the build container); five of those MPS files are committed as fixtures for everywhere else
(tests/golden/mps/*.mps against data/netlib_norm.npz, which is byte-identical to the reference's tensors).

Rules (numpy / plain Python; pure restatement, no solver):
  parse   sections ROWS / COLUMNS / RHS / RANGES / BOUNDS / ENDATA (whitespace tokens; an RHS or RANGES line with an
          even token count has a blank set name); the first N row is the objective; BOUNDS and OBJSENSE are read and
          ignored (the reference's tensors carry no bounds); rows keep the ROWS order, columns the order of first
          appearance in COLUMNS; an RHS entry of the objective row is dropped.
  raw     A (m x n), c, b; every RANGES entry of a constraint row appends a column with one entry in that row:
          +1 for an L row, -1 for a G row, sign(R) for an E row.
  norm    every L / G row WITHOUT a range gets a slack column (+1 / -1) behind all others, in row order;
          each row is scaled by s = 1 / ||row incl. slack||_2, and where |b s| > 5 by s = 5 / b instead (signed: such a
          row ends with right-hand side +5); c is scaled to unit 2-norm and zero-padded for the slacks.
  labels  basis = [v ; c[rows with a slack]] from the solver outputs (fixtures: there is no LP solver here).
"""
import numpy as np
import scipy.sparse as sp


def parse_mps(path):
    sec, rows, rtype, obj, cols, colidx = None, [], {}, None, [], {}
    ent, rhs, ranges, bounds, objsense = [], {}, {}, [], None
    with open(path) as fh:
        for line in fh:
            if not line.strip() or line[0] == "*":
                continue
            if line[0] != " ":
                sec = line.split()[0]
                continue
            f = line.split()
            if sec == "ROWS":
                t, nm = f[0], f[1]
                rtype[nm] = t
                if t == "N":
                    if obj is None:
                        obj = nm
                else:
                    rows.append(nm)
            elif sec == "COLUMNS":
                if "'MARKER'" in f:
                    continue
                c = f[0]
                if c not in colidx:
                    colidx[c] = len(cols)
                    cols.append(c)
                for k in range(1, len(f) - 1, 2):
                    ent.append((f[k], c, float(f[k + 1])))
            elif sec in ("RHS", "RANGES"):
                o = 1 if len(f) % 2 == 1 else 0
                tgt = rhs if sec == "RHS" else ranges
                for k in range(o, len(f) - 1, 2):
                    tgt[f[k]] = float(f[k + 1])
            elif sec == "BOUNDS":
                bounds.append(f)
            elif sec == "OBJSENSE":
                objsense = f[0]
    return dict(rows=rows, rtype=rtype, obj=obj, cols=cols, colidx=colidx, ent=ent, rhs=rhs, ranges=ranges,
                bounds=bounds, objsense=objsense)


def raw_tensors(p):
    """(A csr m x n', c, b): the `dataset/netlib_mps` stage, range columns included."""
    ridx = {r: i for i, r in enumerate(p["rows"])}
    m, n = len(p["rows"]), len(p["cols"])
    A = sp.lil_matrix((m, n))
    c = np.zeros(n)
    for r, cname, v in p["ent"]:
        j = p["colidx"][cname]
        if r == p["obj"]:
            c[j] = v
        elif r in ridx:
            A[ridx[r], j] = v
    b = np.zeros(m)
    for r, v in p["rhs"].items():
        if r in ridx:
            b[ridx[r]] = v
    extra = []
    for r, v in p["ranges"].items():
        i = ridx.get(r)
        if i is None:
            continue
        t = p["rtype"][r]
        extra.append((i, 1.0 if t == "L" else (-1.0 if t == "G" else (1.0 if v >= 0 else -1.0))))
    A = A.tocsr()
    if extra:
        E = sp.lil_matrix((m, len(extra)))
        for k, (i, s) in enumerate(extra):
            E[i, k] = s
        A = sp.hstack([A, E.tocsr()]).tocsr()
        c = np.concatenate([c, np.zeros(len(extra))])
    return A, c, b


def slack_rows(p):
    """Row ids (in row order) that receive a slack column in the normalized stage, and the slack's sign."""
    out = []
    for i, r in enumerate(p["rows"]):
        if p["rtype"][r] in "LG" and r not in p["ranges"]:
            out.append((i, 1.0 if p["rtype"][r] == "L" else -1.0))
    return out


def normalize(p, A, c, b):
    """The `dataset/netlib_mps_norm` stage: (A' csr m x (n' + slacks), c', b')."""
    m = A.shape[0]
    sl = slack_rows(p)
    S = sp.lil_matrix((m, len(sl)))
    for k, (i, sg) in enumerate(sl):
        S[i, k] = sg
    B = sp.hstack([A, S.tocsr()]).tocsr()
    nrm = np.sqrt(np.asarray(B.multiply(B).sum(1)).ravel())
    s = np.where(nrm > 0, 1.0 / np.where(nrm > 0, nrm, 1.0), 1.0)
    over = np.abs(b * s) > 5
    s = np.where(over, 5.0 / np.where(over, b, 1.0), s)
    B = (sp.diags(s) @ B).tocsr()
    B.sort_indices()
    cn = np.linalg.norm(c)
    return B, np.concatenate([c / (cn if cn > 0 else 1.0), np.zeros(len(sl))]), b * s


def basis_labels(p, v, cstat):
    """basis = [v ; c[rows with a slack]] (solver outputs `_v.npy`, `_c.npy` of the raw stage)."""
    rows = [i for i, _ in slack_rows(p)]
    return np.concatenate([np.asarray(v), np.asarray(cstat)[rows]]).astype(np.int32)


def mps_to_normalized(path):
    p = parse_mps(path)
    A, c, b = raw_tensors(p)
    return normalize(p, A, c, b) + (p,)
