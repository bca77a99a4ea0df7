"""ORACLE #2 (test infrastructure -- never imported by the product path).

numpy restatement of the SAME layer as `oracle/pyg_restatement.py::transformer_conv`, but in the
refactored "attention-weighted CSR SpMM + per-node GEMM" form that the HIP kernels implement
(SURVEY.md Appendix A.3 forward, A.4 backward), with a hand-derived backward.  It exists so that
three independent implementations can be compared: (1) the literal edge-list form under torch
autograd, (2) this file, (3) the HIP kernels.  (1) == (2) is checked in tests/test_oracle.py on
CPU; (2) mirrors the kernel decomposition one to one, so a GPU mismatch can be localised to a
single stage (every intermediate the kernels write is returned here under the same name).

PARITY UNPINNED with respect to PyG itself -- see the header of pyg_restatement.py.

Reference call sites restated: GNNModel.forward, linear_program_methods.py:238-251;
BCEWithLogitsLoss, linear_program_experiment.py:41,139-141.
"""
import numpy as np
import scipy.sparse as sp

from .pyg_restatement import CONV_CIN, state_dict_spec

FEAT = 16


class BatchCSR:
    """Block-diagonal batch of LP constraint matrices, both orientations.

    csr: rows = constraints (M), columns = variables (N); csc = the transpose's CSR.
    Variable ids are offset by the sum of previous n, constraint ids by the sum of previous m
    (BipartiteData.__inc__, reference linear_program_methods.py:68-70)."""

    def __init__(self, instances):
        rp, ci, va, n_off, m_off, e_off = [np.zeros(1, np.int64)], [], [], 0, 0, 0
        self.inst_m, self.inst_n = [], []
        for it in instances:
            rp.append(it.indptr[1:] + e_off)
            ci.append(it.indices.astype(np.int64) + n_off)
            va.append(it.values)
            n_off += it.n
            m_off += it.m
            e_off += it.nnz
            self.inst_m.append(it.m)
            self.inst_n.append(it.n)
        self.M, self.N, self.nnz = m_off, n_off, e_off
        self.rp = np.concatenate(rp).astype(np.int64)
        self.ci = np.concatenate(ci) if ci else np.zeros(0, np.int64)
        self.va = np.concatenate(va).astype(np.float64) if va else np.zeros(0)
        A = sp.csr_matrix((self.va, self.ci, self.rp), shape=(self.M, self.N))
        At = A.T.tocsr()
        At.sort_indices()
        self.cp, self.ri, self.cv = At.indptr.astype(np.int64), At.indices.astype(np.int64), At.data.astype(np.float64)
        self.x1 = np.concatenate([it.coefs for it in instances]) if instances else np.zeros(0)
        self.x2 = np.concatenate([it.rhs for it in instances]) if instances else np.zeros(0)
        self.basis = np.concatenate([it.basis for it in instances]).astype(np.float64) if instances else np.zeros(0)
        # per-variable loss weight 1 / (n_k * B): sum over instances of mean BCE, divided by B
        B = max(len(instances), 1)
        self.wnode = np.concatenate([np.full(it.n, 1.0 / (it.n * B)) for it in instances]) if instances else np.zeros(0)

    def orient(self, dst_is_var):
        """(ptr, idx, val, n_dst, n_src) for dst-major traversal."""
        if dst_is_var:      # w2s convs: destination = variables -> traverse A^T
            return self.cp, self.ri, self.cv, self.N, self.M
        return self.rp, self.ci, self.va, self.M, self.N


def spmm(ptr, idx, val, H):
    """Plain CSR SpMM  Y[r,:] = sum_e val[e] * H[idx[e],:]  (the roofline kernel's oracle)."""
    n_rows = len(ptr) - 1
    A = sp.csr_matrix((val, idx, ptr), shape=(n_rows, H.shape[0]))
    return A @ H


def _seg_matrix(ptr):
    n_rows, nnz = len(ptr) - 1, int(ptr[-1])
    rows = np.repeat(np.arange(n_rows), np.diff(ptr))
    return sp.csr_matrix((np.ones(nnz), (rows, np.arange(nnz))), shape=(n_rows, nnz)), rows


def conv_params(sd, prefix):
    g = lambda k: np.asarray(sd[f"{prefix}.{k}"], dtype=np.float64)
    return dict(Wk=g("lin_key.weight"), bk=g("lin_key.bias"), Wq=g("lin_query.weight"), bq=g("lin_query.bias"),
                Wv=g("lin_value.weight"), bv=g("lin_value.bias"), we=g("lin_edge.weight")[:, 0],
                Ws=g("lin_skip.weight"), bs=g("lin_skip.bias"))


def derive(p):
    """Per-step folded weights (the `param_prep` kernel): logits need only q' = Wk^T q / 4 and
    t = <q, w_e> / 4; the key-bias term <q, bk> is constant per destination and cancels in softmax."""
    return dict(Pq=p["Wk"].T @ p["Wq"] / 4.0,      # (Cs, Cd)
                pq0=p["Wk"].T @ p["bq"] / 4.0,     # (Cs,)
                Pt=p["Wq"].T @ p["we"] / 4.0,      # (Cd,)
                pt0=float(p["bq"] @ p["we"]) / 4.0,
                Pb=p["Wq"].T @ p["bk"] / 4.0)      # (Cd,)  backward only


def conv_fwd(p, ptr, idx, val, X_src, x_dst):
    d = derive(p)
    S_mat, rows = _seg_matrix(ptr)
    qp = x_dst @ d["Pq"].T + d["pq0"]                       # (Nd, Cs)
    t = x_dst @ d["Pt"] + d["pt0"]                          # (Nd,)
    Xe = X_src[idx]                                         # (E, Cs)
    l = np.einsum("ec,ec->e", qp[rows], Xe) + val * t[rows]
    n_dst = len(ptr) - 1
    mx = np.full(n_dst, -np.inf)
    np.maximum.at(mx, rows, l)
    deg = np.diff(ptr)
    mx_safe = np.where(deg > 0, mx, 0.0)
    pe = np.exp(l - mx_safe[rows])
    L = S_mat @ pe
    rinv = 1.0 / (L + 1e-16)
    Z = (S_mat @ (pe[:, None] * Xe)) * rinv[:, None]
    u = (S_mat @ (pe * val)) * rinv
    S = L * rinv
    o = Z @ p["Wv"].T + S[:, None] * p["bv"] + u[:, None] * p["we"] + x_dst @ p["Ws"].T + p["bs"]
    h = np.maximum(o, 0.0)
    return h, dict(qp=qp, t=t, mx=mx_safe, rinv=rinv, Z=Z, u=u, S=S, h=h, alpha=pe * rinv[rows])


def conv_bwd(p, ptr, idx, val, X_src, x_dst, saved, dh, need_input_grads=True):
    """Returns (param-grad dict, dx_dst, dX_src, intermediates)."""
    d = derive(p)
    S_mat, rows = _seg_matrix(ptr)
    g = dh * (saved["h"] > 0)
    gv = g @ p["Wv"]                                        # (Nd, Cs)
    ge = g @ p["we"]
    gb = g @ p["bv"]
    D = np.einsum("nc,nc->n", gv, saved["Z"]) + gb * saved["S"] + ge * saved["u"]
    c = gb - D
    Xe = X_src[idx]
    l = np.einsum("ec,ec->e", saved["qp"][rows], Xe) + val * saved["t"][rows]
    alpha = np.exp(l - saved["mx"][rows]) * saved["rinv"][rows]
    dl = alpha * (np.einsum("ec,ec->e", gv[rows], Xe) + val * ge[rows] + c[rows])
    dqp = S_mat @ (dl[:, None] * Xe)                        # (Nd, Cs)
    ds = S_mat @ dl
    dt = S_mat @ (dl * val)
    dx_dst = dX_src = None
    if need_input_grads:
        dx_dst = g @ p["Ws"] + dqp @ d["Pq"] + ds[:, None] * d["Pb"] + dt[:, None] * d["Pt"]
        contrib = alpha[:, None] * gv[rows] + dl[:, None] * saved["qp"][rows]
        dX_src = np.zeros_like(X_src)
        np.add.at(dX_src, idx, contrib)
    # statistics the `param_stats` kernel reduces over nodes
    A_gx, A_gZ = g.T @ x_dst, g.T @ saved["Z"]
    s_g, s_gS, s_gu = g.sum(0), (g * saved["S"][:, None]).sum(0), (g * saved["u"][:, None]).sum(0)
    A_dx, s_dqp = dqp.T @ x_dst, dqp.sum(0)
    v_ds, v_dt, s_ds, s_dt = ds @ x_dst, dt @ x_dst, ds.sum(), dt.sum()
    grads = {
        "lin_skip.weight": A_gx, "lin_skip.bias": s_g,
        "lin_value.weight": A_gZ, "lin_value.bias": s_gS,
        "lin_key.weight": (p["Wq"] @ A_dx.T + np.outer(p["bq"], s_dqp)) / 4.0,
        "lin_key.bias": (p["Wq"] @ v_ds + p["bq"] * s_ds) / 4.0,
        "lin_edge.weight": (s_gu + (p["Wq"] @ v_dt + p["bq"] * s_dt) / 4.0)[:, None],
        "lin_query.weight": (p["Wk"] @ A_dx + np.outer(p["bk"], v_ds) + np.outer(p["we"], v_dt)) / 4.0,
        "lin_query.bias": (p["Wk"] @ s_dqp + p["bk"] * s_ds + p["we"] * s_dt) / 4.0,
    }
    return grads, dx_dst, dX_src, dict(g=g, gv=gv, ge=ge, c=c, dqp=dqp, ds=ds, dt=dt, alpha=alpha, dl=dl)


def gnn_forward_backward(sd, batch: BatchCSR, want_grads=True):
    """Whole model in the kernels' decomposition.  Returns dict(logits, loss, grads (flat, state_dict
    order), hidden...).  Loss = sum_i wnode_i * BCE(z_i, y_i)  ==  (1/B) sum_k mean-BCE(instance k)."""
    x1 = batch.x1[:, None]
    x2 = batch.x2[:, None]
    P = {name: conv_params(sd, name) for name in CONV_CIN}
    ov, oc = batch.orient(True), batch.orient(False)
    h1v, s1v = conv_fwd(P["gconv1_w2s"], *ov[:3], x2, x1)
    h1c, s1c = conv_fwd(P["gconv1_s2w"], *oc[:3], x1, x2)
    h2v, s2v = conv_fwd(P["gconv2_w2s"], *ov[:3], h1c, h1v)
    h2c, s2c = conv_fwd(P["gconv2_s2w"], *oc[:3], h1v, h1c)
    h3v, s3v = conv_fwd(P["gconv3_w2s"], *ov[:3], h2c, h2v)
    wfc = np.asarray(sd["fc.weight"], dtype=np.float64)[0]
    bfc = float(np.asarray(sd["fc.bias"], dtype=np.float64)[0])
    z = h3v @ wfc + bfc
    y = batch.basis
    bce = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    loss = float((batch.wnode * bce).sum())
    out = dict(logits=z, loss=loss, h1v=h1v, h1c=h1c, h2v=h2v, h2c=h2c, h3v=h3v,
               saved=dict(s1v=s1v, s1c=s1c, s2v=s2v, s2c=s2c, s3v=s3v))
    if not want_grads:
        return out
    dz = batch.wnode * (1.0 / (1.0 + np.exp(-z)) - y)
    G = {}
    G["fc.weight"] = (dz @ h3v)[None, :]
    G["fc.bias"] = np.array([dz.sum()])
    dh3v = dz[:, None] * wfc[None, :]
    g3, d_h2v, d_h2c, _ = conv_bwd(P["gconv3_w2s"], *ov[:3], h2c, h2v, s3v, dh3v)
    g2v, d_h1v_a, d_h1c_a, _ = conv_bwd(P["gconv2_w2s"], *ov[:3], h1c, h1v, s2v, d_h2v)
    g2c, d_h1c_b, d_h1v_b, _ = conv_bwd(P["gconv2_s2w"], *oc[:3], h1v, h1c, s2c, d_h2c)
    d_h1v, d_h1c = d_h1v_a + d_h1v_b, d_h1c_a + d_h1c_b
    g1v, _, _, _ = conv_bwd(P["gconv1_w2s"], *ov[:3], x2, x1, s1v, d_h1v, need_input_grads=False)
    g1c, _, _, _ = conv_bwd(P["gconv1_s2w"], *oc[:3], x1, x2, s1c, d_h1c, need_input_grads=False)
    for name, gd in (("gconv3_w2s", g3), ("gconv2_w2s", g2v), ("gconv2_s2w", g2c),
                     ("gconv1_w2s", g1v), ("gconv1_s2w", g1c)):
        for k, v in gd.items():
            G[f"{name}.{k}"] = v
    flat = []
    for key, shape in state_dict_spec():
        flat.append(np.asarray(G.get(key, np.zeros(shape)), dtype=np.float64).reshape(-1))
    out["grads"] = np.concatenate(flat)
    out["grad_dict"] = G
    out["d_hidden"] = dict(dh3v=dh3v, d_h2v=d_h2v, d_h2c=d_h2c, d_h1v=d_h1v, d_h1c=d_h1c)
    return out


def adam_step(params, grads, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad) on flat arrays; `step` is 1-based.
    reference linear_program_experiment.py:119,143."""
    m[:] = b1 * m + (1 - b1) * grads
    v[:] = b2 * v + (1 - b2) * grads * grads
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    params -= (lr / bc1) * m / denom
    return params
