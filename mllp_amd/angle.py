"""the `angleNet` method of the reference -- `AngleModel` on the complete "angle" graph of one
LP instance.

Reference surface mirrored here (same names, arguments and outputs):
  get_netlib_dataset_dense   linear_program_data.py:22-55      (in mllp_amd/data.py, which imports the work from here)
  build_graph_from_Q_sets    linear_program_methods.py:119-130
  get_netlib_dataloader      linear_program_methods.py:111-117
  AngleModel                 linear_program_methods.py:187-200
  training loop              linear_program_experiment.py:81-114  (mllp_amd/experiment.py::train_angle)

The reference builds N (N - 1) PyG edges with one Python call of `cosine_similarity` per edge and runs three
TransformerConv layers over them.  On a complete graph that is dense attention with a scalar edge bias, so the graph is
kept as the dense [N, N] cosine matrix on the device and the model runs in libmllp_hip.so (mllp_amd/csrc/angle.hip:
flash-attention-style kernels and GEMMs on the fp32 matrix cores, hand-derived backward that recomputes the attention
weights; no N x N intermediate, no BLAS library).  feat_dim must be 16, 32, 64, 128 or 256.  There is no CPU path.
"""
import ctypes
from ctypes import c_int64
from typing import List

import numpy as np
import torch

from . import _lib
from .model import _TConvParams


def dense_instance_tensors(inst):
    """linear_program_data.py:34-49 for one instance: Q of [A | b]^T, coefs with a 0 appended, basis."""
    m, n = inst.m, inst.n
    A = np.zeros((m, n + 1), dtype=np.float64)
    rows = np.repeat(np.arange(m), np.diff(inst.indptr))
    A[rows, inst.indices] = inst.values
    A[:, n] = inst.rhs
    Q, _ = np.linalg.qr(A.T)                                  # [(n + 1), min(n + 1, m)]
    coefs = np.concatenate([np.asarray(inst.coefs, np.float64), np.array([0.0])])
    return np.array(Q), coefs, np.asarray(inst.basis)


def cosine_matrix(Q):
    """edge_attr of build_graph_from_Q_sets as a dense matrix: cos[i, j] = cosine_similarity(Q[i], Q[j]) with the
    reference's guard (0 when either norm <= 1e-6, linear_program_methods.py:105-108)."""
    Q = np.asarray(Q, dtype=np.float64)
    nrm = np.linalg.norm(Q, axis=1)
    ok = nrm > 1e-6
    Qn = np.where(ok[:, None], Q / np.where(ok, nrm, 1.0)[:, None], 0.0)
    cos = Qn @ Qn.T
    return 0.5 * (cos + cos.T), nrm       # exactly symmetric (the reference's per-edge dot products are): the kernels read A[j][i] for A[i][j]


class AngleGraph:
    """What the reference's `pyg.data.Data(x, edge_index, edge_attr, name, basis_opt, basis_num, var_num)` carries,
    with the complete graph held as its dense cosine matrix.  `edge_index` / `edge_attr` materialise the PyG edge list
    on demand (N (N - 1) edges: tests and small graphs only)."""

    def __init__(self, x, cos, name, basis_opt, basis_num, var_num):
        self.x, self.cos = x, cos
        self.name, self.basis_opt, self.basis_num, self.var_num = name, basis_opt, basis_num, var_num
        self._ws = None
        self._token = 0

    @property
    def num_nodes(self):
        return int(self.x.shape[0])

    @property
    def edge_index(self):
        N = self.num_nodes
        src, dst = np.where(np.ones((N, N)) - np.eye(N))      # the reference's np.where(ones - diag)
        return torch.tensor(np.stack([src, dst]), dtype=torch.long, device=self.x.device)

    @property
    def edge_attr(self):
        ei = self.edge_index
        # edge (src -> dst): attribute cos(Q[src], Q[dst]); the matrix is symmetric, row = target
        return self.cos[ei[1], ei[0]].unsqueeze(-1)

    def workspace(self, feat_dim):
        n = c_int64()
        _lib.check(_lib.lib().mllp_angle_workspace_floats(self.num_nodes, int(feat_dim), ctypes.byref(n)))
        if self._ws is None or self._ws.numel() != n.value:
            self._ws = torch.empty(n.value, dtype=torch.float32, device=self.x.device)
        return self._ws


def build_graph_from_Q_sets(Q, coefs, device, name, basis_opt):
    """reference linear_program_methods.py:119-130 (same arguments)."""
    cos, nrm = cosine_matrix(Q)
    x = torch.tensor(np.stack([np.asarray(coefs, np.float64), nrm], axis=1), dtype=torch.float, device=device)
    cos_t = torch.tensor(cos, dtype=torch.float, device=device).contiguous()
    var_num = x.shape[0]
    return AngleGraph(x, cos_t, name, basis_opt, int(np.asarray(Q).shape[1]), var_num - 1)


def get_netlib_dataloader(train_dataset, device) -> List[AngleGraph]:
    """reference :111-117: a DataLoader(batch_size=1) over the graphs -- here simply the list of graphs."""
    out = []
    for name, constr_Q, coefs, basis_opt in train_dataset:
        print("basis opt size", np.asarray(basis_opt).shape)
        out.append(build_graph_from_Q_sets(constr_Q, coefs, device, name, basis_opt))
    return out


class _AngleFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, g, feat_dim):
        flat = flat.contiguous()
        N = g.num_nodes
        ws = g.workspace(feat_dim)
        logits = torch.empty(N - 1, dtype=torch.float32, device=flat.device)
        _lib.check(_lib.lib().mllp_angle_forward(N, int(feat_dim), _lib.ptr(g.cos), _lib.ptr(g.x), _lib.ptr(flat),
                                                 _lib.ptr(ws), _lib.ptr(logits), _lib.current_stream()))
        g._token += 1
        ctx.g, ctx.token, ctx.feat_dim = g, g._token, int(feat_dim)
        ctx.save_for_backward(flat)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        (flat,) = ctx.saved_tensors
        g = ctx.g
        if g._token != ctx.token:
            raise RuntimeError("AngleModel: backward after another forward on the same graph (the saved activations "
                               "live in the graph's workspace)")
        grads = torch.empty_like(flat)
        d = dlogits.contiguous().float()
        _lib.check(_lib.lib().mllp_angle_backward(g.num_nodes, ctx.feat_dim, _lib.ptr(g.cos), _lib.ptr(g.x),
                                                  _lib.ptr(flat), _lib.ptr(g.workspace(ctx.feat_dim)), _lib.ptr(d),
                                                  _lib.ptr(grads), _lib.current_stream()))
        return grads, None, None


class AngleStepper:
    """The reference's training step for `AngleModel` (linear_program_experiment.py:88-96: BCEWithLogitsLoss, backward,
    Adam) on FLAT parameters without autograd: forward and backward through the C ABI, the library's Adam kernel
    (trainer.FlatAdam = torch.optim.Adam's arithmetic).  Same numbers as the module + torch.optim loop (tests), about a
    dozen small torch kernels fewer per step."""

    def __init__(self, model: "AngleModel", lr=1e-3):
        from .trainer import FlatAdam
        self.feat_dim = model.feat_dim
        self.params = model.flat_parameters().detach().clone().float().contiguous()
        self.opt = FlatAdam(self.params, lr=lr)
        self.grads = torch.empty_like(self.params)

    def step(self, g: "AngleGraph", y: torch.Tensor):
        """One Adam step on graph `g` with labels y [N - 1]; returns (loss [1], logits [N - 1])."""
        L, N = _lib.lib(), g.num_nodes
        ws = g.workspace(self.feat_dim)
        logits = torch.empty(N - 1, dtype=torch.float32, device=self.params.device)
        _lib.check(L.mllp_angle_forward(N, self.feat_dim, _lib.ptr(g.cos), _lib.ptr(g.x), _lib.ptr(self.params), _lib.ptr(ws),
                                        _lib.ptr(logits), _lib.current_stream()))
        g._token += 1
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, y)
        dlogits = (torch.sigmoid(logits) - y) / float(N - 1)             # d mean-BCE / d logits
        _lib.check(L.mllp_angle_backward(N, self.feat_dim, _lib.ptr(g.cos), _lib.ptr(g.x), _lib.ptr(self.params), _lib.ptr(ws),
                                         _lib.ptr(dlogits), _lib.ptr(self.grads), _lib.current_stream()))
        self.opt.step(self.grads)
        return loss, logits


class AngleModel(torch.nn.Module):
    """reference linear_program_methods.py:187-200: gconv1 = TransformerConv(2, F, edge_dim=1), gconv2 and gconv3 =
    TransformerConv(F, F, edge_dim=1), fc = Linear(F, 1); forward applies gconv1, gconv2, gconv2 (gconv3 is never
    called) with ReLU, then fc, and returns all nodes but the last."""

    def __init__(self, feat_dim=16):
        super().__init__()
        self.feat_dim = int(feat_dim)
        self.gconv1 = _TConvParams(2, self.feat_dim)
        self.gconv2 = _TConvParams(self.feat_dim, self.feat_dim)
        self.gconv3 = _TConvParams(self.feat_dim, self.feat_dim)
        self.fc = torch.nn.Linear(self.feat_dim, 1)

    def flat_parameters(self):
        return torch.cat([p.reshape(-1) for p in self.parameters()])

    def load_flat(self, flat):
        off = 0
        with torch.no_grad():
            for p in self.parameters():
                k = p.numel()
                p.copy_(torch.as_tensor(flat[off:off + k]).reshape(p.shape).to(p.device, p.dtype))
                off += k

    def forward(self, g):
        flat = self.flat_parameters()
        if not flat.is_cuda:
            raise _lib.MllpError("AngleModel runs on the MI355X HIP path only: call model.to('cuda') "
                                 "(there is no CPU fallback)")
        n = c_int64()
        _lib.check(_lib.lib().mllp_angle_num_params(self.feat_dim, ctypes.byref(n)))
        assert flat.numel() == n.value
        return _AngleFunction.apply(flat, g, self.feat_dim)
