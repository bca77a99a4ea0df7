"""Drop-in model surface of the reference's `linear_program_methods.py` for the sparse bipartite path:
`set_seed`, `BipartiteData`, `build_graph_from_weights_sets`, `GNNModel` -- same names, argument order
and state_dict keys, with the arithmetic in libmllp_hip.so (no torch_geometric, no CPU fallback).

reference: linear_program_methods.py:15-24 (set_seed), :60-72 (BipartiteData), :89-103 (graph build),
:202-251 (GNNModel).
"""
import os
import random

import numpy as np
import torch

from . import _lib
from .data import LPInstance
from .graph import LPBatch


def set_seed(seed: int = 42) -> None:
    """reference linear_program_methods.py:15-24"""
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
    os.environ["PYTHONHASHSEED"] = str(seed)


class BipartiteData:
    """Container with the attributes of the reference's PyG `Data` subclass
    (linear_program_methods.py:60-72): `edge_index` (2,E) rows [variable ; constraint], `x1` (n,1),
    `x2` (m,1), `edge_attr` (E,1).  `__inc__` keeps the block-diagonal batching rule; `batch()` applies
    it to a list of graphs.  The HBM-resident `LPBatch` is built lazily, once, and cached."""

    def __init__(self, edge_index, x_src, x_dst, edge_attr):
        self.edge_index = edge_index
        self.x1 = x_src
        self.x2 = x_dst
        self.edge_attr = edge_attr
        self._lp_batch = None
        self._sizes = None   # [(m_k, n_k)] when this object is a batch of several instances

    def __inc__(self, key, value=None):
        if key == "edge_index":
            return torch.tensor([[self.x1.size(0)], [self.x2.size(0)]])
        return 0

    @staticmethod
    def batch(graphs):
        eis, x1s, x2s, eas, n_off, m_off, sizes = [], [], [], [], 0, 0, []
        for g in graphs:
            inc = torch.tensor([[n_off], [m_off]], dtype=g.edge_index.dtype, device=g.edge_index.device)
            eis.append(g.edge_index + inc)
            x1s.append(g.x1)
            x2s.append(g.x2)
            eas.append(g.edge_attr)
            step = g.__inc__("edge_index")
            n_off += int(step[0, 0])
            m_off += int(step[1, 0])
            sizes.append((g.x2.size(0), g.x1.size(0)))
        out = BipartiteData(torch.cat(eis, 1), torch.cat(x1s), torch.cat(x2s), torch.cat(eas))
        out._sizes = sizes
        return out

    def to(self, device):
        self.edge_index = self.edge_index.to(device)
        self.x1, self.x2, self.edge_attr = self.x1.to(device), self.x2.to(device), self.edge_attr.to(device)
        return self

    def lp_batch(self) -> LPBatch:
        if self._lp_batch is None:
            sizes = self._sizes or [(self.x2.size(0), self.x1.size(0))]
            ei = self.edge_index.detach().cpu().numpy()
            ea = self.edge_attr.detach().cpu().numpy().reshape(-1).astype(np.float64)
            x1 = self.x1.detach().cpu().numpy().reshape(-1).astype(np.float64)
            x2 = self.x2.detach().cpu().numpy().reshape(-1).astype(np.float64)
            var, con = ei[0].astype(np.int64), ei[1].astype(np.int64)
            order = np.lexsort((var, con))                     # CSR order whatever the edge order was
            var, con, ea = var[order], con[order], ea[order]
            insts, m_off, n_off = [], 0, 0
            starts = np.searchsorted(con, np.cumsum([0] + [s[0] for s in sizes]))
            for k, (m, n) in enumerate(sizes):
                e0, e1 = starts[k], starts[k + 1]
                indptr = np.zeros(m + 1, dtype=np.int64)
                np.add.at(indptr, con[e0:e1] - m_off + 1, 1)
                indptr = np.cumsum(indptr)
                insts.append(LPInstance(f"g{k}", indptr, (var[e0:e1] - n_off).astype(np.int32), ea[e0:e1],
                                        x1[n_off:n_off + n], x2[m_off:m_off + m], np.zeros(n, np.int32)))
                m_off += m
                n_off += n
            self._lp_batch = LPBatch.from_instances(insts)
        return self._lp_batch


def build_graph_from_weights_sets(constrs, constr_weights, rhs, coefs, device=torch.device("cpu")):
    """Drop-in for reference linear_program_methods.py:89-103: x1 = coefs (n,1), x2 = rhs (m,1),
    edge_index = [variable ; constraint] in CSR order, edge_attr = a_ij (E,1); fp32 casts as the
    reference.  (Vectorised; the reference's Python double loop costs 0.83 s/epoch on full Netlib.)"""
    m = len(constrs)
    lens = np.fromiter((len(r) for r in constrs), dtype=np.int64, count=m)
    var = (np.concatenate([np.asarray(r, dtype=np.int64) for r in constrs]) if lens.sum() else np.zeros(0, np.int64))
    con = np.repeat(np.arange(m, dtype=np.int64), lens)
    x_src = torch.tensor(np.asarray(coefs), dtype=torch.float, device=device).unsqueeze(-1)
    x_tgt = torch.tensor(np.asarray(rhs), dtype=torch.float, device=device).unsqueeze(-1)
    weights = torch.tensor(np.asarray(constr_weights), dtype=torch.float, device=device).unsqueeze(-1)
    edge_index = torch.tensor(np.stack([var, con]), device=device)
    return BipartiteData(edge_index, x_src, x_tgt, weights)


class _TConvParams(torch.nn.Module):
    """Parameter holder with PyG TransformerConv's attribute names and registration order
    (lin_key, lin_query, lin_value, lin_edge[no bias], lin_skip): SURVEY.md appendix A.2."""

    def __init__(self, cin, cout=16):
        super().__init__()
        self.lin_key = torch.nn.Linear(cin, cout)
        self.lin_query = torch.nn.Linear(cin, cout)
        self.lin_value = torch.nn.Linear(cin, cout)
        self.lin_edge = torch.nn.Linear(1, cout, bias=False)
        self.lin_skip = torch.nn.Linear(cin, cout)


class _GNNFunction(torch.autograd.Function):
    """logits = GNN(params; batch) through mllp_gnn_forward / mllp_gnn_backward."""

    @staticmethod
    def forward(ctx, flat, batch):
        flat = flat.contiguous()
        logits = batch.forward(flat)
        batch._fwd_token = getattr(batch, "_fwd_token", 0) + 1
        ctx.batch, ctx.token = batch, batch._fwd_token
        ctx.save_for_backward(flat)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        (flat,) = ctx.saved_tensors
        if ctx.batch._fwd_token != ctx.token:
            raise RuntimeError("GNNModel: backward after another forward on the same graph -- the saved "
                               "activations live in the graph's workspace and were overwritten")
        return ctx.batch.backward(flat, dlogits), None


class GNNModel(torch.nn.Module):
    """reference linear_program_methods.py:202-251: six TransformerConv((c,c),16,edge_dim=1) holders
    (gconv3_s2w is declared but never called, :248) and fc = Linear(16,1).  forward(g) -> (n,) logits."""

    def __init__(self):
        super().__init__()
        self.gconv1_w2s = _TConvParams(1)
        self.gconv1_s2w = _TConvParams(1)
        self.gconv2_w2s = _TConvParams(16)
        self.gconv2_s2w = _TConvParams(16)
        self.gconv3_w2s = _TConvParams(16)
        self.gconv3_s2w = _TConvParams(16)
        self.fc = torch.nn.Linear(16, 1)

    def flat_parameters(self):
        """All 4721 parameters as one differentiable flat tensor in state_dict order."""
        return torch.cat([p.reshape(-1) for p in self.parameters()])

    def load_flat(self, flat):
        off = 0
        with torch.no_grad():
            for p in self.parameters():
                n = p.numel()
                p.copy_(torch.as_tensor(flat[off:off + n]).reshape(p.shape).to(p.device, p.dtype))
                off += n
        assert off == _lib.NUM_PARAMS

    def forward(self, g):
        assert type(g) == BipartiteData
        flat = self.flat_parameters()
        if not flat.is_cuda:
            raise _lib.MllpError("GNNModel runs on the MI355X HIP path only: call model.to('cuda') "
                                 "(there is no CPU fallback)")
        return _GNNFunction.apply(flat, g.lp_batch())
