"""ORACLE (test infrastructure -- never imported by the product path).

CPU restatement, in plain PyTorch, of the reference's learned-LP hot path:

  * graph build            reference linear_program_methods.py:89-103
  * BipartiteData batching reference linear_program_methods.py:60-72  (edge_index += [[n],[m]])
  * GNNModel wiring        reference linear_program_methods.py:202-251
  * loss / step / metrics  reference linear_program_experiment.py:41,119-153

The arithmetic of each layer lives in a third-party dependency that is NOT in
/root/reference and is not installable here: `torch_geometric.nn.TransformerConv`
(PyG; version unpinned by the reference -- no requirements file).  `transformer_conv`
below restates PyG 2.x's published algorithm op for op (lin_query/lin_key/lin_value/
lin_edge/lin_skip, `key_j + edge`, dot / sqrt(C), `torch_geometric.utils.softmax`
with its detached max and `+ 1e-16`, `value_j + edge`, add-aggregation, root skip).

PARITY UNPINNED: the reference ships no tests, golden vectors or checkpoints for
this path, and PyG cannot be imported to generate any, so nothing pins this
restatement to PyG's actual output.  What IS pinned: inputs (captured through the
imported reference loader, tests/golden/), the model wiring (read from the
reference source), and the analytic known-answer cases in tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

FEAT = 16
CONV_NAMES = ["gconv1_w2s", "gconv1_s2w", "gconv2_w2s", "gconv2_s2w", "gconv3_w2s", "gconv3_s2w"]
CONV_CIN = {"gconv1_w2s": 1, "gconv1_s2w": 1, "gconv2_w2s": 16, "gconv2_s2w": 16,
            "gconv3_w2s": 16, "gconv3_s2w": 16}


def state_dict_spec():
    """(key, shape) in registration order: PyG TransformerConv registers lin_key, lin_query,
    lin_value, lin_edge (no bias), lin_skip; GNNModel registers the six convs then fc
    (reference linear_program_methods.py:206-211,220).  4721 parameters in total."""
    spec = []
    for p in CONV_NAMES:
        c = CONV_CIN[p]
        spec += [(f"{p}.lin_key.weight", (FEAT, c)), (f"{p}.lin_key.bias", (FEAT,)),
                 (f"{p}.lin_query.weight", (FEAT, c)), (f"{p}.lin_query.bias", (FEAT,)),
                 (f"{p}.lin_value.weight", (FEAT, c)), (f"{p}.lin_value.bias", (FEAT,)),
                 (f"{p}.lin_edge.weight", (FEAT, 1)),
                 (f"{p}.lin_skip.weight", (FEAT, c)), (f"{p}.lin_skip.bias", (FEAT,))]
    spec += [("fc.weight", (1, FEAT)), ("fc.bias", (1,))]
    return spec


def init_state(seed=42, dtype=torch.float32):
    """Random parameters with the distributions of PyG/torch Linear.reset_parameters
    (kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)) for weight and bias).  The draw ORDER
    is this file's own; parity tests load weights, they never re-derive them from a seed."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for key, shape in state_dict_spec():
        fan_in = shape[1] if len(shape) == 2 else None
        if fan_in is None:  # bias: bound from the owning layer's fan_in
            owner = key.rsplit(".", 1)[0] + ".weight"
            fan_in = sd[owner].shape[1]
        bound = 1.0 / math.sqrt(fan_in)
        sd[key] = ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
    return sd


def flatten_state(sd):
    return torch.cat([sd[k].reshape(-1) for k, _ in state_dict_spec()])


def unflatten_state(flat):
    sd, off = OrderedDict(), 0
    for k, shape in state_dict_spec():
        cnt = int(np.prod(shape))
        sd[k] = flat[off:off + cnt].reshape(shape)
        off += cnt
    assert off == flat.numel()
    return sd


# ------------------------------------------------------------------------------------------
# graph build (reference linear_program_methods.py:89-103)
# ------------------------------------------------------------------------------------------
def build_graph_literal(constrs, constr_weights, rhs, coefs, dtype=torch.float32):
    """The reference's Python double loop, verbatim semantics (small inputs only)."""
    x_src = torch.tensor(np.asarray(coefs), dtype=dtype).unsqueeze(-1)      # x1: variables  (n,1)
    x_tgt = torch.tensor(np.asarray(rhs), dtype=dtype).unsqueeze(-1)        # x2: constraints (m,1)
    index_1, index_2 = [], []
    for constr_idx, vars_ in enumerate(constrs):
        for var_index in vars_:
            index_1.append(int(var_index))
            index_2.append(constr_idx)
    weights = torch.tensor(np.asarray(constr_weights), dtype=dtype).unsqueeze(-1)
    edge_index = torch.tensor([index_1, index_2], dtype=torch.long).reshape(2, -1)
    return edge_index, x_src, x_tgt, weights


def build_graph(constrs, constr_weights, rhs, coefs, dtype=torch.float32):
    """Same result as build_graph_literal, vectorised (row-major CSR order)."""
    m = len(constrs)
    lens = np.fromiter((len(r) for r in constrs), dtype=np.int64, count=m)
    var = np.concatenate([np.asarray(r, dtype=np.int64) for r in constrs]) if lens.sum() else np.zeros(0, np.int64)
    con = np.repeat(np.arange(m, dtype=np.int64), lens)
    edge_index = torch.from_numpy(np.stack([var, con]))
    x_src = torch.tensor(np.asarray(coefs), dtype=dtype).unsqueeze(-1)
    x_tgt = torch.tensor(np.asarray(rhs), dtype=dtype).unsqueeze(-1)
    weights = torch.tensor(np.asarray(constr_weights), dtype=dtype).unsqueeze(-1)
    return edge_index, x_src, x_tgt, weights


def batch_graphs(graphs):
    """Block-diagonal batch by BipartiteData.__inc__ (reference linear_program_methods.py:68-70):
    row 0 (variable ids) += sum of previous n, row 1 (constraint ids) += sum of previous m."""
    eis, x1s, x2s, eas, n_off, m_off = [], [], [], [], 0, 0
    for ei, x1, x2, ea in graphs:
        eis.append(ei + torch.tensor([[n_off], [m_off]], dtype=torch.long))
        x1s.append(x1)
        x2s.append(x2)
        eas.append(ea)
        n_off += x1.shape[0]
        m_off += x2.shape[0]
    return torch.cat(eis, dim=1), torch.cat(x1s), torch.cat(x2s), torch.cat(eas)


# ------------------------------------------------------------------------------------------
# PyG TransformerConv(heads=1, concat=True, beta=False, dropout=0, edge_dim=1, root_weight=True)
# ------------------------------------------------------------------------------------------
def segment_softmax(src, index, num_nodes):
    """torch_geometric.utils.softmax: max (detached) / exp / sum + 1e-16, grouped by `index`."""
    src_max = torch.full((num_nodes,) + src.shape[1:], float("-inf"), dtype=src.dtype)
    src_max = src_max.scatter_reduce(0, index.view(-1, *([1] * (src.dim() - 1))).expand_as(src),
                                     src.detach(), reduce="amax", include_self=True)
    out = (src - src_max.index_select(0, index)).exp()
    out_sum = torch.zeros((num_nodes,) + src.shape[1:], dtype=src.dtype).index_add_(0, index, out) + 1e-16
    return out / out_sum.index_select(0, index)


def transformer_conv(sd, prefix, x_src, x_dst, edge_index, edge_attr):
    """edge_index[0] = source j, edge_index[1] = target i (flow source_to_target, aggr add)."""
    W = lambda name: sd[f"{prefix}.{name}"]
    C = W("lin_query.weight").shape[0]      # out_channels (FEAT = 16 in GNNModel, feat_dim in AngleModel)
    query = x_dst @ W("lin_query.weight").T + W("lin_query.bias")
    key = x_src @ W("lin_key.weight").T + W("lin_key.bias")
    value = x_src @ W("lin_value.weight").T + W("lin_value.bias")
    src, dst = edge_index[0], edge_index[1]
    e = edge_attr @ W("lin_edge.weight").T                      # (E, C), no bias
    key_j = key.index_select(0, src) + e
    alpha = (query.index_select(0, dst) * key_j).sum(dim=-1) / math.sqrt(C)
    alpha = segment_softmax(alpha, dst, x_dst.shape[0])
    msg = (value.index_select(0, src) + e) * alpha.unsqueeze(-1)
    out = torch.zeros((x_dst.shape[0], C), dtype=msg.dtype).index_add_(0, dst, msg)
    return out + x_dst @ W("lin_skip.weight").T + W("lin_skip.bias")


def gnn_forward(sd, edge_index, x1, x2, edge_attr, return_hidden=False):
    """GNNModel.forward (reference linear_program_methods.py:238-251)."""
    rev = torch.stack((edge_index[1], edge_index[0]), dim=0)
    n1 = torch.relu(transformer_conv(sd, "gconv1_w2s", x2, x1, rev, edge_attr))
    n2 = torch.relu(transformer_conv(sd, "gconv1_s2w", x1, x2, edge_index, edge_attr))
    a1, a2 = n1, n2
    n1 = torch.relu(transformer_conv(sd, "gconv2_w2s", a2, a1, rev, edge_attr))
    n2 = torch.relu(transformer_conv(sd, "gconv2_s2w", a1, a2, edge_index, edge_attr))
    b1, b2 = n1, n2
    n1 = torch.relu(transformer_conv(sd, "gconv3_w2s", b2, b1, rev, edge_attr))
    out = (n1 @ sd["fc.weight"].T + sd["fc.bias"]).squeeze(-1)
    if return_hidden:
        return out, dict(h1v=a1, h1c=a2, h2v=b1, h2c=b2, h3v=n1)
    return out


def bce_with_logits(z, t):
    """torch.nn.BCEWithLogitsLoss(), mean reduction (reference linear_program_experiment.py:41)."""
    return torch.nn.functional.binary_cross_entropy_with_logits(z, t)


def topk_metrics(logits, m, basis):
    """reference linear_program_experiment.py:146-151: top-m logits -> 0/1 prediction ->
    (correct_num, f1) with sklearn's binary F1 = 2TP / (2TP + FP + FN)."""
    logits = np.asarray(logits)
    basis = np.asarray(basis).astype(np.float64)
    k = int(m)
    idx = torch.topk(torch.as_tensor(logits), k=k)[-1].numpy()
    pred = np.zeros(basis.shape[0])
    pred[idx] = 1
    tp = float(pred @ basis)
    fp = float(pred.sum() - tp)
    fn = float(basis.sum() - tp)
    f1 = 0.0 if (2 * tp + fp + fn) == 0 else 2 * tp / (2 * tp + fp + fn)
    return tp, f1


# ------------------------------------------------------------------------------------------
# whole-step helpers used by parity tests, golden generation and the cpu_baseline
# ------------------------------------------------------------------------------------------
def instance_graph(inst, dtype=torch.float32):
    """LPInstance (mllp_amd.data) -> (edge_index, x1, x2, edge_attr) in reference order."""
    name, constrs, w, coefs, rhs, basis = inst.as_reference_tuple()
    return build_graph(constrs, w, rhs, coefs, dtype)


def batch_loss_and_grads(sd, instances, dtype=torch.float64, global_count=None):
    """Loss = (1/B) * sum_k mean-BCE(instance k); returns (loss, logits list, flat grad).

    Matches the build's batched step; for B == 1 it is exactly the reference's per-instance
    objective (linear_program_experiment.py:139-141)."""
    sd = OrderedDict((k, v.detach().to(dtype).clone().requires_grad_(True)) for k, v in sd.items())
    graphs = [instance_graph(i, dtype) for i in instances]
    ei, x1, x2, ea = batch_graphs(graphs)
    z = gnn_forward(sd, ei, x1, x2, ea)
    B = global_count or len(instances)
    loss, off, logits = 0.0, 0, []
    for inst in instances:
        zi = z[off:off + inst.n]
        loss = loss + bce_with_logits(zi, torch.tensor(inst.basis, dtype=dtype)) / B
        logits.append(zi.detach())
        off += inst.n
    loss.backward()
    grads = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v))) for k, v in sd.items())
    return loss.detach(), logits, flatten_state(grads).detach()


class ReferenceTrainer:
    """The reference's loop (linear_program_experiment.py:117-153): one Adam step PER INSTANCE,
    graph rebuilt every step (`rebuild_graph=True` reproduces experiment.py:124; False caches)."""

    def __init__(self, sd, lr=1e-3, dtype=torch.float32, rebuild_graph=True):
        self.sd = OrderedDict((k, v.detach().to(dtype).clone().requires_grad_(True)) for k, v in sd.items())
        self.opt = torch.optim.Adam(list(self.sd.values()), lr=lr)
        self.dtype = dtype
        self.rebuild = rebuild_graph
        self._cache = {}

    def step(self, inst):
        if self.rebuild:
            name, constrs, w, coefs, rhs, basis = inst.as_reference_tuple()
            g = build_graph_literal(constrs, w, rhs, coefs, self.dtype)
        else:
            g = self._cache.get(inst.name)
            if g is None:
                g = self._cache[inst.name] = instance_graph(inst, self.dtype)
        z = gnn_forward(self.sd, g[0], g[1], g[2], g[3])
        obj = bce_with_logits(z, torch.tensor(inst.basis, dtype=self.dtype))
        obj.backward()
        self.opt.step()
        self.opt.zero_grad()
        return float(obj.detach()), z.detach()
