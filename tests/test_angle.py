"""the `angleNet` method (AngleModel on the complete angle graph).

CPU: the graph builder against a literal restatement of the reference's per-edge Python loop
(linear_program_methods.py:105-130).  GPU: the dense HIP model (mllp_amd/csrc/angle.hip) against the oracle's literal
PyG TransformerConv run over the explicit N (N - 1) edge list in fp64 with autograd -- logits, loss, every parameter
gradient -- and the training loop of the driver.  Model arithmetic: parity unpinned w.r.t. PyG itself.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import pyg_restatement as o1


def _cosine_similarity(u, v):       # reference linear_program_methods.py:105-108, verbatim semantics
    if np.linalg.norm(u) <= 1e-6 or np.linalg.norm(v) <= 1e-6:
        return 0
    return u @ v / np.linalg.norm(u) / np.linalg.norm(v)


def test_graph_builder_equals_reference_edge_loop():
    from mllp_amd.angle import build_graph_from_Q_sets
    rng = np.random.default_rng(3)
    Q = rng.standard_normal((9, 4))
    Q[5] = 0.0                                              # a zero row: the reference's guard returns 0
    coefs = rng.standard_normal(9)
    g = build_graph_from_Q_sets(Q, coefs, torch.device("cpu"), "t", np.zeros(8))
    assert g.num_nodes == 9 and g.var_num == 8 and g.basis_num == 4
    np.testing.assert_allclose(g.x.numpy(), np.stack([coefs, np.linalg.norm(Q, axis=1)], 1).astype(np.float32))
    ei = np.stack(np.where(np.ones((9, 9)) - np.diag(np.ones(9))))          # reference :126
    assert np.array_equal(g.edge_index.numpy(), ei)
    want = np.array([_cosine_similarity(Q[ei[0, k]], Q[ei[1, k]]) for k in range(ei.shape[1])], dtype=np.float32)
    np.testing.assert_allclose(g.edge_attr.numpy()[:, 0], want, atol=1e-7)


def _oracle_forward(sd, g, dtype=torch.float64):
    """reference linear_program_methods.py:195-200 with the oracle's literal TransformerConv on the edge list"""
    x = g.x.to("cpu", dtype)
    ei = g.edge_index.cpu()
    ea = g.edge_attr.to("cpu", dtype)
    h = torch.relu(o1.transformer_conv(sd, "gconv1", x, x, ei, ea))
    h = torch.relu(o1.transformer_conv(sd, "gconv2", h, h, ei, ea))
    h = torch.relu(o1.transformer_conv(sd, "gconv2", h, h, ei, ea))
    z = (h @ sd["fc.weight"].T + sd["fc.bias"]).squeeze(-1)
    return z[:-1]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["afiro_F256", "random_N300_F64"])
def test_angle_model_forward_backward_vs_oracle(case):
    from mllp_amd.angle import AngleModel, build_graph_from_Q_sets, dense_instance_tensors
    from mllp_amd.data import load_packed
    from mllp_amd.model import set_seed
    set_seed(7)
    if case == "afiro_F256":
        inst = load_packed(["afiro"])[0]
        Q, coefs, basis = dense_instance_tensors(inst)
        F = 256                                             # the reference's feat_dim (experiment.py:83)
    else:
        rng = np.random.default_rng(11)
        Q, _ = np.linalg.qr(rng.standard_normal((300, 40)))
        Q[17] = 0.0
        coefs = rng.standard_normal(300)
        basis = (rng.random(299) < 0.3).astype(np.int32)
        F = 64
    g = build_graph_from_Q_sets(Q, coefs, torch.device("cuda"), case, basis)
    model = AngleModel(feat_dim=F).to("cuda")
    y = torch.tensor(basis, dtype=torch.float, device="cuda")
    crit = torch.nn.BCEWithLogitsLoss()
    logits = model(g)
    loss = crit(logits, y)
    loss.backward()
    # oracle: fp64 autograd over the explicit edge list
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
    z = _oracle_forward(sd, g)
    l_ref = torch.nn.functional.binary_cross_entropy_with_logits(z, y.cpu().double())
    l_ref.backward()

    def close(a, b, rtol, what):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        err = np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)
        assert err <= rtol, (what, err)
    close(logits.detach().cpu().numpy(), z.detach().numpy(), 1e-5, "logits")
    assert abs(float(loss.detach()) - float(l_ref.detach())) <= 1e-5 * abs(float(l_ref.detach()))
    for name, p in model.named_parameters():
        ref = sd[name].grad
        if name.startswith("gconv3"):                      # never called by forward (reference :198)
            assert ref is None and float(p.grad.abs().max()) == 0.0
            continue
        if name.endswith("lin_key.bias"):                  # cancels in the softmax: exactly 0 in exact arithmetic
            assert float(p.grad.abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max()) * 1e6)
            continue
        close(p.grad.cpu().numpy(), ref.numpy(), 5e-5, name)
    # run-to-run bitwise
    model.zero_grad()
    l2 = crit(model(g), y)
    l2.backward()
    assert float(l2.detach()) == float(loss.detach())


@pytest.mark.gpu
def test_angle_driver_end_to_end(tmp_path, monkeypatch, capsys):
    """python linear_program_experiment.py --cfg <yaml with methods: ['angleNet']> -- the reference's loop :81-114"""
    from mllp_amd import experiment
    y = tmp_path / "angle.yaml"
    y.write_text("train_data_type: 'netlib'\ntrain_lr: 1.e-3\ntrain_iter: 3\ninstances: ['afiro']\n"
                 "angle_feat_dim: 32\nmethods:\n  - 'angleNet'\n")
    monkeypatch.chdir(tmp_path)
    assert experiment.main(["--cfg", str(y)]) == 0
    lines = capsys.readouterr().out.splitlines()
    log = json.load(open(tmp_path / "train_log.json"))
    assert len(log["obj"]) == 3 and len(log["afiro.mps"]) == 3 and log["obj"][2] < log["obj"][0]
    sd = torch.load(tmp_path / "linear_program_netlib_angleNet.pt", weights_only=True)
    assert sd["gconv1.lin_key.weight"].shape == (32, 2) and sd["gconv2.lin_edge.weight"].shape == (32, 1)
    assert sd["fc.weight"].shape == (1, 32) and any(s.startswith("epoch 2, obj=") for s in lines)


@pytest.mark.gpu
def test_angle_stepper_equals_module_loop():
    """AngleStepper (flat parameters, C ABI forward / backward, the library's Adam kernel) follows the nn.Module +
    autograd + torch.optim.Adam loop of the reference (linear_program_experiment.py:88-96) step for step; feat_dim
    outside {16, 32, 64, 128, 256} is refused loudly."""
    from mllp_amd import _lib
    from mllp_amd.angle import AngleModel, AngleStepper, build_graph_from_Q_sets
    from mllp_amd.model import set_seed
    rng = np.random.default_rng(5)
    Q, _ = np.linalg.qr(rng.standard_normal((150, 30)))
    coefs = rng.standard_normal(150)
    basis = (rng.random(149) < 0.3).astype(np.int32)
    g = build_graph_from_Q_sets(Q, coefs, torch.device("cuda"), "stepper", basis)
    assert torch.equal(g.cos, g.cos.T)
    y = torch.tensor(basis, dtype=torch.float, device="cuda")
    set_seed(3)
    model = AngleModel(feat_dim=32).to("cuda")
    st = AngleStepper(model, lr=1e-3)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = torch.nn.BCEWithLogitsLoss()
    for _ in range(4):
        opt.zero_grad()
        loss = crit(model(g), y)
        loss.backward()
        opt.step()
        loss2, _ = st.step(g, y)
        assert abs(float(loss.detach()) - float(loss2)) <= 2e-6 * abs(float(loss.detach()))
    a, b = model.flat_parameters().detach(), st.params
    keep = (a - b).abs() <= 2e-5 * a.abs().max()
    assert bool(keep.all())
    with pytest.raises(_lib.MllpError, match="feat_dim"):
        AngleModel(feat_dim=24).to("cuda")(g)
