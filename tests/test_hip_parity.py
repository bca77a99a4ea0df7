"""GPU parity tests: the HIP path (through the C ABI) against the oracle on the same inputs.

Tolerance: BASELINE.json's north_star asks for predicted solutions within 1e-5 relative of the CPU
path in fp32.  Logits / activations / losses are checked at 1e-5 (`close`: relative to the tensor's max
magnitude, the natural scale of an fp32 accumulation; `close_elementwise`: every element within
1e-5 * |ref| + 1e-5 * rms(ref), so small logits are held to the same absolute error as typical ones),
gradients at 5e-5, all against the fp64 oracle.
`lin_key.bias` gradients are excluded from relative checks: a per-destination constant cancels in the
softmax, so that gradient is exactly 0 in exact arithmetic and rounding noise in fp32 on both sides.
"""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
pytestmark = pytest.mark.gpu

from mllp_amd.data import SUBSET5, LPInstance, load_packed, synthetic_instance  # noqa: E402
from oracle import pyg_restatement as o1  # noqa: E402
from oracle import spmm_form as o2  # noqa: E402

RTOL_ACT, RTOL_GRAD = 1e-5, 5e-5


def close(got, want, rtol, what=""):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, what
    assert np.isfinite(got).all(), what
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got - want).max()) / scale
    assert err < rtol, f"{what}: max|diff|/max|ref| = {err:.3e} >= {rtol}"


def close_elementwise(got, want, rtol, what=""):
    """every element: |got - want| <= rtol * (|want| + rms(want))"""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, what
    tol = rtol * (np.abs(want) + max(float(np.sqrt(np.mean(want ** 2))), 1e-30))
    bad = np.abs(got - want) > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} elements off, worst {float((np.abs(got - want) / tol).max()):.2f} x tol"


def grad_mask():
    keep, off = np.ones(4721, bool), 0
    for k, s in o1.state_dict_spec():
        c = int(np.prod(s))
        if k.endswith("lin_key.bias"):
            keep[off:off + c] = False
        off += c
    return keep


@pytest.fixture(scope="module", params=["auto", "generic"])
def LPBatch(request):
    """Every test that builds batches runs twice: with the library's own choice of whole-model kernels (the fused
    latency-regime path for everything below 32 M nonzeros) and with the generic / LDS-tiled sweeps forced."""
    from mllp_amd import _lib
    _lib.lib()                      # fail loudly: no fallback
    assert torch.cuda.is_available()
    from mllp_amd.graph import LPBatch as cls
    cls.default_path = 1 if request.param == "generic" else 0
    yield cls
    cls.default_path = 0


@pytest.fixture(scope="module")
def weights(golden):
    flat = golden["weights_flat"]
    sd = {k: v.numpy() for k, v in o1.unflatten_state(torch.tensor(flat)).items()}
    return flat, sd, torch.tensor(flat, dtype=torch.float32, device="cuda")


# ---------------------------------------------------------------------------------------------------
def test_graph_arrays_match_scipy_transpose(LPBatch, subset5):
    b = LPBatch.from_instances(subset5)
    ob = o2.BatchCSR(subset5)
    np.testing.assert_array_equal(b.export(0), ob.rp)
    np.testing.assert_array_equal(b.export(1), ob.ci)
    np.testing.assert_array_equal(b.export(2), ob.va.astype(np.float32))
    np.testing.assert_array_equal(b.export(3), ob.cp)
    np.testing.assert_array_equal(b.export(4), ob.ri)
    np.testing.assert_array_equal(b.export(5), ob.cv.astype(np.float32))
    inv_n = b.export(6)
    np.testing.assert_allclose(inv_n, ob.wnode * len(subset5), rtol=1e-6)


def test_graph_create_rejects_bad_input(LPBatch):
    from mllp_amd import _lib
    bad = LPInstance("bad", np.array([0, 2]), np.array([3, 1], np.int32), np.ones(2), np.ones(4), np.ones(1),
                     np.zeros(4, np.int32))                # unsorted columns
    with pytest.raises(_lib.MllpError, match="sorted"):
        LPBatch.from_instances([bad])
    oob = LPInstance("oob", np.array([0, 1]), np.array([7], np.int32), np.ones(1), np.ones(4), np.ones(1),
                     np.zeros(4, np.int32))
    with pytest.raises(_lib.MllpError):
        LPBatch.from_instances([oob])


@pytest.mark.parametrize("tiers", [(0, 0), (4, 16), (1, 2)])
def test_spmm_both_orientations(LPBatch, subset5, tiers):
    b = LPBatch.from_instances(subset5, tier_wave=tiers[0], tier_block=tiers[1])
    ob = o2.BatchCSR(subset5)
    rng = np.random.default_rng(1)
    Hn = rng.standard_normal((b.N, 16)).astype(np.float32)
    Hm = rng.standard_normal((b.M, 16)).astype(np.float32)
    Y = b.spmm(torch.tensor(Hn, device="cuda")).cpu().numpy()
    close(Y, o2.spmm(ob.rp, ob.ci, ob.va.astype(np.float32).astype(np.float64), Hn.astype(np.float64)), 1e-6, "A H")
    Yt = b.spmm(torch.tensor(Hm, device="cuda"), transpose=True).cpu().numpy()
    close(Yt, o2.spmm(ob.cp, ob.ri, ob.cv.astype(np.float32).astype(np.float64), Hm.astype(np.float64)), 1e-6, "At H")
    # linearity (size independent property): A (a X + b Y) = a A X + b A Y
    X2 = rng.standard_normal((b.N, 16)).astype(np.float32)
    lhs = b.spmm(torch.tensor(2 * Hn - 3 * X2, device="cuda")).cpu().numpy()
    rhs = 2 * Y - 3 * b.spmm(torch.tensor(X2, device="cuda")).cpu().numpy()
    close(lhs, rhs, 2e-6, "linearity")


@pytest.mark.parametrize("tiers", [(0, 0), (4, 16)])
@pytest.mark.parametrize("name,dst_is_var,cin,off", [("gconv1_w2s", True, 1, 0), ("gconv1_s2w", False, 1, 144),
                                                     ("gconv2_w2s", True, 16, 288), ("gconv2_s2w", False, 16, 1392)])
def test_single_layer_forward_backward(LPBatch, subset5, weights, tiers, name, dst_is_var, cin, off):
    flat, sd, flat_gpu = weights
    b = LPBatch.from_instances(subset5, tier_wave=tiers[0], tier_block=tiers[1])
    ob = o2.BatchCSR(subset5)
    rng = np.random.default_rng(3)
    p = o2.conv_params(sd, name)
    ptr, idx, val, nd, ns = ob.orient(dst_is_var)
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    xs, xd, dh = r32(rng.standard_normal((ns, cin))), r32(rng.standard_normal((nd, cin))), r32(rng.standard_normal((nd, 16)))
    h_ref, saved = o2.conv_fwd(p, ptr, idx, val, xs, xd)
    cp = flat_gpu[off:off + (144 if cin == 1 else 1104)].contiguous()
    ws = b.tconv_workspace(dst_is_var, cin)
    xs_t = torch.tensor(xs, dtype=torch.float32, device="cuda")
    xd_t = torch.tensor(xd, dtype=torch.float32, device="cuda")
    h = b.tconv_fwd(dst_is_var, cin, cp, xs_t, xd_t, ws)
    close(h.cpu().numpy(), h_ref, RTOL_ACT, "h")
    grads, dxd, dxs, inter = o2.conv_bwd(p, ptr, idx, val, xs, xd, saved, dh, need_input_grads=(cin == 16))
    pg, dxd_g, dxs_g, g = b.tconv_bwd(dst_is_var, cin, cp, xs_t, xd_t, h, ws,
                                      torch.tensor(dh, dtype=torch.float32, device="cuda"))
    close(g.cpu().numpy(), inter["g"], 1e-7, "masked dh")
    if cin == 16:
        close(dxd_g.cpu().numpy(), dxd, RTOL_GRAD, "dx_dst")
        close(dxs_g.cpu().numpy(), dxs, RTOL_GRAD, "dx_src")
    pgn, o3 = pg.cpu().numpy(), 0
    for key in ("lin_key.weight", "lin_key.bias", "lin_query.weight", "lin_query.bias", "lin_value.weight",
                "lin_value.bias", "lin_edge.weight", "lin_skip.weight", "lin_skip.bias"):
        ref = np.asarray(grads[key]).reshape(-1)
        if key == "lin_key.bias":
            assert np.abs(pgn[o3:o3 + ref.size]).max() < 1e-5      # ~0 (noise)
        else:
            close(pgn[o3:o3 + ref.size], ref, RTOL_GRAD, key)
        o3 += ref.size


def test_online_softmax_rescale_is_exercised(LPBatch):
    """A long row whose logits INCREASE along the row forces the running max to move in every pass of
    the 16-lane loop (guide rule 26: a rare data-dependent branch needs an input that forces it)."""
    n, m = 700, 3
    rng = np.random.default_rng(5)
    dense = np.zeros((m, n))
    dense[0, :] = np.linspace(-1.0, 1.0, n)            # increasing a_ij -> increasing a_ij * t_i term
    dense[1, ::7] = rng.standard_normal(len(range(0, n, 7)))
    dense[2, :40] = -np.linspace(0.1, 1.0, 40)         # decreasing: max found in the first pass
    import scipy.sparse as sp
    A = sp.csr_matrix(dense)
    A.sort_indices()
    inst = LPInstance("ramp", A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data,
                      np.linspace(-1, 1, n), np.array([3.0, 0.0, 1.0]), (rng.random(n) < 0.4).astype(np.int32))
    sd = o1.init_state(9, torch.float64)
    for k in sd:                                        # large query/edge weights -> logits spread over +-30
        if "lin_query" in k or "lin_edge" in k:
            sd[k] = sd[k] * 6.0
    flat = o1.flatten_state(sd)
    r = o2.gnn_forward_backward({k: v.numpy() for k, v in sd.items()}, o2.BatchCSR([inst]))
    for tiers in [(1024, 4096), (64, 256), (8, 64)]:
        b = LPBatch.from_instances([inst], tier_wave=tiers[0], tier_block=tiers[1])
        loss, logits, grads = b.loss_step(flat.float().cuda())
        close(logits.cpu().numpy(), r["logits"], RTOL_ACT, f"logits tiers={tiers}")
        close(grads.cpu().numpy()[grad_mask()], r["grads"][grad_mask()], RTOL_GRAD, f"grads tiers={tiers}")


def test_empty_rows_columns_and_zero_instance(LPBatch):
    rng = np.random.default_rng(7)
    m, n = 9, 40
    dense = (rng.random((m, n)) < 0.15) * rng.standard_normal((m, n))
    dense[3, :] = 0.0
    dense[:, 5] = 0.0
    dense[6, :] = rng.standard_normal(n) * (np.arange(n) != 5)
    import scipy.sparse as sp
    A = sp.csr_matrix(dense)
    A.sort_indices()
    inst = LPInstance("toy", A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data,
                      rng.standard_normal(n), np.abs(rng.standard_normal(m)), (rng.random(n) < 0.4).astype(np.int32))
    empty = LPInstance("nonz", np.zeros(3, np.int64), np.zeros(0, np.int32), np.zeros(0), rng.standard_normal(4),
                       np.ones(2), np.array([1, 0, 0, 1], np.int32))        # an instance with no nonzeros at all
    sd = o1.init_state(11, torch.float64)
    insts = [inst, empty, inst]
    r = o2.gnn_forward_backward({k: v.numpy() for k, v in sd.items()}, o2.BatchCSR(insts))
    b = LPBatch.from_instances(insts, tier_wave=4, tier_block=16)
    loss, logits, grads = b.loss_step(o1.flatten_state(sd).float().cuda())
    close(logits.cpu().numpy(), r["logits"], RTOL_ACT, "logits")
    close(loss.cpu().numpy(), [r["loss"]], RTOL_ACT, "loss")
    close(grads.cpu().numpy()[grad_mask()], r["grads"][grad_mask()], RTOL_GRAD, "grads")


def test_whole_model_against_golden(LPBatch, subset5, golden, weights):
    """the committed fp64 golden vectors (tests/golden/subset5.npz): batch of 5 and afiro alone"""
    flat, sd, flat_gpu = weights
    b = LPBatch.from_instances(subset5)
    logits = b.forward(flat_gpu)
    close(logits.cpu().numpy(), golden["batch_logits"], RTOL_ACT, "logits")
    loss, logits2, grads = b.loss_step(flat_gpu)
    assert torch.equal(logits, logits2)                       # same kernels, same order: bitwise
    close(loss.cpu().numpy(), [float(golden["batch_loss"])], RTOL_ACT, "loss")
    close(grads.cpu().numpy()[grad_mask()], golden["batch_grads"][grad_mask()], RTOL_GRAD, "grads")
    afiro = [i for i in subset5 if i.name == "afiro.mps"]
    ba = LPBatch.from_instances(afiro)
    loss, logits, grads = ba.loss_step(flat_gpu)
    close(logits.cpu().numpy(), golden["afiro_logits"], RTOL_ACT, "afiro logits")
    close(grads.cpu().numpy()[grad_mask()], golden["afiro_grads"][grad_mask()], RTOL_GRAD, "afiro grads")
    # block-diagonal batch == per-instance results (BipartiteData.__inc__ semantics)
    np.testing.assert_allclose(logits.cpu().numpy(), logits2.cpu().numpy()[138:138 + 51], rtol=1e-6, atol=1e-7)
    # determinism: same call twice gives the same bits (no atomics anywhere)
    l2, z2, g2 = ba.loss_step(flat_gpu)
    assert torch.equal(z2, logits) and torch.equal(g2, grads)


def test_full_netlib_batch(LPBatch, weights):
    """BASELINE.json configs[2]: all 97 instances as one batch, logits and gradients vs the fp64 oracle"""
    flat, sd, flat_gpu = weights
    inst = load_packed()
    r = o2.gnn_forward_backward(sd, o2.BatchCSR(inst))
    b = LPBatch.from_instances(inst)
    d = b.dims()
    assert d["nnz"] == 1074147 and d["A_wave"] > 0 and d["A_block"] > 0 and d["At_block"] > 0   # all tiers in play
    assert d["A_split"] > 0 and d["At_split"] > 0                                                 # incl. rows split over workgroups
    loss, logits, grads = b.loss_step(flat_gpu)
    close(logits.cpu().numpy(), r["logits"], RTOL_ACT, "logits")
    close(loss.cpu().numpy(), [r["loss"]], RTOL_ACT, "loss")
    close(grads.cpu().numpy()[grad_mask()], r["grads"][grad_mask()], RTOL_GRAD, "grads")
    # per-instance relative check of the predicted logits (north_star: within 1e-5 relative)
    for z, zr in zip(b.logits_per_instance(logits.cpu().numpy()), b.logits_per_instance(r["logits"])):
        close(z, zr, RTOL_ACT, "per-instance logits")
        close_elementwise(z, zr, RTOL_ACT, "per-instance logits, element-wise")
    met = b.topm_metrics(logits).cpu().numpy()
    zs = b.logits_per_instance(logits.cpu().numpy())
    for k, (z, i) in enumerate(zip(zs, inst)):
        # the kernel's documented rule: m largest logits, ties at the threshold taken in index order
        order = np.argsort(-z.astype(np.float64), kind="stable")[:i.m]
        tp = float(i.basis[order].sum())
        assert met[k, 0] == tp, i.name
        f1 = 0.0 if tp == 0 else 2 * tp / (2 * tp + (i.m - tp) + (i.basis.sum() - tp))
        assert abs(met[k, 1] - f1) < 1e-5
        thr = np.sort(z)[::-1][i.m - 1]
        if (z == thr).sum() == 1:          # no tie: torch.topk (the reference's call) must agree exactly
            assert o1.topk_metrics(z, i.m, i.basis)[0] == tp, i.name


def test_adam_and_trainer_follow_reference_loop(LPBatch, subset5, golden, weights):
    """reference semantics: one Adam step per instance (experiment.py:123-144); golden = fp64 oracle"""
    from mllp_amd.trainer import LPTrainer
    flat, sd, flat_gpu = weights
    afiro = LPBatch.from_instances([i for i in subset5 if i.name == "afiro.mps"])
    for use_graph in (False, True):
        tr = LPTrainer(flat_gpu, lr=1e-3, use_hip_graph=use_graph)
        losses = [float(tr.step(afiro)[0][0]) for _ in range(3)]
        np.testing.assert_allclose(losses, golden["afiro_adam3_losses"], rtol=1e-5)
        keep = grad_mask()
        # Adam divides by sqrt(v): parameters whose gradient is at rounding-noise level move by a noisy O(lr);
        # 1e-5 absolute is 0.3 % of the 3e-3 a parameter can travel in three steps at lr = 1e-3
        np.testing.assert_allclose(tr.params.cpu().numpy()[keep], golden["afiro_adam3_weights"][keep],
                                   rtol=1e-4, atol=1e-5)
        assert float(tr.opt.state[0]) == 3.0
    # gconv3_s2w is never called (reference methods.py:248): it receives zero gradient and never moves
    np.testing.assert_array_equal(tr.params.cpu().numpy()[3600:4704], flat.astype(np.float32)[3600:4704])


def test_dropin_gnnmodel_autograd(subset5, golden, weights):
    """GNNModel(g) with torch BCE + torch Adam on top (the reference's own loop shape, on cuda)"""
    from mllp_amd.model import GNNModel, build_graph_from_weights_sets
    flat, sd, flat_gpu = weights
    model = GNNModel().to("cuda")
    model.load_flat(flat_gpu)
    inst = [i for i in subset5 if i.name == "afiro.mps"][0]
    name, constrs, w, coefs, rhs, basis = inst.as_reference_tuple()
    g = build_graph_from_weights_sets(constrs, w, rhs, coefs, torch.device("cuda"))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    crit = torch.nn.BCEWithLogitsLoss()
    losses = []
    for _ in range(3):
        z = model(g)
        assert z.shape == (inst.n,)
        obj = crit(z, torch.tensor(basis, dtype=torch.float, device="cuda"))
        obj.backward()
        if not losses:
            got = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                             for p in model.parameters()]).cpu().numpy()
            close(got[grad_mask()], golden["afiro_grads"][grad_mask()], RTOL_GRAD, "autograd grads")
            close(z.detach().cpu().numpy(), golden["afiro_logits"], RTOL_ACT, "logits")
        opt.step()
        opt.zero_grad()
        losses.append(float(obj))
    np.testing.assert_allclose(losses, golden["afiro_adam3_losses"], rtol=1e-5)


def test_synthetic_device_batch_matches_host_build(LPBatch):
    """device generator + device transpose (mllp_graph_create_device) == host build of the same CSR;
    and a size-independent property at scale: A^T (A^T)^T consistency via <A x, y> == <x, A^T y>."""
    from mllp_amd.graph import synthetic_batch
    sb = synthetic_batch(n_inst=3, m=500, n=900, mean_row_nnz=20.0, seed=77, chunk=2)
    ptr, idx, val = sb.export(0), sb.export(1), sb.export(2)
    assert ptr[-1] == sb.nnz and (np.diff(ptr) >= 1).all()
    insts = []
    for k in range(3):
        r0, r1 = k * 500, (k + 1) * 500
        e0, e1 = ptr[r0], ptr[r1]
        cols = idx[e0:e1] - k * 900
        assert cols.min() >= 0 and cols.max() < 900             # block diagonal
        insts.append(LPInstance(f"s{k}", (ptr[r0:r1 + 1] - e0).astype(np.int64), cols.astype(np.int32),
                                val[e0:e1].astype(np.float64), sb.x1[k * 900:(k + 1) * 900].cpu().numpy().astype(np.float64),
                                sb.x2[r0:r1].cpu().numpy().astype(np.float64),
                                sb.labels[k * 900:(k + 1) * 900].cpu().numpy().astype(np.int32)))
    hb = LPBatch.from_instances(insts)
    for w in range(6):
        np.testing.assert_array_equal(sb.export(w), hb.export(w))
    row_norm = np.sqrt(np.add.reduceat(val.astype(np.float64) ** 2, ptr[:-1]))
    np.testing.assert_allclose(row_norm, 1.0, rtol=1e-5)
    x = torch.randn(sb.N, 16, device="cuda")
    y = torch.randn(sb.M, 16, device="cuda")
    lhs = (sb.spmm(x) * y).sum().item()
    rhs = (x * sb.spmm(y, transpose=True)).sum().item()
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)
    sd = o1.init_state(5, torch.float64)
    r = o2.gnn_forward_backward({k: v.numpy() for k, v in sd.items()}, o2.BatchCSR(insts))
    loss, logits, grads = sb.loss_step(o1.flatten_state(sd).float().cuda())
    close(logits.cpu().numpy(), r["logits"], RTOL_ACT, "synthetic logits")
    close(grads.cpu().numpy()[grad_mask()], r["grads"][grad_mask()], RTOL_GRAD, "synthetic grads")


def test_experiment_driver_end_to_end(tmp_path, monkeypatch, golden):
    """python linear_program_experiment.py --cfg <yaml> with batch_size 1: the update sequence is the
    reference's (one Adam step per instance, experiment.py:123-144); outputs: train_log.json, .pt state_dict."""
    import json
    from mllp_amd import experiment
    from mllp_amd.model import GNNModel
    names = ["afiro.mps", "sc50a.mps", "kb2.mps"]
    y = tmp_path / "cfg.yaml"
    y.write_text("train_data_type: 'netlib'\ntrain_lr: 1.e-3\ntrain_iter: 2\nmethods:\n  - 'gs-topk'\n"
                 f"instances: {names}\nbatch_size: 1\nuse_hip_graph: True\n")
    monkeypatch.chdir(tmp_path)
    # the driver seeds torch (set_seed) and builds GNNModel(): capture those initial weights for the oracle
    experiment.set_seed()
    init = GNNModel().flat_parameters().detach().double()
    assert experiment.main(["--cfg", str(y)]) == 0
    log = json.load(open(tmp_path / "train_log.json"))
    assert set(log) == {"obj", "afiro.mps", "kb2.mps", "sc50a.mps"} and len(log["obj"]) == 2
    sd = torch.load(tmp_path / "linear_program_netlib_gs-topk.pt", weights_only=True)
    assert list(sd.keys()) == [k for k, _ in o1.state_dict_spec()]
    # oracle: same loop in fp64 (the yaml's instance order), metrics by torch.topk
    inst = load_packed(names)
    tr = o1.ReferenceTrainer(o1.unflatten_state(init), lr=1e-3, dtype=torch.float64, rebuild_graph=False)
    for epoch in range(2):
        objs = []
        for i in inst:
            loss, z = tr.step(i)
            objs.append(loss)
        np.testing.assert_allclose(log["obj"][epoch], np.mean(objs), rtol=1e-5)
    got = torch.cat([v.reshape(-1) for v in sd.values()]).cpu().numpy()
    keep = grad_mask()
    np.testing.assert_allclose(got[keep], o1.flatten_state(tr.sd).detach().numpy()[keep], rtol=2e-4, atol=3e-6)
    assert all(len(log[n]) == 2 for n in names)                      # one correct-count per instance and epoch


def test_tiled_spmm_equals_generic_and_oracle(LPBatch):
    """LDS-tiled SpMM (row tiles x column blocks, tiles and blocks straddling instances) vs the generic sweep
    and vs scipy, both orientations; unqualified matrices are refused."""
    from mllp_amd.graph import synthetic_batch
    sb = synthetic_batch(n_inst=5, m=700, n=1300, mean_row_nnz=24.0, seed=31, chunk=2)
    rng = np.random.default_rng(2)
    for transpose in (False, True):
        n_in, n_out = (sb.M, sb.N) if transpose else (sb.N, sb.M)
        H = torch.tensor(rng.standard_normal((n_in, 16)).astype(np.float32), device="cuda")
        ref = sb.spmm(H, transpose=transpose).cpu().numpy()
        info = sb.enable_tiled(transpose)
        assert info is not None and info["n_tiles"] == (n_out + info["rows_per_tile"] - 1) // info["rows_per_tile"]
        got = sb.spmm(H, transpose=transpose).cpu().numpy()
        close(got, ref, 2e-6, f"tiled vs generic (transpose={transpose})")
        base = 3 if transpose else 0
        ptr, idx, val = sb.export(base), sb.export(base + 1), sb.export(base + 2)
        close(got, o2.spmm(ptr, idx, val.astype(np.float64), H.cpu().numpy().astype(np.float64)), 2e-6, "tiled vs scipy")
        sb.disable_tiled(transpose)
        again = sb.spmm(H, transpose=transpose).cpu().numpy()
        np.testing.assert_array_equal(again, ref)
    # a (tile, block) segment longer than the LDS window (6144 entries) is processed in several windows
    dense_rows = synthetic_batch(n_inst=1, m=1024, n=1000, mean_row_nnz=40.0, seed=3, chunk=1)
    info = dense_rows.enable_tiled(False)
    assert info is not None and info["max_run"] > 2 * 6144
    H = torch.randn(dense_rows.N, 16, device="cuda")
    got = dense_rows.spmm(H).cpu().numpy()
    dense_rows.disable_tiled(False)
    close(got, dense_rows.spmm(H).cpu().numpy(), 2e-6, "multi-window segment")


def test_bf16_feature_image_spmm(LPBatch):
    """Opt-in `mllp_spmm_csr_bf16` (H as bf16, fp32 accumulate).  Two bounds, both
    documented in include/mllp_hip.h: (a) against the fp32 kernel fed the SAME rounded features it is the same
    arithmetic up to summation order (2e-6 of the tensor's max); (b) against the fp32 product of the unrounded features
    the error is the rounding of H to 8 mantissa bits: 2^-8 of sum |a_ij| |h_j| per element (asserted elementwise).
    Ragged tiles and blocks, both orientations; without the tiled copy the call is refused."""
    from mllp_amd import _lib
    from mllp_amd.graph import synthetic_batch
    sb = synthetic_batch(n_inst=5, m=700, n=1300, mean_row_nnz=24.0, seed=31, chunk=2)
    rng = np.random.default_rng(4)
    for transpose in (False, True):
        n_in = sb.M if transpose else sb.N
        H = torch.tensor(rng.standard_normal((n_in, 16)).astype(np.float32), device="cuda")
        Hb = H.to(torch.bfloat16).contiguous()
        with pytest.raises(_lib.MllpError, match="tiled"):
            sb.spmm_bf16(Hb, transpose=transpose)
        assert sb.enable_tiled(transpose) is not None
        got = sb.spmm_bf16(Hb, transpose=transpose).cpu().numpy()
        same = sb.spmm(Hb.float().contiguous(), transpose=transpose).cpu().numpy()
        close(got, same, 2e-6, f"bf16 image vs fp32 kernel on the rounded features (transpose={transpose})")
        base = 3 if transpose else 0
        ptr, idx, val = sb.export(base), sb.export(base + 1), sb.export(base + 2)
        H64 = H.cpu().numpy().astype(np.float64)
        exact = o2.spmm(ptr, idx, val.astype(np.float64), H64)
        bound = o2.spmm(ptr, idx, np.abs(val).astype(np.float64), np.abs(H64)) * 2.0 ** -8 + 1e-6
        assert np.all(np.abs(got - exact) <= bound), float(np.max(np.abs(got - exact) / bound))
        sb.disable_tiled(transpose)


def test_dropin_batched_graph_equals_per_graph(subset5, weights):
    """GNNModel on BipartiteData.batch([...]) (the __inc__ rule, reference methods.py:68-70) == per-graph calls;
    edge order of the input does not matter (the graph is re-sorted into CSR order when it is built)."""
    from mllp_amd.model import BipartiteData, GNNModel, build_graph_from_weights_sets
    flat, sd, flat_gpu = weights
    model = GNNModel().to("cuda")
    model.load_flat(flat_gpu)
    graphs = []
    for inst in subset5[:3]:
        name, constrs, w, coefs, rhs, basis = inst.as_reference_tuple()
        graphs.append(build_graph_from_weights_sets(constrs, w, rhs, coefs, torch.device("cuda")))
    with torch.no_grad():
        singles = torch.cat([model(g) for g in graphs])
        batched = model(BipartiteData.batch(graphs))
        g0 = graphs[0]
        perm = torch.randperm(g0.edge_index.shape[1], generator=torch.Generator().manual_seed(3)).to("cuda")
        shuffled = model(BipartiteData(g0.edge_index[:, perm], g0.x1, g0.x2, g0.edge_attr[perm]))
    np.testing.assert_allclose(batched.cpu().numpy(), singles.cpu().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(shuffled.cpu().numpy(), singles[:subset5[0].n].cpu().numpy(), rtol=0, atol=0)
    ref = o1.gnn_forward(o1.unflatten_state(torch.tensor(flat)), *o1.batch_graphs(
        [o1.instance_graph(i, torch.float64) for i in subset5[:3]])).numpy()
    close(batched.cpu().numpy(), ref, RTOL_ACT, "batched drop-in logits")


def test_c_abi_argument_errors(LPBatch, subset5):
    """error behaviour of the boundary: negative code + message, nothing launched"""
    from mllp_amd import _lib
    L = _lib.lib()
    b = LPBatch.from_instances(subset5[:1])
    H = torch.zeros(b.N, 16, device="cuda")
    assert L.mllp_spmm_csr_f32(b._h, 0, _lib.ptr(H), None, None) == -1 and b"null" in L.mllp_last_error()
    n = __import__("ctypes").c_int64()
    assert L.mllp_tconv_workspace_floats(b._h, 0, 7, __import__("ctypes").byref(n)) == -1
    assert b"cin must be 1 or 16" in L.mllp_last_error()
    with pytest.raises(AssertionError):
        b.spmm(torch.zeros(b.N + 1, 16, device="cuda"))
    with pytest.raises(_lib.MllpError):
        _lib.check(L.mllp_graph_attach_tiled(b._h, 0, 0, 3, 5, 1, _lib.ptr(H), _lib.ptr(H), _lib.ptr(H), _lib.ptr(H),
                                             _lib.ptr(H)))            # wrong tile count for this graph


def test_tiled_attention_forward_equals_generic_and_oracle(LPBatch, weights):
    """LDS-tiled attention forward sweep (variant 1) vs the generic sweep (h, Z, aux bit-for-bit close) and,
    through the whole model, vs the fp64 oracle; both orientations, tiles straddling instances."""
    from mllp_amd.graph import synthetic_batch
    flat, sd, flat_gpu = weights
    sb = synthetic_batch(n_inst=4, m=600, n=1100, mean_row_nnz=30.0, seed=41, chunk=2)
    rng = np.random.default_rng(4)
    for dst_is_var, off in ((False, 1392), (True, 288)):
        nd, ns = (sb.N, sb.M) if dst_is_var else (sb.M, sb.N)
        cp = flat_gpu[off:off + 1104].contiguous()
        xs = torch.tensor(rng.standard_normal((ns, 16)).astype(np.float32), device="cuda")
        xd = torch.tensor(rng.standard_normal((nd, 16)).astype(np.float32), device="cuda")
        ws_a, ws_b = sb.tconv_workspace(dst_is_var, 16), sb.tconv_workspace(dst_is_var, 16)
        h_ref = sb.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws_a)
        assert sb.enable_tiled(dst_is_var, variant=1) is not None
        h_til = sb.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws_b)
        close(h_til.cpu().numpy(), h_ref.cpu().numpy(), 2e-6, "h tiled vs generic")
        up16 = lambda v: (v + 15) // 16 * 16
        o_ = up16(1088) + up16(nd * 16) + up16(nd)            # skip derived, q', t  (api.cpp::conv_ws_carve)
        Za, Zb = ws_a[o_:o_ + nd * 16].cpu().numpy(), ws_b[o_:o_ + nd * 16].cpu().numpy()
        close(Zb, Za, 2e-6, "Z tiled vs generic")
        o_ += up16(nd * 16)
        close(ws_b[o_:o_ + nd * 4].cpu().numpy(), ws_a[o_:o_ + nd * 4].cpu().numpy(), 2e-6, "aux tiled vs generic")
        # backward consumes what the tiled forward saved
        dh = torch.tensor(rng.standard_normal((nd, 16)).astype(np.float32), device="cuda")
        pg_b = sb.tconv_bwd(dst_is_var, 16, cp, xs, xd, h_til, ws_b, dh)[0].cpu().numpy()
        pg_a = sb.tconv_bwd(dst_is_var, 16, cp, xs, xd, h_ref, ws_a, dh)[0].cpu().numpy()
        keep = np.ones(1104, bool)
        keep[256:272] = False                                   # lin_key.bias
        close(pg_b[keep], pg_a[keep], 2e-5, "param grads after tiled forward")
    # whole model with both tiled attention copies attached
    ptr, idx, val = sb.export(0), sb.export(1), sb.export(2)
    insts = []
    for k in range(4):
        r0, r1 = k * 600, (k + 1) * 600
        e0, e1 = ptr[r0], ptr[r1]
        insts.append(LPInstance(f"s{k}", (ptr[r0:r1 + 1] - e0).astype(np.int64), (idx[e0:e1] - k * 1100).astype(np.int32),
                                val[e0:e1].astype(np.float64), sb.x1[k * 1100:(k + 1) * 1100].cpu().numpy().astype(np.float64),
                                sb.x2[r0:r1].cpu().numpy().astype(np.float64),
                                sb.labels[k * 1100:(k + 1) * 1100].cpu().numpy().astype(np.int32)))
    r = o2.gnn_forward_backward(sd, o2.BatchCSR(insts))
    loss, logits, grads = sb.loss_step(flat_gpu)
    close(logits.cpu().numpy(), r["logits"], RTOL_ACT, "logits with tiled attention forward")
    close(grads.cpu().numpy()[grad_mask()], r["grads"][grad_mask()], RTOL_GRAD, "grads with tiled attention forward")


def test_tiled_attention_backward_src_equals_generic_and_oracle(LPBatch, weights):
    """LDS-tiled source-major attention backward (variant 2: staged 160-byte records) vs the generic gather sweep
    on dx_src (plain and accumulating), both orientations, tiles straddling instances; then the whole model's
    gradients with the tiled forward and backward copies attached vs the fp64 oracle."""
    from mllp_amd.graph import synthetic_batch
    flat, sd, flat_gpu = weights
    sb = synthetic_batch(n_inst=4, m=600, n=1100, mean_row_nnz=30.0, seed=43, chunk=2)
    rng = np.random.default_rng(5)
    for dst_is_var, off in ((False, 1392), (True, 288)):
        nd, ns = (sb.N, sb.M) if dst_is_var else (sb.M, sb.N)
        cp = flat_gpu[off:off + 1104].contiguous()
        xs = torch.tensor(rng.standard_normal((ns, 16)).astype(np.float32), device="cuda")
        xd = torch.tensor(rng.standard_normal((nd, 16)).astype(np.float32), device="cuda")
        dh = torch.tensor(rng.standard_normal((nd, 16)).astype(np.float32), device="cuda")
        ws = sb.tconv_workspace(dst_is_var, 16)
        h = sb.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
        pg_a, dxd_a, dxs_a, _ = sb.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh)
        # the source-major sweep walks the orientation whose rows are the conv's source nodes
        assert sb.enable_tiled(not dst_is_var, variant=2) is not None
        pg_b, dxd_b, dxs_b, _ = sb.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh)
        close(dxs_b.cpu().numpy(), dxs_a.cpu().numpy(), 2e-6, "dx_src tiled vs generic")
        close(dxd_b.cpu().numpy(), dxd_a.cpu().numpy(), 1e-7, "dx_dst unchanged")
        close(pg_b.cpu().numpy(), pg_a.cpu().numpy(), 1e-7, "param grads unchanged")
        assert torch.equal(sb.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh)[2], dxs_b)      # run-to-run bitwise
        # destination-major backward sweep on its own tiled copy (variant 4, the conv's own orientation)
        assert sb.enable_tiled(dst_is_var, variant=4) is not None
        pg_c, dxd_c, dxs_c, _ = sb.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh)
        close(dxd_c.cpu().numpy(), dxd_a.cpu().numpy(), 2e-6, "dx_dst tiled vs generic")
        keep = np.ones(1104, bool)
        keep[256:272] = False                                   # lin_key.bias
        close(pg_c.cpu().numpy()[keep], pg_a.cpu().numpy()[keep], 2e-5, "param grads, tiled destination sweep")
        assert torch.equal(dxs_c, dxs_b)
        assert torch.equal(sb.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh)[1], dxd_c)      # run-to-run bitwise
    ptr, idx, val = sb.export(0), sb.export(1), sb.export(2)
    insts = []
    for k in range(4):
        r0, r1 = k * 600, (k + 1) * 600
        e0, e1 = ptr[r0], ptr[r1]
        insts.append(LPInstance(f"s{k}", (ptr[r0:r1 + 1] - e0).astype(np.int64), (idx[e0:e1] - k * 1100).astype(np.int32),
                                val[e0:e1].astype(np.float64), sb.x1[k * 1100:(k + 1) * 1100].cpu().numpy().astype(np.float64),
                                sb.x2[r0:r1].cpu().numpy().astype(np.float64),
                                sb.labels[k * 1100:(k + 1) * 1100].cpu().numpy().astype(np.int32)))
    assert sb.enable_tiled(False, variant=1) is not None and sb.enable_tiled(True, variant=1) is not None
    r = o2.gnn_forward_backward(sd, o2.BatchCSR(insts))
    loss, logits, grads = sb.loss_step(flat_gpu)
    close(logits.cpu().numpy(), r["logits"], RTOL_ACT, "logits with tiled forward + backward copies")
    close(grads.cpu().numpy()[grad_mask()], r["grads"][grad_mask()], RTOL_GRAD, "grads with tiled forward + backward copies")


def test_tiled_layer1_sweeps_equal_generic_and_oracle(LPBatch, weights):
    """LDS-tiled layer-1 (one channel) attention sweeps (variant 3: lane per row) vs the generic sweeps: forward
    h / Z / aux, backward parameter gradients, both orientations, ragged last column block and tiles straddling
    instances; then the whole model with every tiled copy attached vs the fp64 oracle."""
    from mllp_amd.graph import synthetic_batch
    flat, sd, flat_gpu = weights
    sb = synthetic_batch(n_inst=4, m=600, n=1101, mean_row_nnz=30.0, seed=47, chunk=2)
    rng = np.random.default_rng(6)
    up16 = lambda v: (v + 15) // 16 * 16
    for dst_is_var, off in ((False, 144), (True, 0)):
        nd, ns = (sb.N, sb.M) if dst_is_var else (sb.M, sb.N)
        cp = flat_gpu[off:off + 144].contiguous()
        xs = torch.tensor(rng.standard_normal(ns).astype(np.float32), device="cuda")
        xd = torch.tensor(rng.standard_normal(nd).astype(np.float32), device="cuda")
        dh = torch.tensor(rng.standard_normal((nd, 16)).astype(np.float32), device="cuda")
        ws_a, ws_b = sb.tconv_workspace(dst_is_var, 1), sb.tconv_workspace(dst_is_var, 1)
        h_a = sb.tconv_fwd(dst_is_var, 1, cp, xs, xd, ws_a)
        pg_a = sb.tconv_bwd(dst_is_var, 1, cp, xs, xd, h_a, ws_a, dh)[0]
        assert sb.enable_tiled(dst_is_var, variant=3) is not None
        h_b = sb.tconv_fwd(dst_is_var, 1, cp, xs, xd, ws_b)
        close(h_b.cpu().numpy(), h_a.cpu().numpy(), 2e-6, "layer-1 h tiled vs generic")
        o_ = up16(1088) + up16(nd) + up16(nd)                  # skip derived, q', t  (api.cpp::conv_ws_carve, cin = 1)
        close(ws_b[o_:o_ + nd].cpu().numpy(), ws_a[o_:o_ + nd].cpu().numpy(), 2e-6, "layer-1 Z tiled vs generic")
        o_ += up16(nd)
        close(ws_b[o_:o_ + nd * 4].cpu().numpy(), ws_a[o_:o_ + nd * 4].cpu().numpy(), 2e-6, "layer-1 aux tiled vs generic")
        pg_b = sb.tconv_bwd(dst_is_var, 1, cp, xs, xd, h_b, ws_b, dh)[0]
        keep = np.ones(144, bool)
        keep[16:32] = False                                     # lin_key.bias (cancels in the softmax)
        close(pg_b.cpu().numpy()[keep], pg_a.cpu().numpy()[keep], 2e-5, "layer-1 param grads tiled vs generic")
        assert torch.equal(sb.tconv_fwd(dst_is_var, 1, cp, xs, xd, ws_b), h_b)                  # run-to-run bitwise
    ptr, idx, val = sb.export(0), sb.export(1), sb.export(2)
    insts = []
    for k in range(4):
        r0, r1 = k * 600, (k + 1) * 600
        e0, e1 = ptr[r0], ptr[r1]
        insts.append(LPInstance(f"s{k}", (ptr[r0:r1 + 1] - e0).astype(np.int64), (idx[e0:e1] - k * 1101).astype(np.int32),
                                val[e0:e1].astype(np.float64), sb.x1[k * 1101:(k + 1) * 1101].cpu().numpy().astype(np.float64),
                                sb.x2[r0:r1].cpu().numpy().astype(np.float64),
                                sb.labels[k * 1101:(k + 1) * 1101].cpu().numpy().astype(np.int32)))
    for tr in (False, True):
        for v in (0, 1, 2):
            assert sb.enable_tiled(tr, variant=v) is not None
    r = o2.gnn_forward_backward(sd, o2.BatchCSR(insts))
    loss, logits, grads = sb.loss_step(flat_gpu)
    close(logits.cpu().numpy(), r["logits"], RTOL_ACT, "logits with every tiled copy")
    close(grads.cpu().numpy()[grad_mask()], r["grads"][grad_mask()], RTOL_GRAD, "grads with every tiled copy")


def _holes_instance(seed, m, n):
    """Random LP with empty rows, empty columns and a few dense rows (ragged input for the tiled copies)."""
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(m):
        u = rng.random()
        k = 0 if u < 0.3 else (int(rng.integers(1, 4)) if u < 0.6 else (n // 2 if u > 0.98 else int(rng.poisson(12)) + 1))
        cols = np.sort(rng.choice(n - n // 10, size=min(k, n - n // 10), replace=False)).astype(np.int32)   # last 10 % of the columns stay empty
        rows.append(cols)
    indptr = np.zeros(m + 1, np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.concatenate(rows).astype(np.int32) if indptr[-1] else np.zeros(0, np.int32)
    values = rng.standard_normal(indptr[-1])
    return LPInstance(f"holes{seed}", indptr, indices, values, rng.standard_normal(n), rng.random(m) * 5,
                      (rng.random(n) < 0.37).astype(np.int32))


def test_tiled_copies_edge_cases(LPBatch, weights):
    """Every tiled variant on ragged inputs: empty rows / columns / whole column blocks, dense rows, a 3 x 5 instance,
    tiles straddling instances, and segments longer than the LDS window (several windows per block, odd and even
    item counts).  Whole-model loss step and plain SpMM with all copies attached == generic sweeps == fp64 oracle."""
    from mllp_amd.graph import synthetic_batch
    flat, sd, flat_gpu = weights
    insts = [_holes_instance(1, 700, 900), _holes_instance(2, 3, 5), _holes_instance(3, 515, 2300)]
    b = LPBatch.from_instances(insts)
    H = torch.randn(b.N, 16, device="cuda")
    Ht = torch.randn(b.M, 16, device="cuda")
    y_ref, yt_ref = b.spmm(H).clone(), b.spmm(Ht, transpose=True).clone()
    loss_a, logits_a, grads_a = [t.clone() for t in b.loss_step(flat_gpu)]
    for tr in (False, True):
        for v in (0, 1, 2, 3, 4):
            assert b.enable_tiled(tr, variant=v) is not None
    close(b.spmm(H).cpu().numpy(), y_ref.cpu().numpy(), 2e-6, "A H with holes")
    close(b.spmm(Ht, transpose=True).cpu().numpy(), yt_ref.cpu().numpy(), 2e-6, "At H with holes")
    loss_b, logits_b, grads_b = b.loss_step(flat_gpu)
    close(logits_b.cpu().numpy(), logits_a.cpu().numpy(), 5e-6, "logits, tiled vs generic, ragged batch")
    close(grads_b.cpu().numpy()[grad_mask()], grads_a.cpu().numpy()[grad_mask()], 5e-5, "grads, tiled vs generic, ragged batch")
    r = o2.gnn_forward_backward(sd, o2.BatchCSR(insts))
    close(logits_b.cpu().numpy(), r["logits"], RTOL_ACT, "logits, tiled, ragged batch vs oracle")
    close(grads_b.cpu().numpy()[grad_mask()], r["grads"][grad_mask()], RTOL_GRAD, "grads, tiled, ragged batch vs oracle")
    # multi-window segments in every variant (40 nonzeros per row on 1000 columns)
    d = synthetic_batch(n_inst=1, m=1024, n=1000, mean_row_nnz=40.0, seed=3, chunk=1)
    la, lga, ga = [t.clone() for t in d.loss_step(flat_gpu)]
    for tr in (False, True):
        for v in (0, 1, 2, 3, 4):
            info = d.enable_tiled(tr, variant=v)
            assert info is not None
            if not tr:
                assert info["max_run"] > 3072, (v, info)
    lb, lgb, gb = d.loss_step(flat_gpu)
    close(lgb.cpu().numpy(), lga.cpu().numpy(), 5e-6, "logits, multi-window segments")
    close(gb.cpu().numpy()[grad_mask()], ga.cpu().numpy()[grad_mask()], 5e-5, "grads, multi-window segments")


def test_trainer_attaches_tiled_copies_and_matches_generic(LPBatch, weights):
    """LPTrainer(tiled_copies=True) attaches the re-blocked copies of the training step on first use -- the streamed copies
    of the 16-channel attention sweeps (round 4: geometries 1-3) and the lane-per-row copies (geometry 4) of the layer-1
    sweeps; with stream_copies=False the LDS-tiled variants 1-4 of rounds 2-3; the plain SpMM's copies are not part of the step; the
    `auto` policy does the same for batches of >= 32 M nonzeros -- and trains like the generic path: losses of three Adam
    steps agree."""
    from mllp_amd.graph import synthetic_batch
    from mllp_amd.trainer import LPTrainer
    flat, sd, flat_gpu = weights
    losses = {}
    for mode in (False, True, "tiled"):
        sb = synthetic_batch(n_inst=3, m=700, n=1300, mean_row_nnz=20.0, seed=77, chunk=1)
        tr = LPTrainer(flat_gpu, lr=1e-3, use_hip_graph=False, tiled_copies=bool(mode), stream_copies=mode is True)
        out = []
        for _ in range(3):
            loss, _ = tr.step(sb)
            out.append(float(loss[0]))
        assert bool(getattr(sb, "_tiled", None)) == (mode == "tiled")
        if mode is True:
            assert sb.stream_build_s > 0
            assert sorted(sb._streams) == [(tr_, g_) for tr_ in (False, True) for g_ in (1, 2, 3, 4)]
            assert all(sb.stream_copy_info(tr_, g_)["n_tiles"] > 0 for tr_ in (False, True) for g_ in (1, 2, 3, 4))
        if mode == "tiled":
            assert sorted(sb._tiled) == [(tr_, v) for tr_ in (False, True) for v in (1, 2, 3, 4)]
            assert sb.tiled_build_s > 0 and not getattr(sb, "_streams", None)
        losses[mode] = out
    np.testing.assert_allclose(losses[True], losses[False], rtol=2e-6, atol=0)
    np.testing.assert_allclose(losses["tiled"], losses[False], rtol=2e-6, atol=0)
    auto = LPTrainer(flat_gpu, tiled_copies="auto")
    assert auto.TILED_NNZ_MIN == 32 << 20


def test_throughput_regime_32M_nonzeros(LPBatch, weights):
    """BASELINE.json configs[3] geometry at the size where the library changes regime (>= 32 M nonzeros: LPTrainer
    attaches the LDS-tiled copies of the attention sweeps by itself, graph.cpp::choose_tiers switches to the throughput
    thresholds): 17 x (m = 10 000, n = 20 000).  Streamed copy vs generic for A H and A^T H, scipy on the exported CSR,
    the adjoint identity <A x, y> = <x, A^T y>, and tiled vs generic logits / loss / gradients of the whole training step."""
    import scipy.sparse as sp
    from mllp_amd.graph import synthetic_batch
    from mllp_amd.trainer import LPTrainer
    flat, sd, flat_gpu = weights
    sb = synthetic_batch(n_inst=17, seed=4321)
    assert sb.nnz >= LPTrainer.TILED_NNZ_MIN and not getattr(sb, "_tiled", None)
    g = torch.Generator(device="cuda").manual_seed(3)
    Hn = torch.randn(sb.N, 16, device="cuda", generator=g)
    Hm = torch.randn(sb.M, 16, device="cuda", generator=g)
    Ya, Yat = sb.spmm(Hn), sb.spmm(Hm, transpose=True)                 # generic sweeps
    la, za, ga = [t.clone() for t in sb.loss_step(flat_gpu)]
    A = sp.csr_matrix((sb.export(2).astype(np.float64), sb.export(1), sb.export(0)), shape=(sb.M, sb.N))
    close(Ya.cpu().numpy(), A @ Hn.double().cpu().numpy(), RTOL_ACT, "generic A H vs scipy at 34 M nnz")
    tr = LPTrainer(flat_gpu, lr=1e-3, tiled_copies="auto")
    loss1, logits1 = tr.step(sb)                                       # attaches the copies, then one fused step
    assert not getattr(sb, "_tiled", None) and sorted(sb._streams) == [(t_, g_) for t_ in (False, True) for g_ in (1, 2, 3, 4)]
    close(logits1.cpu().numpy(), za.cpu().numpy(), RTOL_ACT, "trainer logits (tiled) vs generic")
    close(loss1.cpu().numpy(), la.cpu().numpy(), RTOL_ACT, "trainer loss (tiled) vs generic")
    assert sb.build_spmm_copy(False)["n_tiles"] > 0 and sb.build_spmm_copy(True)["n_tiles"] > 0
    Yb, Ybt = sb.spmm(Hn), sb.spmm(Hm, transpose=True)                 # streamed copy (library-owned, device builder)
    close(Yb.cpu().numpy(), Ya.cpu().numpy(), RTOL_ACT, "tiled A H vs generic")
    close(Ybt.cpu().numpy(), Yat.cpu().numpy(), RTOL_ACT, "tiled A^T H vs generic")
    close(Ybt.cpu().numpy(), A.T.tocsr() @ Hm.double().cpu().numpy(), RTOL_ACT, "tiled A^T H vs scipy")
    lhs = float((Yb.double() * Hm.double()).sum())
    rhs = float((Hn.double() * Ybt.double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), float(Yb.double().norm() * Hm.double().norm()) * 1e-3), (lhs, rhs)
    lb, zb, gb = sb.loss_step(flat_gpu)
    close(zb.cpu().numpy(), za.cpu().numpy(), RTOL_ACT, "loss_step logits, tiled vs generic")
    close(lb.cpu().numpy(), la.cpu().numpy(), RTOL_ACT, "loss_step loss, tiled vs generic")
    close(gb.cpu().numpy()[grad_mask()], ga.cpu().numpy()[grad_mask()], RTOL_GRAD, "loss_step grads, tiled vs generic")
    l2, z2, g2 = sb.loss_step(flat_gpu)                                # bitwise run-to-run on the tiled path too
    assert torch.equal(z2, zb) and torch.equal(g2, gb)
    tr.release(sb)
    assert sb.token not in tr._plans


def test_experiment_driver_resume(tmp_path, monkeypatch):
    """2 epochs straight == 1 epoch, then `resume: True` for the second: identical train_log.json,
    bit-identical .pt weights and Adam moments / step counter (kernels are deterministic: no atomics)."""
    import json
    from mllp_amd import experiment
    names = ["afiro.mps", "sc50a.mps", "adlittle.mps", "blend.mps"]
    base = ("train_data_type: 'netlib'\ntrain_lr: 1.e-3\nmethods:\n  - 'gs-topk'\n"
            f"instances: {names}\nbatch_size: 2\n")

    def run(d, iters, resume, save_every=0):
        d.mkdir(exist_ok=True)
        (d / "cfg.yaml").write_text(base + f"train_iter: {iters}\nresume: {resume}\nsave_every: {save_every}\n")
        monkeypatch.chdir(d)
        assert experiment.main(["--cfg", str(d / "cfg.yaml")]) == 0
        ck = torch.load(d / "linear_program_netlib_gs-topk.ckpt", map_location="cpu", weights_only=True)
        sd = torch.load(d / "linear_program_netlib_gs-topk.pt", map_location="cpu", weights_only=True)
        return json.load(open(d / "train_log.json")), ck, sd

    log_s, ck_s, sd_s = run(tmp_path / "straight", 2, False)
    log_1, ck_1, _ = run(tmp_path / "resumed", 1, False, save_every=1)
    assert ck_1["epoch"] == 0 and len(log_1["obj"]) == 1
    log_r, ck_r, sd_r = run(tmp_path / "resumed", 2, True)
    assert log_r == log_s and len(log_r["obj"]) == 2 and all(len(log_r[n]) == 2 for n in names)
    assert ck_r["epoch"] == ck_s["epoch"] == 1
    assert torch.equal(ck_r["params"], ck_s["params"])
    for k in ("m", "v", "state"):
        assert torch.equal(ck_r["opt"][k], ck_s["opt"][k]), k
    assert list(sd_r) == list(sd_s) and all(torch.equal(sd_r[k], sd_s[k]) for k in sd_s)


def test_fused_path_equals_generic_path(LPBatch, weights):
    """The two whole-model paths on the same batches: all 97 Netlib instances (every tier of both: rows of 0 .. 6184
    nonzeros) and a ragged synthetic batch; forward-only logits, loss-step logits / loss / gradients, and the
    backward-from-dlogits entry point that GNNModel's autograd function uses."""
    flat, sd, flat_gpu = weights
    from mllp_amd.graph import synthetic_batch
    inst = load_packed()
    for make in (lambda: LPBatch.from_instances(inst),
                 lambda: LPBatch.from_instances([_holes_instance(11, 300, 700), _holes_instance(12, 1, 40),
                                                 _holes_instance(13, 900, 30)]),
                 lambda: synthetic_batch(n_inst=3, m=700, n=1300, mean_row_nnz=40.0, seed=5, chunk=2)):
        bg, bf = make().set_path(1), make().set_path(2)
        lg, zg, gg = [t.clone() for t in bg.loss_step(flat_gpu)]
        lf, zf, gf = [t.clone() for t in bf.loss_step(flat_gpu)]
        close(zf.cpu().numpy(), zg.cpu().numpy(), RTOL_ACT, "logits fused vs generic")
        close(lf.cpu().numpy(), lg.cpu().numpy(), RTOL_ACT, "loss fused vs generic")
        close(gf.cpu().numpy()[grad_mask()], gg.cpu().numpy()[grad_mask()], RTOL_GRAD, "grads fused vs generic")
        l2, z2, g2 = bf.loss_step(flat_gpu)
        assert torch.equal(z2, zf) and torch.equal(g2, gf) and torch.equal(l2, lf)       # bitwise run to run
        close(bf.forward(flat_gpu).cpu().numpy(), zg.cpu().numpy(), RTOL_ACT, "forward-only logits")
        dz = torch.randn(bf.N, device="cuda") / bf.N
        bg.forward(flat_gpu)
        close(bf.backward(flat_gpu, dz).cpu().numpy()[grad_mask()], bg.backward(flat_gpu, dz).cpu().numpy()[grad_mask()],
              RTOL_GRAD, "backward from dlogits, fused vs generic")


def test_fused_path_follows_in_place_input_changes(LPBatch, subset5, weights):
    """the fused path caches renumbered copies of x1 / x2 / labels keyed on the pointers.  Inputs changed
    in place (same pointers) must be picked up -- through torch's version counters in `LPBatch`, through
    mllp_graph_invalidate_inputs at the C ABI -- and backward refuses a workspace written by the other path."""
    from mllp_amd import _lib
    flat, sd, flat_gpu = weights
    insts = list(subset5)
    bg, bf = LPBatch.from_instances(insts).set_path(1), LPBatch.from_instances(insts).set_path(2)
    z0 = bf.forward(flat_gpu).clone()
    for b in (bg, bf):
        b.x1.mul_(-1.5)                      # same tensors, same addresses, new contents
        b.x2.add_(0.25)
        b.labels.copy_(1.0 - b.labels)
    lg, zg, gg = [t.clone() for t in bg.loss_step(flat_gpu)]
    lf, zf, gf = [t.clone() for t in bf.loss_step(flat_gpu)]
    assert not torch.allclose(zf, z0)
    close(zf.cpu().numpy(), zg.cpu().numpy(), RTOL_ACT, "logits after in-place input change, fused vs generic")
    close(lf.cpu().numpy(), lg.cpu().numpy(), RTOL_ACT, "loss after in-place input change")
    close(gf.cpu().numpy()[grad_mask()], gg.cpu().numpy()[grad_mask()], RTOL_GRAD, "grads after in-place input change")
    # the C ABI itself: without the invalidate call the cached copies are used (documented contract), with it the new ones
    L = _lib.lib()
    logits = torch.empty(bf.N, device="cuda")
    bf.x1.mul_(2.0)
    args = (bf._h, _lib.ptr(flat_gpu), _lib.ptr(bf.x1), _lib.ptr(bf.x2), _lib.ptr(bf.workspace()), _lib.ptr(logits), _lib.current_stream())
    _lib.check(L.mllp_gnn_forward(*args))
    stale = logits.clone()
    _lib.check(L.mllp_graph_invalidate_inputs(bf._h))
    _lib.check(L.mllp_gnn_forward(*args))
    bg.x1.mul_(2.0)
    close(logits.cpu().numpy(), bg.forward(flat_gpu).cpu().numpy(), RTOL_ACT, "C ABI after mllp_graph_invalidate_inputs")
    close(stale.cpu().numpy(), zf.cpu().numpy(), RTOL_ACT, "C ABI without invalidate: the cached inputs (contract)")
    # backward on a workspace that the other path wrote: refused
    dz = torch.randn(bf.N, device="cuda")
    bf.forward(flat_gpu)
    bf.set_path(1)
    with pytest.raises(_lib.MllpError):
        bf.backward(flat_gpu, dz)
    bf.forward(flat_gpu)
    bf.backward(flat_gpu, dz)                # forward and backward on the same path again: fine


def test_device_tiled_builder_layout_and_parity(LPBatch, weights):
    """mllp_graph_build_tiled (tiled_build.hip) against the layout contract of include/mllp_hip.h and the torch
    reference builder: all five arrays bit-identical, entries a permutation of the CSR with the
    right byte offsets (checked like tests/test_hostlogic.py checks the torch builder: ragged last tile, empty rows,
    an empty instance in the middle), and the training step on device-built copies = on torch-built copies."""
    from mllp_amd.graph import synthetic_batch
    from test_hostlogic import _row_entries
    flat, sd, flat_gpu = weights
    sb = synthetic_batch(n_inst=3, m=700, n=1300, mean_row_nnz=12.0, seed=91, chunk=1)
    mult = {0: 64, 1: 64, 2: 160, 3: 4, 4: 64}
    for tr in (False, True):
        ptr, idx, val = (t.cpu().numpy() for t in sb._device_orientation(tr))
        n_dst = sb.N if tr else sb.M
        want = {(r, int(idx[e])): val[e] for r in range(n_dst) for e in range(ptr[r], ptr[r + 1])}
        for v in (0, 1, 2, 3, 4):
            ref = sb.enable_tiled(tr, variant=v, builder="torch")
            t_arr = {k: a.cpu().numpy().copy() for k, a in sb._tiled[(tr, v)].items()}
            info = sb.enable_tiled(tr, variant=v)
            assert info["builder"] == "device" and sb._tiled[(tr, v)] is None
            for k in ("n_tiles", "n_tb", "max_run", "rows_per_tile", "cols_per_block"):
                assert info[k] == ref[k], (tr, v, k)
            d_arr = {k: a.cpu().numpy() for k, a in sb.export_tiled(tr, v).items()}
            for k in ("tile_blk", "blk_id", "ptr2", "perm", "ent"):          # bit for bit, entry order included
                np.testing.assert_array_equal(d_arr[k], t_arr[k], err_msg=f"{k} tr={tr} variant={v}")
            R, CB = info["rows_per_tile"], info["cols_per_block"]
            tile_blk, blk_id, ptr2, perm, ent = (d_arr[k] for k in ("tile_blk", "blk_id", "ptr2", "perm", "ent"))
            assert ent[-1].tolist() == [0, 0]
            got = {}
            for t in range(info["n_tiles"]):
                for tb in range(tile_blk[t], tile_blk[t + 1]):
                    runs = _row_entries(ptr2, tb, R, v)
                    for k in range(R):
                        row = t * R + int(perm[tb * R + k])
                        for e in runs[k]:
                            off = int(ent[e, 0])
                            assert off % mult[v] == 0 and 0 <= off // mult[v] < CB
                            key = (row, int(blk_id[tb]) * CB + off // mult[v])
                            assert key not in got
                            got[key] = ent[e:e + 1, 1].view(np.float32)[0]
            assert got.keys() == want.keys(), (tr, v)
            assert all(got[k] == want[k] for k in want), (tr, v)
    # whole step: device-built copies vs torch-built copies
    out = {}
    for builder in ("torch", "device"):
        b = synthetic_batch(n_inst=3, m=700, n=1300, mean_row_nnz=12.0, seed=91, chunk=1)
        for tr in (False, True):
            for v in (0, 1, 2, 3, 4):
                assert b.enable_tiled(tr, variant=v, builder=builder) is not None
        H = torch.randn(b.N, 16, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
        out[builder] = [t.clone().cpu().numpy() for t in b.loss_step(flat_gpu)] + [b.spmm(H).cpu().numpy()]
    close(out["device"][1], out["torch"][1], 5e-6, "logits, device-built vs torch-built tiled copies")
    close(out["device"][2][grad_mask()], out["torch"][2][grad_mask()], 5e-5, "grads, device-built vs torch-built")
    close(out["device"][3], out["torch"][3], 2e-6, "A H, device-built vs torch-built")
    # a rebuild replaces (and frees) the library-owned arrays; detaching twice is harmless
    assert b.enable_tiled(False, variant=1) is not None
    b.disable_tiled(False, variant=1)
    b.disable_tiled(False, variant=1)
    assert [t.clone().cpu().numpy() for t in b.loss_step(flat_gpu)][1].shape == out["device"][1].shape


def test_mps_files_to_logits_end_to_end(LPBatch, subset5, golden, weights):
    """SURVEY 8f-2 on the GPU box: tests/golden/mps/<name>.mps -> mllp_mps_read (library parser + normalisation) ->
    LPBatch -> logits of the whole model == the committed fp64 golden logits of the same five instances (labels come
    from the packed fixtures: they need an LP solver, which is not part of the path)."""
    from mllp_amd.mps import read_mps
    flat, sd, flat_gpu = weights
    by_name = {i.name: i for i in subset5}
    insts = []
    for name in [str(n) for n in golden["names"]]:
        inst, info = read_mps(os.path.join(ROOT, "tests", "golden", "mps", name), normalize=True, name=name)
        ref = by_name[name]
        assert (inst.m, inst.n, inst.nnz) == (ref.m, ref.n, ref.nnz)
        inst.basis = np.asarray(ref.basis)
        insts.append(inst)
    b = LPBatch.from_instances(insts)
    close(b.forward(flat_gpu).cpu().numpy(), golden["batch_logits"], RTOL_ACT, "logits from the MPS files")
    loss, _, grads = b.loss_step(flat_gpu)
    close(loss.cpu().numpy(), [float(golden["batch_loss"])], RTOL_ACT, "loss from the MPS files")
    close(grads.cpu().numpy()[grad_mask()], golden["batch_grads"][grad_mask()], RTOL_GRAD, "grads from the MPS files")


def test_cfg2_trajectory_100_adam_steps_vs_oracle(LPBatch, subset5, weights):
    """BASELINE.json configs[1] / SURVEY 8d: the 5-instance batch trained for 100 Adam steps at lr 1e-3 (loop shape of
    linear_program_experiment.py:120-144, one step per batch), HIP fp32 against the fp64 oracle trainer
    (tests/oracle_trainer.py: the literal restatement + the same Adam).  The loss curve must agree to 1e-5 relative at
    every step (measured: 2.1e-7).  Weights: Adam divides by sqrt(v), so a parameter whose gradient is rounding noise
    still moves by ~lr per step in a noise-determined direction; after 100 steps such a parameter can have travelled
    100 * lr = 0.1.  The test asks for 1e-4 absolute (0.1 % of that travel) on the parameters that receive gradient
    (measured: 1.7e-5 at most, 6e-8 on average)."""
    from mllp_amd.trainer import LPTrainer
    from oracle_trainer import CpuBatch, OracleTrainer
    flat, sd, flat_gpu = weights
    b = LPBatch.from_instances(subset5)
    tr = LPTrainer(flat_gpu, lr=1e-3, use_hip_graph=False)
    ot = OracleTrainer(torch.from_numpy(np.asarray(flat, dtype=np.float64)), lr=1e-3)
    cb = CpuBatch(subset5)
    got, want = [], []
    for _ in range(100):
        got.append(float(tr.step(b)[0][0]))
        want.append(float(ot.step(cb)[0][0]))
    got, want = np.array(got), np.array(want)
    assert want[-1] < 0.9 * want[0]                                    # it trains
    dev = np.abs(got - want) / np.abs(want)
    print(f"cfg2 trajectory: loss {want[0]:.6f} -> {want[-1]:.6f}, max rel deviation {dev.max():.2e} at step {dev.argmax()}")
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=0)
    keep = grad_mask()
    dw = np.abs(tr.params.cpu().numpy().astype(np.float64) - ot.params.numpy())[keep]
    print(f"cfg2 trajectory: max |dw| {dw.max():.2e}, mean {dw.mean():.2e}")
    assert dw.max() < 1e-4


_HIP_DP_WORKER = """
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from mllp_amd.data import load_packed, SUBSET5
from mllp_amd.graph import LPBatch
from mllp_amd.trainer import LPTrainer, allreduce_sum_, shard_instances
dist.init_process_group("gloo")            # one GPU on the test box: both ranks use cuda:0, gloo carries the all-reduce
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
inst = load_packed(SUBSET5)
golden = np.load(os.path.join({root!r}, "tests", "golden", "subset5.npz"), allow_pickle=False)
params = torch.tensor(golden["weights_flat"], dtype=torch.float32).cuda()       # the `weights` fixture of the tests
out = {{}}
for mode, use_graph in (("eager", False), ("graph", True)):
    shard = [inst[i] for i in shard_instances([i.nnz for i in inst], world)[rank]]
    batch = LPBatch.from_instances(shard)
    tr = LPTrainer(params, lr=1e-3, use_hip_graph=use_graph, global_instances=len(inst))
    losses = []
    for _ in range(4):
        loss, _ = tr.step(batch)
        total = loss.clone()
        allreduce_sum_(total)
        losses.append(float(total[0]))
    # a global batch of ONE instance: rank 1 owns nothing and takes the empty step (experiment.py's per-instance loop)
    one = LPBatch.from_instances([inst[0]])
    tr.global_instances = 1
    for _ in range(2):
        if rank == 0:
            tr.step(one)
        else:
            tr.step_empty()
    assert tr.uses_graph(batch) == use_graph
    out[mode + "_losses"] = np.array(losses)
    out[mode + "_params"] = tr.params.cpu().numpy()
    out[mode + "_adam_steps"] = np.array([float(tr.opt.state[0])])
np.savez({out!r} + str(rank) + ".npz", **out)
dist.barrier()
dist.destroy_process_group()
"""


def test_hip_trainer_two_ranks_equals_one_process(tmp_path, subset5, weights):
    """The HIP LPTrainer with TWO data-parallel ranks: LPT shards of the 5-instance batch, the gradient
    all-reduce between `mllp_gnn_loss_step` and `mllp_adam_step`, the two-graph capture path, and `step_empty` of a
    rank that owns no instance -- against the same steps in one process.  The test box has one GPU, so both ranks run
    their kernels on cuda:0 and gloo carries the 18.9 KB all-reduce (RCCL needs one device per rank; the kernels, the
    sharding and the step logic are the ones the multi-GPU run uses)."""
    import subprocess
    import sys
    from mllp_amd.trainer import LPTrainer
    flat, sd, flat_gpu = weights
    script = tmp_path / "w.py"
    out = str(tmp_path / "dp_rank")
    script.write_text(_HIP_DP_WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29641", str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = [np.load(out + f"{k}.npz") for k in (0, 1)]
    # one process, same schedule
    full = LPBatch_from(subset5)
    one = LPBatch_from([subset5[0]])
    tr = LPTrainer(flat_gpu, lr=1e-3, use_hip_graph=False, global_instances=len(subset5))
    losses = [float(tr.step(full)[0][0]) for _ in range(4)]
    tr.global_instances = 1
    for _ in range(2):
        tr.step(one)
    want = tr.params.cpu().numpy()
    keep = grad_mask()
    for mode in ("eager", "graph"):
        for k in (0, 1):
            np.testing.assert_allclose(got[k][mode + "_losses"], losses, rtol=1e-5, err_msg=f"{mode} rank {k}")
            assert got[k][mode + "_adam_steps"][0] == 6.0
        # replicated weights: identical on both ranks (same all-reduced gradient, same Adam), equal to one process up to
        # the summation order of the two shard gradients
        np.testing.assert_array_equal(got[0][mode + "_params"], got[1][mode + "_params"])
        np.testing.assert_allclose(got[0][mode + "_params"][keep], want[keep], rtol=1e-4, atol=2e-5, err_msg=mode)


def LPBatch_from(instances):
    from mllp_amd.graph import LPBatch as _B
    return _B.from_instances(list(instances))


def test_train_step_equals_loss_step_plus_adam(LPBatch, subset5, weights):
    """mllp_gnn_train_step (one library call per step; on the fused path its tail -- reduce, conv gradients, Adam, the
    next step's folded weights -- is ONE launch with grid barriers) against mllp_gnn_loss_step + mllp_adam_step: the
    same bits in parameters, moments, step counter, loss, logits and gradients over six steps, with the folded weights
    reused (param_gen bookkeeping) and not reused, and after an in-place change of the parameters in between."""
    from mllp_amd.graph import adam_step
    flat, sd, flat_gpu = weights

    def fresh():
        p = flat_gpu.clone()
        return p, torch.zeros_like(p), torch.zeros_like(p), torch.tensor([0.0, 1e-3, 0.9, 0.999], device="cuda")

    b = LPBatch.from_instances(subset5)
    p0, m0, v0, s0 = fresh()
    ref = []
    for k in range(6):
        if k == 3:
            p0.mul_(1.0001)                                       # somebody else writes the parameters
        loss, logits, grads = b.loss_step(p0)
        adam_step(p0, grads, m0, v0, s0, 1e-8)
        ref.append((loss.clone(), logits.clone(), grads.clone(), p0.clone(), m0.clone(), v0.clone(), s0.clone()))
    for reuse in (True, False):
        b2 = LPBatch.from_instances(subset5)
        p1, m1, v1, s1 = fresh()
        gen = 0
        for k in range(6):
            if k == 3:
                p1.mul_(1.0001)                                   # (bumps the tensor's version: the folds are stale)
            loss, logits, grads = b2.train_step(p1, m1, v1, s1, 1e-8, param_gen=gen if reuse else None)
            gen += 1
            if reuse and k not in (0, 3):
                assert b2._folded is not None
            for got, want, what in zip((loss, logits, grads, p1, m1, v1, s1), ref[k],
                                       ("loss", "logits", "grads", "params", "exp_avg", "exp_avg_sq", "state")):
                assert torch.equal(got, want), (what, k, reuse)
    # a forward with other parameters in between overwrites the folds: the next train_step must fold again
    b3 = LPBatch.from_instances(subset5)
    p2, m2, v2, s2 = fresh()
    b3.train_step(p2, m2, v2, s2, 1e-8, param_gen=0)
    b3.forward(flat_gpu * 0.5)
    assert b3._folded is None
    loss, logits, grads = b3.train_step(p2, m2, v2, s2, 1e-8, param_gen=1)
    p3, m3, v3, s3 = fresh()
    b.loss_step(p3); adam_step(p3, b.loss_step(p3)[2], m3, v3, s3, 1e-8)
    want = b.loss_step(p3)
    assert torch.equal(logits, want[1]) and torch.equal(grads, want[2])
    # a path switch in between: the generic branch leaves only pre-Adam folds behind, so a fused train_step
    # after it must fold again -- in the wrapper (set_path drops _folded) AND in the library, which ignores flags bit 0
    # unless its own record says the last call was a fused train_step on this workspace and these parameters
    from ctypes import c_void_p
    from mllp_amd import _lib
    for raw_flag in (False, True):
        b4 = LPBatch.from_instances(subset5)
        p4, m4, v4, s4 = fresh()
        b4.set_path(1)
        b4.train_step(p4, m4, v4, s4, 1e-8, param_gen=0)
        b4.set_path(2)
        assert b4._folded is None
        if raw_flag:        # a caller of the C ABI that claims "folded" after the generic step
            logits4 = torch.empty(b4.N, device="cuda"); loss4 = torch.empty(1, device="cuda")
            grads4 = torch.empty(_lib.NUM_PARAMS, device="cuda")
            _lib.check(_lib.lib().mllp_gnn_train_step(b4._h, _lib.ptr(p4), _lib.ptr(b4.x1), _lib.ptr(b4.x2), _lib.ptr(b4.labels),
                                                      1.0 / b4.n_inst, _lib.ptr(b4.workspace()), _lib.ptr(logits4),
                                                      _lib.ptr(loss4), _lib.ptr(grads4), _lib.ptr(m4), _lib.ptr(v4),
                                                      _lib.ptr(s4), 1e-8, 1, _lib.current_stream()))
        else:
            loss4, logits4, grads4 = b4.train_step(p4, m4, v4, s4, 1e-8, param_gen=1)
        # reference: two plain steps (generic, then fused) with loss_step + adam_step
        b5 = LPBatch.from_instances(subset5)
        p5, m5, v5, s5 = fresh()
        b5.set_path(1); adam_step(p5, b5.loss_step(p5)[2], m5, v5, s5, 1e-8)
        b5.set_path(2); l5, z5, g5 = b5.loss_step(p5); adam_step(p5, g5, m5, v5, s5, 1e-8)
        assert torch.equal(logits4, z5) and torch.equal(grads4, g5) and torch.equal(p4, p5), raw_flag


def test_device_transposition_equals_stable_sort():
    """mllp_csr_transpose_device (transpose.hip: integer-atomic counting and scatter, then every column ordered by row id)
    against the stable sort by column it replaces: row pointers, row ids and values bit-identical -- on a synthetic batch,
    on a matrix with empty rows / empty columns / a dense column of 5 000 entries (the workgroup path for columns longer
    than one wavefront's LDS window) and a dense row, on the empty matrix; and the batches built either way agree."""
    from ctypes import c_void_p
    from mllp_amd import _lib
    from mllp_amd.graph import synthetic_batch

    def torch_transpose(ptr, idx, val, M, N):
        rows = torch.repeat_interleave(torch.arange(M, device="cuda", dtype=torch.int32), (ptr[1:] - ptr[:-1]).long())
        order = torch.sort(idx.long(), stable=True)[1]
        tptr = torch.zeros(N + 1, dtype=torch.int32, device="cuda")
        tptr[1:] = torch.cumsum(torch.bincount(idx.long(), minlength=N), 0).to(torch.int32)
        return tptr, rows[order].contiguous(), val[order].contiguous()

    def lib_transpose(ptr, idx, val, M, N):
        nnz = int(idx.numel())
        tptr = torch.empty(N + 1, dtype=torch.int32, device="cuda")
        tidx = torch.full((max(nnz, 1),), -7, dtype=torch.int32, device="cuda")
        tval = torch.full((max(nnz, 1),), float("nan"), device="cuda")
        _lib.check(_lib.lib().mllp_csr_transpose_device(M, N, nnz, _lib.ptr(ptr), _lib.ptr(idx) if nnz else c_void_p(0),
                                                        _lib.ptr(val) if nnz else c_void_p(0), _lib.ptr(tptr), _lib.ptr(tidx),
                                                        _lib.ptr(tval), _lib.current_stream()))
        return tptr, tidx[:nnz], tval[:nnz]

    rng = np.random.default_rng(3)
    M, N = 6000, 2100
    rows = []
    for r in range(M):
        k = 0 if r % 7 == 3 else int(rng.integers(1, 9))
        c = set(rng.choice(N - 100, size=k, replace=False).tolist())      # (the last 100 columns stay empty, except ...)
        if r < 5000:
            c.add(N - 1)                                                  # ... a dense column of 5 000 entries
        rows.append(np.sort(np.fromiter(c, dtype=np.int64)))
    rows[11] = np.arange(0, N - 100, 1)                                   # a dense row
    ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    idx = np.concatenate(rows).astype(np.int32)
    cases = [(torch.tensor(ptr, device="cuda"), torch.tensor(idx, device="cuda"),
              torch.tensor(rng.standard_normal(len(idx)).astype(np.float32), device="cuda"), M, N)]
    sb = synthetic_batch(n_inst=3, m=700, n=1300, mean_row_nnz=20.0, seed=5, chunk=1)
    cases.append((*sb._device_orientation(False), sb.M, sb.N))
    cases.append((torch.zeros(5, dtype=torch.int32, device="cuda"), torch.zeros(0, dtype=torch.int32, device="cuda"),
                  torch.zeros(0, device="cuda"), 4, 6))
    for p, i, v, m, n in cases:
        want, got = torch_transpose(p, i, v, m, n), lib_transpose(p, i, v, m, n)
        for a, b, what in zip(got, want, ("ptr", "idx", "val")):
            assert torch.equal(a, b), (what, m, n)
        again = lib_transpose(p, i, v, m, n)                               # whatever order the atomics took this time
        assert all(torch.equal(a, b) for a, b in zip(again, got))
    # input the per-column ordering could not keep apart is refused before anything is written (MLLP_EINVAL)
    def dev(a):
        return torch.tensor(a, dtype=torch.int32, device="cuda")
    bad = {"duplicate entry": ([0, 3, 5], [1, 4, 4, 0, 2]), "descending ids": ([0, 3, 5], [1, 5, 4, 0, 2]),
           "id out of range": ([0, 3, 5], [1, 4, 6, 0, 2]), "negative id": ([0, 3, 5], [-1, 4, 5, 0, 2]),
           "row pointers descend": ([0, 4, 3, 5], [0, 1, 2, 3, 4]), "row pointers do not end at nnz": ([0, 3, 4], [1, 4, 5, 0, 2])}
    for what, (p, i) in bad.items():
        p, i = dev(p), dev(i)
        with pytest.raises(_lib.MllpError, match="ascending"):
            lib_transpose(p, i, torch.ones(5, device="cuda"), p.numel() - 1, 6)
    ok = lib_transpose(dev([0, 3, 5]), dev([1, 4, 5, 0, 2]), torch.ones(5, device="cuda"), 2, 6)   # and the library still works
    assert ok[0].tolist() == [0, 1, 2, 3, 3, 4, 5] and ok[1].tolist() == [1, 0, 1, 0, 0]
    # the two ways to build a batch give the same graph arrays
    a = synthetic_batch(n_inst=2, m=300, n=500, mean_row_nnz=10.0, seed=9, chunk=1)
    for k in range(3, 6):
        np.testing.assert_array_equal(a.export(k), np.asarray(torch_transpose(*a._device_orientation(False), a.M, a.N)[k - 3].cpu()))
