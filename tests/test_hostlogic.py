"""CPU: host-side logic of the drop-in surface -- config, loader, graph build, sharding, flat Adam,
and the data-parallel step (world_size 2 over gloo, gradients from the oracle)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from mllp_amd import config as cfgmod
from mllp_amd import data as datamod
from mllp_amd.trainer import FlatAdam, shard_instances
from oracle import pyg_restatement as o1
from oracle import spmm_form as o2

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_config_roundtrip(tmp_path):
    y = tmp_path / "c.yaml"
    y.write_text("train_data_type: 'netlib'\ntrain_lr: 1.e-3\ntrain_iter: 3\nmethods:\n  - 'gs-topk'\nnest:\n  a: 1\n")
    cfg = cfgmod.cfg_from_file(str(y))
    assert cfg.train_data_type == "netlib" and cfg.train_lr == 1e-3 and cfg.methods[0] == "gs-topk"
    assert cfg.nest.a == 1 and cfg.get_default("batch_size") == 1
    with pytest.raises(AttributeError):
        cfg.missing_key
    with pytest.raises(ValueError, match="Please specify path to the configuration file!"):
        cfgmod.load_config([])
    assert cfgmod.load_config(["--config", str(y)]).train_iter == 3
    base = cfgmod.AttrDict({"train_lr": 0.5, "nest": {"a": 0, "b": 2}})
    cfgmod.cfg_from_file(str(y), base)
    assert base.train_lr == 1e-3 and base.nest.a == 1 and base.nest.b == 2


def test_shipped_yaml_selects_the_sparse_path():
    cfg = cfgmod.cfg_from_file(os.path.join(ROOT, "linear_program_netlib.yaml"))
    assert cfg.train_data_type == "netlib" and cfg.methods[0] in ("gs-topk", "soft-topk")
    assert cfg.train_lr == 1e-3 and cfg.train_iter > 0


def test_loader_contract(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)          # no reference layout here -> packed fixture
    dataset, train_dict = datamod.get_netlib_dataset(normalize=True)
    assert len(dataset) == 97 and list(train_dict)[0] == "obj" and len(train_dict) == 98
    names = [t[0] for t in dataset]
    assert names == sorted(names)
    name, constrs, w, coefs, rhs, basis = dataset[names.index("afiro.mps")]
    assert len(constrs) == 27 == len(rhs) and len(coefs) == 51 == len(basis) and len(w) == 102
    assert constrs[0].dtype == np.int32 and w.dtype == np.float64 and basis.dtype == np.int32
    assert set(np.unique(basis)) <= {0, 1}
    dense, tdict = datamod.get_netlib_dataset_dense(names=["afiro"])       # ONE instance, Q of [A | b]^T
    assert len(dense) == 1 and list(tdict) == ["obj", "afiro.mps"]
    nm, Q, cf, bs = dense[0]
    assert nm == "afiro.mps" and Q.shape == (52, 27) and cf.shape == (52,) and cf[-1] == 0.0 and bs.shape == (51,)
    np.testing.assert_allclose(Q.T @ Q, np.eye(27), atol=1e-12)


def test_loader_reads_reference_layout_when_present(tmp_path, monkeypatch):
    """same files, same naming as reference linear_program_data.py:59-77"""
    import scipy.sparse as sp
    inst = datamod.load_packed(["afiro.mps", "sc50a.mps"])
    os.makedirs(tmp_path / "netlib_mps")
    os.makedirs(tmp_path / "dataset" / "netlib_mps_norm")
    for i in inst:
        (tmp_path / "netlib_mps" / i.name).write_text("")
        base = str(tmp_path / "dataset" / "netlib_mps_norm" / i.name)
        np.save(base + "_basis.npy", i.basis)
        np.save(base + "_coefs.npy", i.coefs)
        np.save(base + "_rhs.npy", i.rhs)
        sp.save_npz(base + "_constrs.npz", sp.csr_matrix((i.values, i.indices, i.indptr), shape=(i.m, i.n)))
    monkeypatch.chdir(tmp_path)
    dataset, _ = datamod.get_netlib_dataset(True)
    assert [t[0] for t in dataset] == ["afiro.mps", "sc50a.mps"]
    for t, i in zip(dataset, inst):
        got = datamod.LPInstance.from_reference_tuple(t)
        np.testing.assert_array_equal(got.indptr, i.indptr)
        np.testing.assert_array_equal(got.indices, i.indices)
        np.testing.assert_array_equal(got.values, i.values)


def test_build_graph_matches_oracle_and_batches(subset5):
    from mllp_amd.model import BipartiteData, build_graph_from_weights_sets
    graphs = []
    for inst in subset5:
        name, constrs, w, coefs, rhs, basis = inst.as_reference_tuple()
        g = build_graph_from_weights_sets(constrs, w, rhs, coefs)
        ei, x1, x2, ea = o1.build_graph_literal(constrs, w, rhs, coefs)
        assert type(g) == BipartiteData
        assert torch.equal(g.edge_index, ei) and torch.equal(g.x1, x1) and torch.equal(g.x2, x2)
        assert torch.equal(g.edge_attr, ea)
        assert g.__inc__("edge_index").tolist() == [[inst.n], [inst.m]]
        graphs.append(g)
    b = BipartiteData.batch(graphs)
    ref = o1.batch_graphs([(g.edge_index, g.x1, g.x2, g.edge_attr) for g in graphs])
    assert torch.equal(b.edge_index, ref[0]) and torch.equal(b.x1, ref[1]) and torch.equal(b.edge_attr, ref[3])


def test_gnnmodel_state_dict_keys_match_pyg_names():
    from mllp_amd.model import GNNModel
    m = GNNModel()
    spec = o1.state_dict_spec()
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _ in spec]
    assert [tuple(v.shape) for v in sd.values()] == [s for _, s in spec]
    assert m.flat_parameters().numel() == 4721
    flat = torch.arange(4721, dtype=torch.float32)
    m.load_flat(flat)
    assert torch.equal(m.flat_parameters(), flat)
    with pytest.raises(Exception):          # no CPU fallback
        from mllp_amd.model import BipartiteData
        m(BipartiteData(torch.zeros(2, 0, dtype=torch.long), torch.zeros(1, 1), torch.zeros(1, 1), torch.zeros(0, 1)))


def test_shard_instances_balanced_and_complete():
    inst = datamod.load_packed()
    sizes = [i.nnz for i in inst]
    for w in (1, 2, 4, 8):
        sh = shard_instances(sizes, w)
        assert sorted(sum(sh, [])) == list(range(len(inst)))
        loads = [sum(sizes[i] for i in s) for s in sh]
        assert max(loads) <= sum(sizes) / w + max(sizes)        # LPT bound
    assert shard_instances(sizes, 8) == shard_instances(sizes, 8)


def test_flat_adam_torch_backend_matches_torch_optim():
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(100, generator=g, dtype=torch.float64)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    mine = FlatAdam(p0.clone(), lr=1e-3, backend="torch")
    for step in range(5):
        gr = torch.randn(100, generator=g, dtype=torch.float64)
        ref.grad = gr.clone()
        opt.step()
        mine.step(gr)
    np.testing.assert_allclose(mine.params.numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-9)   # state[] is fp32
    assert float(mine.state[0]) == 5.0


_DP_WORKER = textwrap.dedent("""
    import os, sys, numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, {root!r})
    from mllp_amd.data import load_packed, SUBSET5
    from mllp_amd.trainer import DataParallelStep, shard_instances
    from oracle import pyg_restatement as o1, spmm_form as o2
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    inst = load_packed(SUBSET5)
    mine = [inst[i] for i in shard_instances([i.nnz for i in inst], world)[rank]]
    params = o1.flatten_state(o1.init_state(42, torch.float64)).clone()
    def grad_fn(p):
        sd = {{k: v.numpy() for k, v in o1.unflatten_state(p).items()}}
        b = o2.BatchCSR(mine)
        b.wnode = b.wnode * len(mine) / len(inst)        # divide by the GLOBAL instance count
        r = o2.gnn_forward_backward(sd, b)
        return torch.tensor([r["loss"]], dtype=torch.float64), torch.tensor(r["grads"])
    dp = DataParallelStep(params, grad_fn, lr=1e-3, adam_backend="torch")
    losses = [float(dp.step()[0]) for _ in range(3)]
    if rank == 0:
        np.savez({out!r}, params=params.numpy(), losses=np.array(losses))
    dist.destroy_process_group()
""")


def test_data_parallel_world2_equals_single_process(tmp_path, subset5):
    """2 gloo ranks, each with its LPT shard of the 5-instance batch: after 3 steps the replicated
    weights equal the 1-process result (sum of per-shard flat grads == batch grads)."""
    out = str(tmp_path / "dp.npz")
    script = tmp_path / "w.py"
    script.write_text(_DP_WORKER.format(root=ROOT, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    params = o1.flatten_state(o1.init_state(42, torch.float64)).numpy().copy()
    m, v, losses = np.zeros_like(params), np.zeros_like(params), []
    for step in (1, 2, 3):
        sd = {k: t.numpy() for k, t in o1.unflatten_state(torch.tensor(params)).items()}
        res = o2.gnn_forward_backward(sd, o2.BatchCSR(subset5))
        losses.append(res["loss"])
        o2.adam_step(params, res["grads"], m, v, step, lr=1e-3)
    np.testing.assert_allclose(got["losses"], losses, rtol=1e-7)   # FlatAdam keeps lr, betas in fp32
    # lin_key.bias gradients are rounding noise (dead parameter); Adam turns noise into O(lr) moves
    keep, off = np.ones(params.size, bool), 0
    for k, s in o1.state_dict_spec():
        c = int(np.prod(s))
        if k.endswith("lin_key.bias"):
            keep[off:off + c] = False
        off += c
    np.testing.assert_allclose(got["params"][keep], params[keep], rtol=1e-5, atol=1e-7)


_LOOP_WORKER = textwrap.dedent("""
    import json, os, sys, numpy as np, torch
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    from mllp_amd.config import AttrDict
    from mllp_amd.data import load_packed, SUBSET5
    from mllp_amd.experiment import plan_batches, run_epochs, save_checkpoint, load_checkpoint
    from oracle import pyg_restatement as o1
    from oracle_trainer import CpuBatch, OracleTrainer
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = 0
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        rank = dist.get_rank()
    os.chdir({cwd!r})
    mode, epochs = sys.argv[1], int(sys.argv[2])
    inst = load_packed(SUBSET5)
    cfg = AttrDict(train_iter=epochs, batch_size=2, log_every=1, save_every=0)
    train_dict = dict(obj=[], **{{i.name: [] for i in inst}})
    tr = OracleTrainer(o1.flatten_state(o1.init_state(42, torch.float64)))
    batches = [(mine, CpuBatch([inst[i] for i in mine]) if mine else None, g)
               for mine, g in plan_batches(inst, 2, rank, world)]
    start = 0
    if mode == "resume":
        start = load_checkpoint("t.ckpt", tr, train_dict, torch.device("cpu"))
    lines = []
    run_epochs(cfg, inst, train_dict, tr, batches, rank, world, torch.device("cpu"), start, lines.append)
    if rank == 0:
        if mode == "first":
            save_checkpoint("t.ckpt", tr, train_dict, epochs - 1)
        np.savez(sys.argv[3], params=tr.params.numpy(), m=tr.opt.m.numpy(), v=tr.opt.v.numpy(),
                 log=json.dumps(train_dict), lines=json.dumps(lines), file_log=open("train_log.json").read())
    else:
        assert lines == [] and not os.path.exists("rank1_wrote_nothing") , lines
    if world > 1:
        dist.destroy_process_group()
""")


def _run_loop(tmp_path, cwd, mode, epochs, out, nproc=1, port=29641):
    script = tmp_path / "loop.py"
    script.write_text(_LOOP_WORKER.format(root=ROOT, cwd=str(cwd)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    if nproc == 1:
        env.pop("WORLD_SIZE", None)
        cmd = [sys.executable, str(script), mode, str(epochs), str(out)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), mode, str(epochs), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


def test_driver_loop_world2_log_equals_single_rank(tmp_path):
    """The experiment driver's epoch loop (shard, step / step_empty, metric gather, log) on 2 gloo ranks writes the
    SAME train_log.json, prints the same lines (rank 0 only) and ends with the same weights as on 1 rank.
    batch_size 2 over 5 instances: the last group has one instance, so one rank takes step_empty()."""
    import json
    d1, d2 = tmp_path / "w1", tmp_path / "w2"
    d1.mkdir(); d2.mkdir()
    one = _run_loop(tmp_path, d1, "straight", 2, tmp_path / "one.npz")
    two = _run_loop(tmp_path, d2, "straight", 2, tmp_path / "two.npz", nproc=2)
    log1, log2 = json.loads(str(one["log"])), json.loads(str(two["log"]))
    assert set(log1) == set(log2) and all(len(v) == 2 for v in log2.values()), log2
    for k in log1:
        if k == "obj":
            np.testing.assert_allclose(log2[k], log1[k], rtol=1e-6)
        else:
            assert log2[k] == log1[k], k          # correct counts of EVERY instance, whichever rank owned it
    assert json.loads(str(two["file_log"])) == log2
    l1, l2 = json.loads(str(one["lines"])), json.loads(str(two["lines"]))
    assert len(l1) == len(l2) == 2 * (5 + 1)
    assert [x for x in l1 if not x.startswith("epoch")] == [x for x in l2 if not x.startswith("epoch")]
    np.testing.assert_allclose(two["params"], one["params"], rtol=1e-9, atol=1e-12)


def test_driver_resume_equals_uninterrupted_run(tmp_path):
    """2 epochs straight == 1 epoch + checkpoint + resume for 1 more: same log, same weights and Adam moments."""
    import json
    d1, d2 = tmp_path / "a", tmp_path / "b"
    d1.mkdir(); d2.mkdir()
    straight = _run_loop(tmp_path, d1, "straight", 2, tmp_path / "s.npz")
    _run_loop(tmp_path, d2, "first", 1, tmp_path / "f.npz")
    resumed = _run_loop(tmp_path, d2, "resume", 2, tmp_path / "r.npz")
    assert json.loads(str(resumed["log"])) == json.loads(str(straight["log"]))
    for k in ("params", "m", "v"):
        np.testing.assert_array_equal(resumed[k], straight[k])


def test_root_shims_and_driver_argument_errors(tmp_path, monkeypatch):
    import config as root_config
    import linear_program_data as root_data
    import linear_program_methods as root_methods
    from mllp_amd import experiment
    assert root_config.load_config is cfgmod.load_config and root_data.get_netlib_dataset is datamod.get_netlib_dataset
    assert root_methods.GNNModel.__name__ == "GNNModel"
    assert root_methods.AngleModel.__name__ == "AngleModel" and callable(root_methods.build_graph_from_Q_sets)
    with pytest.raises(NotImplementedError):
        root_methods.InvariantModel
    with pytest.raises(ValueError, match="Please specify path to the configuration file!"):
        experiment.main([])
    y = tmp_path / "angle.yaml"
    y.write_text("train_data_type: 'netlib'\ntrain_lr: 1.e-3\ntrain_iter: 1\nmethods:\n  - 'invariant'\n")
    with pytest.raises(NotImplementedError, match="invariant"):
        experiment.main(["--cfg", str(y)])
    y.write_text("train_data_type: 'netlib'\ntrain_lr: 1.e-3\ntrain_iter: 1\ndtype: bf16\nmethods:\n  - 'gs-topk'\n")
    with pytest.raises(NotImplementedError, match="bf16"):       # stated, not silently computed in fp32
        experiment.main(["--cfg", str(y)])
    y.write_text("train_data_type: 'twitch'\ntrain_lr: 1.e-3\ntrain_iter: 1\nmethods:\n  - 'gs-topk'\n")
    with pytest.raises(ValueError, match="Unknown training dataset"):
        experiment.main(["--cfg", str(y)])


def _random_csr(rng, m, n, kmax):
    rows = [np.sort(rng.choice(n, size=int(rng.integers(0, kmax)), replace=False)) for _ in range(m)]
    ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    idx = np.concatenate(rows).astype(np.int32) if ptr[-1] else np.zeros(0, np.int32)
    val = rng.standard_normal(len(idx)).astype(np.float32)
    return ptr, idx, val


def _row_entries(ptr2, tb, R, variant):
    """entry indices of every position of (tile, block) tb: contiguous runs (variants 0-3) or, variant 4, entry j of the
    l-th row of a 64-row chunk at chunk_start + sum_l' min(len_l', j) + #{l' < l: len_l' > j} (build_tiled_arrays)"""
    lens = np.diff(ptr2[tb * R:(tb + 1) * R + 1])
    if variant != 4:
        return [list(range(ptr2[tb * R + k], ptr2[tb * R + k + 1])) for k in range(R)]
    out = []
    for c0 in range(0, R, 64):
        ln = lens[c0:c0 + 64]
        start = int(ptr2[tb * R + c0])
        for l in range(64):
            out.append([start + int(np.minimum(ln, j).sum()) + int((ln[:l] > j).sum()) for j in range(int(ln[l]))])
    return out


@pytest.mark.parametrize("variant,R,CB,mult", [(0, 512, 1024, 64), (1, 512, 512, 64), (2, 512, 384, 160), (3, 256, 1024, 4),
                                                (4, 512, 512, 64)])
def test_tiled_reblocking_is_a_permutation_of_the_csr(variant, R, CB, mult):
    """mllp_amd.graph.build_tiled_arrays (the layout mllp_graph_attach_tiled borrows): every nonzero appears exactly
    once with its value, rows of a (tile, block) are ordered by entry count, offsets are consistent, entries carry the
    byte offset of their column's staged item; ragged input (empty rows, a ragged last tile and last column block)."""
    from mllp_amd.graph import build_tiled_arrays
    rng = np.random.default_rng(11 + variant)
    m, n = 1100, 2500
    ptr, idx, val = _random_csr(rng, m, n, 40)
    keep, info = build_tiled_arrays(torch.tensor(ptr), torch.tensor(idx), torch.tensor(val), m, R, CB, variant)
    tile_blk, blk_id, ptr2, perm, ent = (keep[k].numpy() for k in ("tile_blk", "blk_id", "ptr2", "perm", "ent"))
    n_tiles, n_tb = info["n_tiles"], info["n_tb"]
    assert n_tiles == (m + R - 1) // R and tile_blk[0] == 0 and tile_blk[-1] == n_tb and len(blk_id) == n_tb
    assert ptr2[0] == 0 and ptr2[-1] == len(idx) and np.all(np.diff(ptr2) >= 0) and len(perm) == n_tb * R
    got = {}
    for t in range(n_tiles):
        for tb in range(tile_blk[t], tile_blk[t + 1]):
            p = perm[tb * R:(tb + 1) * R]
            assert sorted(p.tolist()) == list(range(R))                       # a permutation of the tile's rows
            lens = np.diff(ptr2[tb * R:(tb + 1) * R + 1])
            if variant == 4:
                assert p.tolist() == list(range(R))                           # a lane owns a row: rows keep their order
            else:
                assert np.all(np.diff(lens) <= 0)                             # longest rows first
            runs = _row_entries(ptr2, tb, R, variant)
            for k in range(R):
                row = t * R + int(p[k])
                for e in runs[k]:
                    off, bits = int(ent[e, 0]), ent[e, 1]
                    assert off % mult == 0 and 0 <= off // mult < CB
                    col = int(blk_id[tb]) * CB + off // mult
                    assert row < m and (row, col) not in got
                    got[(row, col)] = np.array([bits], np.int32).view(np.float32)[0]
    want = {(r, int(idx[e])): val[e] for r in range(m) for e in range(ptr[r], ptr[r + 1])}
    assert got.keys() == want.keys()
    assert all(got[k] == want[k] for k in want)


def test_tiled_joint_entry_order_spreads_bank_quarters():
    """The joint ordering (variants 0 and 1) must beat the per-row ordering on the quantity it is built for: the
    number of LDS cycles of a ds_read_b128 lane group = max multiplicity of (column mod 4) over its four quads."""
    import os
    from mllp_amd.graph import build_tiled_arrays
    rng = np.random.default_rng(5)
    m, n = 1024, 4096
    ptr, idx, val = _random_csr(rng, m, n, 80)
    QG = [[0, 3, 5, 6], [1, 2, 4, 7], [8, 11, 13, 14], [9, 10, 12, 15]]

    def cycles(keep, R):
        ptr2, ent = keep["ptr2"].numpy(), keep["ent"].numpy()
        n_tb = len(keep["blk_id"])
        tot = steps = 0
        for tb in range(n_tb):
            for b in range(R // 16):
                for grp in QG:
                    runs = [ent[ptr2[tb * R + b * 16 + q]:ptr2[tb * R + b * 16 + q + 1], 0] // 64 % 4 for q in grp]
                    for p in range(max(len(r) for r in runs)):
                        cl = [int(r[p]) for r in runs if p < len(r)]
                        tot += max(cl.count(c) for c in range(4))
                        steps += 1
        return tot / steps
    args = (torch.tensor(ptr), torch.tensor(idx), torch.tensor(val), m, 512, 1024, 0)
    joint = cycles(build_tiled_arrays(*args)[0], 512)
    perrow = cycles(build_tiled_arrays(*args, entry_order="perrow")[0], 512)
    assert joint < perrow - 0.1 and joint < 1.6, (joint, perrow)
