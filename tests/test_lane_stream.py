"""GPU parity tests of the layer-1 (one input channel) attention sweeps on the lane-per-row streamed copy
(mllp_amd/csrc/lane_stream.hip; layout lane_layout.h, geometry 4 of mllp_graph_build_stream_copy), through the C ABI,
against the fp64 oracle of the reference's TransformerConv(1, 16, edge_dim=1) (oracle/spmm_form.py::conv_fwd / conv_bwd)
-- reference linear_program_methods.py:90-91, 241-242 -- and against the generic sweeps of the same library.
Covered: the copy (device builder == host reference builder, bit for bit; decoded the way the lanes walk it: a permutation
of the CSR; rows of a tile ordered by length; padding entries), ragged batches (empty rows, rows of 1-3 and of hundreds of
entries, tiny instances, an instance without nonzeros, tiles of exactly 512 / 513 rows), instances wider than one column block (3 blocks per tile), the 97 Netlib
instances, run-to-run determinism."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mllp_amd.data import LPInstance, load_packed  # noqa: E402
from oracle import pyg_restatement as o1  # noqa: E402
from oracle import spmm_form as o2  # noqa: E402
from test_stream_attn import _ragged_instance, close, RTOL_ACT, RTOL_GRAD  # noqa: E402

GEOM = 4


@pytest.fixture(scope="module")
def LPBatch():
    from mllp_amd import _lib
    _lib.lib()                      # fail loudly: no fallback
    assert torch.cuda.is_available()
    from mllp_amd.graph import LPBatch as cls
    cls.default_path = 1            # the streamed sweeps belong to the generic / tiled path
    yield cls
    cls.default_path = 0


@pytest.fixture(scope="module")
def sd9():
    return {k: v.numpy() for k, v in o1.init_state(9, torch.float64).items()}


def _wide_instance(seed, m, n, per_row):
    """Columns spread over the whole width n > 20 000: every tile touches several column blocks."""
    rng = np.random.default_rng(seed)
    rows = [np.sort(rng.choice(n, size=int(rng.integers(0, per_row + 1)), replace=False)).astype(np.int32) for _ in range(m)]
    rows[0] = np.array([0, n - 1], np.int32)
    indptr = np.zeros(m + 1, np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    return LPInstance(f"wide{seed}", indptr, np.concatenate(rows).astype(np.int32), rng.standard_normal(indptr[-1]),
                      rng.standard_normal(n), rng.random(m) * 5, (rng.random(n) < 0.37).astype(np.int32))


def _decode(copy, info, n_dst):
    tile_blk, tile_col, rows, offs, tile_row, whdr, vals = copy
    R, cb, nw = info["row_slots"], info["cols_per_block"], info["wavefronts"]
    out_r, out_c, out_v = [], [], []
    n_pad = 0
    for t in range(len(tile_row) - 1):
        nb = tile_blk[t + 1] - tile_blk[t]
        rr = rows[t]
        nr = tile_row[t + 1] - tile_row[t]
        assert sorted(rr[rr >= 0].tolist()) == list(range(nr)) and (rr[:nr] >= 0).all() and (rr[nr:] == -1).all()
        if nb == 0:
            assert tile_col[t, 1] == -1
            continue
        assert tile_col[t, 0] % 4 == 0 and (tile_col[t, 1] - tile_col[t, 0]) // cb + 1 == nb
        for w in range(nw):
            lane_rows = rr[64 * w:64 * w + 64]
            for b in range(nb):
                g0, ng = (int(v) for v in whdr[tile_blk[t] * nw + w * nb + b])
                if ng == 0:
                    continue
                o = offs[g0:g0 + ng]                                     # [ng, 64, 2]
                o4 = np.stack([o[..., 0] & 0xffff, o[..., 0] >> 16, o[..., 1] & 0xffff, o[..., 1] >> 16], -1).astype(np.int64)
                v4 = vals[g0:g0 + ng]                                    # [ng, 64, 4]
                o4 = o4.transpose(1, 0, 2).reshape(64, -1)               # lane-major: the lane's steps in order
                v4 = v4.transpose(1, 0, 2).reshape(64, -1)
                real = o4 != cb
                assert (v4[~real] == 0).all() and (o4[real] < cb).all()
                assert real.any(axis=0)[: 4 * (ng - 1) + 1].all()         # no group without a real entry in some lane
                # a lane's real entries come first, in ascending column order
                assert (real[:, :-1] >= real[:, 1:]).all()
                assert (lane_rows[real.any(axis=1)] >= 0).all()
                n_pad += int((~real).sum())
                for l in np.nonzero(real.any(axis=1))[0]:
                    k = int(real[l].sum())
                    cols = tile_col[t, 0] + b * cb + o4[l, :k]
                    assert (np.diff(cols) > 0).all()
                    out_r.append(np.full(k, tile_row[t] + lane_rows[l]))
                    out_c.append(cols)
                    out_v.append(v4[l, :k])
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    return cat(out_r, np.int64), cat(out_c, np.int64), cat(out_v, np.float32), n_pad


def test_lane_copy_permutes_the_csr(LPBatch):
    insts = [_ragged_instance(1, 700, 900), _ragged_instance(2, 3, 5), _ragged_instance(3, 1300, 2300, {7: 900, 40: 130}),
             LPInstance("empty", np.zeros(6, np.int64), np.zeros(0, np.int32), np.zeros(0), np.zeros(4), np.zeros(5),
                        np.zeros(4, np.int32)),
             _wide_instance(4, 150, 45011, 40), _ragged_instance(5, 512, 40), _ragged_instance(6, 513, 40)]
    b = LPBatch.from_instances(insts)
    for transpose in (False, True):
        info = b.build_stream_copy(transpose, GEOM)
        assert info["row_slots"] == 512 and info["cols_per_block"] == 20000 and info["wavefronts"] == 8 and info["item_bytes"] == 4
        copy = b.export_stream_copy(transpose, GEOM)
        b.build_stream_copy(transpose, GEOM, "host")                     # the host reference builder: the same bytes
        host = b.export_stream_copy(transpose, GEOM)
        for name, h_, d_ in zip(("tile_blk", "tile_col", "rows", "offs", "tile_row", "whdr", "vals"), host, copy):
            assert h_.shape == d_.shape and np.array_equal(h_, d_), f"device builder differs from the host reference builder in {name}"
        n_dst = b.N if transpose else b.M
        tile_row = copy[4]
        bounds = np.concatenate([[0], np.cumsum(b.inst_n if transpose else b.inst_m)])
        assert (np.diff(tile_row) > 0).all() and (np.diff(tile_row) <= 512).all() and np.isin(bounds, tile_row).all()
        base = 3 if transpose else 0
        ptr, idx, val = b.export(base), b.export(base + 1), b.export(base + 2)
        r, c, v, n_pad = _decode(copy, info, n_dst)
        assert r.size == b.nnz and info["entry_slots"] == b.nnz + n_pad
        if not transpose:
            assert (np.diff(copy[0]) == 3).any()                         # the wide instance: three blocks per tile
        want_r = np.repeat(np.arange(n_dst), np.diff(ptr))
        og, ow = np.lexsort((c, r)), np.lexsort((idx, want_r))
        np.testing.assert_array_equal(r[og], want_r[ow])
        np.testing.assert_array_equal(c[og], idx[ow])
        np.testing.assert_array_equal(v[og], val[ow])
        # rows of a tile are ordered by their number of entries, descending
        lens = np.diff(ptr)
        for t in range(len(tile_row) - 1):
            rr = copy[2][t]
            ln = lens[tile_row[t] + rr[rr >= 0]]
            assert (np.diff(ln) <= 0).all()
        b.drop_stream_copy(transpose, GEOM)
        assert b.stream_copy_info(transpose, GEOM)["n_tiles"] == 0


def _layer1_case(LPBatch, insts, sd, name, dst_is_var, off, seed, what=""):
    b = LPBatch.from_instances(insts)
    ob = o2.BatchCSR(insts)
    rng = np.random.default_rng(seed)
    p = o2.conv_params(sd, name)
    ptr, idx, val, nd, ns = ob.orient(dst_is_var)
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    val = r32(val)
    xs, xd, dh = r32(rng.standard_normal((ns, 1))), r32(rng.standard_normal((nd, 1))), r32(rng.standard_normal((nd, 16)))
    h_ref, saved = o2.conv_fwd(p, ptr, idx, val, xs, xd)
    flat = o1.flatten_state({k: torch.tensor(v) for k, v in sd.items()}).float().cuda()
    cp = flat[off:off + 144].contiguous()
    xs_t = torch.tensor(xs[:, 0], dtype=torch.float32, device="cuda")
    xd_t = torch.tensor(xd[:, 0], dtype=torch.float32, device="cuda")
    dh_t = torch.tensor(dh, dtype=torch.float32, device="cuda")
    ws0 = b.tconv_workspace(dst_is_var, 1)
    h0 = b.tconv_fwd(dst_is_var, 1, cp, xs_t, xd_t, ws0)                       # generic sweeps
    pg0 = b.tconv_bwd(dst_is_var, 1, cp, xs_t, xd_t, h0, ws0, dh_t)[0]
    info = b.build_stream_copy(bool(dst_is_var), GEOM)
    assert info["n_tiles"] > 0
    ws = b.tconv_workspace(dst_is_var, 1)
    h = b.tconv_fwd(dst_is_var, 1, cp, xs_t, xd_t, ws)
    assert torch.equal(h, b.tconv_fwd(dst_is_var, 1, cp, xs_t, xd_t, ws)), f"{what}: two launches must give identical bits"
    close(h.cpu().numpy(), h_ref, RTOL_ACT, f"{what} h vs fp64 oracle")
    close(h.cpu().numpy(), h0.cpu().numpy(), 2e-6, f"{what} h vs generic sweep")
    up16 = lambda v: (v + 15) // 16 * 16
    o_ = up16(1088) + up16(nd) + up16(nd)                      # skip derived, q', t  (api.cpp::conv_ws_carve, cin = 1)
    close(ws[o_:o_ + nd].cpu().numpy(), ws0[o_:o_ + nd].cpu().numpy(), 2e-6, f"{what} Z vs generic")
    o_ += up16(nd)
    close(ws[o_:o_ + nd * 4].cpu().numpy(), ws0[o_:o_ + nd * 4].cpu().numpy(), 2e-6, f"{what} aux vs generic")
    grads, _, _, inter = o2.conv_bwd(p, ptr, idx, val, xs, xd, saved, dh, need_input_grads=False)
    pg = b.tconv_bwd(dst_is_var, 1, cp, xs_t, xd_t, h, ws, dh_t)[0]
    assert torch.equal(pg, b.tconv_bwd(dst_is_var, 1, cp, xs_t, xd_t, h, ws, dh_t)[0])
    pgn, pg0n, o3 = pg.cpu().numpy(), pg0.cpu().numpy(), 0
    for key in ("lin_key.weight", "lin_key.bias", "lin_query.weight", "lin_query.bias", "lin_value.weight",
                "lin_value.bias", "lin_edge.weight", "lin_skip.weight", "lin_skip.bias"):
        ref = np.asarray(grads[key]).reshape(-1)
        if key != "lin_key.bias":
            close(pgn[o3:o3 + ref.size], ref, RTOL_GRAD, f"{what} {key} vs fp64 oracle")
            close(pgn[o3:o3 + ref.size], pg0n[o3:o3 + ref.size], 2e-5, f"{what} {key} vs generic")
        o3 += ref.size


@pytest.mark.parametrize("name,dst_is_var,off", [("gconv1_w2s", True, 0), ("gconv1_s2w", False, 144)])
def test_lane_layer1_ragged_and_wide_batches(LPBatch, sd9, name, dst_is_var, off):
    dense = {i: 30 + 7 * i for i in range(0, 60, 3)}
    cases = [
        [_ragged_instance(21, 400, 700, dense)],
        [_ragged_instance(30 + k, 2 + k % 5, 3 + k % 7) for k in range(40)],
        [_ragged_instance(50, 512, 721), _ragged_instance(51, 513, 1100),
         LPInstance("empty", np.zeros(6, np.int64), np.zeros(0, np.int32), np.zeros(0), np.zeros(4), np.zeros(5), np.zeros(4, np.int32))],
        [_wide_instance(60, 700, 45011, 60), _ragged_instance(61, 90, 130)],
    ]
    for k, insts in enumerate(cases):
        _layer1_case(LPBatch, insts, sd9, name, dst_is_var, off, seed=200 + k, what=f"case {k} {name}")


def test_lane_layer1_full_netlib(LPBatch, sd9):
    """All 97 Netlib instances (rows of up to 6 184 entries next to rows of one), both orientations."""
    insts = load_packed()
    _layer1_case(LPBatch, insts, sd9, "gconv1_w2s", True, 0, seed=7, what="netlib w2s")
    _layer1_case(LPBatch, insts, sd9, "gconv1_s2w", False, 144, seed=8, what="netlib s2w")


def test_skewed_batch_does_not_keep_a_padded_copy(LPBatch):
    """A batch in which every tile has ONE row far longer than the other 63 of its wavefront would cost the lane-per-row copy
    64 slots per entry of that row: `enable_stream_step` drops a copy that needs more than two slots per nonzero, and the
    layer-1 conv then runs on the generic sweeps (same results)."""
    rng = np.random.default_rng(3)
    m, n = 600, 30000
    rows = [np.sort(rng.choice(n, size=2, replace=False)).astype(np.int32) for _ in range(m)]
    for r in range(0, m, 64):
        rows[r] = np.sort(rng.choice(n, size=4000, replace=False)).astype(np.int32)
    indptr = np.zeros(m + 1, np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    inst = LPInstance("skewed", indptr, np.concatenate(rows).astype(np.int32), rng.standard_normal(indptr[-1]),
                      rng.standard_normal(n), rng.random(m) * 5, (rng.random(n) < 0.37).astype(np.int32))
    b = LPBatch.from_instances([inst])
    raw = b.build_stream_copy(False, GEOM)
    assert raw["entry_slots"] > 2 * b.nnz
    infos = b.enable_stream_step()
    assert infos[(False, GEOM)].get("dropped") and b.stream_copy_info(False, GEOM)["n_tiles"] == 0
    for (tr, g), i in infos.items():            # the rule, for every copy of the step
        kept = b.stream_copy_info(tr, g)["n_tiles"] > 0
        assert kept == (i["entry_slots"] <= 2 * b.nnz) and kept == (not i.get("dropped")), (tr, g, i)
