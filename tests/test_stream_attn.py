"""GPU parity tests of the STREAMED attention sweeps (mllp_amd/csrc/stream_attn.hip; layout stream_layout.h, geometries 1
and 2), through the C ABI, against the fp64 oracle of the reference's TransformerConv (oracle/spmm_form.py::conv_fwd /
conv_bwd, which equals autograd of the literal PyG form: tests/test_oracle.py) -- reference
linear_program_methods.py:241-247.  Covered: the copies (device builder == host reference builder, bit for bit; decoded,
a permutation of the CSR), single layers in both orientations on ragged batches (empty rows, rows longer than the
register set of a pass: the slow pass), logits beyond the fast pass's +-64 window (the exact redo), determinism, the
whole training step on a > 32 M-nonzero batch against the generic sweeps."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mllp_amd.data import LPInstance, load_packed  # noqa: E402
from oracle import pyg_restatement as o1  # noqa: E402
from oracle import spmm_form as o2  # noqa: E402

RTOL_ACT, RTOL_GRAD = 1e-5, 5e-5


def close(got, want, rtol, what=""):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, what
    assert np.isfinite(got).all(), what
    err = float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-30)
    assert err < rtol, f"{what}: max|diff|/max|ref| = {err:.3e} >= {rtol}"


@pytest.fixture(scope="module")
def LPBatch():
    from mllp_amd import _lib
    _lib.lib()                      # fail loudly: no fallback
    assert torch.cuda.is_available()
    from mllp_amd.graph import LPBatch as cls
    cls.default_path = 1            # the streamed sweeps belong to the generic / tiled path
    yield cls
    cls.default_path = 0


def _ragged_instance(seed, m, n, dense_rows=(), mean=14):
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(m):
        u = rng.random()
        k = 0 if u < 0.2 else (int(rng.integers(1, 4)) if u < 0.45 else int(rng.poisson(mean)) + 1)
        if i in dense_rows:
            k = dense_rows[i]
        hi = max(1, n - n // 10)                               # the last 10 % of the columns stay empty
        rows.append(np.sort(rng.choice(hi, size=min(k, hi), replace=False)).astype(np.int32))
    indptr = np.zeros(m + 1, np.int64)
    indptr[1:] = np.cumsum([len(r) for r in rows])
    indices = np.concatenate(rows).astype(np.int32) if indptr[-1] else np.zeros(0, np.int32)
    values = rng.standard_normal(indptr[-1])
    return LPInstance(f"ragged{seed}", indptr, indices, values, rng.standard_normal(n), rng.random(m) * 5,
                      (rng.random(n) < 0.37).astype(np.int32))


def _decode(copy, info, n_dst):
    """(row, col, value) of every real entry of an exported copy of any geometry, walked the way the wavefronts walk it."""
    tile_blk, blk_id, rows, ent, tile_row, hdr = copy
    rq, item, cb, nw = info["rows_per_quad"], info["item_bytes"], info["cols_per_block"], info["wavefronts"]
    gs = 8 // rq
    zero_off = cb * item
    out_r, out_c, out_v = [], [], []
    q = np.arange(16)
    for t in range(len(tile_row) - 1):
        for tb in range(tile_blk[t], tile_blk[t + 1]):
            assert tb == tile_blk[t] or blk_id[tb] > blk_id[tb - 1]
            for w in range(nw):
                S, cnt, blk, z = (int(v) for v in hdr[tb, w])
                assert blk == blk_id[tb] and z == 0
                n = (cnt & 0xffff, (cnt >> 16) & 0xffff)
                a = S
                for p in range(2):
                    for st in range(a, a + n[p]):
                        g, half = st // gs, st & 1
                        for slot in range(rq):
                            e = ent[g, q * 4 + ((st % gs) >> 1) * rq + slot]          # [16 quads, 3]
                            off = (e[:, 0].astype(np.int64) >> (16 * half)) & 0xffff
                            val = e[:, 1 + half].view(np.float32)
                            rr = rows[tb, w, :, 2 * p + (slot >> 1)]
                            row = tile_row[t] + ((rr >> (16 * (slot & 1))) & 0xffff)
                            real = off != zero_off
                            assert (val[~real] == 0).all()
                            assert (off[real] % item == 0).all() and (off[real] < zero_off).all()
                            assert (row[real] < tile_row[t + 1]).all() and (row[real] < n_dst).all()
                            out_r.append(row[real])
                            out_c.append(blk * cb + off[real] // item)
                            out_v.append(val[real])
                    a += n[p]
    return np.concatenate(out_r), np.concatenate(out_c), np.concatenate(out_v)


@pytest.mark.parametrize("geom", [1, 2, 3])
def test_attention_copies_device_equals_host_and_permute_the_csr(LPBatch, geom):
    insts = [_ragged_instance(1, 700, 900), _ragged_instance(2, 3, 5), _ragged_instance(3, 1300, 2300, {7: 900, 40: 130}),
             LPInstance("empty", np.zeros(6, np.int64), np.zeros(0, np.int32), np.zeros(0), np.zeros(4), np.zeros(5),
                        np.zeros(4, np.int32))]
    b = LPBatch.from_instances(insts)
    for transpose in (False, True):
        info = b.build_stream_copy(transpose, geom, "device")
        assert info["n_tiles"] >= 1 and info["entry_slots"] >= b.nnz and info["rows_per_quad"] == 2
        dev = b.export_stream_copy(transpose, geom)
        b.build_stream_copy(transpose, geom, "host")
        host = b.export_stream_copy(transpose, geom)
        for name, h, d in zip(("tile_blk", "blk_id", "rows", "ent", "tile_row", "hdr"), host, dev):
            assert h.shape == d.shape and np.array_equal(h, d), f"geometry {geom}: device builder differs from the host reference builder in {name}"
        n_dst = b.N if transpose else b.M
        tile_row = dev[4]
        bounds = np.concatenate([[0], np.cumsum(b.inst_n if transpose else b.inst_m)])
        assert (np.diff(tile_row) > 0).all() and (np.diff(tile_row) <= 480).all() and np.isin(bounds, tile_row).all()
        assert info["cols_per_block"] == {1: 720, 2: 312, 3: 432}[geom] and info["item_bytes"] == (160 if geom == 2 else 64)
        base = 3 if transpose else 0
        ptr, idx, val = b.export(base), b.export(base + 1), b.export(base + 2)
        r, c, v = _decode(dev, info, n_dst)
        assert r.size == b.nnz
        want_r = np.repeat(np.arange(n_dst), np.diff(ptr))
        og, ow = np.lexsort((c, r)), np.lexsort((idx, want_r))
        np.testing.assert_array_equal(r[og], want_r[ow])
        np.testing.assert_array_equal(c[og], idx[ow])
        np.testing.assert_array_equal(v[og], val[ow])
        b.drop_stream_copy(transpose, geom)
        assert b.stream_copy_info(transpose, geom)["n_tiles"] == 0


def _layer_case(LPBatch, insts, sd, name, dst_is_var, off, seed, scale_q=1.0, geoms=(1, 2, 3), what="", rtol_grad=RTOL_GRAD):
    """One 16-channel TransformerConv, forward and backward, with the streamed copies attached, against the fp64 oracle
    and against the generic sweeps (same library, no copies)."""
    b = LPBatch.from_instances(insts)
    ob = o2.BatchCSR(insts)
    rng = np.random.default_rng(seed)
    sd = dict(sd)
    if scale_q != 1.0:
        for k in list(sd):
            if k.startswith(name) and ("lin_query" in k or "lin_edge" in k):
                sd[k] = sd[k] * scale_q
    p = o2.conv_params(sd, name)
    ptr, idx, val, nd, ns = ob.orient(dst_is_var)
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    val = r32(val)
    xs, xd, dh = r32(rng.standard_normal((ns, 16))), r32(rng.standard_normal((nd, 16))), r32(rng.standard_normal((nd, 16)))
    h_ref, saved = o2.conv_fwd(p, ptr, idx, val, xs, xd)
    flat = o1.flatten_state({k: torch.tensor(v) for k, v in sd.items()}).float().cuda()
    cp = flat[off:off + 1104].contiguous()
    xs_t = torch.tensor(xs, dtype=torch.float32, device="cuda")
    xd_t = torch.tensor(xd, dtype=torch.float32, device="cuda")
    dh_t = torch.tensor(dh, dtype=torch.float32, device="cuda")
    ws0 = b.tconv_workspace(dst_is_var, 16)
    h0 = b.tconv_fwd(dst_is_var, 16, cp, xs_t, xd_t, ws0)                      # generic sweeps
    pg0, dxd0, dxs0, _ = b.tconv_bwd(dst_is_var, 16, cp, xs_t, xd_t, h0, ws0, dh_t.clone())
    # the copies: geometries 1 and 3 on the destination-major orientation, geometry 2 on the source-major one
    tr_dst = bool(dst_is_var)
    for gm in geoms:
        b.build_stream_copy(tr_dst if gm != 2 else not tr_dst, gm)
    ws = b.tconv_workspace(dst_is_var, 16)
    h = b.tconv_fwd(dst_is_var, 16, cp, xs_t, xd_t, ws)
    h2 = b.tconv_fwd(dst_is_var, 16, cp, xs_t, xd_t, ws)
    assert torch.equal(h, h2), f"{what}: two launches on the same inputs must give identical bits"
    close(h.cpu().numpy(), h_ref, RTOL_ACT, f"{what} h vs fp64 oracle")
    close(h.cpu().numpy(), h0.cpu().numpy(), RTOL_ACT, f"{what} h vs generic sweep")
    grads, dxd, dxs, inter = o2.conv_bwd(p, ptr, idx, val, xs, xd, saved, dh, need_input_grads=True)
    pg, dxd_g, dxs_g, g = b.tconv_bwd(dst_is_var, 16, cp, xs_t, xd_t, h, ws, dh_t.clone())
    close(dxd_g.cpu().numpy(), dxd, rtol_grad, f"{what} dx_dst")
    close(dxs_g.cpu().numpy(), dxs, rtol_grad, f"{what} dx_src")
    pgn, o3 = pg.cpu().numpy(), 0
    for key in ("lin_key.weight", "lin_key.bias", "lin_query.weight", "lin_query.bias", "lin_value.weight",
                "lin_value.bias", "lin_edge.weight", "lin_skip.weight", "lin_skip.bias"):
        ref = np.asarray(grads[key]).reshape(-1)
        if key != "lin_key.bias":
            close(pgn[o3:o3 + ref.size], ref, rtol_grad, f"{what} {key}")
        o3 += ref.size
    return b


@pytest.fixture(scope="module")
def sd9():
    return {k: v.numpy() for k, v in o1.init_state(9, torch.float64).items()}


@pytest.mark.parametrize("name,dst_is_var,off", [("gconv2_w2s", True, 288), ("gconv2_s2w", False, 1392)])
def test_streamed_layer_ragged_batches(LPBatch, sd9, name, dst_is_var, off):
    """Empty rows, rows of 1-3 entries, rows of dozens of entries inside one column block (longer than the register set
    of a pass: the compiled slow pass), tiny instances, an instance without nonzeros, tiles of exactly 480 / 481 rows."""
    dense = {i: 30 + 7 * i for i in range(0, 60, 3)}
    cases = [
        [_ragged_instance(21, 400, 700, dense)],
        [_ragged_instance(22, 900, 450, {i: 18 + i % 25 for i in range(900)})],      # every row long: pass 1 overflows too
        [_ragged_instance(30 + k, 2 + k % 5, 3 + k % 7) for k in range(40)],
        [_ragged_instance(50, 480, 721), _ragged_instance(51, 481, 1100),
         LPInstance("empty", np.zeros(6, np.int64), np.zeros(0, np.int32), np.zeros(0), np.zeros(4), np.zeros(5), np.zeros(4, np.int32))],
    ]
    for k, insts in enumerate(cases):
        _layer_case(LPBatch, insts, sd9, name, dst_is_var, off, seed=100 + k, what=f"case {k} {name}")


def test_streamed_layer_full_netlib(LPBatch, sd9):
    """All 97 Netlib instances (rows of up to 6 184 entries), both orientations."""
    insts = load_packed()
    _layer_case(LPBatch, insts, sd9, "gconv2_w2s", True, 288, seed=7, what="netlib w2s")
    _layer_case(LPBatch, insts, sd9, "gconv2_s2w", False, 1392, seed=8, what="netlib s2w")


def test_streamed_layer_logits_beyond_the_fast_window(LPBatch, sd9):
    """Query / edge weights scaled so that the logits spread over hundreds of log2 units: the fast pass's |l - m| <= 64 window
    is violated, the pass is redone exactly and the row's reference moves (a rare data-dependent path needs an input that
    forces it)."""
    insts = [_ragged_instance(71, 300, 600, mean=30), _ragged_instance(72, 50, 90)]
    # (at scale 60 the logits are a few hundred: their own fp32 rounding, |l| 6e-8, is 2e-5 of a probability, and the
    # gradients sum thousands of them -- the generic sweeps show the same deviation from the fp64 oracle there)
    for scale, rtol in ((8.0, RTOL_GRAD), (60.0, 4 * RTOL_GRAD)):
        _layer_case(LPBatch, insts, sd9, "gconv2_s2w", False, 1392, seed=9, scale_q=scale, what=f"scale {scale}", rtol_grad=rtol)
        _layer_case(LPBatch, insts, sd9, "gconv2_w2s", True, 288, seed=10, scale_q=scale, what=f"scale {scale}", rtol_grad=rtol)


def test_streamed_training_step_at_32M_nonzeros(LPBatch):
    """The throughput regime (SURVEY.md 8d, configs[3] at 17 instances = 34 M nonzeros): loss step with the streamed
    copies against the generic sweeps -- logits, loss, gradients."""
    from mllp_amd.graph import synthetic_batch
    sb = synthetic_batch(17)
    assert sb.nnz > 32 * 2 ** 20
    params = o1.flatten_state(o1.init_state(3, torch.float64)).float().cuda()
    loss0, logits0, grads0 = sb.loss_step(params)
    loss0, logits0, grads0 = loss0.clone(), logits0.clone(), grads0.clone()
    infos = sb.enable_stream_step()
    for (tr, g), i in infos.items():
        assert i["n_tiles"] > 0 and i["entry_slots"] < 1.35 * sb.nnz, (tr, g, i)
    loss, logits, grads = sb.loss_step(params)
    close(logits.cpu().numpy(), logits0.cpu().numpy(), RTOL_ACT, "logits, streamed vs generic")
    close(loss.cpu().numpy(), loss0.cpu().numpy(), RTOL_ACT, "loss")
    keep = np.ones(4721, bool)
    off = 0
    for k, s in o1.state_dict_spec():
        c = int(np.prod(s))
        if k.endswith("lin_key.bias"):
            keep[off:off + c] = False
        off += c
    close(grads.cpu().numpy()[keep], grads0.cpu().numpy()[keep], RTOL_GRAD, "gradients, streamed vs generic")
    again = sb.loss_step(params)
    assert torch.equal(again[1], logits) and torch.equal(again[2], grads)
