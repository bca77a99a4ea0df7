"""GPU parity tests of the streamed SpMM copy (mllp_amd/csrc/stream_layout.h, stream_spmm.hip, stream_build.hip):
the HIP kernel, through the C ABI, against the fp64 CSR product (oracle/spmm_form.py::spmm) and the generic sweep, on
the Netlib batch, on ragged batches (empty rows / columns / blocks, rows denser than the kernel's register set: the slow
path, tiles that end inside a block range, instances smaller than a tile) and on a > 32 M-nonzero synthetic batch;
the device builder's arrays against the host reference builder's (bit for bit), and the exported copy decoded in
numpy against the CSR it came from (a permutation of the nonzeros + padding entries)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mllp_amd.data import LPInstance, load_packed  # noqa: E402
from oracle import spmm_form as o2  # noqa: E402


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
    return cls


def _ragged_instance(seed, m, n, dense_rows=()):
    """empty rows, very short rows, a few rows denser than a column block's register set, empty trailing columns"""
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(m):
        u = rng.random()
        k = 0 if u < 0.25 else (int(rng.integers(1, 4)) if u < 0.55 else int(rng.poisson(14)) + 1)
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


def _decode(copy, n_dst, rows_per_tile, cols_per_block, waves):
    """(row, col, value) of every real entry of an exported copy, walked the way the kernel's wavefronts walk it;
    vectorised over quads and steps.  Geometry from the layout (stream_layout.h): 2 passes, 4 row slots, 2-step groups."""
    tile_blk, blk_id, rows, ent, tile_row, hdr = copy
    zero_off = cols_per_block * 64
    out_r, out_c, out_v = [], [], []
    q = np.arange(16)
    for t in range(len(tile_row) - 1):
        for tb in range(tile_blk[t], tile_blk[t + 1]):
            assert tb == tile_blk[t] or blk_id[tb] > blk_id[tb - 1]
            for w in range(waves):
                S, cnt, blk, z = (int(v) for v in hdr[tb, w])
                assert blk == blk_id[tb] and z == 0
                n = (cnt & 0xffff, (cnt >> 16) & 0xffff)
                a = S
                for p in range(2):
                    for st in range(a, a + n[p]):
                        g, half = st >> 1, st & 1
                        for slot in range(4):
                            e = ent[g, q * 4 + slot]                                  # [16 quads, 3]
                            off = (e[:, 0].astype(np.int64) >> (16 * half)) & 0xffff
                            val = e[:, 1 + half].view(np.float32)
                            rr = rows[tb, w, :, 2 * p + (slot >> 1)]
                            row = tile_row[t] + ((rr >> (16 * (slot & 1))) & 0xffff)
                            real = off != zero_off
                            assert (val[~real] == 0).all()
                            assert (off[real] % 64 == 0).all() and (off[real] < zero_off).all()
                            assert (row[real] < tile_row[t + 1]).all() and (row[real] < n_dst).all()
                            out_r.append(row[real])
                            out_c.append(blk * cols_per_block + off[real] // 64)
                            out_v.append(val[real])
                    a += n[p]
    return np.concatenate(out_r), np.concatenate(out_c), np.concatenate(out_v)


def _check_orientation(b, transpose, rng, rtol=2e-6, compare_host=True):
    n_in, n_out = (b.M, b.N) if transpose else (b.N, b.M)
    H = torch.tensor(rng.standard_normal((n_in, 16)).astype(np.float32), device="cuda")
    b.drop_spmm_copy(transpose)
    ref_generic = b.spmm(H, transpose=transpose).cpu().numpy()
    base = 3 if transpose else 0
    ptr, idx, val = b.export(base), b.export(base + 1), b.export(base + 2)
    want = o2.spmm(ptr, idx, val.astype(np.float64), H.cpu().numpy().astype(np.float64))
    info = b.build_spmm_copy(transpose, "device")
    assert info["n_tiles"] >= 1 and info["entry_slots"] >= b.nnz
    dev = b.export_spmm_copy(transpose)
    got = b.spmm(H, transpose=transpose)
    again = b.spmm(H, transpose=transpose)
    assert torch.equal(got, again), "two launches on the same inputs must give identical bits"
    close(got.cpu().numpy(), want, rtol, f"streamed copy vs fp64 CSR product (transpose={transpose})")
    close(got.cpu().numpy(), ref_generic, rtol, f"streamed copy vs generic sweep (transpose={transpose})")
    if compare_host:
        b.build_spmm_copy(transpose, "host")
        host = b.export_spmm_copy(transpose)
        for name, h, d in zip(("tile_blk", "blk_id", "rows", "ent", "tile_row", "hdr"), host, dev):
            assert h.shape == d.shape and np.array_equal(h, d), f"device builder differs from the host reference builder in {name}"
        close(b.spmm(H, transpose=transpose).cpu().numpy(), want, rtol, "host-built copy")
    b.drop_spmm_copy(transpose)
    np.testing.assert_array_equal(b.spmm(H, transpose=transpose).cpu().numpy(), ref_generic)     # back on the generic sweep
    return info, dev, (ptr, idx, val)


def test_full_netlib_batch_streamed_spmm(LPBatch):
    """All 97 Netlib instances as one block-diagonal batch (1.07 M nonzeros; instance sizes from 27 rows to 16 k, rows
    of up to 6 184 entries: d6cube's long rows run the kernel's slow path), both orientations."""
    b = LPBatch.from_instances(load_packed())
    rng = np.random.default_rng(5)
    for transpose in (False, True):
        info, dev, _ = _check_orientation(b, transpose, rng)
        # tiles never cross an instance boundary and hold at most 960 rows
        tile_row = dev[4]
        bounds = np.concatenate([[0], np.cumsum(b.inst_n if transpose else b.inst_m)])
        assert tile_row[0] == 0 and tile_row[-1] == (b.N if transpose else b.M)
        assert (np.diff(tile_row) > 0).all() and (np.diff(tile_row) <= 960).all()
        assert np.isin(bounds, tile_row).all()


def test_streamed_copy_is_a_permutation_of_the_csr(LPBatch):
    """The exported copy, decoded in numpy, lists every nonzero exactly once (and nothing else but padding)."""
    insts = [_ragged_instance(1, 700, 900), _ragged_instance(2, 3, 5), _ragged_instance(3, 1300, 2300, {7: 900, 40: 130})]
    b = LPBatch.from_instances(insts)
    rng = np.random.default_rng(11)
    for transpose in (False, True):
        info, dev, (ptr, idx, val) = _check_orientation(b, transpose, rng)
        n_dst = b.N if transpose else b.M
        r, c, v = _decode(dev, n_dst, info["rows_per_tile"], info["cols_per_block"], info["wavefronts"])
        assert r.size == b.nnz
        want_r = np.repeat(np.arange(n_dst), np.diff(ptr))
        order_got = np.lexsort((c, r))
        order_want = np.lexsort((idx, want_r))
        np.testing.assert_array_equal(r[order_got], want_r[order_want])
        np.testing.assert_array_equal(c[order_got], idx[order_want])
        np.testing.assert_array_equal(v[order_got], val[order_want])


def test_streamed_spmm_edge_cases(LPBatch):
    """Rows denser than the register set of a pass (slow path in pass 0 and in pass 1), a batch of many tiny instances
    (every tile a few rows), one instance of exactly 960 and one of 961 rows (one tile / two tiles), a source side that
    is not a multiple of the 750-column block, and an instance without any nonzero."""
    rng = np.random.default_rng(3)
    dense = {i: 60 + 9 * i for i in range(0, 40, 3)}                 # 60 .. 180 entries inside one 750-column block
    cases = [
        [_ragged_instance(21, 400, 700, dense)],
        [_ragged_instance(22, 900, 700, {i: 20 + i % 25 for i in range(900)})],      # every row long: pass 1 overflows too
        [_ragged_instance(30 + k, 2 + k % 5, 3 + k % 7) for k in range(40)],
        [_ragged_instance(50, 960, 751), _ragged_instance(51, 961, 1500)],
        [_ragged_instance(60, 50, 80), LPInstance("empty", np.zeros(6, np.int64), np.zeros(0, np.int32), np.zeros(0),
                                                  np.zeros(4), np.zeros(5), np.zeros(4, np.int32)), _ragged_instance(61, 20, 30)],
    ]
    for insts in cases:
        b = LPBatch.from_instances(insts)
        for transpose in (False, True):
            _check_orientation(b, transpose, rng)


def test_streamed_spmm_at_32M_nonzeros(LPBatch):
    """The throughput regime (SURVEY.md 8d, configs[3] at 17 instances = 34 M nonzeros): streamed copy, device builder,
    both orientations, against the generic sweep; the adjoint identity <A H, G> = <H, A^T G> ties the two orientations."""
    from mllp_amd.graph import synthetic_batch
    sb = synthetic_batch(17)
    assert sb.nnz > 32 * 2 ** 20
    g = torch.Generator(device="cuda").manual_seed(5)
    H = torch.randn(sb.N, 16, device="cuda", generator=g)
    G = torch.randn(sb.M, 16, device="cuda", generator=g)
    ref, ref_t = sb.spmm(H).clone(), sb.spmm(G, transpose=True).clone()
    ia, iat = sb.build_spmm_copy(False), sb.build_spmm_copy(True)
    assert ia["entry_slots"] < 1.15 * sb.nnz and iat["entry_slots"] < 1.15 * sb.nnz       # padding of the step layout
    assert ia["bytes"] < 7.5 * sb.nnz                                                      # 6 bytes per slot + records
    Y, Yt = sb.spmm(H), sb.spmm(G, transpose=True)
    close(Y.cpu().numpy(), ref.cpu().numpy(), 2e-6, "A H, streamed vs generic, 34 M nonzeros")
    close(Yt.cpu().numpy(), ref_t.cpu().numpy(), 2e-6, "At G, streamed vs generic, 34 M nonzeros")
    # (2.7 M products of both signs cancel down to a sum of a few hundred: the fp32 rounding of the elements of Y and Yt,
    # ~1e-7 relative each, adds up to ~1e-6 of that SUM; 5e-6 leaves room for the summation order, which the element-wise
    # checks above do not depend on)
    lhs, rhs = float((Y.double() * G.double()).sum()), float((H.double() * Yt.double()).sum())
    assert abs(lhs - rhs) <= 5e-6 * max(abs(lhs), abs(rhs), 1.0)


def test_streamed_copy_argument_errors(LPBatch):
    from ctypes import c_int64, c_void_p
    from mllp_amd import _lib
    L = _lib.lib()
    b = LPBatch.from_instances([_ragged_instance(70, 30, 40)])
    assert L.mllp_graph_build_spmm_copy(None, 0, 0, None) == -1
    assert L.mllp_graph_build_spmm_copy(b._h, 0, 7, None) == -1 and b"where" in L.mllp_last_error()
    buf = np.zeros(4, np.int32)
    assert L.mllp_graph_export_spmm_copy(b._h, 0, 0, buf.ctypes.data_as(c_void_p), buf.nbytes) == -1   # no copy yet
    b.build_spmm_copy(False)
    assert L.mllp_graph_export_spmm_copy(b._h, 0, 9, buf.ctypes.data_as(c_void_p), buf.nbytes) == -1
    assert L.mllp_graph_export_spmm_copy(b._h, 0, 3, buf.ctypes.data_as(c_void_p), 4) == -1             # too small
    d = (c_int64 * 8)()
    assert L.mllp_graph_spmm_copy_info(b._h, 0, d) == 0 and d[0] >= 1
    b.drop_spmm_copy(False)
    assert L.mllp_graph_spmm_copy_info(b._h, 0, d) == 0 and d[0] == 0
