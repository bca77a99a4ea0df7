import sys, os, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from mllp_amd.graph import synthetic_batch
from mllp_amd.model import GNNModel, set_seed
sb = synthetic_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
set_seed(42)
params = GNNModel().flat_parameters().detach().float().cuda()
for dst_is_var, off in ((False, 1392), (True, 288)):
    nd, ns = (sb.N, sb.M) if dst_is_var else (sb.M, sb.N)
    cp = params[off:off + 1104].contiguous()
    xs = torch.randn(ns, 16, device="cuda"); xd = torch.randn(nd, 16, device="cuda"); dh = torch.randn(nd, 16, device="cuda")
    ws = sb.tconv_workspace(dst_is_var, 16)
    h = sb.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws).clone()
    ref = [t.clone() for t in sb.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh.clone())[:3]]
    sb.enable_tiled(dst_is_var, variant=1); sb.enable_tiled(not dst_is_var, variant=2); sb.enable_tiled(dst_is_var, variant=4)
    sb.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
    til = [t.clone() for t in sb.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh.clone())[:3]]
    sb.disable_tiled(dst_is_var, variant=1); sb.disable_tiled(not dst_is_var, variant=2); sb.disable_tiled(dst_is_var, variant=4)
    for g in (1, 2, 3): sb.build_stream_copy(dst_is_var if g != 2 else not dst_is_var, g)
    sb.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
    strm = [t.clone() for t in sb.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh.clone())[:3]]
    for g in (1, 2, 3): sb.drop_stream_copy(dst_is_var if g != 2 else not dst_is_var, g)
    for name, r, t, s in zip(("pg", "dxd", "dxs"), ref, til, strm):
        m = r.abs().max()
        print(dst_is_var, name, "tiled-ref", float((t - r).abs().max() / m), "stream-ref", float((s - r).abs().max() / m),
              "stream-tiled", float((s - t).abs().max() / m), "equal bits", bool(torch.equal(s, t)))
