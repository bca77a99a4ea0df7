import sys, numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from mllp_amd.graph import LPBatch
from oracle import pyg_restatement as o1, spmm_form as o2
from test_stream_attn import _ragged_instance
LPBatch.default_path = 1
sd = {k: v.numpy() for k, v in o1.init_state(9, torch.float64).items()}
insts = [_ragged_instance(21, 40, 70)]
name, dst_is_var, off = "gconv2_w2s", True, 288
b = LPBatch.from_instances(insts)
ob = o2.BatchCSR(insts)
rng = np.random.default_rng(3)
p = o2.conv_params(sd, name)
ptr, idx, val, nd, ns = ob.orient(dst_is_var)
r32 = lambda a: a.astype(np.float32).astype(np.float64)
val = r32(val)
xs, xd, dh = r32(rng.standard_normal((ns, 16))), r32(rng.standard_normal((nd, 16))), r32(rng.standard_normal((nd, 16)))
h_ref, saved = o2.conv_fwd(p, ptr, idx, val, xs, xd)
grads, dxd, dxs, inter = o2.conv_bwd(p, ptr, idx, val, xs, xd, saved, dh, need_input_grads=True)
flat = o1.flatten_state({k: torch.tensor(v) for k, v in sd.items()}).float().cuda()
cp = flat[off:off + 1104].contiguous()
xs_t = torch.tensor(xs, dtype=torch.float32, device="cuda"); xd_t = torch.tensor(xd, dtype=torch.float32, device="cuda")
dh_t = torch.tensor(dh, dtype=torch.float32, device="cuda")
for geoms in ((), (2,)):
    for gm in geoms: b.build_stream_copy(not dst_is_var, gm)
    ws = b.tconv_workspace(dst_is_var, 16)
    h = b.tconv_fwd(dst_is_var, 16, cp, xs_t, xd_t, ws)
    pg, dxd_g, dxs_g, g = b.tconv_bwd(dst_is_var, 16, cp, xs_t, xd_t, h, ws, dh_t.clone())
    got = dxs_g.cpu().numpy()
    err = np.abs(got - dxs).max(axis=1) / np.abs(dxs).max()
    print("geoms", geoms, "max err", err.max())
    if geoms:
        bad = np.where(err > 1e-4)[0]
        print("bad rows", bad[:40], "of", ns)
        deg = np.bincount(idx, minlength=ns)
        print("deg of bad", deg[bad][:40]); print("deg of good", deg[np.where(err <= 1e-4)[0]][:40])
        r = bad[0] if len(bad) else 0
        print("row", r, "got", got[r][:8], "\nwant", dxs[r][:8], "\nratio", got[r][:8] / dxs[r][:8])
