"""Debug: streamed backward vs generic at growing batch sizes -- which rows differ?"""
import sys, os, numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from mllp_amd.graph import synthetic_batch
from mllp_amd.model import GNNModel, set_seed
n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 128
b = synthetic_batch(n_inst)
set_seed(42)
params = GNNModel().flat_parameters().detach().float().cuda()
dst_is_var, off = False, 1392
nd, ns = b.M, b.N
cp = params[off:off + 1104].contiguous()
g = torch.Generator(device="cuda").manual_seed(1)
xs = torch.randn(ns, 16, device="cuda", generator=g); xd = torch.randn(nd, 16, device="cuda", generator=g)
dh = torch.randn(nd, 16, device="cuda", generator=g)
ws = b.tconv_workspace(dst_is_var, 16)
h0 = b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws).clone()
ref = [t.clone() for t in b.tconv_bwd(dst_is_var, 16, cp, xs, xd, h0, ws, dh.clone())[:3]]
for geoms in ((3,), (2,), (1, 2, 3)):
    for gm in geoms:
        b.build_stream_copy(dst_is_var if gm != 2 else not dst_is_var, gm)
    h = b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
    got = b.tconv_bwd(dst_is_var, 16, cp, xs, xd, h, ws, dh.clone())[:3]
    for name, a, r in zip(("pg", "dx_dst", "dx_src"), got, ref):
        err = (a - r).abs()
        print(geoms, name, "maxrel", float(err.max() / r.abs().max()), "nan", int(torch.isnan(a).sum()))
        if a.dim() == 2:
            rows = (err.max(dim=1).values > 1e-4 * r.abs().max()).nonzero().flatten()
            print("   bad rows", rows.numel(), rows[:12].tolist(), "of", a.shape[0])
            if rows.numel():
                per_inst = a.shape[0] // n_inst
                print("   instance / row-in-instance", [(int(x) // per_inst, int(x) % per_inst) for x in rows[:12]])
    for gm in geoms:
        b.drop_stream_copy(dst_is_var if gm != 2 else not dst_is_var, gm)
# aux / Z of the bad destination rows: streamed forward vs generic forward
def up16(x): return (x + 15) & ~15
n = nd
o_qp = up16(1088); o_t = o_qp + up16(n * 16); o_Z = o_t + up16(n); o_aux = o_Z + up16(n * 16)
bad = [116350, 1901196, 2177863]
b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
aux_g = ws[o_aux:o_aux + n * 4].view(n, 4)[bad].clone(); Z_g = ws[o_Z:o_Z + n * 16].view(n, 16)[bad].clone()
b.build_stream_copy(dst_is_var, 1)
hs = b.tconv_fwd(dst_is_var, 16, cp, xs, xd, ws)
aux_s = ws[o_aux:o_aux + n * 4].view(n, 4)[bad].clone(); Z_s = ws[o_Z:o_Z + n * 16].view(n, 16)[bad].clone()
print("aux generic {un, rowmax, rinv, S}\n", aux_g.cpu().numpy()); print("aux streamed\n", aux_s.cpu().numpy())
print("Z diff", (Z_g - Z_s).abs().max().item(), "h diff rows", (hs[bad] - h0[bad]).abs().max().item())
ptr = b.export(0)
print("degrees", [int(ptr[r + 1] - ptr[r]) for r in bad])
