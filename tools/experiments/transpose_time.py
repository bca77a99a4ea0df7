import sys, time, torch
sys.path.insert(0, "/root/repo")
from mllp_amd.graph import synthetic_batch, LPBatch
import mllp_amd.graph as G
b = synthetic_batch(256)
print("device transposition: graph_build_s", round(b.graph_build_s, 3))
p, i, v = b._device_orientation(False)
from mllp_amd import _lib
for rep in range(2):
    tp = torch.empty(b.N + 1, dtype=torch.int32, device="cuda"); ti = torch.empty(b.nnz, dtype=torch.int32, device="cuda"); tv = torch.empty(b.nnz, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(_lib.lib().mllp_csr_transpose_device(b.M, b.N, b.nnz, _lib.ptr(p), _lib.ptr(i), _lib.ptr(v), _lib.ptr(tp), _lib.ptr(ti), _lib.ptr(tv), _lib.current_stream()))
    torch.cuda.synchronize(); print("mllp_csr_transpose_device alone", round(time.perf_counter() - t0, 4), "s")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    order = torch.sort(i.long(), stable=True)[1]
    rows = torch.repeat_interleave(torch.arange(b.M, device="cuda", dtype=torch.int32), (p[1:] - p[:-1]).long())
    ci = rows[order].contiguous(); cv = v[order].contiguous()
    torch.cuda.synchronize(); print("torch stable sort + gathers", round(time.perf_counter() - t0, 4), "s")
    del order, rows, ci, cv
