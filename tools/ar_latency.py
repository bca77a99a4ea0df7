"""Latency of a tiny gradient all-reduce between two kernels: torch.distributed (own NCCL stream + events)
vs RCCL called directly on the compute stream vs both inside a hipGraph.  World size 1 (one-GPU box)."""
import ctypes, os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.ones(4721, device="cuda")
x = torch.randn(1 << 20, device="cuda")
for _ in range(5):
    dist.all_reduce(t)
torch.cuda.synchronize()


class UID(ctypes.Structure):
    _fields_ = [("b", ctypes.c_char * 128)]


rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
uid = UID()
assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
comm = ctypes.c_void_p()
rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UID, ctypes.c_int]
assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
rccl.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                               ctypes.c_void_p, ctypes.c_void_p]


def direct():
    rc = rccl.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), 7, 0, comm, torch.cuda.current_stream().cuda_stream)
    assert rc == 0


def timeit(label, fn, n=200):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{label:46s} gpu {e0.elapsed_time(e1) * 1000 / n:8.1f} us/iter   wall {(t1 - t0) * 1e6 / n:8.1f} us/iter", flush=True)


timeit("kernel, kernel", lambda: (x.mul_(1.0001), x.mul_(1.0001)))
timeit("torch all_reduce only", lambda: dist.all_reduce(t))
timeit("kernel, torch all_reduce, kernel", lambda: (x.mul_(1.0001), dist.all_reduce(t), x.mul_(1.0001)))
timeit("direct rccl only", direct)
timeit("kernel, direct rccl, kernel", lambda: (x.mul_(1.0001), direct(), x.mul_(1.0001)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    timeit("side stream: kernel, direct rccl, kernel", lambda: (x.mul_(1.0001), direct(), x.mul_(1.0001)))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        x.mul_(1.0001); direct(); x.mul_(1.0001)
    timeit("graph(kernel, direct rccl, kernel)", g.replay)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        x.mul_(1.0001); dist.all_reduce(t); x.mul_(1.0001)
    timeit("graph(kernel, torch all_reduce, kernel)", g2.replay)
print("t[0] =", float(t[0]))
rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
rccl.ncclCommDestroy(comm)
dist.destroy_process_group()
