"""Where does the per-step cost of the 1-rank RCCL path come from?  Netlib batch, LPTrainer pieces."""
import ctypes, os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
from mllp_amd.data import load_packed
from mllp_amd.graph import LPBatch
from mllp_amd.trainer import LPTrainer
from mllp_amd.model import GNNModel, set_seed

PG_FIRST = "--pg-first" in sys.argv
if PG_FIRST:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
params0 = (set_seed(42), GNNModel().flat_parameters().detach().float().cuda())[1]
batch = LPBatch.from_instances(load_packed())


def run(label, between, n=300, graph=False):
    tr = LPTrainer(params0, use_hip_graph=False)
    p = tr._plan(batch)

    def step():
        tr._fwd_bwd(p); between(p["grads"]); tr._opt(p)
    for _ in range(3):
        step()
    fn = step
    if graph:
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        fn = g.replay
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label:52s} {dt / n * 1e3:7.3f} ms/step", flush=True)


run("before init_process_group: no all_reduce", lambda g: None)
if not PG_FIRST:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
run("after init_process_group: no all_reduce", lambda g: None)
run("torch all_reduce", lambda g: dist.all_reduce(g))
run("torch all_reduce (again)", lambda g: dist.all_reduce(g))


class UID(ctypes.Structure):
    _fields_ = [("b", ctypes.c_char * 128)]


rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
uid = UID(); assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
comm = ctypes.c_void_p()
rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UID, ctypes.c_int]
assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
rccl.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                               ctypes.c_void_p, ctypes.c_void_p]


def direct(g):
    assert rccl.ncclAllReduce(g.data_ptr(), g.data_ptr(), g.numel(), 7, 0, comm,
                              torch.cuda.current_stream().cuda_stream) == 0


run("direct rccl on the compute stream", direct)
run("no all_reduce (end)", lambda g: None)
run("graph: no all_reduce", lambda g: None, graph=True)
run("graph: torch all_reduce", lambda g: dist.all_reduce(g), graph=True)
run("graph: direct rccl", direct, graph=True)
dist.destroy_process_group()
