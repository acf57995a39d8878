"""Host cost per tiny all-reduce call: torch ProcessGroupNCCL vs direct RCCL (world size 1).  GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
from crog_amd.rccl import RcclComm
t = torch.zeros(512, device="cuda")
g = dist.new_group()
dist.all_reduce(t, group=g); torch.cuda.synchronize()
def host(fn, n=500):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
print("torch PG  host %.1f us/call, incl. drain %.1f us/call" % host(lambda: dist.all_reduce(t, group=g)))
c = RcclComm(g)
c.all_reduce_sum(t); torch.cuda.synchronize()
print("direct    host %.1f us/call, incl. drain %.1f us/call" % host(lambda: c.all_reduce_sum(t)))
import ctypes
from crog_amd import kernels as K
lib = c._lib; comm = c._comm; p = t.data_ptr(); s = K.stream()
print("raw ncclAllReduce host %.1f us/call, incl. drain %.1f" % host(lambda: lib.ncclAllReduce(p, p, 512, 7, 0, comm, s)))
dist.destroy_process_group()
