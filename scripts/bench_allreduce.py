"""Host and device cost of a tiny torch.distributed (RCCL) all-reduce at world size 1 (GPU box)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.zeros(2048, 2, device="cuda")
for _ in range(20): dist.all_reduce(t)
torch.cuda.synchronize()
N = 500
t0 = time.perf_counter()
for _ in range(N): dist.all_reduce(t)
th = time.perf_counter() - t0
torch.cuda.synchronize(); tg = time.perf_counter() - t0
print(f"host issue {th / N * 1e6:.1f} us / call, end-to-end {tg / N * 1e6:.1f} us / call")
x = torch.randn(4096, 4096, device="cuda")
def work():
    for _ in range(N):
        y = x @ x          # ~55 us of GPU work between collectives
        dist.all_reduce(t)
work(); torch.cuda.synchronize(); t0 = time.perf_counter(); work(); torch.cuda.synchronize(); tw = time.perf_counter() - t0
def work2():
    for _ in range(N):
        y = x @ x
work2(); torch.cuda.synchronize(); t0 = time.perf_counter(); work2(); torch.cuda.synchronize(); tw2 = time.perf_counter() - t0
print(f"matmul+allreduce {tw / N * 1e6:.1f} us, matmul alone {tw2 / N * 1e6:.1f} us -> added {(tw - tw2) / N * 1e6:.1f} us per collective")
dist.destroy_process_group()
