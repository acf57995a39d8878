"""Which HIP streams share a hardware queue?  A long kernel on stream i and a tiny one on stream j: if the tiny one finishes only
after the long one, the two streams are multiplexed onto the same in-order hardware queue.  GPU box."""
import torch, time, os
n = int(os.environ.get("NSTREAMS", "8"))
main = torch.cuda.current_stream()
streams = [main] + [torch.cuda.Stream() for _ in range(n - 1)]
big = torch.empty(1 << 28, device="cuda"); big2 = torch.empty_like(big); small = torch.zeros(8, device="cuda")
for s in streams:                       # first use in creation order
    with torch.cuda.stream(s): small.add_(1)
torch.cuda.synchronize()
def long_on(s):
    with torch.cuda.stream(s):
        for _ in range(4): big2.copy_(big)            # ~2 ms
def shares(i, j):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(streams[j]): e0.record()
    long_on(streams[i])
    with torch.cuda.stream(streams[i]): e2.record()
    with torch.cuda.stream(streams[j]):
        small.add_(1); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), e0.elapsed_time(e2)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "default"))
for i in range(n):
    row = []
    for j in range(n):
        if i == j: row.append("  -  "); continue
        t_small, t_long = shares(i, j)
        row.append("SAME " if t_small > 0.5 * t_long else " .   ")
    print(f"long on stream {i}: " + "".join(row))
