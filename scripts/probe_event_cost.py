"""Does bracketing every launch with events change what the launch costs?  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
dt = torch.bfloat16
B, HW, Cin, Cout = 32, 104, 256, 512
M = B * HW * HW
x = torch.randn(M, Cin, device="cuda").to(dt); w = (torch.randn(Cout, 9 * Cin, device="cuda") * 0.05).to(dt)
y = torch.empty(M, Cout, device="cuda", dtype=dt)
big = torch.empty(1 << 28, device="cuda", dtype=torch.float32); big2 = torch.empty_like(big)
def conv(): K.gemm(1, K.A_IM2COL, K.B_KC, x, w, y, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, conv=(HW, HW, Cin))
def other(): big2.copy_(big)        # 2 GB of traffic between the convs: a cold L2 / Infinity Cache, like the step
def bracket(fn, n, timer_cls):
    pairs = []
    for _ in range(n):
        a, b = timer_cls(), timer_cls()
        a.record(); fn(); b.record(); pairs.append((a, b))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in pairs) / n * 1e3
TorchEv = lambda: torch.cuda.Event(enable_timing=True)
for _ in range(3): conv()
torch.cuda.synchronize()
a, b = TorchEv(), TorchEv(); a.record()
for _ in range(10): conv()
b.record(); torch.cuda.synchronize()
print(f"back to back, one pair around 10: {a.elapsed_time(b)*100:.1f} us each")
print(f"torch events around each launch:    {bracket(conv, 10, TorchEv):.1f} us")
print(f"crog timers around each launch:     {bracket(conv, 10, K.Timer):.1f} us")
def seq(timer_cls, n=6):
    tot = 0.0; pairs = []
    for _ in range(n):
        other()
        a, b = timer_cls(), timer_cls(); a.record(); conv(); b.record(); pairs.append((a, b))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in pairs) / n * 1e3
print(f"after a 2 GB copy, torch events:    {seq(TorchEv):.1f} us")
print(f"after a 2 GB copy, crog timers:     {seq(K.Timer):.1f} us")
# whole-sequence cost with and without per-launch timers
def total(with_timers, n=6):
    torch.cuda.synchronize(); a, b = TorchEv(), TorchEv(); a.record()
    for _ in range(n):
        other()
        if with_timers: t0, t1 = K.Timer(), K.Timer(); t0.record()
        conv()
        if with_timers: t1.record()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
print(f"copy+conv per iteration: plain {total(False):.1f} us, with timers {total(True):.1f} us")
torch.cuda.synchronize(); a, b = TorchEv(), TorchEv(); a.record()
for _ in range(6): other()
b.record(); torch.cuda.synchronize(); print(f"copy alone {a.elapsed_time(b)/6*1e3:.1f} us")
