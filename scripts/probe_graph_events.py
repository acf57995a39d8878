"""Can a timing event recorded INSIDE a captured hipGraph be read back after a replay?  (bench.py's roofline leg would then not need
eager steps.)  Prints the elapsed time of an event pair around a 64 MB copy: recorded eagerly, and recorded in a capture + replayed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from crog_amd import kernels as K
src = torch.empty(64 << 20, device="cuda", dtype=torch.uint8); dst = torch.empty_like(src)
def copy():
    K.check(K.lib().crog_probe_copy(K.ptr(src), K.ptr(dst), src.numel(), 0, K.stream()), "probe_copy")
copy(); torch.cuda.synchronize()
t0, t1 = K.Timer(), K.Timer()
t0.record(); copy(); t1.record(); torch.cuda.synchronize()
print("eager pair ms:", t0.elapsed_time(t1))
for kind in ("crog_timer", "torch_event"):
    try:
        g = torch.cuda.CUDAGraph()
        if kind == "crog_timer":
            a, b = K.Timer(), K.Timer()
        else:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(3): copy()
            a.record(); copy(); b.record()
            for _ in range(3): copy()
        for i in range(3):
            g.replay(); torch.cuda.synchronize()
            print(kind, "in-graph pair ms after replay", i, ":", a.elapsed_time(b))
    except Exception as e:
        print(kind, "FAILED:", repr(e)[:300])
