"""BnLink fusion: cost of the gated-statistics epilogue on the producing data-gradient GEMM vs the bn_bwd_partial pass it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
from crog_amd.functional import stat_replicas
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n      # ms per call
dt = torch.bfloat16
for B, HW, Cin, Cout, ks in [(32, 104, 64, 64, 3), (32, 104, 64, 256, 1), (32, 52, 128, 128, 3), (32, 52, 128, 512, 1), (32, 26, 256, 256, 3), (32, 26, 256, 1024, 1), (32, 208, 32, 32, 3), (32, 208, 32, 64, 3), (32, 13, 512, 512, 3), (32, 13, 512, 2048, 1), (32, 26, 256, 512, 1), (32, 52, 128, 256, 1)]:
    M = B * HW * HW
    # layer L+1: conv Cin -> Cout; its dgrad maps dz [M, Cout] -> dx [M, Cin] = dy of layer L (C = Cin)
    nset = max(1, int(600e6 / (M * (Cin * 2 + Cout) * 2)) + 1)
    dzs = [torch.randn(M, Cout, device="cuda").to(dt) for _ in range(nset)]
    zs = [torch.randn(M, Cin, device="cuda").to(dt) for _ in range(nset)]
    dxs = [torch.empty(M, Cin, device="cuda", dtype=dt) for _ in range(nset)]
    w = (torch.randn(Cout, ks * ks * Cin, device="cuda") * 0.05).to(dt)
    wt = (torch.randn(Cin, ks * ks * Cout, device="cuda") * 0.05).to(dt)
    ss = torch.rand(Cin, 2, device="cuda"); mi = torch.rand(Cin, 2, device="cuda") + 0.5
    R = stat_replicas(K.stat_tiles(M), Cin); sums = torch.zeros(R, Cin, 2, device="cuda")
    it = [0]
    def dgrad(fused):
        i = it[0] = (it[0] + 1) % nset
        kw = dict(col_stats=sums, stat_replicas=R, bwd_z=zs[i], bwd_ss=ss) if fused else {}
        if ks == 1:      # (functional.LIN_DGRAD_T: the [Cin][Cout] copy of the weight, both operands K-contiguous)
            K.gemm(1, K.A_KC, K.B_KC, dzs[i], wt, dxs[i], M, Cin, Cout, Cout, Cout, Cin, **kw)
        else:
            K.gemm(1, K.A_IM2COL, K.B_KC, dzs[i], wt, dxs[i], M, Cin, 9 * Cout, Cout, 9 * Cout, Cin, conv=(HW, HW, Cout), **kw)
    rpb = K.bn_rows_per_block(M)
    def partial():
        i = it[0] = (it[0] + 1) % nset
        K.bn_bwd_partial(dxs[i], None, zs[i], mi, rpb, sums, ss, replicas=R)
    n = max(10, 3 * nset)
    t0, t1, t2 = timeit(lambda: dgrad(False), n) * 1e3, timeit(lambda: dgrad(True), n) * 1e3, timeit(partial, n) * 1e3
    print(f"M={M:7d} C={Cin:4d} <- conv{ks}x{ks} from {Cout:4d}: dgrad {t0:6.1f} us, dgrad+stats {t1:6.1f} us (+{t1-t0:5.1f}), bn_bwd_partial {t2:6.1f} us  -> saves {t2-(t1-t0):6.1f} us")
