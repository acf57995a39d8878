"""The step's large 3x3 weight gradients, HBM-cold (operand sets rotated past the Infinity Cache).  Env: CROG_WGRAD_TARGET_CONV (blocks per launch the split count aims at)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K
from bench_gemm import timeit
dt = torch.bfloat16
tag = os.environ.get("TAG", "")
res = []
for B, HW, Cin, Cout in [(32, 104, 512, 256), (32, 52, 512, 512), (32, 26, 512, 512), (32, 26, 256, 256), (32, 52, 256, 256), (32, 26, 1024, 512), (32, 52, 512, 256), (32, 104, 64, 64), (32, 52, 128, 128)]:
    M = B * HW * HW
    nset = max(1, int(600e6 / (M * (Cin + Cout) * 2)) + 1)
    xs = [torch.randn(M, Cin, device="cuda").to(dt) for _ in range(nset)]; dys = [torch.randn(M, Cout, device="cuda").to(dt) for _ in range(nset)]
    dw = torch.zeros(Cout, 9 * Cin, device="cuda"); fl = 2.0 * M * Cout * 9 * Cin
    sk = K.pick_splitk(Cout, 9 * Cin, M, 32, conv=True)
    it = [0]
    def run():
        i = it[0] = (it[0] + 1) % nset
        K.gemm(1, K.A_MC, K.B_NC_IM2COL, dys[i], xs[i], dw, Cout, 9 * Cin, M, Cout, Cin, 9 * Cin, conv=(HW, HW, Cin), splitk=sk, out_mode=K.OUT_F32_ATOMIC)
    t = timeit(run, max(6, 2 * nset))
    res.append(f"{Cin}->{Cout}@{HW} sk={sk:3d} {t*1e3:7.1f} us {fl/t/1e9:5.0f} TF/s")
    del xs, dys
print(f"{tag:8s} " + " | ".join(res), flush=True)
