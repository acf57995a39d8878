"""Micro-benchmark of the GEMM / implicit-GEMM kernel on CROG-R50 (B=32) layer shapes. GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    dt = torch.bfloat16 if len(sys.argv) < 2 or sys.argv[1] != "f32" else torch.float32
    dc = K.dcode(dt)
    B = 32
    rows = []
    # (name, B,H,W,Cin,Cout,ksize)
    convs = [("stem.conv2 3x3 32->32 @208", 208, 32, 32, 3), ("stem.conv3 3x3 32->64 @208", 208, 32, 64, 3),
             ("l1.conv1 1x1 64->64 @104", 104, 64, 64, 1), ("l1.conv2 3x3 64->64 @104", 104, 64, 64, 3),
             ("l1.conv3 1x1 64->256 @104", 104, 64, 256, 1), ("l2.conv2 3x3 128->128 @104", 104, 128, 128, 3),
             ("l3.conv2 3x3 256->256 @52", 52, 256, 256, 3), ("l4.conv2 3x3 512->512 @26", 26, 512, 512, 3),
             ("l4.conv3 1x1 512->2048 @13", 13, 512, 2048, 1), ("neck 3x3 1024->512 @26", 26, 1024, 512, 3),
             ("proj 3x3 512->512 @52", 52, 512, 512, 3), ("proj 3x3 512->256 @104", 104, 512, 256, 3),
             ("proj 1x1 256->1280 @104", 104, 256, 1280, 1), ("ffn 512->2048 M=21632", None, 512, 2048, 0),
             ("ffn 2048->512 M=21632", None, 2048, 512, 0)]
    for name, HW, Cin, Cout, ks in convs:
        if ks == 0:
            M = 21632
            H = W = 0
        else:
            H = W = HW
            M = B * H * W
        x = torch.randn(M, Cin, device="cuda").to(dt)
        Kd = Cin * (9 if ks == 3 else 1)
        w = (torch.randn(Cout, Kd, device="cuda") * 0.05).to(dt)
        y = torch.empty(M, Cout, device="cuda", dtype=dt)
        dy = torch.randn(M, Cout, device="cuda").to(dt)
        dx = torch.empty(M, Cin, device="cuda", dtype=dt)
        dw = torch.zeros(Cout, Kd, device="cuda")
        flops = 2.0 * M * Cout * Kd
        if ks == 3:
            f = lambda: K.gemm(dc, K.A_IM2COL, K.B_KC, x, w, y, M, Cout, Kd, Cin, Kd, Cout, conv=(H, W, Cin))
            d = lambda: K.gemm(dc, K.A_IM2COL, K.B_NC_DGRAD, dy, w, dx, M, Cin, 9 * Cout, Cout, Cin, Cin, conv=(H, W, Cout))
            sk = K.pick_splitk(Cout, Kd, M, 32)
            g = lambda: K.gemm(dc, K.A_MC, K.B_NC_IM2COL, dy, x, dw, Cout, Kd, M, Cout, Cin, Kd, conv=(H, W, Cin), splitk=sk,
                               out_mode=K.OUT_F32_ATOMIC)
        else:
            f = lambda: K.gemm(dc, K.A_KC, K.B_KC, x, w, y, M, Cout, Kd, Cin, Kd, Cout)
            d = lambda: K.gemm(dc, K.A_KC, K.B_NC, dy, w, dx, M, Cin, Cout, Cout, Cin, Cin)
            sk = K.pick_splitk(Cout, Kd, M, 32)
            g = lambda: K.gemm(dc, K.A_MC, K.B_NC, dy, x, dw, Cout, Kd, M, Cout, Cin, Kd, splitk=sk, out_mode=K.OUT_F32_ATOMIC)
        tf, td, tg = timeit(f), timeit(d), timeit(g)
        esz = 2 if dt == torch.bfloat16 else 4
        bytes_f = (M * Cin + M * Cout) * esz
        print(f"{name:34s} fwd {tf:7.3f} ms {flops/tf/1e9:7.1f} TF/s {bytes_f/tf/1e6:6.0f} GB/s | dgrad {td:7.3f} ms {flops/td/1e9:7.1f} TF/s"
              f" | wgrad(sk={sk:4d}) {tg:7.3f} ms {flops/tg/1e9:7.1f} TF/s", flush=True)
    # plain square GEMM
    for n in (4096,):
        a = torch.randn(n, n, device="cuda").to(dt)
        b = torch.randn(n, n, device="cuda").to(dt)
        c = torch.empty(n, n, device="cuda", dtype=dt)
        t = timeit(lambda: K.gemm(dc, K.A_KC, K.B_KC, a, b, c, n, n, n, n, n, n))
        print(f"square {n}: {t:.3f} ms {2*n**3/t/1e9:.1f} TF/s")
        t = timeit(lambda: torch.matmul(a, b.t()))
        print(f"torch.matmul {n}: {t:.3f} ms {2*n**3/t/1e9:.1f} TF/s")
    # copy bandwidth reference
    src = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
    dst = torch.empty_like(src)
    t = timeit(lambda: dst.copy_(src))
    print(f"copy 1 GiB: {t:.3f} ms {2*src.numel()*4/t/1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
