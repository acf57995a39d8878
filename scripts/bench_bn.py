"""BatchNorm streaming kernels at the CROG-R50 B = 32 layer shapes (GPU box): time and algorithmic TB/s of bn_apply_stats,
bn_bwd_partial and bn_bwd_apply, operand sets rotated past the Infinity Cache.  Run once per library (CROG_LIB=...) to A/B two builds;
`check` compares against float64 first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import kernels as K

dt = torch.bfloat16
dev = "cuda"


def timeit(fn, n, iters=20):
    for i in range(3): fn(i % n)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fn(i % n)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def case(M, C, res, mask):
    nset = max(2, int(700e6 // (M * C * 2 * (3 if res else 2))) + 1)
    R = 4
    z = [torch.randn(M, C, device=dev).to(dt) for _ in range(nset)]
    rs = [torch.randn(M, C, device=dev).to(dt) for _ in range(nset)] if res else [None] * nset
    y = [torch.empty(M, C, device=dev, dtype=dt) for _ in range(nset)]
    dy = [torch.randn(M, C, device=dev).to(dt) for _ in range(nset)]
    dz = [torch.empty(M, C, device=dev, dtype=dt) for _ in range(nset)]
    dres = [torch.empty(M, C, device=dev, dtype=dt) for _ in range(nset)] if res else [None] * nset
    mk = [torch.empty(M * C // 8, device=dev, dtype=torch.uint8) for _ in range(nset)] if mask else [None] * nset
    zf = z[0].double()
    sums = torch.zeros(R, C, 2, device=dev)
    sums[0, :, 0] = zf.sum(0).float(); sums[0, :, 1] = (zf * zf).sum(0).float()
    gamma = torch.rand(C, device=dev) + 0.5; beta = torch.randn(C, device=dev) * 0.1
    rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
    ss = torch.empty(C, 2, device=dev); mi = torch.empty(C, 2, device=dev)
    part = torch.zeros(R, C, 2, device=dev)
    dgam = torch.zeros(C, device=dev); dbet = torch.zeros(C, device=dev)
    fwd = lambda i: K.bn_apply_stats(z[i], sums, R, float(M), gamma, beta, rm, rv, 0.1, 1e-5, ss, mi, rs[i], True, y[i], relu_mask=mk[i])
    rpb = K.bn_rows_per_block(M)
    relu_ss = None if mask else ss
    bp = lambda i: K.bn_bwd_partial(dy[i], None, z[i], mi, rpb, part, relu_ss=relu_ss, replicas=R, relu_mask=mk[i])
    ba = lambda i: K.bn_bwd_apply(dy[i], None, z[i], mi, gamma, part, float(M), dz[i], dres[i], relu_ss=relu_ss, sum_rows=R, dgamma=dgam, dbeta=dbet, relu_mask=mk[i])
    # numerics of set 0 against float64
    fwd(0); torch.cuda.synchronize()
    mean = zf.mean(0); var = zf.var(0, unbiased=False); inv = (var + 1e-5).rsqrt()
    pre = (zf - mean) * inv * gamma.double() + beta.double()
    if res: pre = pre + rs[0].double()
    ref = pre.clamp_min(0)
    e_f = float((y[0].double() - ref).abs().max())
    part.zero_(); bp(0); torch.cuda.synchronize()
    g = dy[0].double() * (y[0] > 0)          # the kernel's own gate: a pre-activation within rounding of 0 flips between implementations
    zh = (zf - mean) * inv
    sg = g.sum(0); sgz = (g * zh).sum(0)
    got = part.sum(0).double()
    e_p = float(((got[:, 0] - sg).abs() / (sg.abs() + 1)).max()), float(((got[:, 1] - sgz).abs() / (sgz.abs() + 1)).max())
    ba(0); torch.cuda.synchronize()
    dzr = gamma.double() * inv * (g - sg / M - zh * sgz / M)
    e_b = float((dz[0].double() - dzr).abs().max() / dzr.abs().max())
    e_r = float((dres[0].double() - g).abs().max()) if res else 0.0
    nb = M * C * 2
    t_f = timeit(fwd, nset); t_p = timeit(bp, nset); t_a = timeit(ba, nset)
    by_f = nb * (3 if res else 2) + (M * C // 8 if mask else 0)
    by_p = nb * 2 + (M * C // 8 if mask else 0)
    by_a = nb * (4 if res else 3) + (M * C // 8 if mask else 0)
    print(f"M={M:8d} C={C:5d} res={int(res)} mask={int(mask)}: apply {t_f*1e3:7.1f} us {by_f/t_f/1e9:5.2f} TB/s | partial {t_p*1e3:7.1f} us {by_p/t_p/1e9:5.2f} TB/s | "
          f"bwd_apply {t_a*1e3:7.1f} us {by_a/t_a/1e9:5.2f} TB/s   err fwd {e_f:.2e} part {e_p[0]:.1e}/{e_p[1]:.1e} dz {e_b:.2e} dres {e_r:.1e}", flush=True)
    assert e_f < 0.07 and e_b < 2e-2 and max(e_p) < 1e-3


cases = [(346112, 64, False, False), (346112, 256, True, True), (1384448, 32, False, False), (86528, 128, False, False), (86528, 512, True, True),
                        (21632, 256, False, False), (21632, 1024, True, True), (5408, 512, False, False), (5408, 2048, True, True), (346112, 128, False, False)]
if os.environ.get("BN_CASES"): cases = [cases[int(i)] for i in os.environ["BN_CASES"].split(",")]
for M, C, res, mask in cases:
    case(M, C, res, mask)
