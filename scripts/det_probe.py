"""Where does the run-to-run difference of deterministic mode WITH side streams enter?  (GPU box)

Full-depth CROG-R50 bf16, deterministic kernels, side streams ON (RT.det_streams).  Every backward node of crog_amd.functional is wrapped:
integer checksums of its incoming gradients and saved tensors (before AND after its launches) and of what it returns stay on the device
(no host sync inside the pass).  N passes from the same weights; for every pass that differs from the first, the first node (in
execution order) whose checksums differ is named, with WHICH of its tensors differ.  LayerNorm backward nodes also keep clones of all
their operands, so that the launch can be repeated stand-alone and a differing row analysed (which of the two results does a quiet
re-run reproduce; does dropping one wave's partial sums explain the other).

usage: det_probe.py [B=4] [dropout=0.0] [N=10] [variant ...]      variants: all (default) | nofork:conv,linear,mha,ln | notext | nowgrad"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crog_amd import functional as Fn, kernels as K
from crog_amd.model import build_crog
from crog_amd.runtime import RT, set_deterministic
from crog_amd.testing import make_cfg, synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
variants = sys.argv[4:] or ["all"]

set_deterministic(True)
RT.det_streams = "all"
torch.manual_seed(0)
cfg = make_cfg(dropout=p)
model, _ = build_crog(cfg); model = model.cuda().prepare(); model.train()
b = {k: v.cuda() for k, v in synthetic_batch(B, 416, cfg.word_len, cfg.clip_arch["vocab_size"], seed=9).items()}
sd = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}

CONTRIB = None  # clones of what AddRowsFn / GatherRowsFn backward hand on (the pieces autograd sums into the text features' gradient)
LAG_MS = float(os.environ.get("PROBE_LAG_MS", "0"))
REC = None      # list of (name, [checksum tensors before], [after], [outputs]) of the pass in flight
LN = None       # list of dicts of clones, LayerNormFn.backward only


def cs(t):
    if t is None or not isinstance(t, torch.Tensor) or not t.is_cuda or t.numel() == 0:
        return None
    t = t.detach()
    if not t.is_contiguous():
        t = t.contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        v = t.view(torch.int16)
    elif t.dtype == torch.float32:
        v = t.view(torch.int32)
    else:
        v = t
    return v.to(torch.int64).sum()


def wrap(cls):
    orig = cls.backward

    def backward(ctx, *grads):
        if REC is None:
            return orig(ctx, *grads)
        saved = list(ctx.saved_tensors)
        before = [cs(g) for g in grads] + [cs(t) for t in saved]
        keep = None
        if cls is Fn.LayerNormFn and LN is not None:
            gamma = ctx.cfg[0]
            _, add_slot = ctx.slots
            parked = add_slot.t if add_slot is not None else None
            keep = dict(dout=grads[0], dout2=grads[1] if len(grads) > 1 else None, x=saved[0], stats=saved[1], gamma=gamma.master(), parked=parked)
            keep = {k: (v.clone() if v is not None else None) for k, v in keep.items()}
            keep["cfg"] = ctx.cfg[2:6] + (ctx.relu_in,)
        out = orig(ctx, *grads)
        after = [cs(g) for g in grads] + [cs(t) for t in saved]
        outs = [cs(o) for o in (out if isinstance(out, tuple) else (out,))]
        REC.append((cls.__name__, before, after, outs))
        if cls in (Fn.AddRowsFn, Fn.GatherRowsFn) and CONTRIB is not None and isinstance(out, tuple) and out[0] is not None:
            CONTRIB.append((cls.__name__, len(REC) - 1, out[0].clone()))
        if keep is not None:
            keep["dx"] = out[0].clone() if out[0] is not None else None
            keep["idx"] = len(REC) - 1
            LN.append(keep)
        return out
    cls.backward = staticmethod(backward)


for name in dir(Fn):
    c = getattr(Fn, name)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c)


def grads(record):
    global REC, LN, CONTRIB
    model.load_state_dict({**model.state_dict(), **sd})
    RT.manual_seed(5)
    model.store.g_clean = False
    model.store.zero_grad()
    REC, LN, CONTRIB = ([], [], []) if record else (None, None, None)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        _, _, loss, _ = model(b["img"], b["word"], b["mask"], b["qua"], b["sin"], b["cos"], b["wid"])
    if AGGR is not None:
        AGGR()      # a burst of one kernel on a stream of its own, beside the whole backward pass (variant aggr:<kind>)
    if LAG_MS and RT._wgrad_stream:
        # hold the weight-gradient stream back: everything forked onto it queues up behind a sleep while the main stream runs ahead
        # (the checksum kernels of this probe slow the main stream down, which otherwise hides a race that needs the side stream to LAG)
        with torch.cuda.stream(RT._wgrad_stream[0]):
            torch.cuda._sleep(int(LAG_MS * 2.0e6))
    loss.backward()
    torch.cuda.synchronize()
    rec, ln = REC, (LN, CONTRIB)
    REC = LN = CONTRIB = None
    return float(loss.detach()), model.store.G.clone(), rec, ln


def flat(rec):
    return [[None if c is None else int(c) for c in part] for _, *parts in rec for part in parts]


def ln_rerun(k):
    """Repeat the LayerNorm backward launch on the kept operands, alone on the device."""
    p_in, seed_in, p_out, seed_out, relu_in = k["cfg"]
    x = k["x"]
    M, C, _ = K.mat(x)
    dx = torch.empty_like(x)
    rpb = K.ln_bwd_rows_per_block(M)
    nb = (M + rpb - 1) // rpb
    partial = torch.empty(nb, C, 2, device=x.device, dtype=torch.float32)
    RT.manual_seed(5)
    K.ln_bwd(K.as_mat(k["dout"]), K.as_mat(k["dout2"]) if k["dout2"] is not None else None, x, k["gamma"], k["stats"], dx, partial, rpb, p_in=p_in,
             seed_in=seed_in, p_out=p_out, seed_out=seed_out, dxadd=K.as_mat(k["parked"]) if k["parked"] is not None else None, relu_in=relu_in)
    torch.cuda.synchronize()
    return dx


def ln_ref(k, r, g):
    """fp32 restatement of one row of the backward for an incoming gradient row g (bit-faithful to the kernel on the rows checked)."""
    M, C, _ = K.mat(k["x"])
    x = k["x"].reshape(M, C)[r].float()
    mean, rstd = k["stats"].reshape(M, 2)[r]
    xh = (x - mean) * rstd
    gy = g.float() * k["gamma"].float()
    d = rstd * (gy - gy.mean() - xh * (gy * xh).mean())
    if k["cfg"][4]:
        d = torch.where(x > 0, d, torch.zeros_like(d))
    if k["parked"] is not None:
        d = d + k["parked"].reshape(M, C)[r].float()
    return d.to(k["x"].dtype)


def analyse_ln(k0, k1, contrib=None):
    names = ("dout", "dout2", "x", "stats", "gamma", "parked")
    same = {n: (k0[n] is None and k1[n] is None) or (k0[n] is not None and k1[n] is not None and torch.equal(k0[n], k1[n])) for n in names}
    print("      operand clones equal between the two passes:", same)
    d0, d1 = k0["dx"], k1["dx"]
    M, C, _ = K.mat(k0["x"])
    a, c = d0.reshape(M, C), d1.reshape(M, C)
    rows = (a != c).any(dim=1).nonzero().flatten()
    print(f"      dx: {int((a != c).sum())} elements in {rows.numel()} of {M} rows differ (C = {C}); rows {rows[:12].tolist()}")
    if all(same.values()):
        re = ln_rerun(k0).reshape(M, C)
        print(f"      quiet re-run equals pass A: {bool(torch.equal(re, a))}, equals pass B: {bool(torch.equal(re, c))}")
        for r in rows[:4].tolist():
            wa, wc = (re[r] != a[r]).nonzero().flatten(), (re[r] != c[r]).nonzero().flatten()
            print(f"      row {r}: vs re-run, pass A differs in {wa.numel()} elements, pass B in {wc.numel()}; columns of the wrong one: "
                  f"{(wa if wa.numel() else wc)[:8].tolist()} ... {(wa if wa.numel() else wc)[-4:].tolist()}")
            bad = a[r] if wa.numel() else c[r]
            good = re[r]
            dd = (bad.float() - good.float())
            print(f"        |wrong - right| max {float(dd.abs().max()):.3e}, rms {float(dd.pow(2).mean().sqrt()):.3e}; |right| rms {float(good.float().pow(2).mean().sqrt()):.3e}; "
                  f"wrong row all zero: {bool((bad == 0).all())}")
            if C == 512 and k0["dout"] is not None:
                # the error as an error of the kernel's two row means: wrong - right = -rstd (dA + xh dB) (+ bf16 rounding); then which lanes'
                # partial sums (lane L owns columns 8 L .. 8 L + 7) would explain (dA, dB)
                x = k0["x"].reshape(M, C)[r].float()
                mean, rstd = k0["stats"].reshape(M, 2)[r]
                g = k0["dout"].reshape(M, C)[r].float()
                if k0["dout2"] is not None:
                    g = g + k0["dout2"].reshape(M, C)[r].float()
                xh = (x - mean) * rstd
                gy = g * k0["gamma"].float()
                A_ = torch.stack([torch.ones_like(xh), xh], dim=1) * (-rstd)
                sol = torch.linalg.lstsq(A_, dd.unsqueeze(1)).solution.flatten()
                resid = dd - A_ @ sol
                pa, pb = gy.reshape(64, 8).sum(1) / C, (gy * xh).reshape(64, 8).sum(1) / C
                print(f"        fit: dA = {float(sol[0]):.4e} (A = {float(gy.mean()):.4e}), dB = {float(sol[1]):.4e} (B = {float((gy * xh).mean()):.4e}); residual rms "
                      f"{float(resid.pow(2).mean().sqrt()):.2e} vs error rms {float(dd.pow(2).mean().sqrt()):.2e}")
                for nm, tgt, part in (("A", sol[0], pa), ("B", sol[1], pb)):
                    best = min(range(64), key=lambda L: abs(float(tgt + part[L])))
                    grp16 = part.reshape(4, 16).sum(1)
                    grp32 = part.reshape(2, 32).sum(1)
                    print(f"          d{nm}: closest single lane dropped: lane {best} (-partial = {-float(part[best]):.4e}); rows of 16 lanes {[-round(float(v), 6) for v in grp16]}; halves {[-round(float(v), 6) for v in grp32]}")
            if contrib:
                # the incoming gradient is autograd's running sum of these pieces (bf16 adds, arrival order): did the kernel see a PARTIAL sum?
                pieces = [(nm, j, t.reshape(M, C)[r]) for nm, j, t in contrib if t.numel() == M * C]
                full = k0["dout"].reshape(M, C)[r]
                print(f"        pieces of the incoming gradient (node index): {[(nm, j) for nm, j, _ in pieces]}")
                import itertools
                for n in range(0, len(pieces) + 1):
                    for sub in itertools.combinations(range(len(pieces)), n):
                        g = torch.zeros_like(full)
                        for q in sub:
                            g = (g.float() + pieces[q][2].float()).to(full.dtype)
                        tag = "FULL " if bool(torch.equal(g, full)) else ""
                        nd = int((ln_ref(k0, r, g) != bad).sum())
                        if nd == 0 or tag:
                            print(f"        {tag}subset {sub}: restated row differs from the WRONG row in {nd} elements")
            # what (a, b) - the two row means of the kernel - would reproduce the wrong row?  least squares on d = rstd (gy - A - xh B)
            x = k0["x"].reshape(M, C)[r].float()
            mean, rstd = k0["stats"].reshape(M, 2)[r]
            g = k0["dout"].reshape(M, C)[r].float()
            if k0["dout2"] is not None:
                g = g + k0["dout2"].reshape(M, C)[r].float()
            xh = (x - mean) * rstd
            gy = g * k0["gamma"].float()
            A_true, B_true = gy.mean(), (gy * xh).mean()
            if C > 512:
                per = C // 4
                parts = [(gy[w * per:(w + 1) * per].sum() / C, (gy[w * per:(w + 1) * per] * xh[w * per:(w + 1) * per]).sum() / C) for w in range(4)]
                for w, (pa, pb) in enumerate(parts):
                    d = rstd * (gy - (A_true - pa) - xh * (B_true - pb))
                    if k0["cfg"][4]:
                        d = torch.where(x > 0, d, torch.zeros_like(d))
                    if k0["parked"] is not None:
                        d = d + k0["parked"].reshape(M, C)[r].float()
                    print(f"        without wave {w}'s partial sums: {int((d.to(bad.dtype) != bad).sum())} of {C} elements differ from the wrong row")
            d = rstd * (gy - A_true - xh * B_true)
            if k0["cfg"][4]:
                d = torch.where(x > 0, d, torch.zeros_like(d))
            if k0["parked"] is not None:
                d = d + k0["parked"].reshape(M, C)[r].float()
            print(f"        torch fp32 restatement: {int((d.to(bad.dtype) != bad).sum())} elements differ from the wrong row, "
                  f"{int((d.to(bad.dtype) != re[r]).sum())} from the re-run")


AGGR = None
_AG = {}


def make_aggressor(kind, n):
    """n back-to-back launches of one kernel of the step on a stream of its own: slab = the 3x3 weight gradient in deterministic form
    (ping-pong transposed-operand kernel, plain slab stores) + crog_splitk_reduce; wg = the same GEMM with atomic adds; red = the reduction
    alone; fwd = the 3x3 forward."""
    bf = torch.bfloat16
    if not _AG:
        Bn, H, W = 32, 26, 26
        P = Bn * H * W
        _AG.update(P=P, H=H, W=W, x=torch.randn(P, 512, device="cuda").to(bf), dy=(torch.randn(P, 512, device="cuda") * 0.1).to(bf),
                   ws=torch.empty(4 * 512 * 4608, device="cuda"), G=torch.zeros(512, 4608, device="cuda"), s=torch.cuda.Stream(),
                   w3=(torch.randn(512, 4608, device="cuda") * 0.02).to(bf), y=torch.empty(P, 512, device="cuda", dtype=bf))
    a = _AG

    def go():
        K.set_stream_override(a["s"].cuda_stream)
        try:
            for _ in range(n):
                if kind in ("slab", "wg"):
                    if kind == "slab":
                        K.gemm(K.BF16, K.A_MC, K.B_NC_IM2COL, a["dy"], a["x"], a["ws"], 512, 4608, a["P"], 512, 512, 4608, splitk=4, out_mode=K.OUT_F32, conv=(a["H"], a["W"], 512))
                        K.splitk_reduce(a["ws"], 4, 512, 4608, 4608, a["G"], 0, 4608, accumulate=True)
                    else:
                        K.gemm(K.BF16, K.A_MC, K.B_NC_IM2COL, a["dy"], a["x"], a["G"], 512, 4608, a["P"], 512, 512, 4608, splitk=4, out_mode=K.OUT_F32_ATOMIC, conv=(a["H"], a["W"], 512))
                elif kind == "red":
                    K.splitk_reduce(a["ws"], 4, 512, 4608, 4608, a["G"], 0, 4608, accumulate=True)
                else:
                    K.gemm(K.BF16, K.A_IM2COL, K.B_KC, a["x"], a["w3"], a["y"], a["P"], 512, 4608, 512, 4608, 512, conv=(a["H"], a["W"], 512))
        finally:
            K.set_stream_override(None)
    return go


def run_variant(v):
    global AGGR
    AGGR = None
    if v.startswith("aggr:"):      # one stream for the step itself; the only neighbour is the aggressor's burst
        kind = v.split(":")[1]
        RT.overlap_wgrad = False
        model.overlap_text = False
        RT.no_fork = set()
        AGGR = make_aggressor(kind, {"slab": 150, "wg": 150, "red": 600, "fwd": 300}[kind])
        return _run_variant_body(v)
    return _run_variant_body(v, set_flags=True)


def _run_variant_body(v, set_flags=False):
    if not set_flags:
        pass
    else:
        _set_flags(v)
    return _body(v)


def _set_flags(v):
    RT.no_fork = set()
    model.overlap_text = True
    RT.overlap_wgrad = True
    if v.startswith("nofork:"):
        RT.no_fork = set(v.split(":", 1)[1].split(","))
    elif v == "notext":
        model.overlap_text = False
    elif v == "nowgrad":
        RT.overlap_wgrad = False


def _body(v):
    global LAG_MS, AGGR
    lag, LAG_MS = LAG_MS, 0.0
    ag, AGGR = AGGR, None
    l0, g0, r0, (ln0, con0) = grads(True)      # the reference pass: no artificial lag, no aggressor
    LAG_MS, AGGR = lag, ag
    f0 = flat(r0)
    bad = 0
    shown = 0
    for i in range(1, N):
        l, g, r, (ln, con) = grads(True)
        diff = g != g0
        if l == l0 and not diff.any():
            continue
        bad += 1
        names = [n for n, p_, o, k, _ in model.store.entries if bool(diff[o:o + k].any())]
        print(f"  [{v}] pass {i}: loss equal {l == l0}; {int(diff.sum())} gradient elements in {len(names)} parameters differ; e.g. {names[:2]} ... {names[-2:]}")
        if shown >= 2:
            continue
        shown += 1
        if len(r) != len(r0):
            print(f"    node counts differ: {len(r0)} vs {len(r)}")
            continue
        for j, ((nm, b0_, a0_, o0_), (_, b1_, a1_, o1_)) in enumerate(zip(r0, r)):
            t0 = [[None if c is None else int(c) for c in part] for part in (b0_, a0_, o0_)]
            t1 = [[None if c is None else int(c) for c in part] for part in (b1_, a1_, o1_)]
            if t0 != t1:
                what = [("inputs-before", "inputs-after", "outputs")[q] + str([z for z, (u, w) in enumerate(zip(t0[q], t1[q])) if u != w]) for q in range(3) if t0[q] != t1[q]]
                selfchg0 = t0[0] != t0[1]
                print(f"    first differing node: #{j} {nm} (of {len(r0)}): {what}; its inputs changed DURING the node in pass A: {selfchg0}")
                prev = [r0[q][0] for q in range(max(0, j - 4), j)]
                print(f"      preceding nodes: {prev}")
                if nm == "LayerNormFn":
                    k0 = next((k for k in ln0 if k["idx"] == j), None)
                    k1 = next((k for k in ln if k["idx"] == j), None)
                    if k0 is not None and k1 is not None:
                        analyse_ln(k0, k1, con0)
                break
        else:
            print("    no node's checksums differ (the difference entered in a launch on a side stream: weight gradients only)")
    print(f"[{v}] B={B} dropout={p}: {bad} of {N - 1} passes differ from the first", flush=True)


for v in variants:
    run_variant(v)
