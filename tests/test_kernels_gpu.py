"""Kernel-level parity: every HIP kernel against a plain torch fp32 computation of the same op.

These run on the GPU box (`-m gpu`) and call through the C ABI (crog_amd.kernels -> libcrog_hip.so).
torch ops are used here only as the checker.
"""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def tol(dt, k=1):
    # fp32 path: exact-f32 MFMA, differences are summation order only. bf16: inputs rounded to 8 bits.
    return (2e-5, 2e-5) if dt == torch.float32 else (2e-2, 2e-2)


def close(a, b, dt, scale=1.0):
    rt, at = tol(dt)
    a = a.float()
    b = b.float()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-6
    assert err <= at * scale * max(ref, 1.0) + rt * ref, f"max err {err} vs ref max {ref}"


@pytest.fixture(scope="module")
def K():
    from crog_amd import kernels
    kernels.lib()
    return kernels


def rnd(*shape, dt=torch.float32, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g).to("cuda").to(dt)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (200, 72, 40), (1000, 320, 256), (37, 9, 16), (676, 676, 64)])
def test_gemm_nt(K, dt, M, N, K_):
    a, b = rnd(M, K_, dt=dt), rnd(N, K_, dt=dt, seed=1)
    c = torch.empty(M, N, device="cuda", dtype=dt) if N % 8 == 0 else torch.empty(M, ((N + 7) // 8) * 8, device="cuda", dtype=dt)
    ldc = c.stride(0)
    K.gemm(K.dcode(dt), K.A_KC, K.B_KC, a, b, c, M, N, K_, K_, K_, ldc)
    ref = a.float() @ b.float().t()
    close(c[:, :N], ref, dt, scale=math.sqrt(K_) / 4)


@pytest.mark.parametrize("dt", DT)
def test_gemm_epilogue(K, dt):
    M, N, K_ = 300, 136, 96
    a, b = rnd(M, K_, dt=dt), rnd(N, K_, dt=dt, seed=1)
    bias = rnd(N, seed=2)
    r = rnd(M, N, dt=dt, seed=3)
    for act in (K.ACT_NONE, K.ACT_RELU, K.ACT_QUICKGELU):
        c = torch.empty(M, N, device="cuda", dtype=dt)
        K.gemm(K.dcode(dt), K.A_KC, K.B_KC, a, b, c, M, N, K_, K_, K_, N, alpha=0.5, bias=bias, act=act, R=r, ldr=N)
        v = 0.5 * (a.float() @ b.float().t()) + bias
        if act == K.ACT_RELU:
            v = v.relu()
        elif act == K.ACT_QUICKGELU:
            v = v * torch.sigmoid(1.702 * v)
        close(c, v + r.float(), dt, scale=3)
    # fp32 output + column statistics
    c32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
    stats = torch.zeros(K.stat_tiles(M), N, 2, device="cuda")
    K.gemm(K.dcode(dt), K.A_KC, K.B_KC, a, b, c32, M, N, K_, K_, K_, N, out_mode=K.OUT_F32, col_stats=stats)
    ref = a.float() @ b.float().t()
    close(c32, ref, dt, scale=3)
    s = stats.sum(0)
    close(s[:, 0], ref.sum(0), dt, scale=30)
    close(s[:, 1], (ref * ref).sum(0), dt, scale=300)


@pytest.mark.parametrize("M,N,K_", [(300, 136, 96), (1000, 64, 64), (4097, 258, 32), (256, 256, 64)])
@pytest.mark.parametrize("with_res", [False, True])
def test_bf16_pair_store_epilogue_equals_the_staged_one(K, monkeypatch, M, N, K_, with_res):
    """The bf16 epilogue stores DPP-paired columns straight from the accumulators; debug bit 2 brings the LDS-staged form back.
    Without a residual both round the same fp32 value once (bit-equal); with one the pair form adds in fp32 (one rounding
    instead of two): apart by no more than those roundings, and never further from the fp32 result."""
    dt = torch.bfloat16
    a, b = rnd(M, K_, dt=dt), rnd(N, K_, dt=dt, seed=1)
    ld = (N + 7) // 8 * 8 + 8                                                  # rows 16-byte aligned, canary columns behind N
    r = rnd(M, ld, dt=dt, seed=3) if with_res else None
    ref = a.float() @ b.float().t() + (r[:, :N].float() if with_res else 0)
    out = {}
    for flag in (0, 4):
        monkeypatch.setattr(K, "DEBUG_FLAGS", flag)
        c = torch.full((M + 1, ld), 7.0, device="cuda", dtype=dt)              # canary row and columns
        K.gemm(K.dcode(dt), K.A_KC, K.B_KC, a, b, c, M, N, K_, K_, K_, ld, R=r, ldr=ld)
        assert (c[M] == 7).all() and (c[:, N:] == 7).all(), "epilogue wrote outside the M x N block"
        out[flag] = c[:M, :N].float()
    if not with_res:
        assert torch.equal(out[0], out[4])
    else:
        prod = a.float() @ b.float().t()          # the staged path rounds the product (<= 2^-8 |prod|), then both round the sum
        assert ((out[0] - out[4]).abs() <= 2.0 ** -7 * (prod.abs() + out[4].abs()) * 1.001).all()
        assert (out[0] - ref).abs().max() <= (out[4] - ref).abs().max() * 1.001 + 1e-6
    close(out[0], ref, dt, scale=math.sqrt(K_) / 4)


@pytest.mark.parametrize("case", [
    # (B, HW, Cin, Cout, 3x3?, debug bits)   bit 6: 16x16x32 on the 256 x 256 tile, + bit 7: on 128 x 128 tiles too
    (40, 26, 64, 512, True, 64),             # M = 27040 = 105.6 tiles of 256 rows (row guard), 212 tiles >= 160: the 256 x 256 kernel
    (8, 52, 64, 256, True, 64 | 128),        # 128 x 128 kernel, 3x3
    (8, 52, 256, 384, False, 64 | 128),      # 128 x 128 kernel, 1x1, three column tiles
])
def test_mfma_16x16x32_variant_equals_the_32x32x16_kernel(K, monkeypatch, case):
    """gemm_dma16_kernel (v_mfma_f32_16x16x32_bf16, de-interleaved B tile, 8-byte stores) against the 32x32x16 kernel (debug bit 8) on
    the same operands: identical output bits (both accumulate the 32 products of a k-tile in fp32 in the same order), the same
    BatchNorm column statistics up to fp32 summation order, nothing written outside the M x N block; and both against float64."""
    B, HW, Cin, Cout, conv3, bits = case
    dt = torch.bfloat16
    M = B * HW * HW
    Kd = 9 * Cin if conv3 else Cin
    x = rnd(M, Cin, dt=dt)
    w = (rnd(Cout, Kd, dt=dt, seed=1) * Kd ** -0.5).to(dt)
    ld = Cout + 8
    out, st = {}, {}
    for flag in (256, bits):
        monkeypatch.setattr(K, "DEBUG_FLAGS", flag)
        y = torch.full((M + 1, ld), 7.0, device="cuda", dtype=dt)
        stats = torch.zeros(3, Cout, 2, device="cuda")
        if conv3:
            K.gemm(1, K.A_IM2COL, K.B_KC, x, w, y, M, Cout, Kd, Cin, Kd, ld, conv=(HW, HW, Cin), col_stats=stats, stat_replicas=3)
        else:
            K.gemm(1, K.A_KC, K.B_KC, x, w, y, M, Cout, Kd, Kd, Kd, ld, col_stats=stats, stat_replicas=3)
        assert (y[M] == 7).all() and (y[:, Cout:] == 7).all(), "epilogue wrote outside the M x N block"
        out[flag], st[flag] = y[:M, :Cout].clone(), stats.sum(0).double()
    monkeypatch.setattr(K, "DEBUG_FLAGS", 0)
    assert torch.equal(out[256], out[bits])
    if conv3:
        xi = x.double().view(B, HW, HW, Cin).permute(0, 3, 1, 2)
        wi = w.double().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(xi, wi, padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    else:
        ref = x.double() @ w.double().t()
    close(out[bits], ref.float(), dt, scale=1.0)
    for s in (st[256], st[bits]):
        assert _rel_l2(s[:, 1], (ref ** 2).sum(0)) < 2e-3
        assert float((s[:, 0] - ref.sum(0)).abs().max()) < 2e-3 * math.sqrt(M)
    assert _rel_l2(st[bits], st[256]) < 1e-5


@pytest.mark.parametrize("case", [
    # (B, HW, Cin, Cout, 3x3?, rows bit, distance bits)   bit 9: 256-row tile, bit 10: 192-row tile; bits 12-14: DMA distance
    (40, 26, 64, 512, True, 512, 0),               # M = 27040: ragged last row tile, nine k-tiles (odd: one all-zero tile)
    (40, 26, 64, 512, True, 1024, 0),              # the same on 192-row tiles (ragged too)
    (8, 52, 128, 256, True, 512, 3 << 12),         # 18 k-tiles, distance 3
    (8, 52, 128, 256, True, 1024, 6 << 12),        # 192 rows, distance 6
    (3, 26, 192, 256, True, 512, 7 << 12),         # 27 k-tiles, M = 2028 (7.9 tiles), distance 7
    (8, 52, 256, 512, False, 512, 0),              # 1x1 / linear form, K = 256
    (8, 52, 320, 256, False, 1024, 0),             # K = 320: five k-tiles
    (1, 9, 64, 256, True, 512, 0),                 # M = 81: a single, mostly empty tile
    (8, 52, 512, 256, False, 524288, 0),           # 128-row tile (bit 19), linear form
    (5, 26, 128, 256, True, 524288, 0),            # 128-row tile, 3x3 form, ragged last row tile
])
def test_ping_pong_256x256x64_kernel(K, monkeypatch, case):
    """gemm_pp_kernel (csrc/gemm_pp.hip: two wave groups one barrier apart, 64-deep k-tiles in four half-tiles) against float64 and
    against the gemm_dma16_kernel / 128 x 128 path (debug bit 11) on the same operands: outputs within bf16 rounding of the float64
    result and no further from it than the reference kernel, BatchNorm column statistics (replica mode, and for the 256-row tile the
    deterministic slab mode), nothing written outside the M x N block."""
    B, HW, Cin, Cout, conv3, rows_bit, dist = case
    dt = torch.bfloat16
    M = B * HW * HW
    Kd = 9 * Cin if conv3 else Cin
    x = rnd(M, Cin, dt=dt)
    w = (rnd(Cout, Kd, dt=dt, seed=1) * Kd ** -0.5).to(dt)
    ld = Cout + 8
    out, st = {}, {}
    for flag in (2048, rows_bit | dist):
        monkeypatch.setattr(K, "DEBUG_FLAGS", flag)
        y = torch.full((M + 1, ld), 7.0, device="cuda", dtype=dt)
        stats = torch.zeros(3, Cout, 2, device="cuda")
        if conv3:
            K.gemm(1, K.A_IM2COL, K.B_KC, x, w, y, M, Cout, Kd, Cin, Kd, ld, conv=(HW, HW, Cin), col_stats=stats, stat_replicas=3)
        else:
            K.gemm(1, K.A_KC, K.B_KC, x, w, y, M, Cout, Kd, Kd, Kd, ld, col_stats=stats, stat_replicas=3)
        assert (y[M] == 7).all() and (y[:, Cout:] == 7).all(), "epilogue wrote outside the M x N block"
        out[flag], st[flag] = y[:M, :Cout].clone(), stats.sum(0).double()
    if conv3:
        xi = x.double().view(B, HW, HW, Cin).permute(0, 3, 1, 2)
        wi = w.double().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(xi, wi, padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    else:
        ref = x.double() @ w.double().t()
    new, old = out[rows_bit | dist], out[2048]
    close(new, ref.float(), dt, scale=1.0)
    assert _rel_l2(new.double(), ref) <= 1.05 * _rel_l2(old.double(), ref) + 1e-6
    for s_ in (st[2048], st[rows_bit | dist]):
        assert _rel_l2(s_[:, 1], (ref ** 2).sum(0)) < 2e-3
        assert float((s_[:, 0] - ref.sum(0)).abs().max()) < 2e-3 * math.sqrt(M)
    assert _rel_l2(st[rows_bit | dist], st[2048]) < 1e-4
    if rows_bit == 512:      # slab mode: one (sum, sum of squares) row per 128 matrix rows, plain stores
        monkeypatch.setattr(K, "DEBUG_FLAGS", rows_bit | dist)
        y = torch.empty(M, Cout, device="cuda", dtype=dt)
        slab = torch.full((K.stat_tiles(M) + 1, Cout, 2), 3.0, device="cuda")
        if conv3:
            K.gemm(1, K.A_IM2COL, K.B_KC, x, w, y, M, Cout, Kd, Cin, Kd, Cout, conv=(HW, HW, Cin), col_stats=slab, stat_replicas=0)
        else:
            K.gemm(1, K.A_KC, K.B_KC, x, w, y, M, Cout, Kd, Kd, Kd, Cout, col_stats=slab, stat_replicas=0)
        assert (slab[-1] == 3).all()
        assert torch.equal(y, new)
        assert _rel_l2(slab[:-1].sum(0).double(), st[rows_bit | dist]) < 1e-5
        r0 = 128 * (K.stat_tiles(M) - 1)
        assert _rel_l2(slab[-2, :, 1].double(), (ref[r0:] ** 2).sum(0)) < 2e-3
    monkeypatch.setattr(K, "DEBUG_FLAGS", 0)


@pytest.mark.parametrize("case", [
    # (B, H, W, Cin, Cout, statistics)
    (2, 20, 32, 32, 32, True), (3, 5, 40, 64, 64, True), (1, 7, 208, 32, 64, False), (2, 9, 24, 64, 32, True), (2, 13, 104, 64, 64, True),
    (1, 1, 16, 32, 32, True), (1, 5, 208, 64, 32, False), (2, 11, 208, 64, 64, True),      # (208 x 64 channels: single-row groups)
])
def test_sliding_window_small_channel_convolution(K, monkeypatch, case):
    """conv_sw_kernel (csrc/conv_sw.hip: input rows once through an LDS ring, weights in registers as the MFMA's A operand) against
    float64 and against the implicit-GEMM kernels (debug bit 20) on the same operands: image borders inside a strip (strips cross
    images), ragged last pixel block, 1-row images, BatchNorm statistics in replica mode, nothing written outside the output."""
    B, H, W, Cin, Cout, with_stats = case
    dt = torch.bfloat16
    M = B * H * W
    x = rnd(M, Cin, dt=dt)
    w = (rnd(Cout, 9 * Cin, dt=dt, seed=1) * (9 * Cin) ** -0.5).to(dt)
    ld = Cout + 8
    out, st = {}, {}
    for flag in (1048576, 2097152):      # implicit GEMM / sliding window forced (bit 21 lifts the size threshold)
        monkeypatch.setattr(K, "DEBUG_FLAGS", flag)
        y = torch.full((M + 1, ld), 7.0, device="cuda", dtype=dt)
        stats = torch.zeros(3, Cout, 2, device="cuda") if with_stats else None
        K.gemm(1, K.A_IM2COL, K.B_KC, x, w, y, M, Cout, 9 * Cin, Cin, 9 * Cin, ld, conv=(H, W, Cin), col_stats=stats, stat_replicas=3 if with_stats else 0)
        assert (y[M] == 7).all() and (y[:, Cout:] == 7).all(), "wrote outside the M x N block"
        out[flag] = y[:M, :Cout].clone()
        st[flag] = stats.sum(0).double() if with_stats else None
    monkeypatch.setattr(K, "DEBUG_FLAGS", 0)
    xi = x.double().view(B, H, W, Cin).permute(0, 3, 1, 2)
    wi = w.double().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xi, wi, padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    new, old = out[2097152], out[1048576]
    close(new, ref.float(), dt, scale=1.0)
    assert _rel_l2(new.double(), ref) <= 1.05 * _rel_l2(old.double(), ref) + 1e-6
    if with_stats:
        for s_ in (st[1048576], st[2097152]):
            assert _rel_l2(s_[:, 1], (ref ** 2).sum(0)) < 2e-3
            assert float((s_[:, 0] - ref.sum(0)).abs().max()) < 2e-3 * math.sqrt(M) + 1e-3
        assert _rel_l2(st[2097152], st[1048576]) < 1e-3


@pytest.mark.parametrize("M,with_stats", [(70001, True), (16384, False), (16384 + 4 * 32 * 4 + 31, True), (1384448, True)])
def test_streamed_skinny_gemm_of_the_stem(K, M, with_stats):
    """gemm_skinny32_kernel (csrc/gemm_skinny.hip: [M][32] x [32][32]^T, weights in registers, A streamed into the MFMA without an LDS stage)
    against float64 and against the tiled kernel on the same operands (M below the dispatch threshold is not possible for the same M, so
    the tiled kernel runs on a row slice): ragged last tile, a last batch of tiles that is partly past the end, BatchNorm statistics in
    replica mode, nothing written outside the output."""
    dt = torch.bfloat16
    a = rnd(M, 32, dt=dt)
    a[:, 27:] = 0          # (the stem's 27 taps padded to 32)
    w = (rnd(32, 32, dt=dt, seed=1) * 27 ** -0.5).to(dt)
    y = torch.full((M + 1, 32), 7.0, device="cuda", dtype=dt)
    stats = torch.zeros(5, 32, 2, device="cuda") if with_stats else None
    K.gemm(1, K.A_KC, K.B_KC, a, w, y, M, 32, 32, 32, 32, 32, col_stats=stats, stat_replicas=5 if with_stats else 0)
    assert (y[M] == 7).all(), "wrote past the last row"
    ref = a.double() @ w.double().t()
    close(y[:M], ref.float(), dt, scale=1.0)
    n = 8192                # below the dispatch threshold: the tiled LDS-DMA kernel
    y2 = torch.empty(n, 32, device="cuda", dtype=dt)
    K.gemm(1, K.A_KC, K.B_KC, a[:n], w, y2, n, 32, 32, 32, 32, 32)
    assert _rel_l2(y[:n].double(), ref[:n]) <= 1.05 * _rel_l2(y2.double(), ref[:n]) + 1e-6
    if with_stats:
        st = stats.sum(0).double()
        assert _rel_l2(st[:, 1], (ref ** 2).sum(0)) < 2e-3
        assert float((st[:, 0] - ref.sum(0)).abs().max()) < 2e-3 * math.sqrt(M) + 1e-3


@pytest.mark.parametrize("case", [(2, 20, 32, 32, 32), (3, 5, 40, 64, 64), (1, 7, 208, 32, 64), (2, 9, 24, 64, 32), (2, 13, 104, 64, 64), (1, 1, 16, 32, 32)])
def test_sliding_window_small_channel_weight_gradient(K, monkeypatch, case):
    """wgrad_sw_kernel (csrc/wgrad_sw.hip: one wave per tap, operands through LDS row rings, transposed fragment reads) against float64
    and against the implicit-GEMM weight gradient (debug bit 22) on the same operands; accumulates onto what the gradient held."""
    B, H, W, Cin, Cout = case
    dt = torch.bfloat16
    M = B * H * W
    x = rnd(M, Cin, dt=dt)
    dy = (rnd(M, Cout, dt=dt, seed=1) * 0.1).to(dt)
    N = 9 * Cin
    out = {}
    for flag in (4194304, 8388608):      # implicit GEMM / sliding window forced (bit 23 lifts the size threshold)
        monkeypatch.setattr(K, "DEBUG_FLAGS", flag)
        g = torch.full((Cout + 1, N + 8), 0.5, device="cuda")
        K.gemm(1, K.A_MC, K.B_NC_IM2COL, dy, x, g, Cout, N, M, Cout, Cin, N + 8, splitk=4, out_mode=K.OUT_F32_ATOMIC, conv=(H, W, Cin))
        assert (g[Cout] == 0.5).all() and (g[:, N:] == 0.5).all(), "wrote outside the M x N block"
        out[flag] = g[:Cout, :N].double() - 0.5
    monkeypatch.setattr(K, "DEBUG_FLAGS", 0)
    xi = x.double().view(B, H, W, Cin).permute(0, 3, 1, 2)
    cols = torch.nn.functional.unfold(xi, 3, padding=1).view(B, Cin, 9, H * W).permute(0, 3, 2, 1).reshape(M, 9 * Cin)
    ref = dy.double().t() @ cols
    assert _rel_l2(out[8388608], ref) < 2e-3, _rel_l2(out[8388608], ref)
    assert _rel_l2(out[4194304], ref) < 2e-3
    # slab form: 7 strips (more than rows for the 1-row case: empty strips store zeros) + the ordered reduction, twice: bit-identical
    monkeypatch.setattr(K, "DEBUG_FLAGS", 8388608)
    res = []
    for _ in range(2):
        ws = torch.full((7, Cout, N + 4), float("nan"), device="cuda")
        K.gemm(1, K.A_MC, K.B_NC_IM2COL, dy, x, ws, Cout, N, M, Cout, Cin, N + 4, splitk=7, out_mode=K.OUT_F32, conv=(H, W, Cin))
        g = torch.full((Cout, N), 0.25, device="cuda")
        K.splitk_reduce(ws, 7, Cout, N, N + 4, g, 0, N, accumulate=True)
        res.append(g)
    monkeypatch.setattr(K, "DEBUG_FLAGS", 0)
    assert torch.equal(res[0], res[1])
    assert _rel_l2(res[0].double() - 0.25, ref) < 2e-3


def test_grouped_weight_gradients_in_one_launch(K):
    """crog_gemm_group: dense and 3x3 weight gradients of different sizes and splits side by side in one launch of the ping-pong
    weight-gradient kernel; every output equals the float64 product (and what was in the gradient before), nothing else is touched."""
    dt = torch.bfloat16
    probs = [  # (pixels (B, H, W), Cin, Cout, 3x3?, splitk[, bias gradient rides along])
        ((8, 26, 26), 512, 512, False, 4), ((8, 26, 26), 256, 256, True, 3), ((8, 26, 26), 1024, 264, False, 2),
        ((2, 52, 52), 64, 256, True, 5), ((3, 13, 13), 512, 264, True, 1), ((8, 26, 26), 256, 1024, False, 4),
        # round 5: a_sum blocks (bias gradients) and short unsplit reductions (the text tower's 640 token rows: plain read-modify-write)
        ((8, 26, 26), 512, 520, False, 4, True), ((32, 20, 1), 512, 1536, False, 1, True), ((32, 20, 1), 2048, 512, False, 1, True),
        ((1, 13, 10), 512, 264, False, 1, True), ((32, 20, 1), 512, 512, False, 1)]
    sink, keep, want, biases = [], [], [], []
    K.GROUP_SINK = sink
    try:
        for i, ((B, H, W), cin, cout, conv3, sk, *rest) in enumerate(probs):
            Kd = B * H * W
            N = 9 * cin if conv3 else cin
            x = rnd(Kd, cin, dt=dt, seed=2 * i)
            dy = (rnd(Kd, cout, dt=dt, seed=2 * i + 1) * 0.1).to(dt)
            g = torch.full((cout + 1, N + 8), 0.5, device="cuda")
            bg = torch.full((cout + 8,), 0.25, device="cuda") if rest and rest[0] else None
            biases.append((bg, dy.double().sum(0)))
            K.gemm(1, K.A_MC, K.B_NC_IM2COL if conv3 else K.B_NC, dy, x, g, cout, N, Kd, cout, cin, N + 8, splitk=sk,
                   out_mode=K.OUT_F32_ATOMIC, conv=(H, W, cin) if conv3 else (0, 0, 0), a_sum=bg)
            if conv3:
                xi = x.double().view(B, H, W, cin).permute(0, 3, 1, 2)
                cols = torch.nn.functional.unfold(xi, 3, padding=1).view(B, cin, 9, H * W).permute(0, 3, 2, 1).reshape(Kd, 9 * cin)
            else:
                cols = x.double()
            keep.append((x, dy, g))
            want.append(dy.double().t() @ cols)
    finally:
        K.GROUP_SINK = None
    assert len(sink) == len(probs)
    K.gemm_group(sink)
    torch.cuda.synchronize()
    for (x, dy, g), ref, ((B, H, W), cin, cout, conv3, sk, *rest), (bg, bref) in zip(keep, want, probs, biases):
        N = ref.shape[1]
        assert (g[cout] == 0.5).all() and (g[:, N:] == 0.5).all(), "wrote outside the M x N block"
        got = g[:cout, :N].double() - 0.5
        assert _rel_l2(got, ref) < 2e-3, (cin, cout, conv3, _rel_l2(got, ref))
        if bg is not None:
            assert (bg[cout:] == 0.25).all(), "the bias sum wrote past its M columns"
            assert _rel_l2(bg[:cout].double() - 0.25, bref) < 2e-3, (cin, cout, _rel_l2(bg[:cout].double() - 0.25, bref))


@pytest.mark.parametrize("case", [
    # (M, K, N, 3x3 geometry or None, rows bit, bias, act, residual, statistics)
    (2100, 256, 512, None, 512, True, "relu", False, False),          # linear + bias + ReLU (decoder FFN), ragged last row tile
    (2100, 256, 512, None, 1024, True, "none", True, False),          # bias + residual on 192-row tiles
    (5408, 512, 256, None, 512, False, "none", True, True),           # 1x1 data gradient + identity gradient, statistics of the sum's producer
    (1352, 128, 256, None, 1024, False, "relu_post", True, False),    # ReLU after the residual
    (2704, 2048, 256, None, 512, True, "quickgelu", False, False),    # 32 k-tiles
    (2 * 26 * 26, 9 * 64, 256, (26, 26, 64), 512, True, "relu", True, False),   # 3x3 form with everything
    (2704, 512, 2048, None, 524288, True, "relu", False, False),      # 128-row tile (bit 19) with bias + ReLU: the decoder FFN at B = 4
    (2100, 256, 256, None, 524288, False, "relu_post", True, True),   # 128-row tile, residual + ReLU after it, statistics, ragged
])
def test_ping_pong_kernel_full_epilogue(K, monkeypatch, case):
    """gemm_pp_kernel<..., EPI = true>: bias, activation, residual and the ReLU after the residual in crog_gemm's order (+ bias,
    statistics, activation, + R, ReLU), against float64 and against the 128 x 128 kernel (debug bit 17) on the same operands."""
    M, Kd, N, geom, rows_bit, with_bias, act, with_res, with_stats = case
    dt = torch.bfloat16
    cin = geom[2] if geom else Kd
    x = rnd(M, cin, dt=dt)
    w = (rnd(N, Kd, dt=dt, seed=1) * Kd ** -0.5).to(dt)
    bias = rnd(N, seed=2) if with_bias else None
    res = rnd(M, N + 8, dt=dt, seed=3) if with_res else None
    code = dict(none=K.ACT_NONE, relu=K.ACT_RELU, relu_post=K.ACT_RELU_POST, quickgelu=K.ACT_QUICKGELU)[act]
    ld = N + 8
    out, st = {}, {}
    for flag in (131072, rows_bit):
        monkeypatch.setattr(K, "DEBUG_FLAGS", flag)
        y = torch.full((M + 1, ld), 7.0, device="cuda", dtype=dt)
        stats = torch.zeros(3, N, 2, device="cuda") if with_stats else None
        kw = dict(bias=bias, act=code, R=res, ldr=N + 8 if with_res else 0, col_stats=stats, stat_replicas=3 if with_stats else 0)
        if geom:
            K.gemm(1, K.A_IM2COL, K.B_KC, x, w, y, M, N, Kd, cin, Kd, ld, conv=geom, **kw)
        else:
            K.gemm(1, K.A_KC, K.B_KC, x, w, y, M, N, Kd, Kd, Kd, ld, **kw)
        assert (y[M] == 7).all() and (y[:, N:] == 7).all(), "epilogue wrote outside the M x N block"
        out[flag] = y[:M, :N].clone()
        st[flag] = stats.sum(0).double() if with_stats else None
    if geom:
        H, W, C = geom
        xi = x.double().view(-1, H, W, C).permute(0, 3, 1, 2)
        wi = w.double().view(N, 3, 3, C).permute(0, 3, 1, 2)
        ref = torch.nn.functional.conv2d(xi, wi, padding=1).permute(0, 2, 3, 1).reshape(M, N)
    else:
        ref = x.double() @ w.double().t()
    if with_bias:
        ref = ref + bias.double()
    pre = ref.clone()
    if act == "relu":
        ref = ref.clamp_min(0)
    elif act == "quickgelu":
        ref = ref * torch.sigmoid(1.702 * ref)
    if with_res:
        ref = ref + res[:, :N].double()
    if act == "relu_post":
        ref = ref.clamp_min(0)
    new, old = out[rows_bit], out[131072]
    close(new, ref.float(), dt, scale=1.5)
    assert _rel_l2(new.double(), ref) <= 1.05 * _rel_l2(old.double(), ref) + 1e-6
    if with_stats:      # the statistics are those of the product + bias, before activation and residual
        for s_ in (st[131072], st[rows_bit]):
            assert _rel_l2(s_[:, 1], (pre ** 2).sum(0)) < 2e-3
        assert _rel_l2(st[rows_bit], st[131072]) < 1e-4
    monkeypatch.setattr(K, "DEBUG_FLAGS", 0)


@pytest.mark.parametrize("case", [
    # (pixels as (B, H, W), Cin, Cout, 3x3?, splitk, DMA distance)
    ((8, 26, 26), 256, 256, True, 3, 0),        # one M tile, nine N tiles; 84.5 k-tiles over 3 slices (ragged last k-tile)
    ((8, 26, 26), 256, 512, True, 1, 5),        # no split
    ((2, 52, 52), 64, 256, True, 2, 3),         # Cin = 64: a 128-column half-tile spans two taps; N = 576 (ragged column tile)
    ((3, 13, 13), 512, 264, True, 2, 7),        # M = 264 (ragged row tile), H W = 169 < 2 k-tiles: several image wraps per k-tile
    ((32, 26, 26), 512, 512, False, 4, 0),      # 1x1 / linear form
    ((5, 26, 26), 2048, 264, False, 3, 6),      # linear, ragged M, eight column tiles
    ((1, 20, 20), 320, 256, False, 7, 4),       # 6.25 k-tiles over 7 slices: the last slices are empty
])
def test_ping_pong_weight_gradient_kernel(K, monkeypatch, case):
    """gemm_ppt_kernel (csrc/gemm_ppt.hip: transposed LDS reads, 64-deep k-tiles, split-K) against float64 and against the previous
    weight-gradient kernels (debug bit 16) on the same operands, in both output forms: fp32 atomic adds onto a non-zero gradient, and
    split-K slabs + crog_splitk_reduce, which must give the same bits on every run."""
    (B, H, W), Cin, Cout, conv3, sk, dist = case
    dt = torch.bfloat16
    Mpix = B * H * W
    N = 9 * Cin if conv3 else Cin
    x = rnd(Mpix, Cin, dt=dt)
    dy = (rnd(Mpix, Cout, dt=dt, seed=1) * 0.1).to(dt)
    if conv3:
        xi = x.double().view(B, H, W, Cin).permute(0, 3, 1, 2)
        cols = torch.nn.functional.unfold(xi, 3, padding=1).view(B, Cin, 9, H * W).permute(0, 3, 2, 1).reshape(Mpix, 9 * Cin)
        ref = dy.double().t() @ cols
    else:
        ref = dy.double().t() @ x.double()
    ldc = N + 4
    g0 = torch.randn(Cout + 1, ldc, device="cuda")
    out = {}
    for flag in (65536, 32768 | dist << 12):
        monkeypatch.setattr(K, "DEBUG_FLAGS", flag)
        g = g0.clone()
        K.gemm(1, K.A_MC, K.B_NC_IM2COL if conv3 else K.B_NC, dy, x, g, Cout, N, Mpix, Cout, Cin, ldc, splitk=sk, out_mode=K.OUT_F32_ATOMIC,
               conv=(H, W, Cin) if conv3 else (0, 0, 0))
        assert torch.equal(g[Cout], g0[Cout]) and torch.equal(g[:, N:], g0[:, N:]), "epilogue wrote outside the M x N block"
        out[flag] = (g[:Cout, :N] - g0[:Cout, :N]).double()
    new, old = out[32768 | dist << 12], out[65536]
    assert _rel_l2(new, ref) < 2e-5 and _rel_l2(new, ref) <= 1.5 * _rel_l2(old, ref) + 1e-6     # (fp32 adds onto O(1) values: 1e-7 relative each)
    # slab form: [splitk][M][ldc] workspace, then the ordered reduction onto the gradient
    monkeypatch.setattr(K, "DEBUG_FLAGS", 32768 | dist << 12)
    res = []
    for _ in range(2):
        ws = torch.full((sk, Cout, ldc), float("nan"), device="cuda")
        K.gemm(1, K.A_MC, K.B_NC_IM2COL if conv3 else K.B_NC, dy, x, ws, Cout, N, Mpix, Cout, Cin, ldc, splitk=sk, out_mode=K.OUT_F32,
               conv=(H, W, Cin) if conv3 else (0, 0, 0))
        assert not torch.isnan(ws[:, :, :N]).any() and torch.isnan(ws[:, :, N:]).all()
        g = g0.clone()
        K.splitk_reduce(ws, sk, Cout, N, ldc, g, 0, ldc, accumulate=True)
        assert torch.equal(g[Cout], g0[Cout]) and torch.equal(g[:, N:], g0[:, N:])
        res.append(g[:Cout, :N].clone())
    assert torch.equal(res[0], res[1]), "the slab form must be bit-reproducible"
    assert _rel_l2((res[0] - g0[:Cout, :N]).double(), ref) < 2e-5
    monkeypatch.setattr(K, "DEBUG_FLAGS", 0)


@pytest.mark.parametrize("kind", ["ppt_conv3", "ppt_dense", "tile128"])
def test_transposed_operand_kernels_keep_their_bits_beside_a_bandwidth_hungry_neighbour(K, monkeypatch, kind):
    """The LDS-DMA request is inline assembly (csrc/gemm_dma.h): the compiler no longer drains the ring in front of the transposed LDS reads, so the
    counted vmcnt / barrier pairs of the schedules are the ONLY thing between a fill and the read of its slot.  A missing wait shows when fills
    land late: split-K slabs (plain stores, one deterministic result) of production-size weight gradients, alone and 12 times beside a second
    stream that streams 1 GiB copies through HBM, must be the same bits every time."""
    dt = torch.bfloat16
    if kind == "ppt_conv3":
        (B, H, W), Cin, Cout, conv3, flag = (8, 52, 52), 256, 256, True, 32768
    elif kind == "ppt_dense":
        (B, H, W), Cin, Cout, conv3, flag = (8, 52, 52), 1024, 512, False, 32768
    else:      # the 128 x 128 tile of gemm_dma_kernel<A_MC, B_NC>: the ping-pong kernel switched off (debug bit 16)
        (B, H, W), Cin, Cout, conv3, flag = (8, 52, 52), 256, 128, False, 65536
    Mpix = B * H * W
    N = 9 * Cin if conv3 else Cin
    x = rnd(Mpix, Cin, dt=dt)
    dy = (rnd(Mpix, Cout, dt=dt, seed=1) * 0.1).to(dt)
    sk = 8
    monkeypatch.setattr(K, "DEBUG_FLAGS", flag)

    def run():
        ws = torch.full((sk, Cout, N), float("nan"), device="cuda")
        K.gemm(1, K.A_MC, K.B_NC_IM2COL if conv3 else K.B_NC, dy, x, ws, Cout, N, Mpix, Cout, Cin, N, splitk=sk, out_mode=K.OUT_F32,
               conv=(H, W, Cin) if conv3 else (0, 0, 0))
        return ws

    ref = run()
    torch.cuda.synchronize()
    assert not torch.isnan(ref).any()
    side = torch.cuda.Stream()
    a, b = torch.empty(1 << 28, device="cuda"), torch.empty(1 << 28, device="cuda")
    for i in range(12):
        with torch.cuda.stream(side):
            for _ in range(3):
                b.copy_(a)
        got = run()
        torch.cuda.synchronize()
        assert torch.equal(got, ref), f"{kind}: run {i} beside the copies differs from the run alone"
    monkeypatch.setattr(K, "DEBUG_FLAGS", 0)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("shape", [(2, 13, 64, 72, 3, 64, True), (3, 10, 32, 40, 1, 32, True), (2, 9, 27, 64, 3, 32, False), (2, 12, 64, 64, 1, 64, False)])
def test_eval_batchnorm_folded_into_the_convolution(K, dt, shape):
    """crog_bn_fold_weights + the bias / ReLU / residual epilogue (CROG_ACT_RELU_POST: the ReLU after the residual) against
    conv -> BatchNorm(running statistics) -> (+ identity) -> ReLU in float64 (clip.py:44-57 under model.eval()), incl. a 3x3 weight
    whose ragged Cin is zero-padded while it is folded."""
    B, HW, Cin, Cout, k, Cpad, with_res = shape
    M = B * HW * HW
    x = torch.zeros(B, HW, HW, Cpad, device="cuda", dtype=dt)
    x[..., :Cin] = rnd(B, HW, HW, Cin, dt=dt)
    w = rnd(Cout, k, k, Cin, seed=1) * (k * k * Cin) ** -0.5                                   # fp32 master, [Cout][ky][kx][Cin]
    gamma, beta = rnd(Cout, seed=2).abs() + 0.5, rnd(Cout, seed=3) * 0.3
    mean, var = rnd(Cout, seed=4) * 0.2, rnd(Cout, seed=5).abs() + 0.3
    res = rnd(M, Cout, dt=dt, seed=6) if with_res else None
    rows, rpc = (Cout * k * k, k * k) if Cpad != Cin else (Cout, 1)
    src_cols = Cin if Cpad != Cin else k * k * Cin
    dst_cols = Cpad if Cpad != Cin else k * k * Cin
    wf = torch.empty(rows * dst_cols, device="cuda", dtype=dt)
    shift = torch.empty(Cout, device="cuda")
    K.bn_fold_weights(w.contiguous().view(-1), 0, src_cols, rpc, gamma, beta, mean, var, 1e-5, wf, dst_cols, rows, shift)
    y = torch.empty(M, Cout, device="cuda", dtype=dt)
    act = K.ACT_RELU_POST if with_res else K.ACT_RELU
    Kd = k * k * Cpad
    if k == 3:
        K.gemm(K.dcode(dt), K.A_IM2COL, K.B_KC, x, wf, y, M, Cout, Kd, Cpad, Kd, Cout, conv=(HW, HW, Cpad), bias=shift, act=act, R=res, ldr=Cout)
    else:
        K.gemm(K.dcode(dt), K.A_KC, K.B_KC, x, wf, y, M, Cout, Kd, Cpad, Kd, Cout, bias=shift, act=act, R=res, ldr=Cout)
    xi = x[..., :Cin].double().permute(0, 3, 1, 2)
    wi = w.double().permute(0, 3, 1, 2)
    z = torch.nn.functional.conv2d(xi, wi, padding=k // 2)
    sc = gamma.double() / (var.double() + 1e-5).sqrt()
    ref = (z * sc.view(1, -1, 1, 1) + (beta.double() - mean.double() * sc).view(1, -1, 1, 1)).permute(0, 2, 3, 1).reshape(M, Cout)
    if with_res:
        ref = ref + res.double()
    close(y, ref.relu().float(), dt, scale=2.0)
    assert float(y.float().min()) >= 0.0


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K_", [(128, 128, 64), (250, 72, 44), (676, 64, 676), (64, 2048, 49)])
def test_gemm_nn(K, dt, M, N, K_):
    # A[m][k] k-contig (K padded to a multiple of 8 with zeros), B_mem[k][n] n-contig
    Kp = ((K_ + 7) // 8) * 8
    a = torch.zeros(M, Kp, device="cuda", dtype=dt)
    a[:, :K_] = rnd(M, K_, dt=dt)
    b = rnd(K_, N, dt=dt, seed=1)
    c = torch.empty(M, N, device="cuda", dtype=dt)
    # K = Kp for the K-contiguous operand; rows >= K_ of B do not exist -> pass K_ and rely on A's zero pad
    bp = torch.zeros(Kp, N, device="cuda", dtype=dt)
    bp[:K_] = b
    K.gemm(K.dcode(dt), K.A_KC, K.B_NC, a, bp, c, M, N, Kp, Kp, N, N)
    close(c, a[:, :K_].float() @ b.float(), dt, scale=math.sqrt(K_) / 4)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K_,split", [(64, 64, 5000, 8), (200, 136, 333, 1), (256, 512, 2048, 4), (56, 2048, 169, 1)])
def test_gemm_tn(K, dt, M, N, K_, split):
    # wgrad shape: C[m][n] = sum_k A_mem[k][m] * B_mem[k][n]
    a, b = rnd(K_, M, dt=dt), rnd(K_, N, dt=dt, seed=1)
    c = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    asum = torch.zeros(M + 3, device="cuda")   # bias gradient folded into the weight-gradient launch (a_sum[m] += sum_k A(m, k))
    K.gemm(K.dcode(dt), K.A_MC, K.B_NC, a, b, c, M, N, K_, M, N, N, splitk=split,
           out_mode=K.OUT_F32_ATOMIC if split > 1 else K.OUT_F32, a_sum=asum, a_sum_off=1)
    close(c, a.float().t() @ b.float(), dt, scale=math.sqrt(K_) / 4)
    close(asum[1:M + 1], a.float().sum(0), dt, scale=math.sqrt(K_) / 4)
    assert float(asum[0]) == 0.0 and float(asum[M + 1:].abs().max()) == 0.0


@pytest.mark.parametrize("dt", DT)
def test_gemm_tn_kc(K, dt):
    M, N, K_ = 72, 200, 512
    a, b = rnd(K_, M, dt=dt), rnd(N, K_, dt=dt, seed=1)
    c = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    K.gemm(K.dcode(dt), K.A_MC, K.B_KC, a, b, c, M, N, K_, M, K_, N, out_mode=K.OUT_F32)
    close(c, a.float().t() @ b.float().t(), dt, scale=math.sqrt(K_) / 4)


@pytest.mark.parametrize("dt", DT)
def test_gemm_batched_heads(K, dt):
    # attention-style: q,k from a packed [B*L, 3E] buffer; scores S[b,h] = q k^T; out = P v
    B, H, L, dh = 3, 4, 50, 64
    E = H * dh
    qkv = rnd(B * L, 3 * E, dt=dt)
    Lp = ((L + 7) // 8) * 8
    S = torch.zeros(B * H, L, Lp, device="cuda", dtype=dt)
    K.gemm(K.dcode(dt), K.A_KC, K.B_KC, qkv, qkv, S, L, L, dh, 3 * E, 3 * E, Lp, batch=B * H, batch_inner=H,
           sA=(L * 3 * E, dh), sB=(L * 3 * E, dh), sC=(H * L * Lp, L * Lp), b_off=E, alpha=0.125)
    q = qkv[:, :E].float().view(B, L, H, dh).permute(0, 2, 1, 3)
    k = qkv[:, E:2 * E].float().view(B, L, H, dh).permute(0, 2, 1, 3)
    v = qkv[:, 2 * E:].float().view(B, L, H, dh).permute(0, 2, 1, 3)
    ref = 0.125 * q @ k.transpose(-1, -2)
    close(S.view(B, H, L, Lp)[..., :L], ref, dt, scale=2)
    # O = P V with P = S (row stride Lp, pad cols are zero), V n-contiguous.  The reduction length is the true key count L, as
    # MhaFn passes it: rows L .. Lp-1 of the last batch element's V lie past the end of the buffer (K = Lp multiplied whatever the
    # allocator had left there by P's zero pad, which is NaN when that memory holds a NaN)
    O = torch.empty(B * L, E, device="cuda", dtype=dt)
    K.gemm(K.dcode(dt), K.A_KC, K.B_NC, S, qkv, O, L, dh, L, Lp, 3 * E, E, batch=B * H, batch_inner=H,
           sA=(H * L * Lp, L * Lp), sB=(L * 3 * E, dh), sC=(L * E, dh), b_off=2 * E)
    refo = (S.view(B, H, L, Lp)[..., :L].float() @ v).permute(0, 2, 1, 3).reshape(B * L, E)
    close(O, refo, dt, scale=8)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 13, 13, 64, 96), (1, 26, 20, 32, 32), (3, 8, 8, 128, 256), (2, 20, 20, 64, 64)])
def test_conv3x3_fwd_dgrad_wgrad(K, dt, B, H, W, Cin, Cout):
    x = rnd(B, H, W, Cin, dt=dt)
    w = (rnd(Cout, 3, 3, Cin, dt=dt, seed=1) * 0.1).to(dt)  # KRSC
    M = B * H * W
    y = torch.empty(B, H, W, Cout, device="cuda", dtype=dt)
    stats = torch.zeros(K.stat_tiles(M), Cout, 2, device="cuda")
    K.gemm(K.dcode(dt), K.A_IM2COL, K.B_KC, x, w, y, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, conv=(H, W, Cin), col_stats=stats)
    xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wt = w.float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.conv2d(xt, wt, padding=1)
    close(y, ref.permute(0, 2, 3, 1), dt, scale=math.sqrt(9 * Cin) / 4)
    close(stats.sum(0)[:, 0], ref.sum((0, 2, 3)), dt, scale=40)
    dy = rnd(B, H, W, Cout, dt=dt, seed=5)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    dx = torch.empty(B, H, W, Cin, device="cuda", dtype=dt)
    K.gemm(K.dcode(dt), K.A_IM2COL, K.B_NC_DGRAD, dy, w, dx, M, Cin, 9 * Cout, Cout, Cin, Cin, conv=(H, W, Cout))
    close(dx, xt.grad.permute(0, 2, 3, 1), dt, scale=math.sqrt(9 * Cout) / 4)
    dw = torch.zeros(Cout, 9 * Cin, device="cuda", dtype=torch.float32)
    K.gemm(K.dcode(dt), K.A_MC, K.B_NC_IM2COL, dy, x, dw, Cout, 9 * Cin, M, Cout, Cin, 9 * Cin, conv=(H, W, Cin), splitk=3,
           out_mode=K.OUT_F32_ATOMIC)
    close(dw.view(Cout, 3, 3, Cin), wt.grad.permute(0, 2, 3, 1), dt, scale=math.sqrt(M) / 2)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DT)
def test_batchnorm_train_fwd_bwd(K, dt):
    M, C = 1000, 64
    z = (rnd(M, C, dt=dt) * 2 + 0.5).to(dt)
    res = rnd(M, C, dt=dt, seed=3)
    gamma, beta = rnd(C, seed=1).abs() + 0.5, rnd(C, seed=2)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    rpb = 64
    nb = (M + rpb - 1) // rpb
    partial = torch.empty(nb, C, 2, device="cuda")
    K.bn_partial_stats(z, partial, rpb)
    sums = torch.empty(C, 2, device="cuda")
    K.reduce_pairs(partial, nb, C, sums)
    ss, mi = torch.empty(C, 2, device="cuda"), torch.empty(C, 2, device="cuda")
    K.bn_finalize(sums, M, gamma, beta, rm, rv, 0.1, 1e-5, C, ss, mi)
    y = torch.empty_like(z)
    K.bn_apply(z, ss, res, True, y)
    zt = z.float().requires_grad_(True)
    rt = res.float().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm2, rv2 = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    ref = F.relu(F.batch_norm(zt, rm2, rv2, g_, b_, True, 0.1, 1e-5) + rt)
    close(y, ref, dt, scale=2)
    assert torch.allclose(rm, rm2, atol=1e-4) and torch.allclose(rv, rv2, atol=1e-3)
    dy = rnd(M, C, dt=dt, seed=9)
    # use the kernel's own y for the relu mask so that rounding of y near 0 cannot flip the comparison
    ref2 = F.relu(F.batch_norm(zt, None, None, g_, b_, True, 0.1, 1e-5) + rt)
    mask = (y.float() > 0).float()
    (ref2 * 0 + (F.batch_norm(zt, None, None, g_, b_, True, 0.1, 1e-5) + rt) * mask).backward(dy.float())
    K.bn_bwd_partial(dy, y, z, mi, rpb, partial)
    dbeta, dgamma = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")   # gradient vectors: the kernels ADD into them
    K.reduce_split(partial, nb, C, sums, dbeta, dgamma)
    assert torch.equal(dbeta, sums[:, 0]) and torch.equal(dgamma, sums[:, 1])
    K.reduce_split(partial, nb, C, None, dbeta, dgamma)     # second use of a shared parameter accumulates
    assert torch.equal(dbeta, 2 * sums[:, 0]) and torch.equal(dgamma, 2 * sums[:, 1])
    dbeta, dgamma = sums[:, 0].clone(), sums[:, 1].clone()
    dz, dres = torch.empty_like(z), torch.empty_like(z)
    K.bn_bwd_apply(dy, y, z, mi, gamma, sums, M, dz, dres)
    # ReLU sign bits written by the forward instead of re-reading y: same partial sums, dz and dres, bit for bit
    vec = 8 if dt == torch.bfloat16 else 4
    bits = K.relu_mask_like(y)
    y_b = torch.empty_like(z)
    K.bn_apply(z, ss, res, True, y_b, relu_mask=bits)
    assert torch.equal(y_b, y) and bits.shape == (M, C // vec)
    want = ((y.float().view(M, C // vec, vec) > 0).to(torch.int32) << torch.arange(vec, device="cuda", dtype=torch.int32)).sum(-1)
    assert torch.equal(bits.to(torch.int32), want)
    part_b = torch.empty_like(partial)
    K.bn_bwd_partial(dy, None, z, mi, rpb, part_b, relu_mask=bits)
    assert torch.equal(part_b, partial)
    dz_b, dres_b = torch.empty_like(z), torch.empty_like(z)
    K.bn_bwd_apply(dy, None, z, mi, gamma, sums, M, dz_b, dres_b, relu_mask=bits)
    assert torch.equal(dz_b, dz) and torch.equal(dres_b, dres)
    # atomic replicas + in-kernel totals (no reduction launch): same dz / dres, parameter gradients stored by the apply kernel
    for R in (1, 4):
        acc = torch.zeros(R, C, 2, device="cuda")
        K.bn_bwd_partial(dy, y, z, mi, rpb, acc, replicas=R)
        assert torch.allclose(acc.sum(0), sums, rtol=1e-4, atol=1e-3)
        dz2, dres2 = torch.empty_like(z), torch.empty_like(z)
        dg2, db2 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        K.bn_bwd_apply(dy, y, z, mi, gamma, acc, M, dz2, dres2, sum_rows=R, dgamma=dg2, dbeta=db2)
        assert torch.allclose(db2, dbeta, rtol=1e-4, atol=1e-3) and torch.allclose(dg2, dgamma, rtol=1e-4, atol=1e-3)
        # SyncBatchNorm form: the rows hold GLOBAL totals, the parameter gradients are stored times 1 / world
        dg4, db4 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        K.bn_bwd_apply(dy, y, z, mi, gamma, acc, M, dz2, dres2, sum_rows=R, dgamma=dg4, dbeta=db4, param_grad_scale=0.25)
        assert torch.allclose(db4, 0.25 * db2, rtol=1e-6, atol=1e-6) and torch.allclose(dg4, 0.25 * dg2, rtol=1e-6, atol=1e-6)
        close(dz2, dz.float(), dt, scale=1)
        assert torch.equal(dres2, dres)
    # fused single-replica forward statistics == two-step path
    ss2, mi2 = torch.empty(C, 2, device="cuda"), torch.empty(C, 2, device="cuda")
    K.bn_partial_stats(z, partial, rpb)
    K.bn_reduce_finalize(partial, nb, M, gamma, beta, None, None, 0.1, 1e-5, C, ss2, mi2)
    assert torch.allclose(ss2, ss, rtol=1e-5, atol=1e-6) and torch.allclose(mi2, mi, rtol=1e-5, atol=1e-6)
    close(dz, zt.grad, dt, scale=4)
    close(dres, rt.grad, dt)
    close(sums[:, 0], b_.grad, dt, scale=30)
    close(sums[:, 1], g_.grad, dt, scale=30)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("shape", [(2, 8, 12, 64), (3, 26, 26, 128), (1, 4, 4, 32)])
def test_batchnorm_relu_avgpool_in_one_pass_each_way(K, dt, shape):
    """crog_bn_apply_stats_pool / crog_bn_bwd_partial_pool / crog_bn_bwd_apply_pool (clip.py:49-50, 213-214: AvgPool2d(2) after
    bn + relu): the pooled output and the gradient of z from the POOLED gradient, against torch's batch_norm -> relu -> avg_pool2d."""
    B, H, W, C = shape
    M = B * H * W
    z = rnd(M, C, dt=dt)
    gamma, beta = rnd(C, seed=1) * 0.5 + 1.0, rnd(C, seed=2) * 0.1
    zf = z.float()
    sums = torch.zeros(2, C, 2, device="cuda")
    sums[0, :, 0], sums[0, :, 1] = zf.sum(0), (zf * zf).sum(0)      # what the GEMM epilogue leaves (one replica used, one empty)
    ss, mi = torch.empty(C, 2, device="cuda"), torch.empty(C, 2, device="cuda")
    y = torch.empty(B, H // 2, W // 2, C, device="cuda", dtype=dt)
    K.bn_apply_stats(z, sums, 2, float(M), gamma, beta, None, None, 0.1, 1e-5, ss, mi, None, True, y, pool=(H, W))
    zt = zf.view(B, H, W, C).permute(0, 3, 1, 2).clone().requires_grad_(True)
    gt, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.avg_pool2d(F.relu(F.batch_norm(zt, None, None, gt, bt, True, 0.0, 1e-5)), 2)
    close(y, ref.permute(0, 2, 3, 1), dt, scale=2)
    dyp = rnd(B * (H // 2) * (W // 2), C, dt=dt, seed=3)
    ref.backward(dyp.float().view(B, H // 2, W // 2, C).permute(0, 3, 1, 2))
    rpb = K.bn_rows_per_block(M)
    nb = (M + rpb - 1) // rpb
    from crog_amd.functional import stat_replicas
    R = stat_replicas(nb, C)
    part = torch.zeros(R, C, 2, device="cuda")
    K.bn_bwd_partial(dyp, None, z, mi, rpb, part, ss, replicas=R, pool=(H, W))
    dz = torch.empty_like(z)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    K.bn_bwd_apply(dyp, None, z, mi, gamma, part, float(M), dz, None, ss, sum_rows=R, dgamma=dg, dbeta=db, pool=(H, W))
    close(dz, zt.grad.permute(0, 2, 3, 1).reshape(M, C), dt, scale=4)
    close(db, bt.grad, dt, scale=30)
    close(dg, gt.grad, dt, scale=30)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("C", [512, 2048, 768, 1536])      # (> 128 vectors per row: the one-row-per-block kernels)
def test_layernorm(K, dt, C):
    M, R = 300, 100
    x, res, pos = rnd(M, C, dt=dt), rnd(M, C, dt=dt, seed=1), rnd(R, C, dt=dt, seed=2)
    g, b = rnd(C, seed=3), rnd(C, seed=4)
    out, out2 = torch.empty_like(x), torch.empty_like(x)
    stats = torch.empty(M, 2, device="cuda")
    K.ln_fwd(x, g, b, 1e-5, out, stats, res=res, out2=out2, pos=pos)
    xt = x.float().requires_grad_(True)
    gt, bt = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xt, (C,), gt, bt, 1e-5) + res.float()
    close(out, ref, dt, scale=2)
    close(out2, ref + pos.float().repeat(M // R, 1), dt, scale=2)
    d1, d2 = rnd(M, C, dt=dt, seed=5), rnd(M, C, dt=dt, seed=6)
    ref.backward(d1.float() + d2.float())
    rpb = K.ln_bwd_rows_per_block(M)
    nb = (M + rpb - 1) // rpb
    partial = torch.empty(nb, C, 2, device="cuda")
    dx = torch.empty_like(x)
    K.ln_bwd(d1, d2, x, g, stats, dx, partial, rpb)
    close(dx, xt.grad, dt, scale=4)
    sums = torch.empty(C, 2, device="cuda")
    K.reduce_pairs(partial, nb, C, sums)
    close(sums[:, 0], gt.grad, dt, scale=40)
    close(sums[:, 1], bt.grad, dt, scale=40)
    # atomic form: the blocks add their sums straight into (pre-existing) gradient vectors — same dx, same totals, accumulating
    dg, db = torch.full((C,), 0.5, device="cuda"), torch.full((C,), -0.25, device="cuda")
    dx2 = torch.empty_like(x)
    K.ln_bwd(d1, d2, x, g, stats, dx2, None, rpb, dgamma=dg, dbeta=db)
    assert torch.equal(dx2, dx)
    assert torch.allclose(dg - 0.5, sums[:, 0], rtol=1e-4, atol=1e-3) and torch.allclose(db + 0.25, sums[:, 1], rtol=1e-4, atol=1e-3)
    # dxadd: the gradient of a residual branch around the norm is added to dx in fp32 inside the kernel (same parameter sums)
    extra = rnd(M, C, dt=dt, seed=7)
    dx3, partial3 = torch.empty_like(x), torch.empty_like(partial)
    K.ln_bwd(d1, d2, x, g, stats, dx3, partial3, rpb, dxadd=extra)
    assert torch.equal(partial3, partial)
    close(dx3, xt.grad + extra.float(), dt, scale=4)
    if dt == torch.float32:
        assert torch.equal(dx3, dx + extra)          # fp32: bit-identical to the separate accumulation pass it replaces
    # relu_in: x is a ReLU output, dx is the gradient of the ReLU's input (crog_ln_bwd_relu)
    xr = x.clamp_min(0)
    stats_r = torch.empty(M, 2, device="cuda")
    K.ln_fwd(xr, g, b, 1e-5, torch.empty_like(x), stats_r)
    dx4, dx5 = torch.empty_like(x), torch.empty_like(x)
    K.ln_bwd(d1, d2, xr, g, stats_r, dx4, partial, rpb)
    K.ln_bwd(d1, d2, xr, g, stats_r, dx5, torch.empty_like(partial), rpb, relu_in=True)
    assert torch.equal(dx5, torch.where(xr > 0, dx4, torch.zeros_like(dx4)))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("C", [512, 2048])
def test_layernorm_dropout_consistency(K, dt, C):
    # dropout masks are recomputed in backward from (seed, index): d(out)/d(x) must use the same mask
    M = 64
    x = rnd(M, C, dt=dt)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    out = torch.empty_like(x)
    stats = torch.empty(M, 2, device="cuda")
    K.ln_fwd(x, g, b, 1e-5, out, stats, p_out=0.5, seed_out=123)
    frac = (out.float() == 0).float().mean().item()
    assert 0.4 < frac < 0.6
    dx = torch.empty_like(x)
    partial = torch.empty(16, C, 2, device="cuda")
    K.ln_bwd(torch.ones_like(x), None, x, g, stats, dx, partial, 4, p_out=0.5, seed_out=123)
    sums = torch.empty(C, 2, device="cuda")
    K.reduce_pairs(partial, 16, C, sums)
    kept = (out.float() != 0).float()
    close(sums[:, 1], 2.0 * kept.sum(0), dt, scale=2)  # dbeta = sum of kept * 1/(1-p)


@pytest.mark.parametrize("dt", DT)
def test_softmax(K, dt):
    B, H, Lq, Lk = 2, 3, 40, 20
    ldp = 24
    S = torch.zeros(B * H * Lq, ldp, device="cuda", dtype=dt)
    S[:, :Lk] = rnd(B * H * Lq, Lk, dt=dt)
    kpm = torch.zeros(B, Lk, dtype=torch.bool, device="cuda")
    kpm[0, 15:] = True
    kpm[1, 7:] = True
    s0 = S.clone()
    K.softmax_fwd(S, B * H * Lq, Lq, Lk, ldp, H, False, kpm, None, 0.0, 0)
    st = s0[:, :Lk].float().view(B, H, Lq, Lk).requires_grad_(True)
    ref = torch.softmax(st.masked_fill(kpm[:, None, None, :], float("-inf")), -1)
    close(S[:, :Lk].view(B, H, Lq, Lk), ref, dt)
    assert (S[:, Lk:] == 0).all()
    dP = torch.zeros_like(S)
    dP[:, :Lk] = rnd(B * H * Lq, Lk, dt=dt, seed=4)
    ref.backward(dP[:, :Lk].float().view(B, H, Lq, Lk))
    K.softmax_bwd(S, dP, B * H * Lq, Lk, ldp, 0.0, 0)
    close(dP[:, :Lk].view(B, H, Lq, Lk), st.grad, dt)
    # causal
    L = 20
    S2 = torch.zeros(H * L, ldp, device="cuda", dtype=dt)
    S2[:, :L] = rnd(H * L, L, dt=dt, seed=8)
    ref2 = torch.softmax(S2[:, :L].float().view(H, L, L) + torch.full((L, L), float("-inf"), device="cuda").triu(1), -1)
    K.softmax_fwd(S2, H * L, L, L, ldp, H, True, None, None, 0.0, 0)
    close(S2[:, :L].view(H, L, L), ref2, dt)


@pytest.mark.parametrize("dt", DT)
def test_pool_upsample(K, dt):
    B, H, W, C = 2, 6, 10, 32
    x = rnd(B, H, W, C, dt=dt)
    xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    y = torch.empty(B, H // 2, W // 2, C, device="cuda", dtype=dt)
    K.avgpool2_fwd(x, y)
    ref = F.avg_pool2d(xt, 2)
    close(y, ref.permute(0, 2, 3, 1), dt)
    dy = rnd(B, H // 2, W // 2, C, dt=dt, seed=2)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    dx = torch.empty_like(x)
    K.avgpool2_bwd(dy, dx)
    close(dx, xt.grad.permute(0, 2, 3, 1), dt)
    other = rnd(B, H, W, C, dt=dt, seed=5)          # the gradient another consumer of x already produced: added while dx is written
    dx2 = torch.empty_like(x)
    K.avgpool2_bwd(dy, dx2, add=other)
    close(dx2, xt.grad.permute(0, 2, 3, 1) + other.float(), dt)
    xt.grad = None
    up = torch.empty(B, 2 * H, 2 * W, C, device="cuda", dtype=dt)
    K.upsample2_fwd(x, up)
    ref = F.interpolate(xt, scale_factor=2, mode="bilinear")
    close(up, ref.permute(0, 2, 3, 1), dt)
    dup = rnd(B, 2 * H, 2 * W, C, dt=dt, seed=3)
    ref.backward(dup.float().permute(0, 3, 1, 2))
    K.upsample2_bwd(dup, dx)
    close(dx, xt.grad.permute(0, 2, 3, 1), dt, scale=2)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("shape", [(2, 13, 13, 64), (1, 5, 7, 24), (3, 1, 1, 8), (2, 1, 4, 16), (1, 4, 1, 40), (2, 26, 26, 512)])
def test_upsample_borders_odd_sizes_and_ragged_channel_counts(K, dt, shape):
    """The patch (forward) / pair (backward) kernels against F.interpolate on odd widths (a pair without its second member), single-row and
    single-column maps (both border rules on one pixel) and channel counts whose vector count is not a power of two."""
    B, H, W, C = shape
    x = rnd(B, H, W, C, dt=dt, seed=11)
    xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    up = torch.empty(B, 2 * H, 2 * W, C, device="cuda", dtype=dt)
    K.upsample2_fwd(x, up)
    ref = F.interpolate(xt, scale_factor=2, mode="bilinear")
    close(up, ref.permute(0, 2, 3, 1), dt)
    dup = rnd(B, 2 * H, 2 * W, C, dt=dt, seed=12)
    ref.backward(dup.float().permute(0, 3, 1, 2))
    dx = torch.full_like(x, float("nan"))
    K.upsample2_bwd(dup, dx)
    close(dx, xt.grad.permute(0, 2, 3, 1), dt, scale=2)


@pytest.mark.parametrize("dt", DT)
def test_embedding_gather(K, dt):
    B, L, C, V = 4, 20, 512, 1000
    tok, pos = rnd(V, C, dt=dt), rnd(77, C, dt=dt, seed=1)
    word = torch.randint(0, V, (B, L), device="cuda")
    out = torch.empty(B * L, C, device="cuda", dtype=dt)
    K.embedding_fwd(word, tok, pos, out, L, V)
    close(out.view(B, L, C), tok.float()[word] + pos.float()[:L], dt)
    dout = rnd(B * L, C, dt=dt, seed=2)
    dtok, dpos = torch.zeros(V, C, device="cuda"), torch.zeros(77, C, device="cuda")
    K.embedding_bwd(word, dout, dtok, dpos, L, V)
    ref = torch.zeros(V, C, device="cuda").index_add_(0, word.view(-1), dout.float())
    close(dtok, ref, dt)
    close(dpos[:L], dout.float().view(B, L, C).sum(0), dt)
    idx = torch.tensor([3, 25, 47, 79], device="cuda")
    g = torch.empty(4, C, device="cuda", dtype=dt)
    K.gather_rows(out, idx, g)
    assert torch.equal(g, out[idx])
    dx = torch.zeros_like(out)
    K.scatter_rows(g, idx, dx)
    assert torch.equal(dx[idx], g) and torch.count_nonzero(dx) == torch.count_nonzero(g)   # rows outside idx stay zero


@pytest.mark.parametrize("dt", DT)
def test_bcast_ops(K, dt):
    B, P, C = 3, 20, 64
    x, s = rnd(B * P, C, dt=dt), rnd(B, C, dt=dt, seed=1)
    z = torch.empty_like(x)
    K.mul_bcast_fwd(x, s, z, B, P)
    close(z.view(B, P, C), x.float().view(B, P, C) * s.float()[:, None], dt)
    dz = rnd(B * P, C, dt=dt, seed=2)
    dx, ds = torch.empty_like(x), torch.empty_like(s)
    K.mul_bcast_bwd(dz, x, s, dx, ds, B, P)
    close(dx.view(B, P, C), dz.float().view(B, P, C) * s.float()[:, None], dt)
    close(ds, (dz.float() * x.float()).view(B, P, C).sum(1), dt, scale=5)
    pos = rnd(P, C, dt=dt, seed=3)
    o = torch.empty_like(x)
    K.add_rows(x, pos, o)
    close(o.view(B, P, C), x.float().view(B, P, C) + pos.float(), dt)
    acc = torch.zeros(P, C, device="cuda")
    K.sum_over_batch(x, acc, B)
    close(acc, x.float().view(B, P, C).sum(0), dt)
    o2 = torch.empty_like(x)
    K.add_dropout(x, dz, o2, 0.0, 0)
    close(o2, x.float() + dz.float(), dt)
    K.add_dropout(None, torch.ones_like(x), o2, 0.25, 77)
    v = o2.float()
    assert set(torch.unique(v).tolist()) <= {0.0, float(torch.tensor(1 / 0.75).to(dt))}
    assert 0.15 < (v == 0).float().mean().item() < 0.35
    u = rnd(B * P, C, dt=dt, seed=7)
    a = torch.empty_like(u)
    K.quickgelu_fwd(u, a)
    ut = u.float().requires_grad_(True)
    ref = ut * torch.sigmoid(1.702 * ut)
    close(a, ref, dt)
    ref.backward(dz.float())
    du = torch.empty_like(u)
    K.act_bwd(dz, u, du, 1)
    close(du, ut.grad, dt)
    K.act_bwd(dz, u, du, 0)
    close(du, dz.float() * (u.float() > 0), dt)


@pytest.mark.parametrize("dt", DT)
def test_stem_im2col_and_casts(K, dt):
    B, H, W = 2, 16, 12
    img = rnd(B, 3, H, W)
    w = rnd(32, 3, 3, 3, seed=1)
    out = torch.empty(B * (H // 2) * (W // 2), 32, device="cuda", dtype=dt)
    K.stem_im2col(img, out)
    wk = torch.zeros(32, 32, device="cuda", dtype=dt)
    wkrsc = w.permute(0, 2, 3, 1).contiguous().view(32, 27)
    K.cast_pad2d(wkrsc, 27, 27, wk, 32, 32, 32)
    y = out.float() @ wk.float().t()
    ref = F.conv2d(img.to(dt).float(), w.to(dt).float(), stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, 32)
    close(y, ref, torch.float32, scale=10)
    n = 1000 * 8 + 5
    src = rnd(n)
    dst = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    K.cast_f32_to_bf16(src, dst, n)
    assert torch.equal(dst, src.bfloat16())
    buf = torch.full((2, 5, 7, 48), 9.0, device="cuda", dtype=dt)
    K.coord_fill(buf, 32, 48)
    xs = torch.linspace(-1, 1, 7, device="cuda")
    ys = torch.linspace(-1, 1, 5, device="cuda")
    close(buf[0, :, :, 32], xs[None, :].expand(5, 7), dt)
    close(buf[1, :, :, 33], ys[:, None].expand(5, 7), dt)
    assert (buf[..., 34:] == 0).all() and (buf[..., :32] == 9).all()


def test_adam_matches_torch(K):
    n = 10007
    p = rnd(n)
    p2 = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([p2], lr=1e-3, weight_decay=0.0)
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    shadow = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    for step in range(1, 4):
        g = rnd(n, seed=step)
        p2.grad = g.clone()
        opt.step()
        K.adam_step(p, g, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, shadow)
    assert torch.allclose(p, p2.detach(), atol=1e-6, rtol=1e-5)
    assert torch.equal(shadow, p.bfloat16())


def test_head_and_loss(K):
    B, heads, H, W, C = 2, 5, 12, 10, 32
    dt = torch.float32
    x5 = rnd(B, H, W, heads * C)
    word = rnd(B, C * 9 + 1 + 7, seed=1)  # ld padded
    wpad = torch.empty(B, C, 16, device="cuda", dtype=dt)
    K.head_pack_weights(word, wpad, B, C)
    P = H * W
    t = torch.empty(B * P * heads, 16, device="cuda")
    K.gemm(K.F32, K.A_KC, K.B_NC, x5, wpad, t, P * heads, 16, C, C, 16, 16, batch=B, sA=(P * heads * C, 0), sB=(C * 16, 0),
           sC=(P * heads * 16, 0), out_mode=K.OUT_F32)
    out = torch.empty(B, heads, H, W, device="cuda")
    K.head_stencil_fwd(t, word, C * 9, out, B, heads, H, W)
    wt = word[:, :C * 9].reshape(B, C, 3, 3)
    xs = x5.permute(0, 3, 1, 2)
    refs = []
    for h in range(heads):
        xi = xs[:, h * C:(h + 1) * C].reshape(1, B * C, H, W)
        refs.append(F.conv2d(xi, wt, padding=1, groups=B, bias=word[:, C * 9]).transpose(0, 1))
    ref = torch.cat(refs, 1)
    close(out, ref, dt, scale=4)
    # loss
    Hin, Win = 4 * H, 4 * W
    tg = [(rnd(B, 1, Hin, Win, seed=10 + i) > 0.5).float() if i == 0 else rnd(B, 1, Hin, Win, seed=10 + i) for i in range(5)]
    pred = out.clone().requires_grad_(True)
    small = [F.interpolate(t_, (H, W), mode="nearest") for t_ in tg]
    l0 = F.binary_cross_entropy_with_logits(pred[:, 0:1], small[0], weight=small[0] * 0.5 + 1)
    ls = [l0] + [F.smooth_l1_loss(pred[:, i:i + 1], small[i]) for i in range(1, 5)]
    total = sum(ls)
    total.backward()
    tgt_small = torch.empty(5, B, H, W, device="cuda")
    sums = torch.empty(5, device="cuda")
    dpred = torch.empty_like(out)
    K.head_loss(out, tg, Hin, Win, True, tgt_small, sums, dpred)
    for i in range(5):
        assert torch.allclose(sums[i], ls[i].detach(), rtol=1e-4, atol=1e-6), (i, sums[i].item(), ls[i].item())
        assert torch.equal(tgt_small[i], small[i][:, 0])
    assert torch.allclose(dpred, pred.grad, rtol=1e-4, atol=1e-7)
    # stencil backward
    dtb = torch.empty(B * P * heads, 16, device="cuda")
    dbias = torch.empty(B, device="cuda")
    K.head_stencil_bwd(dpred, dtb, dbias, B, heads, H, W)
    assert torch.allclose(dbias, dpred.sum((1, 2, 3)), rtol=1e-4, atol=1e-7)
    tt = t.clone().requires_grad_(True)
    # reference stencil via autograd on a torch restatement
    tv = tt.view(B, H, W, heads, 16)
    acc = torch.zeros(B, heads, H, W, device="cuda")
    for tap in range(9):
        dy_, dx_ = tap // 3 - 1, tap % 3 - 1
        sh = torch.zeros(B, H, W, heads, device="cuda")
        ys = slice(max(0, -dy_), H - max(0, dy_))
        xs_ = slice(max(0, -dx_), W - max(0, dx_))
        yd = slice(max(0, dy_), H - max(0, -dy_))
        xd = slice(max(0, dx_), W - max(0, -dx_))
        sh[:, ys, xs_] = tv[:, yd, xd, :, tap]
        acc = acc + sh.permute(0, 3, 1, 2)
    close(acc + word[:, C * 9][:, None, None, None], out, dt, scale=4)
    acc.backward(dpred)
    assert torch.allclose(dtb, tt.grad, rtol=1e-4, atol=1e-7)
    # metric
    counts, m2 = torch.empty(B, 2, device="cuda"), torch.empty(2, device="cuda")
    K.train_metric(out, heads * P, tgt_small[0], B, P, 0.35, 0.5, counts, m2)
    o = (torch.sigmoid(out[:, 0]).flatten(1) >= 0.35)
    tb = tgt_small[0].flatten(1).bool()
    ious = (o & tb).sum(1) / ((o | tb).sum(1) + 1e-6)
    assert torch.allclose(m2[0], 100 * ious.mean(), atol=1e-3) and torch.allclose(m2[1], 100 * (ious > 0.5).float().mean(), atol=1e-3)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,C", [(1000, 512), (37, 1536), (4, 2305), (21632, 64)])
def test_colsum(K, dt, M, C):
    ld = ((C + 7) // 8) * 8
    buf = rnd(M, ld, dt=dt)
    x = buf[:, :C]
    out = torch.ones(C + 3, device="cuda")
    K.colsum(x, out, out_off=0)
    close(out[:C], 1.0 + x.float().sum(0), dt, scale=math.sqrt(M))
    assert (out[C:] == 1).all()


@pytest.mark.parametrize("dt", DT)
def test_bn_backward_relu_mask_from_z_equals_mask_from_y(K, dt):
    M, C = 777, 128
    z = (rnd(M, C, dt=dt) * 1.5 - 0.2).to(dt)
    gamma, beta = rnd(C, seed=1), rnd(C, seed=2)          # negative gammas included
    rpb = 64
    nb = (M + rpb - 1) // rpb
    partial = torch.empty(nb, C, 2, device="cuda")
    K.bn_partial_stats(z, partial, rpb)
    ss, mi = torch.empty(C, 2, device="cuda"), torch.empty(C, 2, device="cuda")
    K.bn_reduce_finalize(partial, nb, M, gamma, beta, None, None, 0.1, 1e-5, C, ss, mi)
    y = torch.empty_like(z)
    K.bn_apply(z, ss, None, True, y)
    dy = rnd(M, C, dt=dt, seed=5)
    outs = []
    for use_z in (False, True):
        K.bn_bwd_partial(dy, None if use_z else y, z, mi, rpb, partial, ss if use_z else None)
        sums = torch.empty(C, 2, device="cuda")
        K.reduce_split(partial, nb, C, sums, None, None)
        dz = torch.empty_like(z)
        K.bn_bwd_apply(dy, None if use_z else y, z, mi, gamma, sums, M, dz, None, ss if use_z else None)
        outs.append((sums.clone(), dz.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


# ---- SSG / ViT staging kernels ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("H,W,C,k,s,p", [(9, 7, 16, 3, 2, 1), (8, 8, 32, 1, 2, 0), (5, 6, 8, 3, 1, 1), (1, 1, 8, 3, 2, 1), (13, 13, 8, 7, 2, 3)])
def test_im2col_col2im(K, dt, H, W, C, k, s, p):
    """im2col rows == F.unfold in (ky, kx, c) column order; col2im == its exact transpose (fold)."""
    B = 2
    x = rnd(B, H, W, C, dt=dt)
    OH, OW = K.conv_out(H, k, s, p), K.conv_out(W, k, s, p)
    col = torch.empty(B, OH, OW, k * k * C, device="cuda", dtype=dt)
    K.im2col_nhwc(x, col, k, k, s, p)
    xt = x.float().permute(0, 3, 1, 2)
    ref = F.unfold(xt, k, padding=p, stride=s).view(B, C, k * k, OH * OW).permute(0, 3, 2, 1).reshape(B, OH, OW, k * k * C)
    assert torch.equal(col.float(), ref)          # pure data movement: bit exact
    dcol = rnd(B, OH, OW, k * k * C, dt=dt, seed=5)
    dx = torch.empty_like(x)
    K.col2im_nhwc(dcol, dx, k, k, s, p)
    fold_in = dcol.float().view(B, OH * OW, k * k, C).permute(0, 3, 2, 1).reshape(B, C * k * k, OH * OW)
    refdx = F.fold(fold_in, (H, W), k, padding=p, stride=s).permute(0, 2, 3, 1)
    close(dx, refdx, dt)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("Cimg", [3, 4])
def test_im2col_image_and_patchify(K, dt, Cimg):
    B, H, W = 2, 18, 22
    img = rnd(B, Cimg, H, W)
    k, s, p = 7, 2, 3
    OH, OW = K.conv_out(H, k, s, p), K.conv_out(W, k, s, p)
    kc = k * k * Cimg
    kp = (kc + 7) // 8 * 8
    col = torch.full((B, OH, OW, kp), 7.0, device="cuda", dtype=dt)
    K.im2col_image(img, col, k, k, s, p)
    ref = F.unfold(img, k, padding=p, stride=s).view(B, Cimg, k * k, OH * OW).permute(0, 3, 2, 1).reshape(B, OH, OW, kc)
    assert torch.equal(col[..., :kc].float(), ref.to(dt).float())
    assert float(col[..., kc:].abs().max()) == 0.0 if kp > kc else True
    if Cimg == 3:
        P = 2
        out = torch.empty(B * (H // P) * (W // P), 3 * P * P, device="cuda", dtype=dt)
        K.patchify(img, out, P)
        refp = F.unfold(img, P, stride=P).view(B, 3, P * P, -1).permute(0, 3, 2, 1).reshape(-1, 3 * P * P)
        assert torch.equal(out.float(), refp.to(dt).float())


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("H,W", [(8, 8), (7, 9), (1, 3)])
def test_maxpool3s2(K, dt, H, W):
    B, C = 2, 16
    x = rnd(B, H, W, C, dt=dt)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, OH, OW, C, device="cuda", dtype=dt)
    arg = torch.empty(B, OH, OW, C, device="cuda", dtype=torch.uint8)
    K.maxpool3s2_fwd(x, y, arg)
    xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.max_pool2d(xt, 3, 2, 1)
    assert torch.equal(y.float(), ref.permute(0, 2, 3, 1))
    dy = rnd(B, OH, OW, C, dt=dt, seed=3)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    dx = torch.empty_like(x)
    K.maxpool3s2_bwd(dy, arg, dx)
    close(dx, xt.grad.permute(0, 2, 3, 1), dt)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("H,W", [(6, 10), (17, 17), (1, 2)])
def test_upsample2_align_corners(K, dt, H, W):
    B, C = 2, 16
    x = rnd(B, H, W, C, dt=dt)
    xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    up = torch.empty(B, 2 * H, 2 * W, C, device="cuda", dtype=dt)
    K.upsample2ac_fwd(x, up)
    ref = F.interpolate(xt, scale_factor=2, mode="bilinear", align_corners=True)
    close(up, ref.permute(0, 2, 3, 1), dt)
    dup = rnd(B, 2 * H, 2 * W, C, dt=dt, seed=3)
    ref.backward(dup.float().permute(0, 3, 1, 2))
    dx = torch.empty_like(x)
    K.upsample2ac_bwd(dup, dx)
    close(dx, xt.grad.permute(0, 2, 3, 1), dt, scale=2)


@pytest.mark.parametrize("dt", DT)
def test_gemm_tanh_epilogue_and_act_bwd(K, dt):
    M, N, Kd = 100, 24, 40
    x, w = rnd(M, Kd, dt=dt), rnd(N, Kd, dt=dt, seed=1) * 0.2
    b = rnd(N, seed=2)
    y = torch.empty(M, N, device="cuda", dtype=dt)
    K.gemm(K.dcode(dt), K.A_KC, K.B_KC, x, w, y, M, N, Kd, Kd, Kd, N, bias=b, act=K.ACT_TANH)
    ref = torch.tanh(x.float() @ w.float().t() + b)
    close(y, ref, dt)
    dy = rnd(M, N, dt=dt, seed=4)
    dz = torch.empty_like(dy)
    K.act_bwd(dy, y, dz, 2)
    close(dz, dy.float() * (1 - y.float() ** 2), dt)


# ---- fused attention -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,Lq,Lk,causal", [(2, 3, 197, 197, False), (1, 2, 300, 676, False), (2, 1, 64, 70, False), (1, 4, 33, 129, False),
                                               (4, 8, 20, 20, True), (2, 2, 77, 77, True), (1, 3, 160, 160, True)])
def test_flash_attention_matches_torch(K, B, H, Lq, Lk, causal):
    """crog_flash_attn_fwd/bwd (bf16, head_dim 64, no dropout; unmasked, or with the text tower's causal mask) against fp32 softmax attention
    on the same bf16 inputs; Q/K/V are column slices of one packed buffer, as the MHA projections produce them."""
    dh, E = 64, 64 * H
    dt = torch.bfloat16
    qkv = (rnd(B * max(Lq, Lk), 3 * E, dt=dt) * 0.7).contiguous()
    dO = rnd(B * Lq, E, dt=dt, seed=3)
    scale = dh ** -0.5
    q = qkv[:B * Lq].view(B, Lq, 3 * E)[:, :, :E] if Lq == Lk else None
    # separate buffers when Lq != Lk (cross-shaped), packed slices otherwise
    if Lq == Lk:
        qs, ks, vs = (qkv, 0, 3 * E), (qkv, E, 3 * E), (qkv, 2 * E, 3 * E)
        Qf = qkv[:, :E].float().view(B, Lq, H, dh).permute(0, 2, 1, 3)
        Kf = qkv[:, E:2 * E].float().view(B, Lk, H, dh).permute(0, 2, 1, 3)
        Vf = qkv[:, 2 * E:].float().view(B, Lk, H, dh).permute(0, 2, 1, 3)
    else:
        qbuf = rnd(B * Lq, E, dt=dt, seed=7) * 0.7
        kvbuf = rnd(B * Lk, 2 * E, dt=dt, seed=8) * 0.7
        qs, ks, vs = (qbuf, 0, E), (kvbuf, 0, 2 * E), (kvbuf, E, 2 * E)
        Qf = qbuf.float().view(B, Lq, H, dh).permute(0, 2, 1, 3)
        Kf = kvbuf[:, :E].float().view(B, Lk, H, dh).permute(0, 2, 1, 3)
        Vf = kvbuf[:, E:].float().view(B, Lk, H, dh).permute(0, 2, 1, 3)
    Qf, Kf, Vf = Qf.contiguous().requires_grad_(True), Kf.contiguous().requires_grad_(True), Vf.contiguous().requires_grad_(True)
    S = scale * Qf @ Kf.transpose(-1, -2)
    if causal:
        S = S + torch.full((Lq, Lk), float("-inf"), device="cuda").triu(1)
    ref = (torch.softmax(S, -1) @ Vf)
    ref.backward(dO.float().view(B, Lq, H, dh).permute(0, 2, 1, 3))
    O = torch.empty(B * Lq, E, device="cuda", dtype=dt)
    lse = torch.empty(B * H * Lq, device="cuda")
    Lkp = (Lk + 7) // 8 * 8
    K.flash_attn_fwd(qs, ks, vs, (O, 0, E), lse, B, H, Lq, Lk, dh, scale, 0.0, 0, Lkp, causal=causal)
    close(O.view(B, Lq, H, dh).permute(0, 2, 1, 3), ref, dt)
    close(lse.view(B, H, Lq), torch.logsumexp(S, -1), torch.float32, scale=50)
    D = torch.empty_like(lse)
    if Lq == Lk:
        dqkv = torch.zeros_like(qkv)
        dq, dk, dv = (dqkv, 0, 3 * E), (dqkv, E, 3 * E), (dqkv, 2 * E, 3 * E)
    else:
        dqb, dkvb = torch.zeros_like(qs[0]), torch.zeros_like(ks[0])
        dq, dk, dv = (dqb, 0, E), (dkvb, 0, 2 * E), (dkvb, E, 2 * E)
    K.flash_attn_bwd(qs, ks, vs, (O, 0, E), (dO, 0, E), lse, D, dq, dk, dv, B, H, Lq, Lk, dh, scale, 0.0, 0, Lkp, causal=causal)
    def sl(t3, L):
        t, c, ld = t3
        return t[:B * L, c:c + E].float().view(B, L, H, dh).permute(0, 2, 1, 3)
    close(sl(dq, Lq), Qf.grad, dt, scale=2)
    close(sl(dk, Lk), Kf.grad, dt, scale=2)
    close(sl(dv, Lk), Vf.grad, dt, scale=2)


@pytest.mark.parametrize("B,H,Lq,Lk,causal", [(2, 3, 197, 197, False), (1, 2, 300, 676, False), (1, 4, 33, 129, False), (2, 2, 77, 77, True),
                                               (1, 1, 676, 676, False), (2, 2, 300, 20, False)])
def test_flash_attention_keep_bits_reproduce_the_hash(K, B, H, Lq, Lk, causal):
    """crog_flash_attn_fwd_bits / bwd_bits: the forward's dropout decisions kept as a bit map.  (1) the map says what the hash says:
    decoded with the layout include/crog_hip.h documents it equals the elements crog_softmax_fwd (the unfused path, same seed and
    index) keeps; (2) outputs and all three gradients are the SAME BITS whether the backward kernels read the map or hash again."""
    dh, E, dt, p, seed = 64, 64 * H, torch.bfloat16, 0.1, 4242
    qbuf = rnd(B * Lq, E, dt=dt, seed=7) * 0.7
    kvbuf = rnd(B * Lk, 2 * E, dt=dt, seed=8) * 0.7
    dO = rnd(B * Lq, E, dt=dt, seed=3)
    qs, ks, vs = (qbuf, 0, E), (kvbuf, 0, 2 * E), (kvbuf, E, 2 * E)
    scale, Lkp = dh ** -0.5, (Lk + 7) // 8 * 8
    nkt = (Lk + 31) // 32
    res = []
    for use_bits in (False, True):
        keep = torch.full((K.flash_keep_words(B, H, Lq, Lk),), -1, device="cuda", dtype=torch.int32) if use_bits else None
        O = torch.empty(B * Lq, E, device="cuda", dtype=dt)
        lse = torch.empty(B * H * Lq, device="cuda")
        D = torch.empty_like(lse)
        K.flash_attn_fwd(qs, ks, vs, (O, 0, E), lse, B, H, Lq, Lk, dh, scale, p, seed, Lkp, causal=causal, keep=keep)
        dqb, dkvb = torch.zeros_like(qbuf), torch.zeros_like(kvbuf)
        K.flash_attn_bwd(qs, ks, vs, (O, 0, E), (dO, 0, E), lse, D, (dqb, 0, E), (dkvb, 0, 2 * E), (dkvb, E, 2 * E), B, H, Lq, Lk, dh, scale,
                         p, seed, Lkp, causal=causal, keep=keep)
        torch.cuda.synchronize()
        res.append((O, lse, dqb, dkvb, keep))
    for a, b, name in zip(res[0][:4], res[1][:4], ("O", "lse", "dQ", "dK | dV")):
        assert torch.equal(a, b), f"{name}: differs between hashed and bit-map backward"
    assert res[0][2].float().abs().max() > 0 and res[0][3].float().abs().max() > 0
    # decode the map: word ((b*H + h)*nkt + t)*Lq + q, bit 16*half + r <-> key 32*t + (r & 3) + 8*(r >> 2) + 4*half
    words = res[1][4].view(B * H, nkt, Lq).to(torch.int64) & 0xffffffff
    bit = torch.arange(32, device="cuda")
    half, r = bit >> 4, bit & 15
    key_of_bit = (r & 3) + 8 * (r >> 2) + 4 * half                                      # [32] within the tile
    kept = ((words[..., None] >> bit) & 1).bool()                                        # [BH, nkt, Lq, 32 bits]
    dec = torch.zeros(B * H, nkt, Lq, 32, dtype=torch.bool, device="cuda")
    dec[..., key_of_bit] = kept
    dec = dec.permute(0, 2, 1, 3).reshape(B * H, Lq, nkt * 32)[..., :Lk]               # [BH, Lq, Lk]
    S = torch.zeros(B * H, Lq, Lkp, device="cuda", dtype=dt)
    Pd = torch.empty_like(S)
    K.softmax_fwd(S, B * H * Lq, Lq, Lk, Lkp, H, False, None, Pd, p, seed)
    ref = Pd[..., :Lk] != 0
    assert torch.equal(dec, ref), f"{int((dec != ref).sum())} keep bits differ from the unfused path's decisions"
    rate = 1.0 - dec.float().mean().item()
    assert abs(rate - p) < 0.01, rate


@pytest.mark.parametrize("B,H,Lq,Lk", [(3, 2, 300, 20), (2, 3, 130, 70), (2, 1, 676, 17)])
def test_flash_attention_with_key_padding_mask_matches_torch(K, B, H, Lq, Lk):
    """crog_flash_attn_fwd_bits / bwd_bits with a key padding mask (the decoder's vision-to-text cross-attention: 20 word keys, crog.py:55
    pad_mask, layers.py:329-332) against fp32 softmax attention with the padded keys at -inf; padded keys get exactly zero dK / dV."""
    dh, E, dt = 64, 64 * H, torch.bfloat16
    qbuf = rnd(B * Lq, E, dt=dt, seed=7) * 0.7
    kvbuf = rnd(B * Lk, 2 * E, dt=dt, seed=8) * 0.7
    dO = rnd(B * Lq, E, dt=dt, seed=3)
    g = torch.Generator().manual_seed(5)
    nvalid = torch.randint(1, Lk + 1, (B,), generator=g)
    nvalid[0] = Lk
    kpm = (torch.arange(Lk)[None, :] >= nvalid[:, None]).cuda().contiguous()             # True = padding (trailing, as word == 0 is)
    kpm[-1, 1] = Lk > 2                                                                   # ... and one hole in the middle
    qs, ks, vs = (qbuf, 0, E), (kvbuf, 0, 2 * E), (kvbuf, E, 2 * E)
    Qf = qbuf.float().view(B, Lq, H, dh).permute(0, 2, 1, 3).contiguous().requires_grad_(True)
    Kf = kvbuf[:, :E].float().view(B, Lk, H, dh).permute(0, 2, 1, 3).contiguous().requires_grad_(True)
    Vf = kvbuf[:, E:].float().view(B, Lk, H, dh).permute(0, 2, 1, 3).contiguous().requires_grad_(True)
    scale, Lkp = dh ** -0.5, (Lk + 7) // 8 * 8
    S = (scale * Qf @ Kf.transpose(-1, -2)).masked_fill(kpm[:, None, None, :], float("-inf"))
    ref = torch.softmax(S, -1) @ Vf
    ref.backward(dO.float().view(B, Lq, H, dh).permute(0, 2, 1, 3))
    O = torch.empty(B * Lq, E, device="cuda", dtype=dt)
    lse = torch.empty(B * H * Lq, device="cuda")
    D = torch.empty_like(lse)
    K.flash_attn_fwd(qs, ks, vs, (O, 0, E), lse, B, H, Lq, Lk, dh, scale, 0.0, 0, Lkp, kpm=kpm)
    close(O.view(B, Lq, H, dh).permute(0, 2, 1, 3), ref, dt)
    close(lse.view(B, H, Lq), torch.logsumexp(S, -1), torch.float32, scale=50)
    dqb, dkvb = torch.zeros_like(qbuf), torch.full_like(kvbuf, 7.0)
    K.flash_attn_bwd(qs, ks, vs, (O, 0, E), (dO, 0, E), lse, D, (dqb, 0, E), (dkvb, 0, 2 * E), (dkvb, E, 2 * E), B, H, Lq, Lk, dh, scale,
                     0.0, 0, Lkp, kpm=kpm)
    sl = lambda t, c, L: t[:, c:c + E].float().view(B, L, H, dh).permute(0, 2, 1, 3)
    close(sl(dqb, 0, Lq), Qf.grad, dt, scale=2)
    close(sl(dkvb, 0, Lk), Kf.grad, dt, scale=2)
    close(sl(dkvb, E, Lk), Vf.grad, dt, scale=2)
    assert float(dkvb.view(B, Lk, 2 * E)[kpm].abs().max()) == 0.0


@pytest.mark.parametrize("cross", [False, True])
def test_flash_attention_equals_unfused_path_with_dropout(cross):
    """Same seed -> the fused kernels drop exactly the elements the unfused softmax kernel drops: the two MHA paths agree to bf16
    rounding in outputs and every gradient (decoder self-attention shape, p = 0.1; cross: the decoder's cross-attention - queries,
    keys and values from three tensors, 20 keys behind a key padding mask)."""
    import importlib
    from crog_amd import functional as Fn
    from crog_amd.model.blocks import MultiheadAttention, bind_all
    from crog_amd.runtime import RT, ParamStore
    torch.manual_seed(0)
    B, L, E, H = 2, 200, 256, 4
    Lk = 20 if cross else L
    mha = MultiheadAttention(E, H, dropout=0.1)
    store = ParamStore(mha, torch.device("cuda"))
    bind_all(mha, store)
    x0 = (torch.randn(B * L, E, device="cuda") * 0.5).to(torch.bfloat16)
    k0 = (torch.randn(B * Lk, E, device="cuda") * 0.5).to(torch.bfloat16)
    v0 = (torch.randn(B * Lk, E, device="cuda") * 0.5).to(torch.bfloat16)
    kpm = (torch.arange(Lk, device="cuda")[None, :] >= torch.tensor([Lk, 7], device="cuda")[:, None]).contiguous() if cross else None
    res = {}
    for flash in (True, False):
        Fn.FLASH_ATTN = flash
        RT.manual_seed(11)
        store.zero_grad()
        store.invalidate_shadow()
        x = x0.clone().requires_grad_(True)
        if cross:
            xk, xv = k0.clone().requires_grad_(True), v0.clone().requires_grad_(True)
            y = mha(x, xk, xv, B=B, kpm=kpm, training=True)
        else:
            y = mha(x, x, x, B=B, training=True)
        (y.float() * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
        RT.join_streams()
        torch.cuda.synchronize()
        res[flash] = (y.detach().float(), x.grad.float(), store.G.clone()) + ((xk.grad.float(), xv.grad.float()) if cross else ())
    Fn.FLASH_ATTN = True
    for a, b, name in zip(res[True], res[False], ("out", "dx", "param grads", "dxk", "dxv")):
        err = (a - b).abs().max().item()
        ref = b.abs().max().item()
        assert err <= 3e-2 * ref + 1e-3, f"{name}: {err} vs scale {ref}"
        cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
        assert cos > 0.999, (name, cos)


@pytest.mark.parametrize("dt", DT)
def test_fused_head_bias_kernels(K, dt):
    """Helpers of the folded vis.4 + dynamic head (csrc/head.hip) against einsum."""
    B, g, C, P = 3, 5, 32, 101
    b5 = rnd(g * C)
    wpad = (rnd(B, C, 16, dt=dt, seed=1) * 0.3)
    wpad[..., 9:] = 0
    cb = torch.empty(B, g, 16, device="cuda")
    K.head_cb_fwd(b5, 0, wpad, cb, B, g, C)
    close(cb, torch.einsum("hc,bct->bht", b5.view(g, C), wpad.float()), dt)
    dtb = rnd(B * P * g, 16, dt=dt, seed=2)
    dcb = torch.empty(B, g, 16, device="cuda")
    K.head_tap_sums(dtb, dcb, B, g, P)
    close(dcb, dtb.float().view(B, P, g, 16).sum(1), dt, scale=4)
    db5 = torch.zeros(g * C + 2, device="cuda")
    dwpad = torch.zeros(B, C, 16, device="cuda")
    K.head_cb_bwd(b5, 0, wpad, dcb, db5, 1, dwpad, B, g, C)
    close(db5[1:g * C + 1].view(g, C), torch.einsum("bht,bct->hc", dcb, wpad.float()), dt, scale=4)
    close(dwpad[..., :9], torch.einsum("hc,bht->bct", b5.view(g, C), dcb)[..., :9], dt, scale=4)
    assert float(db5[0]) == 0.0 and float(db5[-1]) == 0.0 and float(dwpad[..., 9:].abs().max()) == 0.0


@pytest.mark.parametrize("shape,size,mask", [((2, 5, 13, 13), (52, 52), 0b10011), ((3, 1, 26, 26), (104, 104), 1),
                                             ((1, 2, 7, 9), (30, 17), 0), ((2, 5, 104, 104), (416, 416), 0b10011)])
def test_eval_maps(K, shape, size, mask):
    """crog_eval_maps (sigmoid + bicubic align_corners=True, crog_engine.py:181-211) against the oracle and the ATen routine."""
    sys.path.insert(0, ROOT)
    from oracle import crog_oracle as O
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(5)) * 3
    chans = [c for c in range(shape[1]) if (mask >> c) & 1]
    y = K.eval_maps(x.cuda(), mask, *size).cpu()
    assert (y - O.eval_maps(x, chans, size)).abs().max().item() < 1e-5
    ref = x.clone()
    for c in chans:
        ref[:, c] = torch.sigmoid(ref[:, c])
    ref = F.interpolate(ref, size=size, mode="bicubic", align_corners=True)
    assert (y - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("dt", DT)
def test_conv3_dgrad_weights(K, dt):
    """crog_conv3_dgrad_weights: every [Cout][tap][Cin] block of the flat weight buffer -> [Cin][8 - tap][Cout], all entries in
    one launch: tile-aligned, ragged (96 x 40), sub-vector (3 input channels: element-wise path) and untouched gaps."""
    shapes = [(128, 64), (96, 40), (32, 3), (64, 256)]
    offs, off = [], 8
    for co, ci in shapes:
        offs.append(off)
        off += (co * 9 * ci + 7) // 8 * 8 + 16
    total = off
    src = rnd(total, dt=dt, seed=21)
    dst = torch.full((total,), -7.0, device="cuda", dtype=dt)
    table = torch.tensor([[o, co, ci] for o, (co, ci) in zip(offs, shapes)], dtype=torch.int64, device="cuda")
    K.conv3_dgrad_weights(src, dst, table, len(shapes))
    covered = torch.zeros(total, dtype=torch.bool, device="cuda")
    for o, (co, ci) in zip(offs, shapes):
        w = src[o:o + co * 9 * ci].view(co, 9, ci)
        want = w.flip(1).permute(2, 1, 0).contiguous().view(-1)
        assert torch.equal(dst[o:o + co * 9 * ci], want), (co, ci)
        covered[o:o + co * 9 * ci] = True
    assert torch.all(dst[~covered] == -7.0)


# ------------------------------------------------------------------------------------------------
# Production shapes (CROG-R50 at B = 32, 416 x 416: M = B*H*W rows): what bench.py actually launches.
# The checker is fp32/fp64 torch on the same (bf16-rounded) operands; errors are normalised by the accumulation length.
# ------------------------------------------------------------------------------------------------
def _rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,Cin,Cout", [(346112, 64, 64), (346112, 64, 256), (346112, 256, 64), (86528, 256, 512)])
def test_production_1x1_conv_with_statistics_and_gradients(K, dt, M, Cin, Cout):
    """Bottleneck 1x1 convolutions of layer1 / layer2 at B = 32 (clip.py:17-18,25-26): forward with BatchNorm statistics in the
    epilogue (atomic replica rows, stat_replicas > 1), data gradient, weight gradient with the split-K the host picks."""
    from crog_amd.functional import stat_replicas
    x = rnd(M, Cin, dt=dt)
    w = (rnd(Cout, Cin, dt=dt, seed=1) * (Cin ** -0.5)).to(dt)
    y = torch.empty(M, Cout, device="cuda", dtype=dt)
    R = stat_replicas(K.stat_tiles(M), Cout)
    assert R > 1
    stats = torch.zeros(R, Cout, 2, device="cuda")
    K.gemm(K.dcode(dt), K.A_KC, K.B_KC, x, w, y, M, Cout, Cin, Cin, Cin, Cout, col_stats=stats, stat_replicas=R)
    ref = x.float() @ w.float().t()
    close(y, ref, dt, scale=math.sqrt(Cin) / 4)
    s = stats.sum(0).double()
    assert _rel_l2(s[:, 1], (ref.double() ** 2).sum(0)) < (1e-5 if dt == torch.float32 else 2e-3)
    assert float((s[:, 0] - ref.double().sum(0)).abs().max()) < 2e-3 * math.sqrt(M)    # column sums are O(sqrt(M)); fp32 order noise only
    # plain slab statistics (the fp32 parity mode's form) agree with the atomic rows
    slab = torch.zeros(K.stat_tiles(M), Cout, 2, device="cuda")
    y2 = torch.empty_like(y)
    K.gemm(K.dcode(dt), K.A_KC, K.B_KC, x, w, y2, M, Cout, Cin, Cin, Cin, Cout, col_stats=slab)
    assert torch.equal(y, y2) and _rel_l2(slab.sum(0)[:, 1], s[:, 1]) < 1e-5
    dy = rnd(M, Cout, dt=dt, seed=5)
    dx = torch.empty(M, Cin, device="cuda", dtype=dt)
    K.gemm(K.dcode(dt), K.A_KC, K.B_NC, dy, w, dx, M, Cin, Cout, Cout, Cin, Cin)
    close(dx, dy.float() @ w.float(), dt, scale=math.sqrt(Cout) / 4)
    sk = K.pick_splitk(Cout, Cin, M, 32 if dt == torch.bfloat16 else 16)
    assert sk > 1
    dw = torch.zeros(Cout, Cin, device="cuda")
    K.gemm(K.dcode(dt), K.A_MC, K.B_NC, dy, x, dw, Cout, Cin, M, Cout, Cin, Cin, splitk=sk, out_mode=K.OUT_F32_ATOMIC)
    want = (dy.double().t() @ x.double()).float()
    r = _rel_l2(dw, want)
    print(f"1x1 wgrad {Cin}->{Cout} M={M} splitk={sk}: rel L2 {r:.2e}")
    assert r < (2e-6 if dt == torch.float32 else 1e-5)       # operands are exact in both dtypes; only the fp32 summation order differs


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,Kd", [(21632, 512, 512), (21632, 2048, 512), (21632, 512, 2048), (5408, 2048, 2048)])
def test_production_linear_weight_gradient_split_k(K, dt, M, N, Kd):
    """Decoder / attention-pool linears (layers.py:291-301, clip.py:119-139): dW[N, Kd] = dy[M, N]^T x[M, Kd], reduction 21632
    split across blocks, bias gradient folded into the same launch."""
    x, dy = rnd(M, Kd, dt=dt), rnd(M, N, dt=dt, seed=3)
    sk = K.pick_splitk(N, Kd, M, 32 if dt == torch.bfloat16 else 16)
    dw = torch.zeros(N, Kd, device="cuda")
    db = torch.zeros(N, device="cuda")
    K.gemm(K.dcode(dt), K.A_MC, K.B_NC, dy, x, dw, N, Kd, M, N, Kd, Kd, splitk=sk, out_mode=K.OUT_F32_ATOMIC, a_sum=db)
    want = (dy.double().t() @ x.double()).float()
    r = _rel_l2(dw, want)
    print(f"linear wgrad [{N} x {Kd}] over M={M} splitk={sk}: rel L2 {r:.2e}")
    assert r < 1e-5
    assert _rel_l2(db, dy.double().sum(0)) < 1e-5


@pytest.mark.parametrize("dt", DT)
def test_linear_data_gradient_on_the_transposed_weight_copy(K, dt):
    """crog_dgrad_weights (taps = 1: a plain transpose per table entry, next to a 3x3 entry in the same launch) and the forward-shaped
    data gradient that reads it: lin_dgrad on the copy == lin_dgrad reading W transposed out of LDS == dy @ W, for a whole parameter and
    for a row block of a packed one (nn.MultiheadAttention's in_proj_weight)."""
    from crog_amd import functional as Fn
    from crog_amd.runtime import ParamStore

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(96, 200, bias=False)                  # [200, 96]
            self.packed = torch.nn.Parameter(torch.randn(3 * 64, 64) * 0.1)     # three row blocks of 64
            self.conv = torch.nn.Conv2d(32, 16, 3, bias=False)
            self.ragged = torch.nn.Linear(64, 12, bias=False)                # 12 rows: no 16-byte rows in the copy -> stays on the NN form
    net = Net().cuda()
    store = ParamStore(net, torch.device("cuda"))
    assert id(net.lin.weight) in store.lin_t and id(net.packed) in store.lin_t and id(net.ragged.weight) not in store.lin_t
    T = store.weights_t(dt)
    W = store.weights(dt)
    o = store.off(net.lin.weight)
    assert torch.equal(T[o:o + 200 * 96].view(96, 200), W[o:o + 200 * 96].view(200, 96).t())
    oc = store.off(net.conv.weight)
    wc = W[oc:oc + 16 * 9 * 32].view(16, 9, 32)
    assert torch.equal(T[oc:oc + 16 * 9 * 32].view(32, 9, 16), wc.flip(1).permute(2, 1, 0))
    M = 1000
    cases = [(Fn.WRef(store, net.lin.weight), net.lin.weight), (Fn.WRef(store, net.packed, 64, 64), net.packed[64:128])]
    if dt == torch.float32:      # 12 gradient columns are whole 16-byte vectors only in fp32 (the operator layer pads bf16 rows)
        cases.append((Fn.WRef(store, net.ragged.weight), net.ragged.weight))
    for w, ref_w in cases:
        dy = rnd(M, w.rows, dt=dt, seed=4)
        want = dy.float() @ ref_w.detach().to(dt).float()
        outs = []
        for flag in (True, False):
            Fn.LIN_DGRAD_T = flag
            dx = torch.empty(M, w.cols, device="cuda", dtype=dt)
            Fn.lin_dgrad(dy, w, dx)
            outs.append(dx.float())
        Fn.LIN_DGRAD_T = True
        close(outs[0], want, dt, scale=math.sqrt(w.rows) / 4)
        assert (outs[0] - outs[1]).abs().max().item() <= (1e-5 if dt == torch.float32 else 2e-2) * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("M,N,Kd", [(21632, 2048, 512), (21632, 512, 2048), (86528, 1024, 1024), (21600, 2048, 512)])
def test_wide_tile_weight_gradient_256x256(K, M, N, Kd):
    """Weight gradients whose two output sides are multiples of 256 (1x1 / linear: >= 1 M outputs, reduction >= 8192) take the 8-wave 256 x 256 tile
    with the atomic-only epilogue (csrc/gemm.hip big_wgrad / ShapeDma8A): against float64, incl. a reduction that is no multiple of the
    k-tile, accumulation on top of existing gradients, and the bias gradient the operator layer sums separately for this tile."""
    dt = torch.bfloat16
    assert K.lib().crog_gemm_wgrad_tile(K.BF16, K.A_MC, K.B_NC, N, Kd, M) == 256
    x, dy = rnd(M, Kd, dt=dt), rnd(M, N, dt=dt, seed=3)
    sk = K.pick_splitk(N, Kd, M, 32)
    base = rnd(N, Kd, seed=9)
    dw = base.clone()
    K.gemm(K.BF16, K.A_MC, K.B_NC, dy, x, dw, N, Kd, M, N, Kd, Kd, splitk=sk, out_mode=K.OUT_F32_ATOMIC)
    want = (dy.double().t() @ x.double()).float()
    r = _rel_l2(dw - base, want)
    print(f"wide-tile wgrad [{N} x {Kd}] over M={M} splitk={sk}: rel L2 {r:.2e}")
    assert r < 1e-5
    # through the operator layer: weight + bias gradient of a Linear (the bias sum becomes its own launch for this tile)
    from crog_amd import functional as Fn
    from crog_amd.runtime import ParamStore
    lin = torch.nn.Linear(Kd, N).cuda()
    store = ParamStore(lin, torch.device("cuda"))
    w, b = Fn.WRef(store, lin.weight), Fn.WRef(store, lin.bias, cols=1)
    store.G.zero_()
    Fn.lin_wgrad(dy, x, w, bias=b)
    torch.cuda.synchronize()
    assert _rel_l2(w.grad().view(N, Kd), want) < 1e-5
    assert _rel_l2(b.grad(), dy.double().sum(0)) < 1e-5


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,HW,Cin,Cout", [(32, 104, 64, 64), (32, 52, 128, 128), (8, 104, 512, 256), (32, 26, 512, 512)])
def test_production_conv3x3_fwd_dgrad_wgrad(K, dt, B, HW, Cin, Cout):
    """3x3 convolutions at bench resolution (layer1 conv2, layer2 conv2, the projector's 512 -> 256 at 104 x 104 on a quarter of the
    batch, the neck's 512 -> 512 at 26 x 26): forward with statistics, data gradient on the transposed weight copy, weight gradient."""
    if dt == torch.float32 and Cin * Cout >= 512 * 256:
        B = max(1, B // 8)      # the fp32 parity kernels take minutes at the full batch: same maps, an eighth of the images
    H = W = HW
    M = B * H * W
    x = rnd(B, H, W, Cin, dt=dt)
    w = (rnd(Cout, 3, 3, Cin, dt=dt, seed=1) * (9 * Cin) ** -0.5).to(dt)
    y = torch.empty(B, H, W, Cout, device="cuda", dtype=dt)
    # with BatchNorm statistics in the epilogue, as the training forward launches it (the (8, 104, 512, 256) case takes the 8-wave
    # 256 x 256 tile: 338 tiles; the others stay on 128 x 128)
    from crog_amd.functional import stat_replicas
    R = stat_replicas(K.stat_tiles(M), Cout)
    stats = torch.zeros(R, Cout, 2, device="cuda")
    K.gemm(K.dcode(dt), K.A_IM2COL, K.B_KC, x, w, y, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, conv=(H, W, Cin), col_stats=stats, stat_replicas=R)
    xt = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wt = w.float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.conv2d(xt, wt, padding=1)
    close(y, ref.permute(0, 2, 3, 1), dt, scale=math.sqrt(9 * Cin) / 4)
    st = stats.sum(0).double()
    assert _rel_l2(st[:, 1], (ref.double() ** 2).sum((0, 2, 3))) < (1e-5 if dt == torch.float32 else 2e-3)
    assert float((st[:, 0] - ref.double().sum((0, 2, 3))).abs().max()) < 2e-3 * math.sqrt(M)
    dy = rnd(B, H, W, Cout, dt=dt, seed=5)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    # data gradient as the forward-shaped implicit GEMM on the [Cin][flipped tap][Cout] weight copy (crog_conv3_dgrad_weights)
    wT = torch.empty(Cin * 9 * Cout, device="cuda", dtype=dt)
    table = torch.tensor([0, Cout, Cin], dtype=torch.int64, device="cuda")
    K.conv3_dgrad_weights(w.reshape(-1), wT, table, 1)
    dx = torch.empty(B, H, W, Cin, device="cuda", dtype=dt)
    K.gemm(K.dcode(dt), K.A_IM2COL, K.B_KC, dy, wT, dx, M, Cin, 9 * Cout, Cout, 9 * Cout, Cin, conv=(H, W, Cout))
    close(dx, xt.grad.permute(0, 2, 3, 1), dt, scale=math.sqrt(9 * Cout) / 4)
    sk = K.pick_splitk(Cout, 9 * Cin, M, 32 if dt == torch.bfloat16 else 16, conv=True)
    dw = torch.zeros(Cout, 9 * Cin, device="cuda")
    K.gemm(K.dcode(dt), K.A_MC, K.B_NC_IM2COL, dy, x, dw, Cout, 9 * Cin, M, Cout, Cin, 9 * Cin, conv=(H, W, Cin), splitk=sk,
           out_mode=K.OUT_F32_ATOMIC)
    r = _rel_l2(dw.view(Cout, 3, 3, Cin), wt.grad.permute(0, 2, 3, 1))
    print(f"3x3 {Cin}->{Cout} @{HW} B={B} wgrad splitk={sk}: rel L2 {r:.2e}")
    assert r < (1e-4 if dt == torch.float32 else 1e-4)


def test_operands_beyond_32bit_buffer_extent_take_the_pointer_path(K):
    """gemm.hip addresses an operand of the LDS-DMA kernel through a 32-bit byte offset; an operand of >= 2 GiB must fall back to
    the 64-bit pointer kernel and still be right (last rows are the ones a wrapped offset would miss)."""
    M, Kd, N = (1 << 20) + 128, 1024, 64
    a = torch.zeros(M, Kd, device="cuda", dtype=torch.bfloat16)       # 2.0 GiB + 256 KiB
    a[-300:] = rnd(300, Kd, dt=torch.bfloat16)
    a[:200] = rnd(200, Kd, dt=torch.bfloat16, seed=2)
    b = rnd(N, Kd, dt=torch.bfloat16, seed=1)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    K.gemm(K.BF16, K.A_KC, K.B_KC, a, b, c, M, N, Kd, Kd, Kd, N)
    for sl in (slice(0, 200), slice(M - 300, M), slice(M // 2, M // 2 + 64)):
        close(c[sl], a[sl].float() @ b.float().t(), torch.bfloat16, scale=math.sqrt(Kd) / 4)


def test_timing_only_events_measure_a_launch(K):
    """crog_timer_* (fence-free HIP events): elapsed time between two timers around a known-length launch is positive, of the
    right order, and consistent with torch's own events around the same launch."""
    src = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
    dst = torch.empty_like(src)
    def copy():
        K.check(K.lib().crog_probe_copy(K.ptr(src), K.ptr(dst), src.numel(), 0, K.stream()), "probe_copy")
    copy(); torch.cuda.synchronize()
    t0, t1 = K.Timer(), K.Timer()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); t0.record(); copy(); t1.record(); e1.record()
    torch.cuda.synchronize()
    ms, ms_torch = t0.elapsed_time(t1), e0.elapsed_time(e1)
    assert 0.005 < ms < 5.0                      # 128 MB of traffic: ~30 us at 4.5 TB/s, never milliseconds
    assert ms <= ms_torch * 1.05 + 0.01          # bracketed by the torch pair
    for mode in range(6):                        # every copy shape of the probe moves the bytes (ragged tail included)
        dst.zero_()
        src.random_(0, 255)
        n = src.numel() - 16 * 37
        K.check(K.lib().crog_probe_copy(K.ptr(src), K.ptr(dst), n, mode, K.stream()), "probe_copy")
        assert torch.equal(dst[:n], src[:n]) and int(dst[n:].sum()) == 0, mode


@pytest.mark.parametrize("M,N,K_,layout", [(4096, 256, 64, "nc"), (1000, 72, 96, "kc"), (21632, 1024, 256, "kc"), (2500, 512, 128, "nc"),
                                           (4096, 256, 64, "kc"), (1280, 128, 512, "kc"),
                                           # round 5: the ping-pong tile's bwd_z epilogue (N % 256 == 0, K >= 512, >= 150 tiles): 128-row tiles, 256-row
                                           # tiles, a ragged last row tile
                                           (21632, 256, 1024, "kc"), (43264, 512, 512, "kc"), (21000, 256, 512, "kc")])
def test_dgrad_epilogue_gates_a_residual_layer_with_its_bit_mask(K, M, N, K_, layout):
    """crog_gemm bwd_z + bwd_mask + R: the data gradient of a block's first convolution, plus the gradient of that block's identity
    path, is the gradient of the PREVIOUS block's output relu(bn3(z) + identity): the epilogue adds R, gates with the forward's bit
    mask (crog_bn_apply), stores g and accumulates (sum g, sum g*z) - against torch on the same operands."""
    dt = torch.bfloat16
    from crog_amd.functional import stat_replicas
    a = rnd(M, K_, dt=dt)
    w = (rnd(K_, N, dt=dt, seed=1) * 0.1).to(dt)
    z = (rnd(M, N, dt=dt, seed=2) * 1.5 + 0.3).to(dt)
    res = rnd(M, N + 8, dt=dt, seed=3)
    passed = torch.rand(M, N, device="cuda") > 0.4
    mask = torch.zeros(M, N // 8, device="cuda", dtype=torch.uint8)
    for e in range(8):
        mask |= (passed[:, e::8].to(torch.uint8) << e)
    R = stat_replicas(K.stat_tiles(M), N)
    sums = torch.zeros(R, N, 2, device="cuda")
    dx = torch.full((M + 1, N), 7.0, device="cuda", dtype=dt)
    kw = dict(col_stats=sums, stat_replicas=R, bwd_z=z, bwd_mask=mask, R=res, ldr=N + 8)
    if layout == "nc":
        K.gemm(K.dcode(dt), K.A_KC, K.B_NC, a, w, dx, M, N, K_, K_, N, N, **kw)
    else:
        K.gemm(K.dcode(dt), K.A_KC, K.B_KC, a, w.t().contiguous(), dx, M, N, K_, K_, K_, N, **kw)
    assert (dx[M] == 7).all()
    v = a.float() @ w.float() + res[:, :N].float()
    g = torch.where(passed, v, torch.zeros_like(v))
    close(dx[:M], g, dt, scale=math.sqrt(K_) / 4 + 1)
    assert ((dx[:M] == 0) | passed).all()
    tot = sums.sum(0)
    assert torch.allclose(tot[:, 0], g.sum(0), rtol=2e-3, atol=2e-2 * g.abs().sum(0).max().item() / 100)
    assert torch.allclose(tot[:, 1], (g * z.float()).sum(0), rtol=2e-3, atol=2e-2 * (g * z.float()).abs().sum(0).max().item() / 100)


@pytest.mark.parametrize("M,N,K_,relu", [(21632, 256, 1024, True), (43264, 256, 512, False)])
def test_ping_pong_tile_does_the_first_batchnorm_backward_pass(K, monkeypatch, M, N, K_, relu):
    """The same epilogue on the ping-pong tile (gemm_pp_kernel<..., 2>, K-contiguous operands: the data gradients on the transposed weight
    copies), gate from bwd_ss or none: stored gradient and (sum g, sum g*z) against torch, and against the 128 x 128 tile (CROG_PP_BWDZ
    off = debug bit 11) on the same operands."""
    dt = torch.bfloat16
    from crog_amd.functional import stat_replicas
    a, b = rnd(M, K_, dt=dt), (rnd(N, K_, dt=dt, seed=1) * 0.1).to(dt)
    z = (rnd(M, N, dt=dt, seed=2) * 1.5 + 0.3).to(dt)
    ss = torch.stack([torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda") * 0.3], 1).contiguous()
    R = stat_replicas(K.stat_tiles(M), N)
    outs = []
    for flags in (0, 2048):
        monkeypatch.setattr(K, "DEBUG_FLAGS", flags)
        sums = torch.zeros(R, N, 2, device="cuda")
        dx = torch.full((M + 1, N), 7.0, device="cuda", dtype=dt)
        K.gemm(K.dcode(dt), K.A_KC, K.B_KC, a, b, dx, M, N, K_, K_, K_, N, col_stats=sums, stat_replicas=R, bwd_z=z, bwd_ss=ss if relu else None)
        outs.append((dx, sums.sum(0)))
    monkeypatch.setattr(K, "DEBUG_FLAGS", 0)
    v = a.float() @ b.float().t()
    gate = (z.float() * ss[:, 0] + ss[:, 1] > 0) if relu else torch.ones_like(v, dtype=torch.bool)
    g = torch.where(gate, v, torch.zeros_like(v))
    for dx, tot in outs:
        assert (dx[M] == 7).all()
        close(dx[:M], g, dt, scale=math.sqrt(K_) / 4)
        assert torch.allclose(tot[:, 0], g.sum(0), rtol=2e-3, atol=2e-2 * g.abs().sum(0).max().item() / 100)
        assert torch.allclose(tot[:, 1], (g * z.float()).sum(0), rtol=2e-3, atol=2e-2 * (g * z.float()).abs().sum(0).max().item() / 100)
    assert (outs[0][0][:M] != outs[1][0][:M]).float().mean().item() < 0.05      # (same products, possibly another order inside a k-tile: a few last-bit differences)


@pytest.mark.parametrize("M,N,K_,relu", [(4096, 64, 64, True), (1000, 72, 96, True), (21632, 256, 2304 // 9, False), (2500, 512, 128, True)])
def test_dgrad_epilogue_does_the_first_batchnorm_backward_pass(K, M, N, K_, relu):
    """crog_gemm bwd_z: a data-gradient GEMM whose output is a BatchNorm(+ReLU) layer's dy gates it with the ReLU mask recomputed
    from z, stores the gated gradient and accumulates (sum g, sum g*z) into replica rows; crog_bn_bwd_apply with a negative
    sum_rows turns the raw z-moments into the totals the two-launch path computes.  Checked against torch on the same operands."""
    dt = torch.bfloat16
    from crog_amd.functional import stat_replicas
    a, b = rnd(M, K_, dt=dt), (rnd(K_, N, dt=dt, seed=1) * 0.1).to(dt)            # dy_next [M, K_] @ W [K_, N] (B_NC layout)
    z = (rnd(M, N, dt=dt, seed=2) * 1.5 + 0.3).to(dt)
    ss = torch.stack([torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda") * 0.3], 1).contiguous()
    R = stat_replicas(K.stat_tiles(M), N)
    sums = torch.zeros(R, N, 2, device="cuda")
    dx = torch.full((M + 1, N), 7.0, device="cuda", dtype=dt)
    K.gemm(K.dcode(dt), K.A_KC, K.B_NC, a, b, dx, M, N, K_, K_, N, N, col_stats=sums, stat_replicas=R, bwd_z=z, bwd_ss=ss if relu else None)
    assert (dx[M] == 7).all()
    v = a.float() @ b.float()
    gate = (z.float() * ss[:, 0] + ss[:, 1] > 0) if relu else torch.ones_like(v, dtype=torch.bool)
    g = torch.where(gate, v, torch.zeros_like(v))
    close(dx[:M], g, dt, scale=math.sqrt(K_) / 4)
    tot = sums.sum(0)
    assert torch.allclose(tot[:, 0], g.sum(0), rtol=2e-3, atol=2e-2 * g.abs().sum(0).max().item() / 100)
    assert torch.allclose(tot[:, 1], (g * z.float()).sum(0), rtol=2e-3, atol=2e-2 * (g * z.float()).abs().sum(0).max().item() / 100)
    # second pass from the raw moments == second pass from (sum g, sum g*zhat) of the stored gradient
    mean = z.float().mean(0); var = z.float().var(0, unbiased=False); invstd = (var + 1e-5).rsqrt()
    mi = torch.stack([mean, invstd], 1).contiguous()
    gamma = torch.rand(N, device="cuda") + 0.5
    dg = [torch.zeros(N, device="cuda") for _ in range(4)]
    dz_raw, dz_ref = torch.empty(M, N, device="cuda", dtype=dt), torch.empty(M, N, device="cuda", dtype=dt)
    K.bn_bwd_apply(dx[:M], None, z, mi, gamma, sums, float(M), dz_raw, None, ss if relu else None, sum_rows=-R, dgamma=dg[0], dbeta=dg[1])
    gs = dx[:M].float()                       # the gradient as stored (bf16-rounded): what a separate first pass would have read
    zh = (z.float() - mean) * invstd
    ref_sums = torch.stack([gs.sum(0), (gs * zh).sum(0)], 1).contiguous().view(1, N, 2)
    K.bn_bwd_apply(dx[:M], None, z, mi, gamma, ref_sums, float(M), dz_ref, None, ss if relu else None, sum_rows=1, dgamma=dg[2], dbeta=dg[3])
    scale = max(dz_ref.float().abs().max().item(), 1e-3)
    assert (dz_raw.float() - dz_ref.float()).abs().max().item() <= 2e-2 * scale
    assert torch.allclose(dg[1], dg[3], rtol=5e-3, atol=5e-3 * dg[3].abs().max().item())
    assert torch.allclose(dg[0], dg[2], rtol=5e-3, atol=5e-3 * dg[2].abs().max().item() + 1e-3)


def test_register_only_kernels_are_stable_beside_another_streams_gemm(K):
    """Round 5's root cause of "a kernel's result depends on what runs beside it" (LAB_NOTES section 10, scripts/pk_probe.py): built with the
    packed-fp32 VALU instructions (v_pk_fma_f32 ...: the SLP vectoriser's form of adjacent scalar fp32 operations), the bilinear x2 backward -
    loads, register arithmetic, stores; no LDS, no cross-lane operation - returned wrong values in lanes 48-63 in 2997 of 3000 launches while a
    3x3 weight-gradient GEMM ran on another stream.  The library is built without those instructions (crog_amd/_lib.py NO_PACKED_F32): the
    same launch beside the same GEMM must reproduce its serial result every time; a LayerNorm backward (the kernel rounds 3-4 saw it in) too."""
    B, H, W, C = 8, 13, 13, 512
    torch.manual_seed(0)
    cat = (torch.randn(B, 2 * H, 2 * W, 3 * C, device="cuda") * 0.01).to(torch.bfloat16)
    dy, dx = cat[..., 2 * C:], torch.empty(B, H, W, C, device="cuda", dtype=torch.bfloat16)
    x = (torch.randn(B, 2 * H, 2 * W, C, device="cuda") * 0.5).to(torch.bfloat16)
    G = torch.zeros(C, 9 * C, device="cuda", dtype=torch.float32)
    rows = B * 2 * H * 2 * W
    sk = K.pick_splitk(C, 9 * C, rows, 64, conv=True)
    side = torch.cuda.Stream()
    # the LayerNorm backward of the decoder's token rows
    R_, Cn = 2704, 512
    lx = torch.randn(R_, Cn, device="cuda").to(torch.bfloat16)
    lw = torch.rand(Cn, device="cuda") + 0.5
    ldo = (torch.randn(R_, Cn, device="cuda") * 0.1).to(torch.bfloat16)
    ly, lstats = torch.empty_like(lx), torch.empty(R_, 2, device="cuda")
    K.ln_fwd(lx, lw, torch.zeros(Cn, device="cuda"), 1e-5, ly, lstats)
    rpb = K.ln_bwd_rows_per_block(R_)
    ldx, lpart = torch.empty_like(lx), torch.empty((R_ + rpb - 1) // rpb, Cn, 2, device="cuda")

    def victims():
        K.upsample2_bwd(dy, dx)
        K.ln_bwd(ldo, None, lx, lw, lstats, ldx, lpart, rpb)

    victims()
    torch.cuda.synchronize()
    ref, lref = dx.clone(), ldx.clone()
    bad = 0
    for it in range(300):
        dx.fill_(7.0)
        side.wait_stream(torch.cuda.current_stream())
        K.set_stream_override(side.cuda_stream)
        try:
            K.gemm(K.BF16, K.A_MC, K.B_NC_IM2COL, cat, x, G, C, 9 * C, rows, 3 * C, C, 9 * C, a_off=C, conv=(2 * H, 2 * W, C), splitk=sk, out_mode=K.OUT_F32_ATOMIC)
        finally:
            K.set_stream_override(None)
        victims()
        torch.cuda.synchronize()
        bad += int(not torch.equal(dx, ref)) + int(not torch.equal(ldx, lref))
    assert bad == 0, f"{bad} of 600 launches beside a GEMM differ from their serial result"
