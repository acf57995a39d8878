"""Row N3, host side: the letterbox geometry and OpenCV fixed-point tables of crog_amd/data.py, and the preprocessing oracle
(oracle/preprocess_oracle.py).  cv2 is absent from this image and the reference holds no fixtures for this path, so the oracle's
parity with cv2.warpAffine is UNPINNED (its header says so); what can be checked here is that its geometry and interpolation are
right: against exact-coordinate float64 bicubic / bilinear interpolation of a smooth image the 8-bit result may only differ by the
1/32-pixel coordinate grid and the final rounding."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _smooth(h, w, c, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.zeros((h, w, c))
    for ch in range(c):
        for _ in range(4):
            fx, fy, ph = rng.uniform(0.005, 0.04), rng.uniform(0.005, 0.04), rng.uniform(0, 6.28)
            out[..., ch] += np.sin(fx * xx + fy * yy + ph)
    out = (out - out.min()) / (out.max() - out.min())
    return np.rint(out * 255).astype(np.uint8)


def _cubic_w(t):      # Keys kernel, a = -0.75, taps at -1, 0, 1, 2 (the fourth weight is 1 - the others)
    a = -0.75
    return np.stack([((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a, ((a + 2) * t - (a + 3)) * t * t + 1,
                     ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1], 0)


def _exact_warp(img, M_fwd, S, cubic, border):
    from oracle.preprocess_oracle import invert_affine
    Mi = invert_affine(M_fwd)
    H, W, C = img.shape
    ys, xs = np.mgrid[0:S, 0:S].astype(np.float64)
    sx, sy = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2], Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    tx, ty = sx - x0, sy - y0
    if cubic:
        wx3, wy3 = _cubic_w(tx), _cubic_w(ty)
        wx = list(wx3) + [1 - wx3.sum(0)]
        wy = list(wy3) + [1 - wy3.sum(0)]
        offs = (-1, 0, 1, 2)
    else:
        wx, wy, offs = [1 - tx, tx], [1 - ty, ty], (0, 1)
    out = np.zeros((S, S, C))
    for r, oy in enumerate(offs):
        for c, ox in enumerate(offs):
            yy, xx = y0 + oy, x0 + ox
            inside = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            v = np.where(inside[..., None], img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.float64), np.asarray(border, np.float64)[None, None, :])
            out += v * (wy[r] * wx[c])[..., None]
    return out


def test_letterbox_matrices_and_tables():
    from crog_amd import data as D
    from oracle import preprocess_oracle as O
    fwd, inv = D.letterbox_matrices((480, 640), (416, 416))
    assert np.allclose(fwd, [[0.65, 0, 0], [0, 0.65, 52.0]], atol=1e-12) and np.allclose(inv, [[1 / 0.65, 0, 0], [0, 1 / 0.65, -80.0]], atol=1e-9)
    fo, io = O.get_transform_mat((480, 640), (416, 416))
    assert np.array_equal(fwd, fo) and np.array_equal(inv, io) and np.array_equal(D._invert(fwd), O.invert_affine(fo))
    fwd2, _ = D.letterbox_matrices((640, 480), (224, 224))        # portrait: the other axis is padded
    assert np.allclose(fwd2, [[0.35, 0, 28.0], [0, 0.35, 0]], atol=1e-6)
    for cubic, kind, k in ((True, "cubic", 16), (False, "linear", 4)):
        t = D.interpolation_table(cubic)
        assert t.shape == (1024, k) and t.dtype == np.int16 and (t.astype(np.int64).sum(1) == 32768).all()
        assert np.array_equal(t, O.inter_table(kind))
    lin = D.interpolation_table(False)
    assert lin[16 * 32 + 16].tolist() == [8192, 8192, 8192, 8192] and lin[0].tolist() == [32767, 0, 0, 1]
    cub = D.interpolation_table(True).reshape(32, 32, 4, 4)
    assert cub[0, 16].sum(0).tolist() == [-3072, 19456, 19456, -3072]            # half-pixel bicubic taps for a = -0.75: (-3, 19, 19, -3) / 32


def test_oracle_warp_agrees_with_exact_coordinate_interpolation():
    from oracle import preprocess_oracle as O
    for (h, w), S in (((480, 640), 416), ((300, 500), 224), ((500, 300), 224)):
        img = _smooth(h, w, 3, seed=h + S)
        mat, _ = O.get_transform_mat((h, w), (S, S))
        border = [123, 117, 104]
        got = O.warp_affine_u8(img, mat, (S, S), "cubic", border).astype(np.float64)
        want = _exact_warp(img, mat, S, True, border)
        # interior of the letterboxed picture (the first / last source rows mix with the border colour through the 4-tap footprint)
        sc = min(S / h, S / w)
        y0, y1 = int((S - h * sc) / 2) + 3, int((S + h * sc) / 2) - 3
        x0, x1 = int((S - w * sc) / 2) + 3, int((S + w * sc) / 2) - 3
        d = np.abs(got - np.clip(want, 0, 255))[y0:y1, x0:x1]
        assert d.max() <= 2.0 and d.mean() < 0.5, (d.max(), d.mean())
        assert (got[0, 0] == border).all() or h * sc >= S - 1
        m = img[..., 0]
        gl = O.warp_affine_u8(m, mat, (S, S), "linear", 0.).astype(np.float64)
        wl = _exact_warp(m[..., None], mat, S, False, [0.])[..., 0]
        dl = np.abs(gl - wl)[y0:y1, x0:x1]
        assert dl.max() <= 2.0 and dl.mean() < 0.5, (dl.max(), dl.mean())


def test_oracle_preprocess_outputs():
    from oracle import preprocess_oracle as O
    img = _smooth(480, 640, 3, 1)
    ins = (_smooth(480, 640, 1, 2)[..., 0] > 128).astype(np.uint8)           # {0, 1} mask: preprocess scales it to {0, 255}
    qua, wid = _smooth(480, 640, 1, 3)[..., 0], _smooth(480, 640, 1, 4)[..., 0]
    ang = (_smooth(480, 640, 1, 5)[..., 0].astype(np.int32) * 179 // 255).astype(np.uint8)
    out = O.preprocess(img, ins, qua, ang, wid, 416)
    assert out["img"].shape == (3, 416, 416) and out["img"].dtype == np.float32 and out["mask"].shape == (416, 416)
    # letterbox bars carry the CLIP mean -> (123 / 255 - mean) / std etc.; targets are 0 there, cos(0) = 1
    assert np.allclose(out["img"][:, 0, 0], (np.array([123, 117, 104]) / 255.0 - O.CLIP_MEAN) / O.CLIP_STD, atol=1e-6)
    assert out["mask"][0, 0] == 0 and out["sin"][0, 0] == 0 and out["cos"][0, 0] == 1 and out["mask"].max() == 1.0
    assert np.allclose(out["sin"] ** 2 + out["cos"] ** 2, 1.0, atol=1e-6)
    assert np.allclose(out["inverse"], [[1 / 0.65, 0, 0], [0, 1 / 0.65, -80.0]], atol=1e-9)
