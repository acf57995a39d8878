"""CPU oracle of the training-input preprocessing (TEST INFRASTRUCTURE — never imported by crog_amd): numpy restatement of
OCIDVLGDataset.get_transform_mat / preprocess (utils/dataset.py:824-914) including the arithmetic of the two cv2 calls it makes.

PARITY UNPINNED: the reference delegates the warp to OpenCV (`cv2.getAffineTransform`, `cv2.warpAffine`, pinned by the reference's
environment.yml as opencv-python 4.x), which is absent from this image, and the reference holds no fixtures for this path.  The
functions below restate OpenCV's published algorithm for 8-bit images (modules/imgproc/src/imgwarp.cpp: `warpAffine` ->
`WarpAffineInvoker` -> `remap` with the fixed-point interpolation tables of `initInterTab2D`):

  * the 2 x 3 matrix is inverted in double precision;
  * destination -> source coordinates are formed in 10-bit fixed point (`AB_BITS`), rounded (`cvRound`, half to even) and cut to a
    1/32-pixel grid (`INTER_BITS` = 5);
  * interpolation weights come from a 32 x 32 table of outer products of 1-D coefficients (bicubic: a = -0.75, float arithmetic;
    bilinear), converted to 15-bit fixed point and corrected so that every entry's weights sum to exactly 2^15;
  * taps outside the source take the border value (BORDER_CONSTANT); the weighted sum is rounded back with (s + 2^14) >> 15 and
    saturated to 8 bits.

The HIP kernel (csrc/preprocess.hip) is held bit-exact to this restatement on the integer stages and to 1e-6 on the float stages.
"""
from __future__ import annotations

import numpy as np

INTER_BITS, INTER_TAB_SIZE = 5, 32
AB_BITS, AB_SCALE = 10, 1 << 10
COEF_BITS, COEF_SCALE = 15, 1 << 15
CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073])          # utils/dataset.py:721-724
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711])


def get_transform_mat(img_size, input_size):
    """utils/dataset.py:824-840: letterbox affine (source -> destination) and its inverse, from three point pairs held in float32
    (cv2.getAffineTransform solves the 6 x 6 system in double)."""
    ori_h, ori_w = img_size
    inp_h, inp_w = input_size
    scale = min(inp_h / ori_h, inp_w / ori_w)
    new_h, new_w = ori_h * scale, ori_w * scale
    bias_x, bias_y = (inp_w - new_w) / 2., (inp_h - new_h) / 2.
    src = np.array([[0, 0], [ori_w, 0], [0, ori_h]], np.float32)
    dst = np.array([[bias_x, bias_y], [new_w + bias_x, bias_y], [bias_x, new_h + bias_y]], np.float32)

    def affine(a, b):
        A = np.zeros((6, 6))
        rhs = np.zeros(6)
        for i in range(3):
            A[i, 0:2], A[i, 2] = a[i], 1.0
            A[i + 3, 3:5], A[i + 3, 5] = a[i], 1.0
            rhs[i], rhs[i + 3] = b[i, 0], b[i, 1]
        return np.linalg.solve(A, rhs).reshape(2, 3)
    return affine(src.astype(np.float64), dst.astype(np.float64)), affine(dst.astype(np.float64), src.astype(np.float64))


def invert_affine(M):
    """warpAffine's in-place inversion of the forward matrix (imgwarp.cpp)."""
    M = np.array(M, dtype=np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0] = A11
    M[0, 1] *= -D
    M[1, 0] *= -D
    M[1, 1] = A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def _coeffs_1d(kind: str) -> np.ndarray:
    """initInterTab1D: [32][ksize] float32 coefficients at offsets i / 32."""
    x = (np.arange(INTER_TAB_SIZE, dtype=np.float32) * np.float32(1.0 / INTER_TAB_SIZE)).astype(np.float32)
    if kind == "linear":
        return np.stack([np.float32(1) - x, x], 1).astype(np.float32)
    A = np.float32(-0.75)          # interpolateCubic
    one, two, three, four, five, eight = (np.float32(v) for v in (1, 2, 3, 4, 5, 8))
    c0 = ((A * (x + one) - five * A) * (x + one) + eight * A) * (x + one) - four * A
    c1 = ((A + two) * x - (A + three)) * x * x + one
    c2 = ((A + two) * (one - x) - (A + three)) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], 1).astype(np.float32)


def inter_table(kind: str) -> np.ndarray:
    """initInterTab2D for 8-bit remap: [32 * 32][ksize * ksize] int16 weights, every row summing to exactly 2^15."""
    c = _coeffs_1d(kind)
    k = c.shape[1]
    tab = np.zeros((INTER_TAB_SIZE * INTER_TAB_SIZE, k * k), np.int16)
    for fy in range(INTER_TAB_SIZE):
        for fx in range(INTER_TAB_SIZE):
            w = (c[fy][:, None] * c[fx][None, :]).astype(np.float32)                       # vy * vx in float
            it = np.rint(w.astype(np.float64) * COEF_SCALE).astype(np.int64).clip(-32768, 32767).reshape(-1)     # saturate_cast<short>(v * SCALE)
            diff = int(it.sum()) - COEF_SCALE
            if diff != 0:
                k2 = k // 2
                Mk = mk = k2 * k + k2
                for k1 in (k2, k2 + 1):
                    for kk in (k2, k2 + 1):
                        i = k1 * k + kk
                        v = int(it[i]) if i < it.size else 0      # ksize 2: OpenCV's scan reads on into the next, still zero, entry
                        if v < it[mk]:
                            mk = i
                        elif v > it[Mk]:
                            Mk = i
                if diff < 0:
                    it[Mk] -= diff
                else:
                    it[mk] -= diff
            tab[fy * INTER_TAB_SIZE + fx] = it.astype(np.int16)
    return tab


def warp_coordinates(M_inv, out_w: int, out_h: int):
    """WarpAffineInvoker: integer source position and 1/32-pixel fraction of every destination pixel -> (sx, sy, alpha)."""
    M = np.asarray(M_inv, np.float64).reshape(-1)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    xs, ys = np.arange(out_w, dtype=np.float64), np.arange(out_h, dtype=np.float64)
    adelta = np.rint(M[0] * xs * AB_SCALE).astype(np.int64)
    bdelta = np.rint(M[3] * xs * AB_SCALE).astype(np.int64)
    X0 = np.rint((M[1] * ys + M[2]) * AB_SCALE).astype(np.int64) + round_delta
    Y0 = np.rint((M[4] * ys + M[5]) * AB_SCALE).astype(np.int64) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    alpha = (Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1))
    return sx, sy, alpha


def warp_affine_u8(src: np.ndarray, M_fwd, out_size, kind: str, border_value) -> np.ndarray:
    """cv2.warpAffine(src uint8 [H, W] or [H, W, C], M, (w, h), flags=INTER_LINEAR | INTER_CUBIC, borderValue=...) with the
    default BORDER_CONSTANT."""
    out_w, out_h = out_size
    img = src if src.ndim == 3 else src[:, :, None]
    H, W, C = img.shape
    cval = np.clip(np.rint(np.broadcast_to(np.asarray(border_value, np.float64), (C,))), 0, 255).astype(np.int64)    # saturate_cast<uchar>
    sx, sy, alpha = warp_coordinates(invert_affine(M_fwd), out_w, out_h)
    tab = inter_table(kind).astype(np.int64)
    k = 2 if kind == "linear" else 4
    off = 0 if kind == "linear" else 1
    acc = np.zeros((out_h, out_w, C), np.int64)
    for r in range(k):
        for c in range(k):
            yy, xx = sy - off + r, sx - off + c
            inside = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            v = np.where(inside[..., None], img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64), cval[None, None, :])
            acc += v * tab[alpha, r * k + c][..., None]
    out = np.clip((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255).astype(np.uint8)
    return out if src.ndim == 3 else out[:, :, 0]


def preprocess(img: np.ndarray, ins_mask: np.ndarray, qua: np.ndarray, ang: np.ndarray, wid: np.ndarray, input_size: int):
    """utils/dataset.py:843-914 for one sample: uint8 image [H, W, 3] and uint8 masks [H, W] -> float32 tensors as collate_fn
    stacks them: img [3, S, S] (CLIP-normalised), mask, qua, sin(2 theta), cos(2 theta), wid [S, S]; plus the inverse matrix."""
    size = (input_size, input_size)
    mat, mat_inv = get_transform_mat(img.shape[:2], size)
    if ins_mask.max() <= 1:
        ins_mask = (ins_mask * 255).astype(np.uint8)
    warped = warp_affine_u8(img, mat, size, "cubic", [0.48145466 * 255, 0.4578275 * 255, 0.40821073 * 255])
    x = warped.transpose(2, 0, 1).astype(np.float32)
    x = ((x / np.float32(255.)) - CLIP_MEAN.astype(np.float32).reshape(3, 1, 1)) / CLIP_STD.astype(np.float32).reshape(3, 1, 1)
    m = {k: warp_affine_u8(v, mat, size, "linear", 0.) for k, v in (("mask", ins_mask), ("qua", qua), ("ang", ang), ("wid", wid))}
    theta = m["ang"] * np.pi / 180.
    return dict(img=x.astype(np.float32), mask=(m["mask"] / 255.).astype(np.float32), qua=(m["qua"] / 255.).astype(np.float32),
                sin=np.sin(2 * theta).astype(np.float32), cos=np.cos(2 * theta).astype(np.float32), wid=(m["wid"] / 255.).astype(np.float32),
                inverse=mat_inv, warped_u8=warped, masks_u8=m)
