"""GPU-side input pipeline of the training step: OCIDVLGDataset.preprocess + collate_fn
(utils/dataset.py:843-914,1041-1064) for a whole batch on the MI355X.

The reference warps every sample on the host with cv2 in two DataLoader workers (config yaml `workers: 2`); at the ~850 img/s one
MI355X trains CROG-R50 at, that loader is the bottleneck by an order of magnitude.  Here the loader ships the RAW uint8 arrays
(image + the four uint8 target masks of GraspTransforms.generate_masks, 1.5 MB per 640 x 480 sample instead of 5.5 MB of fp32) and
ONE kernel launch per batch (`crog_preprocess_u8`, csrc/preprocess.hip) produces the tensors `train_with_grasp` feeds the model:
normalised image, instance mask, quality / sin 2theta / cos 2theta / width maps.

Only the geometry lives on the host: the letterbox matrix (three point pairs in float32, solved in double, exactly as
`get_transform_mat` -> cv2.getAffineTransform) and OpenCV's fixed-point interpolation tables, built once.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch

from . import kernels as K

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # utils/dataset.py:721-724
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
INTER_TAB_SIZE, COEF_SCALE = 32, 1 << 15


def letterbox_matrices(img_size: Tuple[int, int], input_size: Tuple[int, int]):
    """get_transform_mat (utils/dataset.py:824-840): (source -> destination, destination -> source) 2 x 3 matrices in double."""
    ori_h, ori_w = img_size
    inp_h, inp_w = input_size
    scale = min(inp_h / ori_h, inp_w / ori_w)
    new_h, new_w = ori_h * scale, ori_w * scale
    bias_x, bias_y = (inp_w - new_w) / 2., (inp_h - new_h) / 2.
    src = np.array([[0, 0], [ori_w, 0], [0, ori_h]], np.float32).astype(np.float64)
    dst = np.array([[bias_x, bias_y], [new_w + bias_x, bias_y], [bias_x, new_h + bias_y]], np.float32).astype(np.float64)

    def solve(a, b):
        A = np.zeros((6, 6))
        A[:3, :2], A[:3, 2], A[3:, 3:5], A[3:, 5] = a, 1.0, a, 1.0
        return np.linalg.solve(A, np.concatenate([b[:, 0], b[:, 1]])).reshape(2, 3)
    return solve(src, dst), solve(dst, src)


def _invert(M: np.ndarray) -> np.ndarray:
    """cv2.warpAffine inverts the forward matrix itself, in double (imgwarp.cpp)."""
    M = M.astype(np.float64).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0], M[1, 1] = A11, A22
    M[0, 1] *= -D
    M[1, 0] *= -D
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def _weights_1d(cubic: bool) -> np.ndarray:
    x = np.arange(INTER_TAB_SIZE, dtype=np.float32) * np.float32(1.0 / INTER_TAB_SIZE)
    one = np.float32(1)
    if not cubic:
        return np.stack([one - x, x], 1)
    a = np.float32(-0.75)
    c0 = ((a * (x + one) - np.float32(5) * a) * (x + one) + np.float32(8) * a) * (x + one) - np.float32(4) * a
    c1 = ((a + np.float32(2)) * x - (a + np.float32(3))) * x * x + one
    c2 = ((a + np.float32(2)) * (one - x) - (a + np.float32(3))) * (one - x) * (one - x) + one
    return np.stack([c0, c1, c2, one - c0 - c1 - c2], 1).astype(np.float32)


def interpolation_table(cubic: bool) -> np.ndarray:
    """OpenCV's 8-bit remap weights: [32 * 32][k * k] int16, outer products of the 1-D coefficients in 15-bit fixed point with the
    rounding residue folded into the largest (or smallest) central weight so that every entry sums to 2^15."""
    c = _weights_1d(cubic)
    k = c.shape[1]
    w = (c[:, None, :, None] * c[None, :, None, :]).astype(np.float32).reshape(INTER_TAB_SIZE * INTER_TAB_SIZE, k * k)   # [fy, fx][r, c]
    it = np.clip(np.rint(w.astype(np.float64) * COEF_SCALE), -32768, 32767).astype(np.int64)
    for row in it:
        diff = int(row.sum()) - COEF_SCALE
        if diff == 0:
            continue
        k2 = k // 2
        Mk = mk = k2 * k + k2
        for k1 in (k2, k2 + 1):
            for kk in (k2, k2 + 1):
                i = k1 * k + kk
                # k == 2: the scan runs past the 2 x 2 entry into the (still zero) next one, as OpenCV's does; only the entry at the
                # exact pixel position (weight 2^15 saturated to 32767) needs the correction, and it lands on its last weight
                v = int(row[i]) if i < row.size else 0
                if v < row[mk]:
                    mk = i
                elif v > row[Mk]:
                    Mk = i
        if diff < 0:
            row[Mk] -= diff
        else:
            row[mk] -= diff
    return it.astype(np.int16)


class Preprocessor:
    """OCIDVLGDataset.preprocess + collate for batches of same-sized raw samples.

        pre = Preprocessor(input_size=416)
        batch = pre(img_u8, masks_u8)     # img_u8 [B, H, W, 3] uint8 (cuda), masks_u8 [B, 4, H, W] uint8: instance, quality, angle, width
        -> {"img": [B, 3, S, S] fp32, "mask": [B, S, S], "grasp_masks": {"qua", "sin", "cos", "wid": [B, S, S]}, "inverse": 2 x 3 ndarray}
    """

    def __init__(self, input_size: int = 416, device="cuda"):
        self.size = int(input_size)
        self.device = torch.device(device)
        self.tab_cubic = torch.from_numpy(interpolation_table(True)).to(self.device)
        self.tab_linear = torch.from_numpy(interpolation_table(False)).to(self.device)
        # borderValue=[mean * 255] saturate_cast to uchar (dataset.py:857-860)
        self.border = np.clip(np.rint(np.array(CLIP_MEAN) * 255.0), 0, 255).astype(np.int32)
        self.mean = np.array(CLIP_MEAN, np.float32)
        self.std = np.array(CLIP_STD, np.float32)
        self._mats: Dict[Tuple[int, int], Tuple[np.ndarray, np.ndarray]] = {}

    def matrices(self, img_size: Tuple[int, int]):
        if img_size not in self._mats:
            fwd, inv = letterbox_matrices(img_size, (self.size, self.size))
            self._mats[img_size] = (np.ascontiguousarray(_invert(fwd)), inv)
        return self._mats[img_size]

    def __call__(self, img_u8: torch.Tensor, masks_u8: torch.Tensor) -> dict:
        if img_u8.dtype != torch.uint8 or masks_u8.dtype != torch.uint8 or img_u8.dim() != 4 or img_u8.shape[-1] != 3:
            raise TypeError("Preprocessor expects uint8 tensors: img [B, H, W, 3], masks [B, 4, H, W]")
        B, H, W, _ = img_u8.shape
        if tuple(masks_u8.shape) != (B, 4, H, W):
            raise ValueError(f"masks must be [B, 4, H, W] = {(B, 4, H, W)}, got {tuple(masks_u8.shape)}")
        img_u8, masks_u8 = img_u8.contiguous(), masks_u8.contiguous()
        minv, inverse = self.matrices((H, W))
        S = self.size
        out_img = torch.empty(B, 3, S, S, device=img_u8.device, dtype=torch.float32)
        out_masks = torch.empty(B, 5, S, S, device=img_u8.device, dtype=torch.float32)
        K.check(K.lib().crog_preprocess_u8(K.ptr(img_u8), K.ptr(masks_u8), B, H, W, minv.ctypes.data, K.ptr(self.tab_cubic), K.ptr(self.tab_linear),
                                           S, self.mean.ctypes.data, self.std.ctypes.data, self.border.ctypes.data, K.ptr(out_img), K.ptr(out_masks),
                                           K.stream()), "preprocess_u8")
        return {"img": out_img, "mask": out_masks[:, 0], "grasp_masks": {"qua": out_masks[:, 1], "sin": out_masks[:, 2], "cos": out_masks[:, 3],
                                                                        "wid": out_masks[:, 4]}, "inverse": inverse}
