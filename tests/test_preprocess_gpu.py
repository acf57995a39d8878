"""Row N3 on the MI355X: crog_preprocess_u8 (one launch per batch) against the preprocessing oracle (oracle/preprocess_oracle.py,
restating utils/dataset.py:824-914 and cv2.warpAffine's 8-bit arithmetic).  Integer stages bit-exact (the warped uint8 image and
masks are recovered from the float outputs), float stages within 1e-6."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _sample(h, w, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img[: h // 3] = np.clip(np.add.outer(np.arange(h // 3), np.arange(w))[..., None] % 256, 0, 255).astype(np.uint8)      # a smooth region
    ins = (rng.random((h, w)) > 0.7).astype(np.uint8) * 255
    qua = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ang = rng.integers(0, 180, (h, w), dtype=np.uint8)
    wid = rng.integers(0, 256, (h, w), dtype=np.uint8)
    return img, ins, qua, ang, wid


@pytest.mark.parametrize("h,w,S,B", [(480, 640, 416, 3), (300, 500, 224, 2), (500, 300, 224, 1), (64, 64, 416, 2)])
def test_preprocess_kernel_matches_oracle(h, w, S, B):
    from crog_amd.data import Preprocessor
    from oracle import preprocess_oracle as O
    samples = [_sample(h, w, 10 * h + i) for i in range(B)]
    img = torch.from_numpy(np.stack([s[0] for s in samples])).cuda()
    masks = torch.from_numpy(np.stack([np.stack(s[1:]) for s in samples])).cuda()
    pre = Preprocessor(S)
    out = pre(img, masks)
    torch.cuda.synchronize()
    mean, std = torch.tensor(O.CLIP_MEAN, dtype=torch.float64).view(3, 1, 1), torch.tensor(O.CLIP_STD, dtype=torch.float64).view(3, 1, 1)
    for i, s in enumerate(samples):
        ref = O.preprocess(*s, S)
        got_u8 = torch.round((out["img"][i].double().cpu() * std + mean) * 255).to(torch.uint8).numpy().transpose(1, 2, 0)
        assert np.array_equal(got_u8, ref["warped_u8"]), ("image warp", int(np.abs(got_u8.astype(int) - ref["warped_u8"].astype(int)).max()))
        assert np.abs(out["img"][i].cpu().numpy() - ref["img"]).max() < 1e-6
        for k, key in (("mask", "mask"), ("qua", "qua"), ("wid", "wid")):
            g = (out["mask"] if k == "mask" else out["grasp_masks"][k])[i].cpu().numpy()
            assert np.array_equal(np.rint(g.astype(np.float64) * 255).astype(np.uint8), ref["masks_u8"][key]), k
            assert np.abs(g - ref[k]).max() < 1e-7, k
        assert np.abs(out["grasp_masks"]["sin"][i].cpu().numpy() - ref["sin"]).max() < 1e-6
        assert np.abs(out["grasp_masks"]["cos"][i].cpu().numpy() - ref["cos"]).max() < 1e-6
        assert np.allclose(out["inverse"], ref["inverse"])
    assert out["img"].shape == (B, 3, S, S) and out["mask"].shape == (B, S, S) and set(out["grasp_masks"]) == {"qua", "sin", "cos", "wid"}


def test_preprocessed_batch_feeds_the_training_step():
    """The dict the Preprocessor returns is the collate format `train_with_grasp` unpacks (crog_engine.py:49-66)."""
    from types import SimpleNamespace
    from crog_amd.data import Preprocessor
    from crog_amd.engine import train_with_grasp
    from crog_amd.model import build_crog
    from crog_amd.optim import FusedAdam
    from crog_amd.testing import tiny_cfg
    cfg = tiny_cfg()
    model, groups = build_crog(cfg)
    model = model.cuda().prepare()
    opt = FusedAdam(groups, lr=1e-5, store=model.store)
    pre = Preprocessor(cfg.input_size)
    s = [_sample(120, 160, 77 + i) for i in range(2)]
    batch = pre(torch.from_numpy(np.stack([x[0] for x in s])).cuda(), torch.from_numpy(np.stack([np.stack(x[1:]) for x in s])).cuda())
    word = torch.zeros(2, cfg.word_len, dtype=torch.long)
    word[:, 0], word[:, 1:4], word[:, 4] = cfg.clip_arch["vocab_size"] - 2, 7, cfg.clip_arch["vocab_size"] - 1
    data = dict(img=batch["img"], word_vec=word, mask=batch["mask"], grasp_masks=batch["grasp_masks"])
    lines = []
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [5])
    train_with_grasp([data], model, opt, sched, None, 1, SimpleNamespace(print_freq=1, epochs=1, max_norm=0.0), log=lines.append)
    torch.cuda.synchronize()
    assert len(lines) == 1 and "nan" not in lines[0].lower() and bool(torch.isfinite(model.store.P).all())
