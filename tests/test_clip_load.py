"""Pretrained-CLIP load path of CROG.__init__ (crog.py:20-23 -> clip.py:503-556) on the host: TorchScript archive ->
state_dict -> architecture inferred from tensor shapes -> fp16 round trip of exactly the tensor classes `convert_weights`
touches -> strict=False load that leaves `attnpool.connect.*` at its (fp16-rounded) random initialisation.

The expectations in tests/golden/clip_load.json were produced by the reference's own build_model(load_weights=True).float()
on the same synthetic checkpoint (oracle/make_golden.py::clip_load_fixture)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crog_amd.testing import clip_load_arch, make_cfg, save_clip_archive, synthetic_clip_checkpoint  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _fixture():
    return json.load(open(os.path.join(GOLD, "clip_load.json")))


def _archive(tmp_path, fx):
    sd = synthetic_clip_checkpoint(clip_load_arch(), seed=fx["seed"])
    path = str(tmp_path / "RN-small.pt")
    save_clip_archive(sd, path)
    return sd, path


def test_architecture_is_inferred_from_the_archive(tmp_path):
    from crog_amd.model.clip import arch_from_state_dict
    fx = _fixture()
    sd, path = _archive(tmp_path, fx)
    loaded = torch.jit.load(path, map_location="cpu").eval().state_dict()
    assert set(loaded) == set(sd) and all(torch.equal(loaded[k], sd[k]) for k in sd)
    arch = arch_from_state_dict(loaded)
    want = fx["arch"]
    for k in ("embed_dim", "image_resolution", "vision_width", "context_length", "vocab_size", "transformer_width",
              "transformer_heads", "transformer_layers"):
        assert arch[k] == want[k], (k, arch[k], want[k])
    assert list(arch["vision_layers"]) == want["vision_layers"] and arch["vision_patch_size"] is None


def test_pretrained_weights_get_the_reference_fp16_round_trip(tmp_path):
    from crog_amd.model import build_crog
    fx = _fixture()
    sd, path = _archive(tmp_path, fx)
    torch.manual_seed(5)
    model, groups = build_crog(make_cfg(clip_pretrain=path, use_pretrained_clip=True))
    got = model.backbone.state_dict()
    flags = fx["flags"]
    assert set(got) == set(flags)
    counts = {0: 0, 1: 0, 2: 0}
    for k, (flag, rep16) in flags.items():
        v = got[k]
        counts[flag] += 1
        if flag == 0:       # BatchNorm / LayerNorm tensors, embeddings, positional / class embeddings, logit_scale: full precision
            assert torch.equal(v, sd[k].float()), k
            if v.is_floating_point() and v.numel() > 1:
                assert not torch.equal(v, sd[k].half().float()), k      # i.e. NOT rounded (the advisor's finding of round 1)
        elif flag == 1:     # conv / linear / attention projections / text_projection: fp16 round trip of the checkpoint value
            assert torch.equal(v, sd[k].half().float()), k
        else:               # attnpool.connect.*: not in CLIP archives -> random init survives (strict=False)
            assert k not in sd and ".connect." in k, k
            if rep16:       # ... but convert_weights rounded the conv's random init through fp16 as well
                assert torch.equal(v, v.half().float()), k
    assert counts[0] > 50 and counts[1] > 50 and counts[2] == 6, counts
    assert model.backbone.visual.attnpool.connect["0"].weight.abs().sum() > 0
    assert len(groups) == 2 and all(p.dtype == torch.float32 for p in model.parameters())   # crog.py:23 `.float()`


def test_without_pretrained_flag_the_archive_only_defines_the_architecture(tmp_path):
    from crog_amd.model import build_crog
    fx = _fixture()
    sd, path = _archive(tmp_path, fx)
    model, _ = build_crog(make_cfg(clip_pretrain=path, use_pretrained_clip=False))
    got = model.backbone.state_dict()
    assert tuple(got["visual.layer2.1.conv1.weight"].shape) == tuple(sd["visual.layer2.1.conv1.weight"].shape)
    same = [k for k in sd if k in got and got[k].is_floating_point() and got[k].numel() > 16
            and (torch.equal(got[k], sd[k].float()) or torch.equal(got[k], sd[k].half().float()))]
    assert not same, same[:4]
    # clip.py:402-408: the last BatchNorm of every Bottleneck starts at zero when the weights are not loaded
    assert float(got["visual.layer1.0.bn3.weight"].abs().sum()) == 0.0


def test_decoder_return_intermediate_flag():
    """layers.py:259-274: return_intermediate=True yields one normalised map per layer (the last is the plain output); CROG.forward
    cannot consume that list in the reference either (crog.py:69 calls .reshape on it)."""
    from crog_amd.model.layers import TransformerDecoder
    dec = TransformerDecoder(num_layers=3, d_model=64, nhead=2, dim_ffn=128, dropout=0.0, return_intermediate=True)
    assert dec.return_intermediate and len(dec.layers) == 3


def test_unsupported_resnet_width_is_refused_by_name_before_any_forward():
    """clip.py:165-185 is width-generic; the HIP path implements width 64 (RN50 / RN101).  An RN50x4-shaped tower (width 80) is
    refused by `check_supported` - which CROG.prepare() calls when the model is bound to the GPU - with the width and the archive
    in the message, not by a kernel argument check deep inside the first forward."""
    import pytest
    from crog_amd.model.clip import ModifiedResNet
    tower = ModifiedResNet((1, 1, 1, 1), 640, 40, 288, width=80)
    with pytest.raises(NotImplementedError, match=r"width 80 \(RN50x4.pt\)"):
        tower.check_supported("RN50x4.pt")
    ModifiedResNet((1, 1, 1, 1), 1024, 32, 224, width=64).check_supported("RN50.pt")
