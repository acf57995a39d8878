"""Deterministic synthetic inputs and weights shared by tests, bench.py and the golden generator.

Nothing here is model code: it only produces tensors.  Weights are generated per parameter name
(seeded by the name), so the reference model, the CPU oracle and the HIP module can all be loaded
with bit-identical values without committing 588 MB of weights.
"""
from __future__ import annotations

import math
import zlib
from types import SimpleNamespace
from typing import Dict, Iterable, Tuple

import torch


def make_cfg(**over):
    """Hot-path keys of config/OCID-VLG/crog_multiple_r50.yaml, word_len = 20 per BASELINE.json."""
    cfg = dict(clip_pretrain="synthetic-RN50", word_len=20, word_dim=1024, vis_dim=512, fpn_in=[512, 1024, 1024],
               fpn_out=[256, 512, 1024], num_layers=3, num_head=8, dim_ffn=2048, dropout=0.1, intermediate=False,
               use_contrastive=True, use_pretrained_clip=False, use_grasp_masks=True, base_lr=1e-4, lr_multi=0.1,
               sync_bn=True, batch_size=32, weight_decay=0.0, milestones=[35], lr_decay=0.1, max_norm=0.0, print_freq=100,
               input_size=416,
               # architecture of the CLIP backbone when no checkpoint is given (RN50 values, clip.py:503-546)
               clip_arch=dict(embed_dim=1024, image_resolution=224, vision_layers=(3, 4, 6, 3), vision_width=64,
                              vision_patch_size=None, context_length=77, vocab_size=49408, transformer_width=512,
                              transformer_heads=8, transformer_layers=12))
    cfg.update(over)
    return SimpleNamespace(**cfg)


def tiny_cfg(**over):
    """Same channel widths as CROG-R50 (kernels see real K/N sizes) but one block per stage, a 2-layer text
    transformer and a small vocabulary, for fast CPU-oracle parity."""
    arch = dict(embed_dim=1024, image_resolution=224, vision_layers=(1, 1, 1, 1), vision_width=64, vision_patch_size=None,
                context_length=77, vocab_size=512, transformer_width=512, transformer_heads=8, transformer_layers=2)
    base = dict(clip_arch=arch, num_layers=1, dropout=0.0, word_len=12, input_size=96)
    base.update(over)
    return make_cfg(**base)


def _gen(name: str, seed: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def seeded_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    """Value recipe keyed on the parameter name; scales chosen so activations stay O(1) through the net."""
    g = _gen(name, seed)
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "running_var":
        return 1.0 + 0.2 * torch.rand(shape, generator=g)
    if name.endswith("logit_scale"):
        return torch.tensor(math.log(1 / 0.07))
    is_norm = any(t in name for t in (".bn", "ln_", ".norm", "_norm", "downsample.1", "connect.1", ".ffn.3")) or \
        (len(shape) == 1 and leaf == "weight")
    if len(shape) == 1:
        if leaf == "weight" and is_norm:
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.05 * torch.randn(shape, generator=g)
    if "positional_embedding" in name or leaf == "class_embedding":
        return 0.02 * torch.randn(shape, generator=g)
    if "token_embedding" in name:
        return 0.05 * torch.randn(shape, generator=g)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    if leaf in ("text_projection", "proj"):
        fan_in = shape[0]
    gain = math.sqrt(2.0) if len(shape) == 4 else 1.0
    if name.endswith("proj.txt.weight"):
        gain = 0.02  # the generated 3x3 kernel is summed over C*9 taps: keep logits O(1)
    return (gain / math.sqrt(fan_in)) * torch.randn(shape, generator=g)


def grad_probe(name: str, numel: int, seed: int = 0) -> torch.Tensor:
    """A fixed unit-normal vector per parameter name: <gradient, probe> is a direction-sensitive scalar of a gradient too large to commit
    (|<e, probe>| ~ ||e|| for an error vector e).  Both the golden generator and the GPU tests flatten the gradient in the reference's
    logical element order ([Cout, Cin, kh, kw] for convolutions)."""
    return torch.randn(numel, generator=_gen("probe::" + name, seed))


def seeded_cotangent(name: str, shape: Tuple[int, ...], seed: int = 0, scale: float = 1e-2, relu: bool = False) -> torch.Tensor:
    """Synthetic stage inputs / upstream gradients of the stage-isolated backward tests (oracle/make_golden.py `stages`), in the reference's
    NCHW element order; relu=True for stand-ins of post-ReLU feature maps."""
    t = torch.randn(shape, generator=_gen("cot::" + name, seed)) * scale
    return t.clamp_min(0) if relu else t


STAGE_OF = (("image", lambda n: n.startswith("backbone.visual.")),
            ("text", lambda n: n.startswith("backbone.") and not n.startswith("backbone.visual.") and n != "backbone.logit_scale"),
            ("neck", lambda n: n.startswith("neck.")), ("decoder", lambda n: n.startswith("decoder.")), ("proj", lambda n: n.startswith("proj.")))


def stage_of(name: str) -> str:
    for st, pred in STAGE_OF:
        if pred(name):
            return st
    return ""


def seeded_state(shapes: Dict[str, Iterable[int]], seed: int = 0, residual_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """`residual_gain` < 1 scales the last norm of every residual branch (bn3 of each Bottleneck), which is how
    trained / zero-init-residual networks look (clip.py:402-408 zero-inits bn3).  With gain 1 the random R50 trunk
    amplifies any perturbation ~5000x end to end (measured: fp32 rounding 1e-7 -> 5e-4 at the logits), which is fine
    for fp32 parity but makes bf16-vs-fp32 comparisons meaningless; bf16 checks use a damped trunk."""
    out = {k: seeded_tensor(k, tuple(v), seed) for k, v in shapes.items()}
    if residual_gain != 1.0:
        for k in out:
            if k.endswith("bn3.weight") and ".layer" in k:
                out[k] = out[k] * residual_gain
    return out


def synthetic_batch(B: int, size: int = 416, L: int = 20, vocab: int = 49408, seed: int = 1234, device="cpu"):
    """img ~ N(0,1); word = [SOT, t1..tk, EOT, 0...]; mask in {0,1}; qua, wid ~ U(0,1);
    sin/cos of 2*theta masked by `mask` (mirrors utils/dataset.py:892-897).  Returns dict of CPU/`device` tensors."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, size, size, generator=g)
    sot, eot = vocab - 2, vocab - 1
    word = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        k = int(torch.randint(3, max(4, L - 2), (1,), generator=g))
        word[b, 0] = sot
        word[b, 1:1 + k] = torch.randint(1, sot, (k,), generator=g)
        word[b, 1 + k] = eot
    # blobby mask: low-res Bernoulli upsampled
    low = (torch.rand(B, 1, size // 16, size // 16, generator=g) > 0.8).float()
    mask = torch.nn.functional.interpolate(low, size=(size, size), mode="nearest")
    theta = (torch.rand(B, 1, size, size, generator=g) - 0.5) * math.pi
    qua = torch.rand(B, 1, size, size, generator=g) * mask
    wid = torch.rand(B, 1, size, size, generator=g) * mask
    sin = torch.sin(2 * theta) * mask
    cos = torch.cos(2 * theta) * mask
    out = dict(img=img, word=word, mask=mask, qua=qua, sin=sin, cos=cos, wid=wid)
    return {k: v.to(device) for k, v in out.items()}


# ---- SSG-R50 (BASELINE config 5) ---------------------------------------------------------------------------------
def ssg_cfg(**over):
    """Model keys of config/OCID-Grasp/ssg_r50.yaml."""
    cfg = dict(backbone="resnet", resnet_layers=[3, 4, 6, 3], fpn_in_channels=[512, 1024, 2048], num_protos=32, num_classes=32,
               anchor_strides=[8, 16, 32, 64, 128], aspect_ratios=[1, 0.5, 2], img_size=544, with_depth=True, with_grasp_masks=True,
               path_to_pretrained_resnet=None, resume=None,
               # loss keys (TRAIN / MODEL sections of the yaml)
               pos_iou_thre=0.5, neg_iou_thre=0.4, alpha_conf=1, alpha_bbox=1.5, alpha_ins=6.126, alpha_sem=1, alpha_grasp=6.125,
               masks_to_train=100, intermidiate_output=True)
    cfg.update(over)
    return SimpleNamespace(**cfg)


def ssg_tiny_cfg(**over):
    """Real channel widths, one Bottleneck per stage, 96x96 input (pyramid 12, 6, 3, 2, 1)."""
    base = dict(resnet_layers=[1, 1, 1, 1], img_size=96)
    base.update(over)
    return ssg_cfg(**base)


def synthetic_ssg_batch(B: int, size: int, with_depth: bool, seed: int = 1234, device="cpu"):
    """rgb ~ U(0,1) (augmentation.py:150 scales to [0,1]), depth ~ U(0,1)."""
    g = torch.Generator().manual_seed(seed)
    out = dict(rgb=torch.rand(B, 3, size, size, generator=g))
    if with_depth:
        out["depth"] = torch.rand(B, 1, size, size, generator=g)
    return {k: v.to(device) for k, v in out.items()}


SSG_OUTPUTS = ("class_pred", "box_pred", "ins_coef_pred", "grasp_coef_pred", "protos", "seg_pred")


def ssg_surrogate_loss(raw: Dict[str, torch.Tensor], seed: int = 0) -> torch.Tensor:
    """Fixed pseudo-random linear functional of the six raw trunk outputs: drives a backward pass through every head
    without the data-dependent matching loss (ssg.py:297-530), so trunk gradients can be pinned against the reference."""
    total = 0.0
    for k in SSG_OUTPUTS:
        t = raw[k]
        r = torch.randn(t.shape, generator=_gen("ssg-loss::" + k, seed)).to(device=t.device, dtype=t.dtype)
        total = total + (t * r).mean()
    return total


def synthetic_ssg_targets(B: int, size: int, num_classes: int, seed: int = 1234, device="cpu"):
    """Ground truth in the collate format of utils/dataset.py:1396-1416: 2-4 objects per image with corner boxes (+ class),
    elliptical instance masks, per-instance grasp maps (quality / sin 2t / cos 2t / width, masked by the instance), a
    single-channel semantic mask and the label list."""
    g = torch.Generator().manual_seed(seed ^ 0x55AA)
    yy, xx = torch.meshgrid(torch.arange(size, dtype=torch.float32), torch.arange(size, dtype=torch.float32), indexing="ij")
    out = dict(bboxes=[], labels=[], ins_masks=[], sem_mask=[], grasp_masks={k: [] for k in ("qua", "sin", "cos", "wid")})
    for _ in range(B):
        n = int(torch.randint(2, 5, (1,), generator=g))
        cx, cy = 0.25 + 0.5 * torch.rand(n, generator=g), 0.25 + 0.5 * torch.rand(n, generator=g)
        w, h = 0.15 + 0.3 * torch.rand(n, generator=g), 0.15 + 0.3 * torch.rand(n, generator=g)
        box = torch.stack([(cx - w / 2).clamp(0.01), (cy - h / 2).clamp(0.01), (cx + w / 2).clamp(max=0.99), (cy + h / 2).clamp(max=0.99)], 1)
        cls = torch.randint(1, num_classes, (n,), generator=g)
        masks = torch.stack([(((xx / size - cx[j]) / (w[j] / 2)) ** 2 + ((yy / size - cy[j]) / (h[j] / 2)) ** 2 <= 1).float() for j in range(n)])
        theta = (torch.rand(n, size, size, generator=g) - 0.5) * math.pi
        out["bboxes"].append(torch.cat([box, cls[:, None].float()], 1))
        out["labels"].append(cls)
        out["ins_masks"].append(masks)
        out["sem_mask"].append(masks.amax(0))
        out["grasp_masks"]["qua"].append(torch.rand(n, size, size, generator=g) * masks)
        out["grasp_masks"]["sin"].append(torch.sin(2 * theta) * masks)
        out["grasp_masks"]["cos"].append(torch.cos(2 * theta) * masks)
        out["grasp_masks"]["wid"].append(torch.rand(n, size, size, generator=g) * masks)
    out["sem_mask"] = torch.stack(out["sem_mask"])
    mv = lambda t: t.to(device)
    return dict(bboxes=[mv(t) for t in out["bboxes"]], labels=[mv(t) for t in out["labels"]], ins_masks=[mv(t) for t in out["ins_masks"]],
                sem_mask=mv(out["sem_mask"]), grasp_masks={k: [mv(t) for t in v] for k, v in out["grasp_masks"].items()})


# ---- CLIP ViT tower (BASELINE config 4) and the pretrained-CLIP load path -----------------------------------------
def vit_seeded_state(shapes: Dict[str, Iterable[int]], seed: int = 0) -> Dict[str, torch.Tensor]:
    """Name-seeded weights of a VisionTransformer (keys as its own state_dict: `conv1.weight`, `class_embedding`, ...); the
    recipe sees them under the `visual.` prefix they carry inside CLIP."""
    st = seeded_state({"visual." + k: tuple(v) for k, v in shapes.items()}, seed=seed)
    return {k[len("visual."):]: v for k, v in st.items()}


def clip_load_arch():
    """A small ModifiedResNet CLIP whose architecture is fully recoverable from tensor shapes (clip.py:503-542)."""
    return dict(embed_dim=64, image_resolution=224, vision_layers=(1, 2, 1, 1), vision_width=16, vision_patch_size=None,
                context_length=77, vocab_size=300, transformer_width=128, transformer_heads=2, transformer_layers=2)


def synthetic_clip_checkpoint(arch: dict, seed: int = 0) -> Dict[str, torch.Tensor]:
    """State dict with the key names and shapes of an OpenAI CLIP archive for `arch` (no `visual.attnpool.connect.*`: that
    module is a CRIS/CROG addition, clip.py:76-78), plus the three metadata entries real archives carry.  Values are not
    fp16-representable on purpose, so a test can tell which tensors went through the fp16 round trip."""
    from .model.clip import build_model
    proto = build_model(arch, arch["context_length"])
    sd = {}
    for k, v in proto.state_dict().items():
        if ".connect." in k:
            continue
        t = seeded_tensor("ckpt::" + k, tuple(v.shape), seed)
        if t.is_floating_point():
            t = t * (1.0 + 2.0 ** -15) + 2.0 ** -20
        sd[k] = t
    sd["input_resolution"] = torch.tensor(arch["image_resolution"])
    sd["context_length"] = torch.tensor(arch["context_length"])
    sd["vocab_size"] = torch.tensor(arch["vocab_size"])
    return sd


def save_clip_archive(sd: Dict[str, torch.Tensor], path: str):
    """Write `sd` as a TorchScript archive (what cfg.clip_pretrain names, crog.py:20): a scripted module tree whose state_dict()
    is exactly `sd` (floating tensors as parameters, the rest as buffers)."""
    root = torch.nn.Module()
    for key, val in sd.items():
        mod = root
        *parts, leaf = key.split(".")
        for p in parts:
            if not hasattr(mod, p):
                mod.add_module(p, torch.nn.Module())
            mod = getattr(mod, p)
        if val.is_floating_point() and not leaf.startswith("running_"):
            mod.register_parameter(leaf, torch.nn.Parameter(val.clone(), requires_grad=False))
        else:
            mod.register_buffer(leaf, val.clone())
    torch.jit.save(torch.jit.script(root), path)


def synthetic_ssg_predictions(B: int, A: int, cfg, seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """Seeded random stand-ins for the six raw SSG predictions at the configuration's sizes (class logits, box offsets, tanh-range
    coefficients, ReLU-range prototypes, semantic logits): inputs for loss-only parity (the loss does not care where they came from)."""
    hp, hs, P, C = cfg.img_size // 4, cfg.img_size // 8, cfg.num_protos, cfg.num_classes
    g = lambda name: _gen("ssg-pred::" + name, seed)
    out = dict(class_pred=torch.randn(B, A, C, generator=g("cls")),
               box_pred=0.5 * torch.randn(B, A, 4, generator=g("box")),
               ins_coef_pred=torch.tanh(torch.randn(B, A, P, generator=g("ins"))),
               grasp_coef_pred=torch.tanh(torch.randn(B, A, 4, P, generator=g("grasp"))),
               protos=torch.relu(torch.randn(B, hp, hp, P, generator=g("protos"))) * 0.5,
               seg_pred=torch.randn(B, C, hs, hs, generator=g("seg")))
    return {k: v.to(device) for k, v in out.items()}


def synthetic_ssg_output(anchors: torch.Tensor, cfg, seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """Seeded eval-mode output_dict of SSG.forward for ONE image (ssg.py:283-293) with a few dozen confident, overlapping
    detections: class probabilities peaked on clusters of neighbouring anchors, small box offsets, tanh-range coefficients."""
    A = anchors.shape[0]
    P, C, hp = cfg.num_protos, cfg.num_classes, cfg.img_size // 4
    g = lambda name: _gen("ssg-out::" + name, seed)
    logits = torch.randn(A, C, generator=g("cls"))
    logits[:, 0] += 4.0                                              # background dominates ...
    centres = torch.randint(0, A, (40,), generator=g("centres"))
    for j, c in enumerate(centres.tolist()):                         # ... except around 40 anchor clusters
        lo, hi = max(0, c - 4), min(A, c + 5)
        logits[lo:hi, 1 + j % (C - 1)] += 7.0 + 0.1 * torch.arange(hi - lo)
    out = dict(anchors=anchors.flatten().tolist(), protos=torch.relu(torch.randn(1, hp, hp, P, generator=g("protos"))) * 0.5,
               cls_pred=torch.softmax(logits, -1)[None], box_pred=0.3 * torch.randn(1, A, 4, generator=g("box")),
               ins_coef_pred=torch.tanh(torch.randn(1, A, P, generator=g("ins"))),
               grasp_coef_pred=torch.tanh(torch.randn(1, A, 4, P, generator=g("grasp"))))
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in out.items()}
