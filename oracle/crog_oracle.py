"""CPU oracle for the CROG training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-fp32 *restatement* of the reference algorithm, written functionally
over a flat {name: tensor} state dict (the reference's own parameter names).  It exists so that
the HIP path can be checked against it; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  The product (crog_amd/) never does.

Pinning: the reference ships no tests or golden vectors.  The oracle is
pinned against outputs of the reference itself, imported in the build container by
oracle/make_golden.py, which wrote the fixtures under tests/golden/ (tests/test_oracle_golden.py
replays them).  All file:line citations are into the reference tree (HilbertXu/CROG).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]

BN_EPS, BN_MOM, LN_EPS = 1e-5, 0.1, 1e-5


# ---------------------------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------------------------
def batchnorm(P: State, pre: str, x: Tensor, training: bool) -> Tensor:
    """nn.BatchNorm2d/1d with torch defaults (eps 1e-5, momentum 0.1); running stats updated in place
    when training (model.train() at engine/crog_engine.py:37 flips every BN to batch statistics)."""
    out = F.batch_norm(x, P[pre + ".running_mean"], P[pre + ".running_var"], P[pre + ".weight"], P[pre + ".bias"],
                       training, BN_MOM, BN_EPS)
    if training and (pre + ".num_batches_tracked") in P:
        P[pre + ".num_batches_tracked"] += 1
    return out


def layernorm(P: State, pre: str, x: Tensor) -> Tensor:
    """clip.py:226-231 (fp32 LayerNorm) and nn.LayerNorm in layers.py; eps 1e-5."""
    w = P[pre + ".weight"]
    xin = x if x.dtype == torch.float64 else x.float()  # float64 only in precision studies (scripts/debug_grads.py)
    return F.layer_norm(xin, (w.shape[0],), w, P[pre + ".bias"], LN_EPS)


def mha(q_in: Tensor, k_in: Tensor, v_in: Tensor, wq, wk, wv, bq, bk, bv, wo, bo, heads: int,
        attn_mask: Optional[Tensor] = None, key_padding_mask: Optional[Tensor] = None) -> Tensor:
    """Multi-head attention as F.multi_head_attention_forward computes it (clip.py:119-139, 246-260;
    layers.py:291-296,324,329-332), sequence-first inputs [L, B, E], dropout omitted (p = 0 in parity runs).
    scores = (q / sqrt(dh)) k^T + attn_mask, key-padding positions -> -inf, softmax, @ v, out-proj."""
    Lq, B, E = q_in.shape
    Lk = k_in.shape[0]
    dh = E // heads
    q = F.linear(q_in, wq, bq).reshape(Lq, B * heads, dh).transpose(0, 1)
    k = F.linear(k_in, wk, bk).reshape(Lk, B * heads, dh).transpose(0, 1)
    v = F.linear(v_in, wv, bv).reshape(Lk, B * heads, dh).transpose(0, 1)
    s = torch.bmm(q * (dh ** -0.5), k.transpose(1, 2))
    if attn_mask is not None:
        s = s + attn_mask
    if key_padding_mask is not None:
        s = s.view(B, heads, Lq, Lk).masked_fill(key_padding_mask[:, None, None, :], float("-inf")).view(B * heads, Lq, Lk)
    p = torch.softmax(s, dim=-1)
    o = torch.bmm(p, v).transpose(0, 1).reshape(Lq, B, E)
    return F.linear(o, wo, bo)


# ---------------------------------------------------------------------------------------------
# CLIP ModifiedResNet image encoder (clip.py:10-223)
# ---------------------------------------------------------------------------------------------
def bottleneck(P: State, pre: str, x: Tensor, stride: int, training: bool) -> Tensor:
    """clip.py:44-57: 1x1-BN-ReLU, 3x3-BN-ReLU, AvgPool(stride), 1x1-BN, (+AvgPool-1x1-BN downsample), add, ReLU."""
    out = F.relu(batchnorm(P, pre + ".bn1", F.conv2d(x, P[pre + ".conv1.weight"]), training))
    out = F.relu(batchnorm(P, pre + ".bn2", F.conv2d(out, P[pre + ".conv2.weight"], padding=1), training))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = batchnorm(P, pre + ".bn3", F.conv2d(out, P[pre + ".conv3.weight"]), training)
    identity = x
    if (pre + ".downsample.0.weight") in P:
        identity = F.avg_pool2d(x, stride) if stride > 1 else x
        identity = batchnorm(P, pre + ".downsample.1", F.conv2d(identity, P[pre + ".downsample.0.weight"]), training)
    return F.relu(out + identity)


def attnpool(P: State, pre: str, x: Tensor, heads: int, training: bool) -> Tensor:
    """clip.py:110-144: residual 1x1+BN branch, bicubic-resized positional table (CLS row dropped,
    clip.py:97-108), single MHA over the HW tokens with separate q/k/v weights, + residual, ReLU."""
    B, C, H, W = x.shape
    res = batchnorm(P, pre + ".connect.1", F.conv2d(x, P[pre + ".connect.0.weight"]), training)
    pe = P[pre + ".positional_embedding"]
    side = int(round(math.sqrt(pe.shape[0] - 1)))
    grid = pe[1:].reshape(1, side, side, C).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=(H, W), mode="bicubic", align_corners=False)
    tok = x.reshape(B, C, H * W) + grid.reshape(1, C, H * W)
    tok = tok.permute(2, 0, 1)  # (HW) B C
    o = mha(tok, tok, tok, P[pre + ".q_proj.weight"], P[pre + ".k_proj.weight"], P[pre + ".v_proj.weight"],
            P[pre + ".q_proj.bias"], P[pre + ".k_proj.bias"], P[pre + ".v_proj.bias"],
            P[pre + ".c_proj.weight"], P[pre + ".c_proj.bias"], heads)
    o = o.permute(1, 2, 0).reshape(B, -1, H, W)
    return F.relu(o + res)


def resnet_layers(P: State, pre: str = "backbone.visual") -> Tuple[int, ...]:
    return tuple(len({k.split(".")[len(pre.split(".")) + 1] for k in P if k.startswith(f"{pre}.layer{i}.")}) for i in (1, 2, 3, 4))


def encode_image(P: State, img: Tensor, training: bool, pre: str = "backbone.visual"):
    """clip.py:207-223: 3-conv stem + AvgPool2, layer1..4, attention pool; returns (layer2, layer3, attnpool(layer4))."""
    x = img
    for i, (s, p) in enumerate(((2, 1), (1, 1), (1, 1)), start=1):
        x = F.relu(batchnorm(P, f"{pre}.bn{i}", F.conv2d(x, P[f"{pre}.conv{i}.weight"], stride=s, padding=p), training))
    x = F.avg_pool2d(x, 2)
    feats = []
    for li, n in enumerate(resnet_layers(P, pre), start=1):
        for b in range(n):
            x = bottleneck(P, f"{pre}.layer{li}.{b}", x, 2 if (li > 1 and b == 0) else 1, training)
        feats.append(x)
    width = P[f"{pre}.layer1.0.conv1.weight"].shape[0]
    heads = width * 32 // 64  # clip.py:356
    x4 = attnpool(P, f"{pre}.attnpool", feats[3], heads, training)
    return feats[1], feats[2], x4


# ---------------------------------------------------------------------------------------------
# CLIP text transformer / ViT blocks (clip.py:239-283, 309-332, 439-456)
# ---------------------------------------------------------------------------------------------
def residual_attention_block(P: State, pre: str, x: Tensor, heads: int, attn_mask: Optional[Tensor]) -> Tensor:
    """clip.py:262-265: x + attn(ln_1(x)); x + c_proj(QuickGELU(c_fc(ln_2(x))))."""
    E = x.shape[-1]
    h = layernorm(P, pre + ".ln_1", x)
    w, b = P[pre + ".attn.in_proj_weight"], P[pre + ".attn.in_proj_bias"]
    x = x + mha(h, h, h, w[:E], w[E:2 * E], w[2 * E:], b[:E], b[E:2 * E], b[2 * E:],
                P[pre + ".attn.out_proj.weight"], P[pre + ".attn.out_proj.bias"], heads, attn_mask=attn_mask)
    h = layernorm(P, pre + ".ln_2", x)
    h = F.linear(h, P[pre + ".mlp.c_fc.weight"], P[pre + ".mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)  # QuickGELU clip.py:234-236
    return x + F.linear(h, P[pre + ".mlp.c_proj.weight"], P[pre + ".mlp.c_proj.bias"])


def n_blocks(P: State, pre: str) -> int:
    d = len(pre.split("."))
    return len({k.split(".")[d] for k in P if k.startswith(pre + ".")})


def encode_text(P: State, text: Tensor, pre: str = "backbone"):
    """clip.py:439-456: embedding + positions, causal transformer (mask clip.py:424-430), ln_final,
    returns (all-token features [B, L, D], EOT feature @ text_projection [B, embed])."""
    B, L = text.shape
    x = P[pre + ".token_embedding.weight"][text] + P[pre + ".positional_embedding"][:L]
    x = x.permute(1, 0, 2)
    width = x.shape[-1]
    heads = width // 64  # clip.py:538
    mask = torch.full((L, L), float("-inf")).triu_(1)
    for i in range(n_blocks(P, pre + ".transformer.resblocks")):
        x = residual_attention_block(P, f"{pre}.transformer.resblocks.{i}", x, heads, mask)
    x = layernorm(P, pre + ".ln_final", x.permute(1, 0, 2))
    state = x[torch.arange(B), text.argmax(dim=-1)] @ P[pre + ".text_projection"]
    return x, state


def encode_image_vit(P: State, img: Tensor, pre: str = "backbone.visual") -> Tensor:
    """clip.py:309-332 VisionTransformer: patch conv, CLS + positions, ln_pre, blocks, ln_post on PATCH tokens, @ proj."""
    w = P[pre + ".conv1.weight"]
    x = F.conv2d(img, w, stride=w.shape[-1])
    B, D = x.shape[0], x.shape[1]
    x = x.reshape(B, D, -1).permute(0, 2, 1)
    x = torch.cat([P[pre + ".class_embedding"].expand(B, 1, D), x], dim=1) + P[pre + ".positional_embedding"]
    x = layernorm(P, pre + ".ln_pre", x).permute(1, 0, 2)
    for i in range(n_blocks(P, pre + ".transformer.resblocks")):
        x = residual_attention_block(P, f"{pre}.transformer.resblocks.{i}", x, D // 64, None)
    x = layernorm(P, pre + ".ln_post", x.permute(1, 0, 2)[:, 1:, :])
    return x @ P[pre + ".proj"]


# ---------------------------------------------------------------------------------------------
# FPN neck (layers.py:342-398)
# ---------------------------------------------------------------------------------------------
def conv_bn_relu(P: State, pre: str, x: Tensor, training: bool) -> Tensor:
    """layers.py:8-11 conv_layer: bias-free conv (k from the weight), BN, ReLU."""
    w = P[pre + ".0.weight"]
    return F.relu(batchnorm(P, pre + ".1", F.conv2d(x, w, padding=w.shape[-1] // 2), training))


def fpn(P: State, vis: Sequence[Tensor], state: Tensor, training: bool, pre: str = "neck") -> Tensor:
    v3, v4, v5 = vis
    s = F.relu(batchnorm(P, pre + ".txt_proj.1", F.linear(state, P[pre + ".txt_proj.0.weight"]), training))  # layers.py:14-16,376
    f5 = conv_bn_relu(P, pre + ".f1_v_proj", v5, training)
    f5 = F.relu(batchnorm(P, pre + ".norm_layer.0", f5 * s[:, :, None, None], training))  # layers.py:379
    f4 = conv_bn_relu(P, pre + ".f2_v_proj", v4, training)
    f5u = F.interpolate(f5, scale_factor=2, mode="bilinear")
    f4 = conv_bn_relu(P, pre + ".f2_cat", torch.cat([f4, f5u], 1), training)
    f3 = F.avg_pool2d(conv_bn_relu(P, pre + ".f3_v_proj", v3, training), 2, 2)
    f3 = conv_bn_relu(P, pre + ".f3_cat", torch.cat([f3, f4], 1), training)
    fq5 = F.interpolate(conv_bn_relu(P, pre + ".f4_proj5", f5, training), scale_factor=2, mode="bilinear")
    fq4 = conv_bn_relu(P, pre + ".f4_proj4", f4, training)
    fq3 = conv_bn_relu(P, pre + ".f4_proj3", f3, training)
    fq = conv_bn_relu(P, pre + ".aggr", torch.cat([fq3, fq4, fq5], 1), training)
    b, _, h, w = fq.shape
    xs = torch.linspace(-1, 1, w).view(1, 1, 1, w).expand(b, 1, h, w)  # CoordConv layers.py:30-39: x first, then y
    ys = torch.linspace(-1, 1, h).view(1, 1, h, 1).expand(b, 1, h, w)
    fq = conv_bn_relu(P, pre + ".coordconv.0.conv1", torch.cat([fq, xs, ys], 1), training)
    return conv_bn_relu(P, pre + ".coordconv.1", fq, training)


# ---------------------------------------------------------------------------------------------
# Transformer decoder (layers.py:176-339)
# ---------------------------------------------------------------------------------------------
def pos1d(d_model: int, length: int) -> Tensor:
    """layers.py:195-212: interleaved sin/cos over positions -> [length, d_model]."""
    pos = torch.arange(length, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(length, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def pos2d(d_model: int, height: int, width: int) -> Tensor:
    """layers.py:214-241: first half of channels encodes x (sin/cos interleaved), second half y -> [H*W, d_model]."""
    half = d_model // 2
    div = torch.exp(torch.arange(0.0, half, 2) * -(math.log(10000.0) / half))
    px = torch.arange(0.0, width)[:, None] * div
    py = torch.arange(0.0, height)[:, None] * div
    pe = torch.zeros(d_model, height, width)
    pe[0:half:2] = torch.sin(px).t()[:, None, :].expand(-1, height, -1)
    pe[1:half:2] = torch.cos(px).t()[:, None, :].expand(-1, height, -1)
    pe[half::2] = torch.sin(py).t()[:, :, None].expand(-1, -1, width)
    pe[half + 1::2] = torch.cos(py).t()[:, :, None].expand(-1, -1, width)
    return pe.reshape(d_model, height * width).t().contiguous()


def decoder_layer(P: State, pre: str, vis: Tensor, txt: Tensor, vis_pos: Tensor, txt_pos: Tensor, pad_mask: Tensor, heads: int) -> Tensor:
    """layers.py:313-339 (dropout = 0): pre-LN self-attn with 2-D positions on q,k then post-LN + residual;
    pre-LN cross-attn to text (1-D positions on keys, key padding mask) then post-LN + residual;
    FFN = Linear, ReLU, LayerNorm(dim_ffn), Linear, + residual."""
    E = vis.shape[-1]
    v2 = layernorm(P, pre + ".norm1", vis)
    qk = v2 + vis_pos[:, None, :]
    w, b = P[pre + ".self_attn.in_proj_weight"], P[pre + ".self_attn.in_proj_bias"]
    v2 = mha(qk, qk, v2, w[:E], w[E:2 * E], w[2 * E:], b[:E], b[E:2 * E], b[2 * E:],
             P[pre + ".self_attn.out_proj.weight"], P[pre + ".self_attn.out_proj.bias"], heads)
    vis = vis + layernorm(P, pre + ".self_attn_norm", v2)
    v2 = layernorm(P, pre + ".norm2", vis)
    w, b = P[pre + ".multihead_attn.in_proj_weight"], P[pre + ".multihead_attn.in_proj_bias"]
    v2 = mha(v2 + vis_pos[:, None, :], txt + txt_pos[:, None, :], txt, w[:E], w[E:2 * E], w[2 * E:], b[:E], b[E:2 * E], b[2 * E:],
             P[pre + ".multihead_attn.out_proj.weight"], P[pre + ".multihead_attn.out_proj.bias"], heads, key_padding_mask=pad_mask)
    vis = vis + layernorm(P, pre + ".cross_attn_norm", v2)
    v2 = layernorm(P, pre + ".norm3", vis)
    v2 = F.relu(F.linear(v2, P[pre + ".ffn.0.weight"], P[pre + ".ffn.0.bias"]))
    v2 = layernorm(P, pre + ".ffn.3", v2)
    v2 = F.linear(v2, P[pre + ".ffn.4.weight"], P[pre + ".ffn.4.bias"])
    return vis + v2


def decoder(P: State, fq: Tensor, word: Tensor, pad_mask: Tensor, heads: int, pre: str = "decoder") -> Tensor:
    """layers.py:243-277: tokens (HW, B, C), layers, final LayerNorm, back to [B, C, HW]."""
    B, C, H, W = fq.shape
    L, D = word.shape[1], word.shape[2]
    vis_pos, txt_pos = pos2d(C, H, W), pos1d(D, L)
    vis = fq.reshape(B, C, H * W).permute(2, 0, 1)
    txt = word.permute(1, 0, 2)
    for i in range(n_blocks(P, pre + ".layers")):
        vis = decoder_layer(P, f"{pre}.layers.{i}", vis, txt, vis_pos, txt_pos, pad_mask, heads)
    return layernorm(P, pre + ".norm", vis).permute(1, 2, 0)


# ---------------------------------------------------------------------------------------------
# Projector heads (layers.py:47-173) and losses (crog.py:76-131)
# ---------------------------------------------------------------------------------------------
def projector(P: State, x: Tensor, state: Tensor, training: bool, pre: str = "proj"):
    """layers.py:64-132: (bilinear x2, conv3x3-BN-ReLU) twice, 1x1 conv with bias to n*C channels, then a
    per-sample 3x3 conv whose C*9 weights + 1 bias come from Linear(state); the same dynamic kernel is applied to
    each of the n = out_channels // C channel groups (n = 5 for MultiTaskProjector, 1 for Projector)."""
    x = F.interpolate(x, scale_factor=2, mode="bilinear")
    x = conv_bn_relu(P, pre + ".vis.1", x, training)
    x = F.interpolate(x, scale_factor=2, mode="bilinear")
    x = conv_bn_relu(P, pre + ".vis.3", x, training)
    x = F.conv2d(x, P[pre + ".vis.4.weight"], P[pre + ".vis.4.bias"])
    C = P[pre + ".vis.3.0.weight"].shape[0]
    n = x.shape[1] // C
    B, _, H, W = x.shape
    dyn = F.linear(state, P[pre + ".txt.weight"], P[pre + ".txt.bias"])
    weight, bias = dyn[:, :-1].reshape(B, C, 3, 3), dyn[:, -1]
    outs = []
    for g in range(n):
        xg = x[:, g * C:(g + 1) * C].reshape(1, B * C, H, W)
        outs.append(F.conv2d(xg, weight, bias, padding=1, groups=B).transpose(0, 1))
    return outs


def losses(preds: Sequence[Tensor], targets: Sequence[Tensor], weighted: bool):
    """crog.py:78-99 (use_grasp_masks) / :121-124: nearest-resize targets to the prediction size, BCE-with-logits
    with weight mask*0.5+1 on the mask head, smooth-L1 (beta 1, mean) on raw qua/sin/cos/wid maps, unweighted sum."""
    small = [F.interpolate(t, preds[0].shape[-2:], mode="nearest") if t.shape[-2:] != preds[0].shape[-2:] else t for t in targets]
    w = small[0] * 0.5 + 1 if weighted else None
    ls = [F.binary_cross_entropy_with_logits(preds[0], small[0], weight=w)]
    for p, t in zip(preds[1:], small[1:]):
        ls.append(F.smooth_l1_loss(p, t))
    return small, ls, sum(ls)


def train_metric(output: Tensor, target: Tensor, threshold: float = 0.35, pr_iou: float = 0.5):
    """utils/misc.py:115-131."""
    o = torch.sigmoid(output.flatten(1)) >= threshold
    t = target.flatten(1).bool()
    ious = (o & t).sum(1) / ((o | t).sum(1) + 1e-6)
    return 100.0 * ious.mean(), 100.0 * (ious > pr_iou).float().mean()


def eval_maps(logits: Tensor, sigmoid_channels: Sequence[int], size: Tuple[int, int]) -> Tensor:
    """Device part of validate_with_grasp, engine/crog_engine.py:181-211 (and :356-361): sigmoid on the listed channels, then
    F.interpolate(mode='bicubic', align_corners=True) to `size`.  Restated with explicit loops over the 4x4 taps (cubic
    convolution, A = -0.75, source index o*(n_in-1)/(n_out-1), taps clamped to the plane) so that it does not lean on the very
    ATen routine the reference calls; tests/test_oracle_golden.py pins it against that routine."""
    x = logits.double().clone()
    for c in sigmoid_channels:
        x[:, c] = torch.sigmoid(x[:, c])
    B, G, h, w = x.shape
    H, W = size
    A = -0.75

    def taps(n_in, n_out):
        scale = torch.tensor((n_in - 1) / (n_out - 1) if n_out > 1 else 0.0, dtype=torch.float32)   # ATen computes the scale in fp32
        src = (scale * torch.arange(n_out, dtype=torch.float32)).double()
        i0 = torch.floor(src)
        t = src - i0
        wts = torch.stack([((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A,
                           ((A + 2) * t - (A + 3)) * t * t + 1,
                           ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1,
                           ((A * (2 - t) - 5 * A) * (2 - t) + 8 * A) * (2 - t) - 4 * A], 1)          # [n_out, 4]
        idx = (i0.long()[:, None] + torch.arange(-1, 3)[None, :]).clamp(0, n_in - 1)             # [n_out, 4]
        return wts, idx

    wy, iy = taps(h, H)
    wx, ix = taps(w, W)
    out = torch.zeros(B, G, H, W, dtype=torch.float64)
    for j in range(4):
        rows = x[:, :, iy[:, j], :]                                   # [B, G, H, w]
        for i in range(4):
            out += wy[:, j][None, None, :, None] * wx[:, i][None, None, None, :] * rows[:, :, :, ix[:, i]]
    return out.float()


# ---------------------------------------------------------------------------------------------
# whole model (crog.py:47-133)
# ---------------------------------------------------------------------------------------------
def crog_forward(P: State, img: Tensor, word: Tensor, targets: Optional[Sequence[Tensor]] = None, *, num_head: int = 8,
                 training: bool = True, use_contrastive: bool = True, use_grasp_masks: bool = True):
    """Returns dict(preds, targets_small, losses, total, vis, word_feat, state, fq)."""
    pad_mask = word == 0  # crog.py:55
    vis = encode_image(P, img, training)
    wfeat, state = encode_text(P, word)
    fq = fpn(P, vis, state, training)
    b, c, h, w = fq.shape
    fq_dec = fq
    if use_contrastive:
        fq_dec = decoder(P, fq, wfeat, pad_mask, num_head).reshape(b, c, h, w)
    preds = projector(P, fq_dec, state, training)
    out = dict(preds=preds, vis=vis, word_feat=wfeat, state=state, fq=fq, fq_dec=fq_dec)
    if targets is not None:
        n = 5 if use_grasp_masks else 1
        small, ls, total = losses(preds[:n], list(targets)[:n], weighted=use_grasp_masks)
        out.update(targets_small=small, losses=ls, total=total)
    return out


def param_groups(names: Sequence[str]):
    """model/__init__.py:10-14: backbone.* minus *positional_embedding* -> group 0 (lr_multi * base_lr), rest -> group 1."""
    g0 = [n for n in names if n.startswith("backbone") and "positional_embedding" not in n]
    g1 = [n for n in names if not (n.startswith("backbone") and "positional_embedding" not in n)]
    return g0, g1
