"""CLIP towers of CROG on the HIP path: ModifiedResNet image encoder, ViT image encoder, text transformer.

Mirrors the module tree (and therefore the state_dict keys) of the reference's model/clip.py; the
arithmetic is crog_amd.functional (HIP kernels).  Activations are channels-last [B, H, W, C] /
token rows [B*L, C] in the compute dtype; `encode_image` returns the reference's NCHW tensors
only at the public boundary (`CLIP.encode_image`), internal callers use the channels-last forms.
"""
from __future__ import annotations

import math
from typing import Tuple, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as TF

from .. import functional as Fn
from .. import kernels as K
from ..functional import WRef
from .blocks import BatchNorm, Bound, Conv2d, ConvBN, LayerNorm, Linear, MultiheadAttention


class Bottleneck(Bound):
    """clip.py:10-57.  Children: conv1,bn1,conv2,bn2,conv3,bn3 and downsample.{0,1} (the reference's
    parameter-free "-1" AvgPool entry has no state and is implicit here)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1, self.bn1 = Conv2d(inplanes, planes, 1), BatchNorm(planes)
        self.conv2, self.bn2 = Conv2d(planes, planes, 3), BatchNorm(planes)
        self.conv3, self.bn3 = Conv2d(planes, planes * 4, 1), BatchNorm(planes * 4)
        self.stride = stride
        self.downsample = None
        if stride > 1 or inplanes != planes * 4:
            self.downsample = nn.ModuleDict({"0": Conv2d(inplanes, planes * 4, 1), "1": BatchNorm(planes * 4)})

    def forward(self, x, res_in=None, res_out=None, fan_slot=None):
        """fan_slot (strided blocks): a GradSlot in which a consumer of x OUTSIDE the tower (the neck) leaves its gradient; the
        downsample branch's pool backward adds it.
        res_in / res_out (Stage.forward): BnLinks to the previous / next block of the stage - this block's first data
        gradient does the previous block's bn3 first backward pass (Fn.BN_RES_FUSED).  The consumer's conv1 data gradient adds the
        gradient of the block's other branch (identity, or avgpool / downsample through the slot), i.e. it writes the COMPLETE gradient
        of the block's input; ConvBnAct.backward only arms the link when that second gradient is really there."""
        tr = self.training
        # x has two consumers (conv1 and the identity / downsample branch): the second branch's gradient rides a slot into conv1's
        # dgrad epilogue (no separate accumulation pass).  Valid because that branch is created after conv1 / conv2, so autograd runs
        # its backward first; without a downsample the producer is conv3's residual gradient, with one the branch's last backward op
        slot = Fn.GradSlot() if (tr and x.requires_grad) else None
        ds = self.downsample is not None
        # bn1's output feeds conv2 and nothing else, bn2's feeds conv3 directly when no AvgPool sits between them: the data gradient
        # of the consuming convolution does the first pass of that BatchNorm's backward in its epilogue (Fn.BnLink)
        l1 = Fn.BnLink() if tr else None
        l2 = Fn.BnLink() if (tr and self.stride == 1) else None
        out = Fn.conv_bn_act(x, self.conv1.w, self.bn1.buffers_ref(), ksize=1, relu=True, training=tr, grad_slot=slot, stat_out=l1,
                             res_in=res_in)
        # (a strided block's AvgPool2d(stride) after bn2 + relu, clip.py:49-50, runs inside the BatchNorm passes: Fn.POOL_FUSED)
        out = Fn.conv_bn_act(out, self.conv2.w, self.bn2.buffers_ref(), ksize=3, relu=True, training=tr, stat_in=l1, stat_out=l2,
                             pool=self.stride > 1)
        identity = x
        if self.downsample is not None:
            if self.stride > 1:
                identity = Fn.avgpool2(x, grad_slot=slot, add_slot=fan_slot if slot is not None else None)
            identity = Fn.conv_bn_act(identity, self.downsample["0"].w, self.downsample["1"].buffers_ref(), ksize=1, relu=False, training=tr,
                                      dx_slot=slot if self.stride == 1 else None)
        return Fn.conv_bn_act(out, self.conv3.w, self.bn3.buffers_ref(), ksize=1, relu=True, res=identity, training=tr,
                              res_slot=None if ds else slot, stat_in=l2, res_out=res_out)


class Stage(nn.Sequential):
    """A stage of Bottlenecks (clip.py:187-195: nn.Sequential, same child names) with a BnLink between consecutive blocks: block b's
    output feeds block b + 1 and nothing else."""

    def forward(self, x, res_in=None, res_out=None, fan_slot=None):
        """fan_slot: see Bottleneck.forward (first block).  res_in / res_out: links to the previous / next STAGE, when this stage's input / output has no other consumer (layer1 ->
        layer2; the outputs of layer2 and layer3 also feed the neck, layer4's the attention pool)."""
        blocks = list(self)
        link = res_in
        for i, blk in enumerate(blocks):
            last = i + 1 == len(blocks)
            nxt = res_out if last else (Fn.BnLink() if self.training else None)
            x = blk(x, res_in=link, res_out=nxt, **({"fan_slot": fan_slot} if i == 0 and fan_slot is not None else {}))
            link = nxt
        return x


_BICUBIC_CACHE = {}


def bicubic_matrix(side: int, H: int, W: int, device, dtype) -> torch.Tensor:
    """Interpolation matrix of F.interpolate(mode='bicubic', align_corners=False) from side x side to H x W
    (clip.py:101-104), columns padded to a multiple of 8.  A constant of the geometry, built once on the host."""
    key = (side, H, W, str(device), dtype)
    if key not in _BICUBIC_CACHE:
        eye = torch.eye(side * side).view(side * side, 1, side, side)
        R = TF.interpolate(eye, size=(H, W), mode="bicubic", align_corners=False).reshape(side * side, H * W).t()
        Kp = (side * side + 7) // 8 * 8
        Rp = torch.zeros(H * W, Kp)
        Rp[:, :side * side] = R
        _BICUBIC_CACHE[key] = Rp.to(device=device, dtype=dtype).contiguous()
    return _BICUBIC_CACHE[key]


class AttentionPool2d(Bound):
    """clip.py:60-144 (CRIS variant: no CLS token, residual `connect` branch, bicubic-resized positions)."""

    def __init__(self, spacial_dim, embed_dim, num_heads, output_dim=None):
        super().__init__()
        self.spacial_dim = spacial_dim
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = Linear(embed_dim, embed_dim)
        self.q_proj = Linear(embed_dim, embed_dim)
        self.v_proj = Linear(embed_dim, embed_dim)
        self.c_proj = Linear(embed_dim, output_dim or embed_dim)
        self.num_heads = num_heads
        self.connect = nn.ModuleDict({"0": Conv2d(embed_dim, output_dim, 1), "1": BatchNorm(output_dim)})

    def _make_refs(self, store):
        n = self.spacial_dim ** 2
        self.pos_rows = WRef(store, self.positional_embedding, 1, n)

    def forward(self, x):
        B, H, W, C = x.shape
        # x feeds the `connect` convolution and the attention tokens: the token branch's gradient (created later, so its backward runs
        # first) rides a slot into the convolution's data-gradient epilogue
        slot = Fn.GradSlot() if (self.training and torch.is_grad_enabled() and x.requires_grad and Fn.FAN_SLOTS) else None
        res = Fn.conv_bn_act(x, self.connect["0"].w, self.connect["1"].buffers_ref(), ksize=1, relu=False, training=self.training,
                             grad_slot=slot)
        R = bicubic_matrix(self.spacial_dim, H, W, x.device, x.dtype)
        pos = Fn.table_matmul(R, self.pos_rows, x.dtype)                      # [H*W, C]
        tok = Fn.add_rows(x.view(B * H * W, C), pos, grad_slot=slot)
        o = Fn.mha(tok, tok, tok, self.q_proj.w, self.k_proj.w, self.v_proj.w, self.q_proj.b, self.k_proj.b, self.v_proj.b,
                   self.c_proj.w, self.c_proj.b, B=B, heads=self.num_heads)
        out = Fn.add_relu(o, res.view(B * H * W, -1))
        return out.view(B, H, W, -1)


class ModifiedResNet(Bound):
    """clip.py:147-223: 3-conv stem + AvgPool, 4 stages of Bottlenecks, attention pool.  Returns (layer2, layer3, pooled layer4)
    as channels-last maps."""

    def __init__(self, layers, output_dim, heads, input_resolution=224, width=64):
        super().__init__()
        self.width = width
        self.output_dim, self.input_resolution = output_dim, input_resolution
        self.conv1, self.bn1 = Conv2d(3, width // 2, 3), BatchNorm(width // 2)
        self.conv2, self.bn2 = Conv2d(width // 2, width // 2, 3), BatchNorm(width // 2)
        self.conv3, self.bn3 = Conv2d(width // 2, width, 3), BatchNorm(width)
        self._inplanes = width
        self.layer1 = self._make_layer(width, layers[0])
        self.layer2 = self._make_layer(width * 2, layers[1], stride=2)
        self.layer3 = self._make_layer(width * 4, layers[2], stride=2)
        self.layer4 = self._make_layer(width * 8, layers[3], stride=2)
        self.attnpool = AttentionPool2d(input_resolution // 32, width * 32, heads, output_dim)
        self.fan = None

    def check_supported(self, source: str = ""):
        """clip.py:165-185 is width-generic (RN50x4 = 80, RN50x16 = 96, RN50x64 = 128).  Widths that are multiples of 64 run (RN50 / RN101
        = 64; RN50x64 = 128 is pinned by `tests/golden/rn_wide`, a reference fixture at depth (1, 1, 1, 1)).  The 3x3 implicit-GEMM kernels
        need Cin % 32 == 0; widths 80 / 96 give 40- / 48- / 80-channel maps that would need zero-padded channel strides through the stem and
        layer1 (the mechanism CoordConv's 514 -> 544 uses) - not built.  Called by CROG.prepare(), i.e. when the model is bound to the GPU,
        so an unsupported archive is refused by name before the first forward."""
        if self.width % 64 != 0:
            raise NotImplementedError(f"crog_amd: CLIP ModifiedResNet of width {self.width}{' (' + source + ')' if source else ''} is not supported: the HIP "
                                      "path implements widths that are multiples of 64 (RN50, RN101, RN50x64 archives); RN50x4 / x16 need padded "
                                      "channel strides")

    def _make_layer(self, planes, blocks, stride=1):
        mods = [Bottleneck(self._inplanes, planes, stride)]
        self._inplanes = planes * 4
        mods += [Bottleneck(self._inplanes, planes) for _ in range(1, blocks)]
        return Stage(*mods)

    def forward(self, img, dtype, after_layer1=None):
        """`after_layer1`: host-side hook called after each of layer1, layer2, layer3 is enqueued (few launches, most of the
        tower's GPU time) — CROG.forward issues a slice of the launch-bound text tower there, so the GPU always has bulk work
        queued while the host works through the ~250 small text launches."""
        tr = self.training
        c1 = self.conv1.weight.shape[0]
        self.check_supported()
        s1, s2 = (Fn.BnLink(), Fn.BnLink()) if tr else (None, None)       # stem: conv1 -> conv2 -> conv3 is a plain chain
        x = Fn.conv_bn_act(img, self.conv1.w, self.bn1.buffers_ref(), ksize="s", relu=True, training=tr, wpad=(27, 32, c1), dtype=dtype,
                           stat_out=s1)
        x = Fn.conv_bn_act(x, self.conv2.w, self.bn2.buffers_ref(), ksize=3, relu=True, training=tr, stat_in=s1, stat_out=s2)
        x = Fn.conv_bn_act(x, self.conv3.w, self.bn3.buffers_ref(), ksize=3, relu=True, training=tr, stat_in=s2, pool=True)      # + the stem's AvgPool2d(2)
        l12 = Fn.BnLink() if tr else None          # layer1's output feeds layer2 and nothing else
        x = self.layer1(x, res_out=l12)
        if after_layer1 is not None:
            after_layer1()
        x2 = self.layer2(x, res_in=l12)
        if after_layer1 is not None:
            after_layer1()
        # layer2's / layer3's outputs also feed the neck (layers.py:373-386), whose backward runs first: its gradients wait in these
        # slots (`self.fan`, handed to FPN.forward by CROG.forward) and are added by the next stage's downsample pool backward
        g = tr and torch.is_grad_enabled() and x2.requires_grad and Fn.FAN_SLOTS and self.layer3[0].stride > 1 and self.layer4[0].stride > 1
        self.fan = (Fn.GradSlot(), Fn.GradSlot()) if g else None
        x3 = self.layer3(x2, fan_slot=self.fan[0] if g else None)
        if after_layer1 is not None:
            after_layer1()
        x4 = self.layer4(x3, fan_slot=self.fan[1] if g else None)
        x4 = self.attnpool(x4)
        return x2, x3, x4


class ResidualAttentionBlock(Bound):
    """clip.py:239-265: x + attn(ln_1(x)); x + c_proj(QuickGELU(c_fc(ln_2(x)))).  Token rows are batch-first."""

    def __init__(self, d_model, n_head, causal):
        super().__init__()
        self.attn = MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.ModuleDict({"c_fc": Linear(d_model, d_model * 4), "c_proj": Linear(d_model * 4, d_model)})
        self.ln_2 = LayerNorm(d_model)
        self.causal = causal

    def forward(self, x, B):
        # both residual streams have two consumers (a norm and the add after the sub-layer): the add's gradient rides a GradSlot into
        # that norm's backward (crog_ln_bwd dxadd) instead of an autograd accumulation pass - 24 launches per text tower
        g = self.training and torch.is_grad_enabled() and x.requires_grad and Fn.LN_GRAD_SLOTS
        s1, s2 = (Fn.GradSlot(), Fn.GradSlot()) if g else (None, None)
        h = self.ln_1(x, add_slot=s1)
        x = self.attn(h, h, h, B=B, causal=self.causal, res=x, training=self.training, res_slot=s1)
        h = self.ln_2(x, add_slot=s2)
        u = Fn.linear(h, self.mlp["c_fc"].w, self.mlp["c_fc"].b)
        a = Fn.quickgelu(u)
        return Fn.linear(a, self.mlp["c_proj"].w, self.mlp["c_proj"].b, res=x, res_slot=s2)


class Transformer(Bound):
    def __init__(self, width, layers, heads, causal):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, causal) for _ in range(layers)])

    def forward(self, x, B):
        for blk in self.resblocks:
            x = blk(x, B)
        return x


class VisionTransformer(Bound):
    """clip.py:286-332.  Patch embedding is an im2col-free GEMM: with stride == kernel the patches are a pure reshape."""

    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.output_dim, self.patch_size = input_resolution, output_dim, patch_size
        self.conv1 = Conv2d(3, width, patch_size)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, causal=False)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def _make_refs(self, store):
        W = self.class_embedding.shape[0]
        self.cls = WRef(store, self.class_embedding, 0, 1, W)
        self.pos = WRef(store, self.positional_embedding)
        self.wproj = WRef(store, self.proj)

    def forward(self, img, dtype):
        """[B, 3, R, R] fp32 -> patch-token features [B, (R/P)^2, output_dim]  (the CRIS variant keeps the patch tokens and
        drops CLS after the blocks, clip.py:326-330)."""
        B, _, H, Wd = img.shape
        P = self.patch_size
        if H != self.input_resolution or Wd != self.input_resolution:
            # the reference adds a fixed-length positional table (clip.py:320): any other grid fails there with a shape error
            raise RuntimeError(f"The size of tensor a ({(H // P) * (Wd // P) + 1}) must match the size of tensor b "
                               f"({self.positional_embedding.shape[0]}) at non-singleton dimension 1")
        G = (H // P) * (Wd // P)
        X = torch.empty(B * G, 3 * P * P, device=img.device, dtype=dtype)
        K.patchify(img, X, P)
        y = Fn.linear(X, self.conv1.w, None)
        x = Fn.vit_tokens(y, self.cls, self.pos, B)
        x = self.ln_pre(x)
        x = self.transformer(x, B)
        T = G + 1
        keep = (torch.arange(B * T, device=img.device).view(B, T)[:, 1:]).reshape(-1)
        x = self.ln_post(Fn.gather_rows(x, keep))
        return Fn.table_matmul(x, self.wproj, dtype).view(B, G, -1)


class CLIP(Bound):
    """clip.py:335-474 (towers + text front end; the unused contrastive `forward` of CLIP is not part of CROG's path)."""

    def __init__(self, embed_dim, image_resolution, vision_layers: Union[Tuple[int, int, int, int], int], vision_width, vision_patch_size,
                 context_length, txt_length, vocab_size, transformer_width, transformer_heads, transformer_layers):
        super().__init__()
        self.context_length = context_length
        if isinstance(vision_layers, (tuple, list)):
            self.visual = ModifiedResNet(vision_layers, embed_dim, vision_width * 32 // 64, image_resolution, vision_width)
        else:
            self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers, vision_width // 64, embed_dim)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads, causal=True)
        self.vocab_size = vocab_size
        self.token_embedding = nn.Module()
        self.token_embedding.weight = nn.Parameter(torch.empty(vocab_size, transformer_width))
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.txt_length = txt_length
        self.initialize_parameters()

    def initialize_parameters(self):
        """clip.py:390-422 restated."""
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        if isinstance(self.visual, ModifiedResNet):
            ap = self.visual.attnpool
            std = ap.c_proj.weight.shape[1] ** -0.5
            for lin in (ap.q_proj, ap.k_proj, ap.v_proj, ap.c_proj):
                nn.init.normal_(lin.weight, std=std)
            for layer in (self.visual.layer1, self.visual.layer2, self.visual.layer3, self.visual.layer4):
                for blk in layer:
                    nn.init.zeros_(blk.bn3.weight)
        width, layers = self.transformer.width, self.transformer.layers
        proj_std, attn_std, fc_std = (width ** -0.5) * ((2 * layers) ** -0.5), width ** -0.5, (2 * width) ** -0.5
        for blk in self.transformer.resblocks:
            nn.init.normal_(blk.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(blk.mlp["c_fc"].weight, std=fc_std)
            nn.init.normal_(blk.mlp["c_proj"].weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=width ** -0.5)

    def _make_refs(self, store):
        self.tok = WRef(store, self.token_embedding.weight)
        self.pos = WRef(store, self.positional_embedding)
        self.tproj = WRef(store, self.text_projection)

    # channels-last internal forms -------------------------------------------------------------
    def image_features(self, image, dtype, after_layer1=None):
        if after_layer1 is not None and isinstance(self.visual, ModifiedResNet):
            return self.visual(image.float().contiguous(), dtype, after_layer1)
        if after_layer1 is not None:
            after_layer1()
        return self.visual(image.float().contiguous(), dtype)

    def text_features(self, text, dtype):
        """clip.py:439-456 -> (token features [B, L, D], state [B, embed])."""
        out = None
        for out in self.text_features_steps(text, dtype, parts=1):
            pass
        return out

    def text_features_steps(self, text, dtype, parts=3):
        """Generator form of text_features: yields None after each 1/parts of the transformer blocks has been enqueued and the
        (token features, state) pair at the end, so a caller can interleave the host-side issue with other work."""
        B, L = text.shape
        if L != self.txt_length:
            raise RuntimeError(f"The shape of the 2D attn_mask is ({self.txt_length}, {self.txt_length}), but should be ({L}, {L}).")
        x = Fn.embedding(text, self.tok, self.pos, dtype)
        blocks = list(self.transformer.resblocks)
        per = max(1, (len(blocks) + parts - 1) // parts)
        for i, blk in enumerate(blocks):
            x = blk(x, B)
            if (i + 1) % per == 0 and i + 1 < len(blocks):
                yield None
        x = self.ln_final(x)
        idx = torch.arange(B, device=text.device) * L + text.argmax(dim=-1)
        state = Fn.table_matmul(Fn.gather_rows(x, idx), self.tproj, dtype)
        yield x.view(B, L, -1), state

    # reference-shaped public API ------------------------------------------------------------------
    def encode_image(self, image, dtype=None):
        dtype = dtype or _auto_dtype()
        feats = self.image_features(image, dtype)
        if isinstance(feats, tuple):
            return tuple(f.permute(0, 3, 1, 2) for f in feats)
        return feats

    def encode_text(self, text, dtype=None):
        return self.text_features(text, dtype or _auto_dtype())


def _auto_dtype():
    return torch.bfloat16 if torch.is_autocast_enabled() else torch.float32


def build_model(arch: dict, txt_length: int) -> CLIP:
    """clip.py:503-556 equivalent when the architecture is given directly (no TorchScript archive on this path)."""
    return CLIP(arch["embed_dim"], arch["image_resolution"], arch["vision_layers"], arch["vision_width"], arch["vision_patch_size"],
                arch["context_length"], txt_length, arch["vocab_size"], arch["transformer_width"], arch["transformer_heads"],
                arch["transformer_layers"])


CLIP_META_KEYS = ("input_resolution", "context_length", "vocab_size")     # clip.py:547-549


def fp16_converted(name: str, dims: dict) -> bool:
    """Does `convert_weights` (clip.py:477-500) cast this tensor to fp16?  It touches Conv/Linear weights and biases, the packed /
    separate projection tensors of nn.MultiheadAttention, `text_projection` and `proj` — NOT BatchNorm / LayerNorm affine
    parameters and statistics, embeddings, positional / class embeddings or `logit_scale`, which keep their fp32 values.
    `dims`: tensor rank by state-dict key (tells a Linear bias from a norm bias through its sibling weight)."""
    leaf = name.rsplit(".", 1)[-1]
    if leaf in ("in_proj_weight", "in_proj_bias", "q_proj_weight", "k_proj_weight", "v_proj_weight", "bias_k", "bias_v"):
        return True
    if leaf in ("text_projection", "proj"):
        return True
    if "token_embedding" in name:
        return False
    if leaf == "weight":
        return dims.get(name, 0) >= 2
    if leaf == "bias":
        return dims.get(name[:-4] + "weight", 0) >= 2
    return False


def load_pretrained_clip(model: "CLIP", sd: dict):
    """`convert_weights(model); model.load_state_dict(state_dict, False)` of clip.py:551-554 on fp32 storage: every tensor of a
    converted class goes through an fp16 round trip — the checkpoint's values AND the random initialisation of modules the archive
    does not carry (`visual.attnpool.connect.*`, a CRIS addition) — everything else is copied at full precision.  strict=False."""
    own = model.state_dict()
    dims = {k: v.dim() for k, v in own.items()}
    with torch.no_grad():
        for k, v in own.items():
            if v.is_floating_point() and fp16_converted(k, dims):
                v.copy_(v.half().float())
    clean = {}
    for k, v in sd.items():
        if k in CLIP_META_KEYS:
            continue
        if torch.is_tensor(v) and v.is_floating_point():
            v = v.half().float() if fp16_converted(k, dims) else v.float()
        clean[k] = v
    return model.load_state_dict(clean, strict=False)


def arch_from_state_dict(sd: dict) -> dict:
    """Architecture inference from checkpoint tensor shapes, as clip.py:503-542 does."""
    vit = "visual.proj" in sd
    if vit:
        vw = sd["visual.conv1.weight"].shape[0]
        vl = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
        ps = sd["visual.conv1.weight"].shape[-1]
        res = ps * round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    else:
        vl = tuple(len({k.split(".")[2] for k in sd if k.startswith(f"visual.layer{b}")}) for b in (1, 2, 3, 4))
        vw = sd["visual.layer1.0.conv1.weight"].shape[0]
        ow = round((sd["visual.attnpool.positional_embedding"].shape[0] - 1) ** 0.5)
        ps, res = None, ow * 32
    tw = sd["ln_final.weight"].shape[0]
    return dict(embed_dim=sd["text_projection"].shape[1], image_resolution=res, vision_layers=vl, vision_width=vw, vision_patch_size=ps,
                context_length=sd["positional_embedding"].shape[0], vocab_size=sd["token_embedding.weight"].shape[0],
                transformer_width=tw, transformer_heads=tw // 64,
                transformer_layers=len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")}))
