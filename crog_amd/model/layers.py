"""Neck, decoder and projector heads of CROG on the HIP path (state_dict-compatible with the reference's model/layers.py).

Channel concatenations never materialise a copy: producers write straight into channel slices of the
destination buffer (row stride = total channels) and `Fn.join` stitches the autograd graph.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import kernels as K
from .blocks import BatchNorm, Bound, Conv2d, ConvBN, LayerNorm, Linear, MultiheadAttention, conv_layer, linear_layer


class CoordConv(Bound):
    """layers.py:19-44: append (x, y) coordinate channels in [-1, 1], then conv3x3-BN-ReLU.
    The 514-channel input is held as a 544-channel buffer (multiple of 32 for the implicit GEMM):
    channels 512/513 carry the constant coordinates, the tail is zero; the weight is zero-padded to match."""

    def __init__(self, cin, cout, k=3):
        super().__init__()
        self.conv1 = conv_layer(cin + 2, cout, k)
        self.cin = cin
        self.cpad = (cin + 2 + 31) // 32 * 32

    def make_input(self, B, H, W, device, dtype):
        buf = torch.empty(B, H, W, self.cpad, device=device, dtype=dtype)
        K.coord_fill(buf, self.cin, self.cpad)
        return buf

    def forward(self, buf, stat_out=None):
        cout = self.conv1.conv.weight.shape[0]
        return self.conv1.run(buf, ksize=3, wpad=(self.cin + 2, self.cpad, cout * 9), stat_out=stat_out)


class FPN(Bound):
    """layers.py:342-398."""

    def __init__(self, in_channels=(512, 1024, 1024), out_channels=(256, 512, 1024)):
        super().__init__()
        i, o = in_channels, out_channels
        self.txt_proj = linear_layer(i[2], o[2])
        self.f1_v_proj = conv_layer(i[2], o[2], 1)
        self.norm_layer = ConvBN(None, BatchNorm(o[2]), names=("_", "0"))
        self.f2_v_proj = conv_layer(i[1], o[1], 3)
        self.f2_cat = conv_layer(o[2] + o[1], o[1], 1)
        self.f3_v_proj = conv_layer(i[0], o[0], 3)
        self.f3_cat = conv_layer(o[0] + o[1], o[1], 1)
        self.f4_proj5 = conv_layer(o[2], o[1], 3)
        self.f4_proj4 = conv_layer(o[1], o[1], 3)
        self.f4_proj3 = conv_layer(o[1], o[1], 3)
        self.aggr = conv_layer(3 * o[1], o[1], 1)
        self.coordconv = nn.Sequential(CoordConv(o[1], o[1], 3), conv_layer(o[1], o[1], 3))
        self.o = tuple(o)

    def text_gate(self, state):
        """layers.py:376: the sentence gate relu(bn(linear(state))) [B, o2] - a function of the text tower's output alone.  CROG.forward
        computes it on the TEXT stream, right behind the tower (round 6): a 32-row linear + a BatchNorm1d over 32 rows are ~0.1 ms of
        latency-bound launches that the main chain otherwise runs between the image tower and the neck, and again in backward."""
        return self.txt_proj.run(state, ksize=1)

    def forward(self, imgs, state, fan=None, gate=None):
        """fan: (slot for v3's gradient, slot for v4's) - ModifiedResNet.fan; the tower's own backward adds them (clip.py Bottleneck).
        gate: text_gate(state) when the caller has already computed it (on another stream)."""
        v3, v4, v5 = imgs                      # [B,52,52,512] [B,26,26,1024] [B,13,13,1024] channels-last
        fan3, fan4 = fan if fan is not None else (None, None)
        o0, o1, o2 = self.o
        dev, dt = v4.device, v4.dtype
        B, H4, W4, _ = v4.shape
        s = gate if gate is not None else self.text_gate(state)                # Linear + BN1d + ReLU -> [B, o2]
        f5 = self.f1_v_proj.run(v5)
        f5 = self.norm_layer.run(Fn.mul_bcast(f5, s), ksize=0)                 # relu(bn(f5 * state))
        # fusion 2: cat([f2_v_proj(v4), up(f5)])
        cat2 = torch.empty(B, H4, W4, o1 + o2, device=dev, dtype=dt)
        a = self.f2_v_proj.run(v4, out=cat2[..., :o1], dx_slot=fan4)
        b = Fn.upsample2(f5, out=Fn.OutRef(cat2[..., o1:]))
        cat3 = torch.empty(B, H4, W4, o0 + o1, device=dev, dtype=dt)
        f4 = self.f2_cat.run(Fn.join(cat2, [a, b]), out=cat3[..., o0:])
        # fusion 3: cat([avgpool(f3_v_proj(v3)), f4])
        f3 = Fn.avgpool2(self.f3_v_proj.run(v3, dx_slot=fan3), out=Fn.OutRef(cat3[..., :o0]))
        # (f3 feeds f4_proj3 and nothing else, the CoordConv output only coordconv[1]: the consumer's data gradient does the producer's
        # first BatchNorm-backward pass, Fn.BnLink)
        tr = self.training
        l3, lc = (Fn.BnLink(), Fn.BnLink()) if tr else (None, None)
        f3 = self.f3_cat.run(Fn.join(cat3, [f3, f4]), stat_out=l3)
        # fusion 4
        catq = torch.empty(B, H4, W4, 3 * o1, device=dev, dtype=dt)
        fq5 = Fn.upsample2(self.f4_proj5.run(f5), out=Fn.OutRef(catq[..., 2 * o1:]))
        fq4 = self.f4_proj4.run(f4, out=catq[..., o1:2 * o1])
        fq3 = self.f4_proj3.run(f3, out=catq[..., :o1], stat_in=l3)
        cc = self.coordconv[0]
        cbuf = cc.make_input(B, H4, W4, dev, dt)
        fq = self.aggr.run(Fn.join(catq, [fq3, fq4, fq5]), out=cbuf[..., :o1])
        fq = cc(Fn.join(cbuf, [fq]), stat_out=lc)
        return self.coordconv[1].run(fq, stat_in=lc)


_POS_CACHE = {}


def pos1d(d_model, length, device, dtype):
    """layers.py:195-212 as a [length, d_model] table (constant of the shape, built once on the host)."""
    key = ("1d", d_model, length, str(device), dtype)
    if key not in _POS_CACHE:
        pos = torch.arange(length, dtype=torch.float32)[:, None]
        div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
        pe = torch.zeros(length, d_model)
        pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
        _POS_CACHE[key] = pe.to(device=device, dtype=dtype).contiguous()
    return _POS_CACHE[key]


def pos2d(d_model, height, width, device, dtype):
    """layers.py:214-241 as a [H*W, d_model] table: first half of the channels encodes x, second half y."""
    key = ("2d", d_model, height, width, str(device), dtype)
    if key not in _POS_CACHE:
        half = d_model // 2
        div = torch.exp(torch.arange(0.0, half, 2) * -(math.log(10000.0) / half))
        px = torch.arange(0.0, width)[:, None] * div
        py = torch.arange(0.0, height)[:, None] * div
        pe = torch.zeros(height, width, d_model)
        pe[:, :, 0:half:2] = torch.sin(px)[None, :, :]
        pe[:, :, 1:half:2] = torch.cos(px)[None, :, :]
        pe[:, :, half::2] = torch.sin(py)[:, None, :]
        pe[:, :, half + 1::2] = torch.cos(py)[:, None, :]
        _POS_CACHE[key] = pe.reshape(height * width, d_model).to(device=device, dtype=dtype).contiguous()
    return _POS_CACHE[key]


class TransformerDecoderLayer(Bound):
    """layers.py:280-339."""

    def __init__(self, d_model=512, nhead=9, dim_feedforward=2048, dropout=0.1):
        super().__init__()
        self.self_attn_norm = LayerNorm(d_model)
        self.cross_attn_norm = LayerNorm(d_model)
        self.self_attn = MultiheadAttention(d_model, nhead, dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout)
        self.ffn = nn.ModuleDict({"0": Linear(d_model, dim_feedforward), "3": LayerNorm(dim_feedforward), "4": Linear(dim_feedforward, d_model)})
        self.norm1, self.norm2, self.norm3 = LayerNorm(d_model), LayerNorm(d_model), LayerNorm(d_model)
        self.p = dropout

    def forward(self, vis, txt, txt_k, vis_pos, pad_mask, B, kv=None):
        p = self.p if self.training else 0.0
        tr = self.training
        # each `vis` below has two consumers, a norm and the residual add after the sub-layer: the residual's gradient travels in a
        # GradSlot to that norm's backward and is added there (crog_ln_bwd dxadd) instead of by an autograd accumulation pass
        g = tr and torch.is_grad_enabled() and vis.requires_grad and Fn.LN_GRAD_SLOTS
        s1, s2, s3 = (Fn.GradSlot(), Fn.GradSlot(), Fn.GradSlot()) if g else (None, None, None)
        v2, qk = self.norm1(vis, pos=vis_pos, want_out2=True, add_slot=s1)
        a = self.self_attn(qk, qk, v2, B=B, training=tr)
        vis = self.self_attn_norm(a, res=vis, p_out=p, res_slot=s1)
        _, q = self.norm2(vis, pos=vis_pos, want_out2=True, add_slot=s2)
        a = self.multihead_attn(q, txt_k, txt, B=B, kpm=pad_mask, training=tr, kv=kv)
        vis = self.cross_attn_norm(a, res=vis, p_out=p, res_slot=s2)
        v2 = self.norm3(vis, add_slot=s3 if p > 0 else None)
        # (the ReLU's backward rides in the LayerNorm's, which holds the ReLU output in registers anyway: no activation-backward pass
        # over the 21632 x 2048 map)
        fuse = tr and torch.is_grad_enabled() and Fn.LN_RELU_FUSED
        h = Fn.linear(v2, self.ffn["0"].w, self.ffn["0"].b, act=K.ACT_RELU, grad_gated=fuse)
        h = self.ffn["3"](h, p_in=p, relu_in=fuse)
        if p > 0:
            return Fn.add_dropout(vis, Fn.linear(h, self.ffn["4"].w, self.ffn["4"].b), p, res_slot=s3)
        return Fn.linear(h, self.ffn["4"].w, self.ffn["4"].b, res=vis)


class TransformerDecoder(Bound):
    """layers.py:176-277."""

    def __init__(self, num_layers, d_model, nhead, dim_ffn, dropout, return_intermediate=False):
        super().__init__()
        self.layers = nn.ModuleList([TransformerDecoderLayer(d_model, nhead, dim_ffn, dropout) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.norm = LayerNorm(d_model)
        self.return_intermediate = return_intermediate

    def text_kv(self, txt):
        """Everything of the decoder that depends on the text side alone (layers.py:252-253, 329-332): the word features with their 1-D positions
        and every layer's key / value projections of the cross-attention.  CROG.forward calls it on the text stream (round 6) and hands the result to
        forward(text=): six 80-block GEMMs forward and six backward leave the main chain."""
        B, L, D = txt.shape
        t = txt.reshape(B * L, D)
        t_k = Fn.add_rows(t, pos1d(D, L, txt.device, txt.dtype))
        return t, t_k, [layer.multihead_attn.project_kv(t_k, t) for layer in self.layers]

    def forward(self, vis, txt, pad_mask, text=None):
        """vis: [B, H, W, C] channels-last; txt: [B, L, D]; pad_mask: [B, L] bool -> [B, H, W, C].  text: text_kv(txt) when already computed."""
        B, H, W, C = vis.shape
        vis_pos = pos2d(C, H, W, vis.device, vis.dtype)
        x = vis.reshape(B * H * W, C)
        if text is not None:
            t, t_k, kvs = text
        else:
            _, L, D = txt.shape
            t = txt.reshape(B * L, D)
            t_k = Fn.add_rows(t, pos1d(D, L, vis.device, vis.dtype))
            kvs = [None] * len(self.layers)
        intermediate = []
        for layer, kv in zip(self.layers, kvs):
            x = layer(x, t, t_k, vis_pos, pad_mask, B, kv=kv)
            if self.return_intermediate:            # layers.py:262-274: the shared final norm applied to every layer's output
                intermediate.append(self.norm(x).view(B, H, W, C))
        if self.return_intermediate:
            return intermediate                     # [output_1, ..., output_n]; the last entry is the plain return value
        return self.norm(x).view(B, H, W, C)


class _ProjectorBase(Bound):
    def __init__(self, word_dim, in_dim, kernel_size, groups):
        super().__init__()
        if kernel_size != 3:
            raise NotImplementedError("dynamic head kernel is 3x3 (layers.py:42,45 pass kernel_size=3)")
        self.in_dim, self.kernel_size, self.groups = in_dim, kernel_size, groups
        self.vis = nn.ModuleDict({"1": conv_layer(in_dim * 2, in_dim * 2, 3), "3": conv_layer(in_dim * 2, in_dim, 3),
                                  "4": Conv2d(in_dim, in_dim * groups, 1, bias=True)})
        self.txt = Linear(word_dim, in_dim * 9 + 1)

    def text_word(self, state):
        """layers.py:90-91: the per-sample dynamic kernel + bias txt(state), fp32 [B, pad8(9 in_dim + 1)] - a function of the text tower's output
        alone (CROG.forward computes it on the text stream); None when the unfused head is selected."""
        return Fn.head_word(state, self.txt.w, self.txt.b, self.in_dim) if Fn.FUSED_HEAD else None

    def forward(self, x, state, word=None, word_stream=None):
        """x: [B, h, w, 2*in_dim] channels-last, state: [B, word_dim] -> fp32 logits [B, groups, 4h, 4w].  word: text_word(state) when the
        caller has already computed it, word_stream: the stream it was computed on (the backward of the head's weight side follows it there)."""
        x = self.vis["1"].run(Fn.upsample2(x))
        x = self.vis["3"].run(Fn.upsample2(x))
        if Fn.FUSED_HEAD:
            if word is None:
                word = self.text_word(state)
            return Fn.fused_head(x, word, self.vis["4"].w, self.vis["4"].b, self.in_dim, self.groups, tail_stream=word_stream)
        x5 = Fn.linear(x, self.vis["4"].w, self.vis["4"].b)
        return Fn.dyn_head(x5, state, self.txt.w, self.txt.b, self.in_dim)


class MultiTaskProjector(_ProjectorBase):
    """layers.py:47-132: five maps (mask, quality, sin, cos, width) from one 1x1 conv, one shared dynamic kernel."""

    def __init__(self, word_dim=1024, in_dim=256, kernel_size=3):
        super().__init__(word_dim, in_dim, kernel_size, 5)


class Projector(_ProjectorBase):
    """layers.py:135-173: segmentation map only."""

    def __init__(self, word_dim=1024, in_dim=256, kernel_size=3):
        super().__init__(word_dim, in_dim, kernel_size, 1)
