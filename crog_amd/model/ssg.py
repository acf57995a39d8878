"""SSG-R50 trunk on the HIP path (BASELINE config 5): reference model/ssg.py:15-293.

Module tree, parameter names and shapes follow the reference (`backbone.layers.1.0.downsample.0.weight`,
`fpn.lat_layers.2.bias`, `proto_net.proto1.4.weight`, `prediction_layers.grasp_coef_layer.0.weight`, `semantic_seg_conv.*`),
so reference checkpoints and optimizer state load unchanged.  Internally every map is channels-last [B, H, W, C]:

  * 7x7/s2 stem, 3x3/s2 and 1x1/s2 convolutions = explicit im2col rows + the GEMM kernel (`Fn.im2col`, `crog_im2col_image`);
    3x3/s1 = implicit GEMM; 1x1/s1 = plain GEMM; BatchNorm statistics come out of the GEMM epilogue as for CROG;
  * bias/ReLU/tanh of the FPN, ProtoNet and prediction heads are GEMM epilogues; the FPN top-down add is the lateral GEMM's
    residual epilogue;
  * the heads' `permute(0, 2, 3, 1).reshape(B, -1, k)` (ssg.py:138-142) is free: channels-last rows already are that layout.

The data-dependent loss (ssg.py:297-530, row S2: per-image anchor matching, OHEM, cropped proto masks) stays PyTorch-ROCm host
logic by the scope table: crog_amd/ssg_loss.py, called from `SSG.compute_loss`.
"""
from __future__ import annotations

import math
from itertools import product
from typing import List

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import kernels as K
from ..runtime import RT, ParamStore
from .blocks import BatchNorm, Bound, Conv2d, bind_all


def _xavier(conv: Conv2d) -> Conv2d:
    """ssg.py:237-241: every Conv2d gets xavier_uniform weights and a zero bias."""
    nn.init.xavier_uniform_(conv.weight.data)
    if conv.bias is not None:
        conv.bias.data.zero_()
    return conv


class Bottleneck(Bound):
    """ssg.py:15-50 (torchvision style: the stride sits on the 3x3 convolution; downsample = strided 1x1 + BN)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1, self.bn1 = _xavier(Conv2d(inplanes, planes, 1)), BatchNorm(planes)
        self.conv2, self.bn2 = _xavier(Conv2d(planes, planes, 3)), BatchNorm(planes)
        self.conv3, self.bn3 = _xavier(Conv2d(planes, planes * 4, 1)), BatchNorm(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.ModuleDict({"0": _xavier(Conv2d(inplanes, planes * 4, 1)), "1": BatchNorm(planes * 4)})
        self.stride = stride

    def forward(self, x):
        tr = self.training
        # bn1 -> conv2 (when conv2 reads bn1's output itself, i.e. stride 1) and bn2 -> conv3: the consuming convolution's data
        # gradient does the first pass of the BatchNorm backward in its epilogue (Fn.BnLink)
        l1 = Fn.BnLink() if (tr and self.stride == 1) else None
        l2 = Fn.BnLink() if tr else None
        out = Fn.conv_bn_act(x, self.conv1.w, self.bn1.buffers_ref(), ksize=1, relu=True, training=tr, stat_out=l1)
        if self.stride == 1:
            out = Fn.conv_bn_act(out, self.conv2.w, self.bn2.buffers_ref(), ksize=3, relu=True, training=tr, stat_in=l1, stat_out=l2)
        else:
            out = Fn.conv_bn_act(Fn.im2col(out, 3, 3, self.stride, 1), self.conv2.w, self.bn2.buffers_ref(), ksize=1, relu=True, training=tr,
                                 stat_out=l2)
        identity = x
        if self.downsample is not None:
            xin = x if self.stride == 1 else Fn.im2col(x, 1, 1, self.stride, 0)
            identity = Fn.conv_bn_act(xin, self.downsample["0"].w, self.downsample["1"].buffers_ref(), ksize=1, relu=False, training=tr)
        return Fn.conv_bn_act(out, self.conv3.w, self.bn3.buffers_ref(), ksize=1, relu=True, res=identity, training=tr, stat_in=l2)


class ResNet(Bound):
    """ssg.py:53-114."""

    def __init__(self, layers, in_channels=3):
        super().__init__()
        self.layers = nn.ModuleList()          # registered first, as in the reference (state_dict / optimizer order)
        self.channels: List[int] = []
        self.inplanes = 64
        self.conv1 = _xavier(Conv2d(in_channels, 64, 7))
        self.bn1 = BatchNorm(64)
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            self._make_layer(planes, n, stride=1 if i == 0 else 2)

    def _make_layer(self, planes, blocks, stride):
        need_ds = stride != 1 or self.inplanes != planes * 4
        mods = [Bottleneck(self.inplanes, planes, stride, need_ds)]
        self.inplanes = planes * 4
        mods += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        self.channels.append(planes * 4)
        self.layers.append(nn.Sequential(*mods))

    def forward(self, img, dtype):
        """img: NCHW fp32 (3 or 4 channels) -> tuple of the four stage outputs, channels-last."""
        B, C, H, W = img.shape
        OH, OW = K.conv_out(H, 7, 2, 3), K.conv_out(W, 7, 2, 3)
        kc = 49 * C
        kp = (kc + 7) // 8 * 8
        col = torch.empty(B, OH, OW, kp, device=img.device, dtype=dtype)
        K.im2col_image(img, col, 7, 7, 2, 3)
        x = Fn.conv_bn_act(col, self.conv1.w, self.bn1.buffers_ref(), ksize=1, relu=True, training=self.training,
                           wpad=(kc, kp, 64) if kp != kc else None, dtype=dtype)
        x = Fn.maxpool3s2(x)
        outs = []
        for layer in self.layers:
            x = layer(x)
            outs.append(x)
        return tuple(outs)


class _Seq(nn.ModuleDict):
    """nn.Sequential whose activation entries hold no state: only the convolution indices exist ('0', '2', '4')."""


class FPN(Bound):
    """ssg.py:172-205."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = list(in_channels)
        self.lat_layers = nn.ModuleList([_xavier(Conv2d(c, 256, 1, bias=True)) for c in self.in_channels])
        self.pred_layers = nn.ModuleList([_Seq({"0": _xavier(Conv2d(256, 256, 3, bias=True))}) for _ in self.in_channels])
        self.downsample_layers = nn.ModuleList([_Seq({"0": _xavier(Conv2d(256, 256, 3, bias=True))}) for _ in range(2)])

    def forward(self, c3, c4, c5):
        lat, pred, down = self.lat_layers, self.pred_layers, self.downsample_layers
        p5_1 = Fn.linear(c5, lat[2].w, lat[2].b)
        p4_1 = Fn.linear(c4, lat[1].w, lat[1].b, res=Fn.upsample2(p5_1))     # lateral 1x1 + top-down add in one epilogue
        p3_1 = Fn.linear(c3, lat[0].w, lat[0].b, res=Fn.upsample2(p4_1))
        p5 = Fn.conv_bias_act(p5_1, pred[2]["0"].w, pred[2]["0"].b, ksize=3, act=K.ACT_RELU)
        p4 = Fn.conv_bias_act(p4_1, pred[1]["0"].w, pred[1]["0"].b, ksize=3, act=K.ACT_RELU)
        p3 = Fn.conv_bias_act(p3_1, pred[0]["0"].w, pred[0]["0"].b, ksize=3, act=K.ACT_RELU)
        p6 = Fn.conv_bias_act(p5, down[0]["0"].w, down[0]["0"].b, ksize=3, stride=2, act=K.ACT_RELU)
        p7 = Fn.conv_bias_act(p6, down[1]["0"].w, down[1]["0"].b, ksize=3, stride=2, act=K.ACT_RELU)
        return p3, p4, p5, p6, p7


class ProtoNet(Bound):
    """ssg.py:150-169."""

    def __init__(self, coef_dim):
        super().__init__()
        self.proto1 = _Seq({k: _xavier(Conv2d(256, 256, 3, bias=True)) for k in ("0", "2", "4")})
        self.proto2 = _Seq({"0": _xavier(Conv2d(256, 256, 3, bias=True)), "2": _xavier(Conv2d(256, coef_dim, 1, bias=True))})

    def forward(self, x):
        for k in ("0", "2", "4"):
            x = Fn.conv_bias_act(x, self.proto1[k].w, self.proto1[k].b, ksize=3, act=K.ACT_RELU)
        x = Fn.upsample2ac(x)
        x = Fn.conv_bias_act(x, self.proto2["0"].w, self.proto2["0"].b, ksize=3, act=K.ACT_RELU)
        return Fn.conv_bias_act(x, self.proto2["2"].w, self.proto2["2"].b, ksize=1, act=K.ACT_RELU)     # [B, 2H, 2W, coef_dim]


class PredictionModule(Bound):
    """ssg.py:117-147: one head shared by the five pyramid levels."""

    def __init__(self, cfg, coef_dim=32):
        super().__init__()
        self.num_classes, self.coef_dim = cfg.num_classes, coef_dim
        na = len(cfg.aspect_ratios)
        self.upfeature = _Seq({"0": _xavier(Conv2d(256, 256, 3, bias=True))})
        self.bbox_layer = _xavier(Conv2d(256, na * 4, 3, bias=True))
        self.conf_layer = _xavier(Conv2d(256, na * self.num_classes, 3, bias=True))
        self.coef_layer = _Seq({"0": _xavier(Conv2d(256, na * coef_dim, 3, bias=True))})
        self.grasp_coef_layer = None
        if cfg.with_grasp_masks:
            self.grasp_coef_layer = _Seq({"0": _xavier(Conv2d(256, na * coef_dim * 4, 3, bias=True))})

    def forward(self, x):
        B = x.shape[0]
        u = Fn.conv_bias_act(x, self.upfeature["0"].w, self.upfeature["0"].b, ksize=3, act=K.ACT_RELU)
        conf = Fn.conv_bias_act(u, self.conf_layer.w, self.conf_layer.b, ksize=3).reshape(B, -1, self.num_classes)
        box = Fn.conv_bias_act(u, self.bbox_layer.w, self.bbox_layer.b, ksize=3).reshape(B, -1, 4)
        coef = Fn.conv_bias_act(u, self.coef_layer["0"].w, self.coef_layer["0"].b, ksize=3, act=K.ACT_TANH).reshape(B, -1, self.coef_dim)
        if self.grasp_coef_layer is None:
            # ssg.py:144 dereferences the layer unconditionally: without grasp masks the reference fails here as well
            raise AttributeError("'PredictionModule' object has no attribute 'grasp_coef_layer'")
        g = self.grasp_coef_layer["0"]
        gcoef = Fn.conv_bias_act(u, g.w, g.b, ksize=3, act=K.ACT_TANH).reshape(B, -1, 4, self.coef_dim)
        return conf, box, coef, gcoef


def make_anchors(cfg, conv_h, conv_w, scale):
    """utils/box_utils.py:88-103: centre-form priors, row-major over the map, one per aspect ratio."""
    out = []
    for j, i in product(range(conv_h), range(conv_w)):
        x, y = (i + 0.5) / conv_w, (j + 0.5) / conv_h
        for ar in cfg.aspect_ratios:
            r = math.sqrt(ar)
            out += [x, y, scale * r / cfg.img_size, scale / r / cfg.img_size]
    return out


class SSG(nn.Module):
    """ssg.py:208-293."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        if cfg.backbone != "resnet":
            raise NotImplementedError
        self.backbone = ResNet(cfg.resnet_layers, in_channels=4 if cfg.with_depth else 3)
        self.fpn = FPN(cfg.fpn_in_channels)
        self.proto_net = ProtoNet(cfg.num_protos)
        self.prediction_layers = PredictionModule(cfg, coef_dim=cfg.num_protos)
        self.anchors: List[float] = []
        scales = [int(cfg.img_size / 544 * a) for a in (24, 48, 96, 192, 384)]
        for s, stride in zip(scales, cfg.anchor_strides):
            n = math.ceil(cfg.img_size / stride)
            self.anchors += make_anchors(cfg, n, n, s)
        self.semantic_seg_conv = _xavier(Conv2d(256, cfg.num_classes, 1, bias=True))
        self.compute_dtype = None
        self._store = None
        self.explicit_grad_ready = True

    # ---- flat parameter storage (as CROG) ---------------------------------------------------------------
    def prepare(self, device=None):
        device = torch.device(device if device is not None else next(self.parameters()).device)
        if device.type != "cuda":
            raise RuntimeError("crog_amd.SSG runs on an MI355X only: there is no CPU path (move the module with .cuda())")
        for name, buf in list(self.named_buffers()):
            if buf.device != device:
                mod = self
                *path, leaf = name.split(".")
                for p in path:
                    mod = getattr(mod, p)
                mod._buffers[leaf] = buf.to(device)
        self._store = ParamStore(self, device)
        self._store.explicit = True
        bind_all(self, self._store)
        self._bns = [m for m in self.modules() if isinstance(m, BatchNorm)]
        return self

    @property
    def store(self) -> ParamStore:
        return self._store

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._store = None
        return out

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        if self._store is not None:
            self._store.invalidate_shadow()
        return out

    # ---- forward ------------------------------------------------------------------------------------------
    def trunk(self, img, dtype=None):
        """NCHW fp32 image (RGB or RGB-D) -> raw predictions (fp32): class logits [B, A, classes], boxes [B, A, 4], instance
        coefficients [B, A, P], grasp coefficients [B, A, 4, P], prototypes [B, H/4, W/4, P], semantic logits [B, classes, H/8, W/8]."""
        dev = img.device
        if self._store is None or not self._store.valid() or self._store.device != dev:
            self.prepare(dev)
        dtype = dtype or self.compute_dtype or (torch.bfloat16 if torch.is_autocast_enabled() else torch.float32)
        st = self._store
        st.forward_begins()
        if self.training and torch.is_grad_enabled():
            RT.join_streams()
            st.fresh_grads_if_dropped()
            RT.begin_step(dev)
        with torch.autocast("cuda", enabled=False):
            RT.streams = [torch.cuda.current_stream()]
            st.weights(dtype)
            c2, c3, c4, c5 = self.backbone(img.float().contiguous(), dtype)
            levels = self.fpn(c3, c4, c5)
            protos = self.proto_net(levels[0])
            per_level = [self.prediction_layers(p) for p in levels]
            conf, box, coef, gcoef = (torch.cat([lv[i] for lv in per_level], dim=1).float() for i in range(4))
            seg = Fn.linear(levels[0], self.semantic_seg_conv.w, self.semantic_seg_conv.b)
            if self.training:
                torch._foreach_add_([m.num_batches_tracked for m in self._bns], 1)
        outs = (conf, box, coef, gcoef, protos.float(), seg.permute(0, 3, 1, 2).float())
        if self.training and torch.is_grad_enabled():
            outs = Fn.backward_begin(st, *outs)     # first node of backward: honours a set_to_none zero_grad after the forward
        return dict(zip(("class_pred", "box_pred", "ins_coef_pred", "grasp_coef_pred", "protos", "seg_pred"), outs))

    def forward(self, data_dict):
        """ssg.py:248-293.  Eval: the reference's output_dict.  Train with targets in `data_dict`: (output_dict, loss_dict);
        train without targets: (output_dict, raw predictions)."""
        img = torch.cat([data_dict["rgb"], data_dict["depth"]], dim=1) if self.cfg.with_depth else data_dict["rgb"]
        raw = self.trunk(img)
        out = {"anchors": self.anchors, "protos": raw["protos"], "cls_pred": torch.softmax(raw["class_pred"], -1),
               "box_pred": raw["box_pred"], "ins_coef_pred": raw["ins_coef_pred"], "grasp_coef_pred": raw["grasp_coef_pred"]}
        if self.training:
            if "bboxes" in data_dict:
                return out, self.compute_loss(raw, data_dict, out)
            return out, raw
        return out

    def compute_loss(self, raw, data_dict, output_dict=None):
        """ssg.py:297-350 -> the reference's eight-entry loss dict (host-side PyTorch logic, crog_amd/ssg_loss.py)."""
        from ..ssg_loss import ssg_loss
        dev = raw["class_pred"].device
        if isinstance(self.anchors, list):
            self.anchors = torch.tensor(self.anchors, device=dev).reshape(-1, 4)
        return ssg_loss(self.cfg, self.anchors.to(dev), raw, data_dict, output_dict)


def build_ssg(cfg):
    return SSG(cfg)
