"""CPU oracle for the SSG-R50 trunk (BASELINE config 5) — TEST INFRASTRUCTURE ONLY.

A functional restatement of the reference's model/ssg.py forward on plain torch CPU ops, parameterised by a state dict with the
reference's key names.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product
(crog_amd/) never does.  Pinned against fixtures captured from the reference itself (oracle/make_golden.py ssg ->
tests/golden/ssg_tiny_*.npz): see tests/test_oracle_golden.py::test_oracle_ssg_trunk_matches_reference.
"""
from __future__ import annotations

import math
from itertools import product
from typing import Dict, List

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]


def _bn(P: State, pre: str, x: Tensor, training: bool) -> Tensor:
    """nn.BatchNorm2d defaults (eps 1e-5, momentum 0.1); updates the running statistics in P when training."""
    return F.batch_norm(x, P[pre + ".running_mean"], P[pre + ".running_var"], P[pre + ".weight"], P[pre + ".bias"], training, 0.1, 1e-5)


def bottleneck(P: State, pre: str, x: Tensor, stride: int, training: bool) -> Tensor:
    """ssg.py:29-50: 1x1-BN-ReLU, 3x3(stride)-BN-ReLU, 1x1-BN, + (strided 1x1-BN downsample | identity), ReLU."""
    out = F.relu(_bn(P, pre + ".bn1", F.conv2d(x, P[pre + ".conv1.weight"]), training))
    out = F.relu(_bn(P, pre + ".bn2", F.conv2d(out, P[pre + ".conv2.weight"], stride=stride, padding=1), training))
    out = _bn(P, pre + ".bn3", F.conv2d(out, P[pre + ".conv3.weight"]), training)
    res = x
    if pre + ".downsample.0.weight" in P:
        res = _bn(P, pre + ".downsample.1", F.conv2d(x, P[pre + ".downsample.0.weight"], stride=stride), training)
    return F.relu(out + res)


def resnet(P: State, img: Tensor, training: bool, pre: str = "backbone") -> List[Tensor]:
    """ssg.py:98-109: 7x7/s2 stem, BN, ReLU, MaxPool(3,2,1), four stages (stride 1,2,2,2 on the first block of each)."""
    x = F.relu(_bn(P, pre + ".bn1", F.conv2d(img, P[pre + ".conv1.weight"], stride=2, padding=3), training))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li in range(4):
        bi = 0
        while f"{pre}.layers.{li}.{bi}.conv1.weight" in P:
            x = bottleneck(P, f"{pre}.layers.{li}.{bi}", x, 2 if (bi == 0 and li > 0) else 1, training)
            bi += 1
        outs.append(x)
    return outs


def _conv(P: State, pre: str, x: Tensor, stride: int = 1) -> Tensor:
    w = P[pre + ".weight"]
    return F.conv2d(x, w, P[pre + ".bias"], stride=stride, padding=w.shape[-1] // 2)


def fpn(P: State, c3: Tensor, c4: Tensor, c5: Tensor, pre: str = "fpn") -> List[Tensor]:
    """ssg.py:192-205."""
    up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
    p5_1 = _conv(P, pre + ".lat_layers.2", c5)
    p4_1 = _conv(P, pre + ".lat_layers.1", c4) + up(p5_1)
    p3_1 = _conv(P, pre + ".lat_layers.0", c3) + up(p4_1)
    p5 = F.relu(_conv(P, pre + ".pred_layers.2.0", p5_1))
    p4 = F.relu(_conv(P, pre + ".pred_layers.1.0", p4_1))
    p3 = F.relu(_conv(P, pre + ".pred_layers.0.0", p3_1))
    p6 = F.relu(_conv(P, pre + ".downsample_layers.0.0", p5, stride=2))
    p7 = F.relu(_conv(P, pre + ".downsample_layers.1.0", p6, stride=2))
    return [p3, p4, p5, p6, p7]


def proto_net(P: State, x: Tensor, pre: str = "proto_net") -> Tensor:
    """ssg.py:166-169 -> channels-last prototypes (ssg.py:256)."""
    for k in ("0", "2", "4"):
        x = F.relu(_conv(P, f"{pre}.proto1.{k}", x))
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    x = F.relu(_conv(P, pre + ".proto2.0", x))
    x = F.relu(_conv(P, pre + ".proto2.2", x))
    return x.permute(0, 2, 3, 1).contiguous()


def prediction(P: State, x: Tensor, num_classes: int, coef_dim: int, pre: str = "prediction_layers"):
    """ssg.py:137-147."""
    B = x.shape[0]
    u = F.relu(_conv(P, pre + ".upfeature.0", x))
    conf = _conv(P, pre + ".conf_layer", u).permute(0, 2, 3, 1).reshape(B, -1, num_classes)
    box = _conv(P, pre + ".bbox_layer", u).permute(0, 2, 3, 1).reshape(B, -1, 4)
    coef = torch.tanh(_conv(P, pre + ".coef_layer.0", u)).permute(0, 2, 3, 1).reshape(B, -1, coef_dim)
    gcoef = torch.tanh(_conv(P, pre + ".grasp_coef_layer.0", u)).permute(0, 2, 3, 1).reshape(B, -1, 4, coef_dim)
    return conf, box, coef, gcoef


def ssg_trunk(P: State, img: Tensor, num_classes: int = 32, coef_dim: int = 32, training: bool = True) -> Dict[str, Tensor]:
    """ssg.py:248-281 up to the raw predictions (img = cat(rgb, depth) when with_depth)."""
    c2, c3, c4, c5 = resnet(P, img, training)
    levels = fpn(P, c3, c4, c5)
    protos = proto_net(P, levels[0])
    per = [prediction(P, lv, num_classes, coef_dim) for lv in levels]
    conf, box, coef, gcoef = (torch.cat([p[i] for p in per], dim=1) for i in range(4))
    seg = _conv(P, "semantic_seg_conv", levels[0])
    return dict(class_pred=conf, box_pred=box, ins_coef_pred=coef, grasp_coef_pred=gcoef, protos=protos, seg_pred=seg)


def make_anchors(aspect_ratios, img_size: int, conv_h: int, conv_w: int, scale: float) -> List[float]:
    """utils/box_utils.py:88-103."""
    out = []
    for j, i in product(range(conv_h), range(conv_w)):
        x, y = (i + 0.5) / conv_w, (j + 0.5) / conv_h
        for ar in aspect_ratios:
            r = math.sqrt(ar)
            out += [x, y, scale * r / img_size, scale / r / img_size]
    return out


def anchors(aspect_ratios, img_size: int, strides) -> List[float]:
    """ssg.py:227-231."""
    scales = [int(img_size / 544 * a) for a in (24, 48, 96, 192, 384)]
    out: List[float] = []
    for s, st in zip(scales, strides):
        n = math.ceil(img_size / st)
        out += make_anchors(aspect_ratios, img_size, n, n, s)
    return out
