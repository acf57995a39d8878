"""SSG training loss and detection post-processing, batched on the device.

Reference: model/ssg.py:297-530 (`compute_loss`), utils/box_utils.py:8-37,57-199 (`box_iou`, `match`, `encode`, `crop`, `ones_crop`),
utils/grasp_eval.py:54-150 (`fast_nms` and the tensor half of `ssg_post_processing`).  The reference walks the batch image by image
(and the ground-truth boxes one by one inside `match`): ~25 small launches per image and loss term, i.e. thousands per step at
B = 64.  Here the ragged ground truth is padded once to [B, Gmax, ...], and

  * anchor <-> box matching, labels and SSD offsets of the whole batch are two launches of csrc/ssg.hip (`crog_ssg_match`);
  * OHEM cross entropy, box regression and the semantic term are single batched expressions;
  * the prototype x coefficient masks of ALL positive anchors are one batched GEMM per map ([B, Pmax, 32] x [B, 32, h*w]) followed
    by batched window / loss / normalisation arithmetic, with padded slots masked out.

Semantics follow the reference term by term (same normalisers, same tie rules, positives in anchor order, the same CPU
`randperm` stream when an image has more than `masks_to_train` positives), pinned by tests/golden/ssg_tiny_*.npz and ssg_b8_rgbd.npz.
Gradients flow back into the HIP trunk through ordinary autograd.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import kernels as K

GRASP_KEYS = ("qua", "sin", "cos", "wid")    # dict order of the collate function (utils/dataset.py:1409-1414) = coefficient index


# ---- ragged ground truth -> padded batch ---------------------------------------------------------------------------
def pad_ground_truth(rows: Sequence[torch.Tensor], device) -> (torch.Tensor, torch.Tensor):
    """list of [G_i, ...] -> ([B, Gmax, ...] zero padded, counts [B] int32)."""
    counts = torch.tensor([r.shape[0] for r in rows], dtype=torch.int32)
    padded = torch.nn.utils.rnn.pad_sequence([r.to(device) for r in rows], batch_first=True)
    return padded, counts.to(device)


def match_batch(cfg, bboxes: Sequence[torch.Tensor], anchors: torch.Tensor):
    """model/ssg.py:313-321 for the whole batch: (offsets [B,A,4], labels [B,A], matched boxes [B,A,4], matched index [B,A])."""
    dev = anchors.device
    gt, ng = pad_ground_truth([b.float() for b in bboxes], dev)
    return K.ssg_match(anchors.float().contiguous(), gt.contiguous(), ng, cfg.pos_iou_thre, cfg.neg_iou_thre)


# ---- loss terms that were batched already ------------------------------------------------------------------------------
def category_loss(cfg, class_pred: torch.Tensor, labels: torch.Tensor, pos: torch.Tensor, np_ratio: int = 3) -> torch.Tensor:
    """Cross entropy over positives + 3:1 hardest negatives (ssg.py:353-386)."""
    B, A, C = class_pred.shape
    flat = class_pred.reshape(-1, C)
    shift = flat.max()
    hardness = (torch.log(torch.exp(flat - shift).sum(1)) + shift - flat[:, 0]).reshape(B, A)
    hardness[pos] = 0
    hardness[labels < 0] = 0
    order = hardness.sort(1, descending=True)[1]
    rank = order.sort(1)[1]
    n_pos = pos.long().sum(1, keepdim=True)
    n_neg = torch.clamp(np_ratio * n_pos, max=A - 1)
    neg = rank < n_neg.expand_as(rank)
    neg[pos] = False
    neg[labels < 0] = False
    chosen = pos | neg
    return cfg.alpha_conf * F.cross_entropy(class_pred[chosen].reshape(-1, C), labels[chosen], reduction="sum") / n_pos.sum()


def box_loss(cfg, box_pred: torch.Tensor, offsets: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """ssg.py:389-394."""
    return cfg.alpha_bbox * F.smooth_l1_loss(box_pred[pos, :], offsets[pos, :], reduction="sum") / pos.sum()


# ---- positives of the whole batch, padded ---------------------------------------------------------------------------------
class Positives:
    """Positive anchors of every image as [B, Pm] index matrices (anchor order, as the reference's boolean indexing yields them).
    An image with more than `limit` positives keeps a random subset drawn with torch.randperm on the CPU generator — one draw per
    call of `select`, in image order — and its terms are re-weighted by n / limit (ssg.py:415-421,446-447)."""

    def __init__(self, pos: torch.Tensor, limit: int):
        self.limit = limit
        self.counts = pos.sum(1)                                   # [B], device
        self.counts_host = self.counts.tolist()                    # the step's one host sync of the loss (the reference has dozens)
        self.pmax = max(self.counts_host) if self.counts_host else 0
        # stable sort: positives first, each group in ascending anchor index
        self.order = torch.argsort(pos.to(torch.int8), dim=1, descending=True, stable=True)[:, :max(self.pmax, 1)]

    def base(self):
        """-> (anchor index [B, Pm], valid [B, Pm]) with Pm = min(max positives, limit); rows of over-limit images hold the first
        `limit` positives until a draw replaces them."""
        dev = self.order.device
        pm = max(min(self.pmax, self.limit), 1)
        kept = torch.tensor([min(n, self.limit) for n in self.counts_host], device=dev)
        return self.order[:, :pm].clone(), torch.arange(pm, device=dev)[None, :] < kept[:, None]

    def over_limit(self) -> List[int]:
        return [i for i, n in enumerate(self.counts_host) if n > self.limit]

    def apply_draws(self, idx: torch.Tensor, draws: Dict[int, torch.Tensor]):
        """Replace the rows of over-limit images by their drawn subsets; returns the re-weighting factors [B] (n / limit)."""
        weight = torch.ones(idx.shape[0], device=idx.device)
        for i, pick in draws.items():
            n = self.counts_host[i]
            idx[i, :self.limit] = self.order[i, :n][pick.to(idx.device)]
            weight[i] = n / self.limit
        return weight

    def select(self):
        """-> (anchor index [B, Pm], valid [B, Pm] bool, weight [B] float), drawing once per over-limit image, in image order."""
        idx, valid = self.base()
        draws = {i: torch.randperm(self.counts_host[i])[:self.limit] for i in self.over_limit()}     # CPU generator, as the reference
        return idx, valid, self.apply_draws(idx, draws)


def _take(t: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """t [B, A, ...] gathered along dim 1 by idx [B, Pm] -> [B, Pm, ...]."""
    view = idx.reshape(idx.shape + (1,) * (t.dim() - 2)).expand(idx.shape + t.shape[2:])
    return t.gather(1, view)


def _windows(h: int, w: int, boxes: torch.Tensor, padding: int = 1) -> torch.Tensor:
    """[B, Pm, h, w] boolean window of each (relative, corner-form) box (box_utils.py:120-169: relative -> absolute, ordered, padded,
    clamped; float bounds, not rounded)."""
    def span(lo, hi, size):
        a, b = lo * size, hi * size
        return (torch.min(a, b) - padding).clamp(min=0), (torch.max(a, b) + padding).clamp(max=size)
    x1, x2 = span(boxes[..., 0], boxes[..., 2], w)
    y1, y2 = span(boxes[..., 1], boxes[..., 3], h)
    col = torch.arange(w, device=boxes.device, dtype=boxes.dtype).view(1, 1, 1, w)
    row = torch.arange(h, device=boxes.device, dtype=boxes.dtype).view(1, 1, h, 1)
    return (col >= x1[..., None, None]) & (col < x2[..., None, None]) & (row >= y1[..., None, None]) & (row < y2[..., None, None])


def _resized_targets(maps: Sequence[torch.Tensor], h: int, w: int, device) -> torch.Tensor:
    """list of [G_i, H, W] -> [B, Gmax, h, w]: ONE bilinear resize (align_corners=False) of the padded stack (ssg.py:405-407,464-465)."""
    padded, _ = pad_ground_truth([m.float() for m in maps], device)
    return F.interpolate(padded, (h, w), mode="bilinear", align_corners=False)


def _mask_maps(protos: torch.Tensor, coef: torch.Tensor) -> torch.Tensor:
    """protos [B, h, w, K], coef [B, Pm, K] -> sigmoid(protos . coef) as [B, Pm, h, w]: one batched GEMM for the whole batch."""
    B, h, w, Kp = protos.shape
    return torch.sigmoid(torch.bmm(coef, protos.reshape(B, h * w, Kp).transpose(1, 2))).reshape(B, coef.shape[1], h, w)


def _area(boxes: torch.Tensor) -> torch.Tensor:
    return (boxes[..., 2] - boxes[..., 0]) * (boxes[..., 3] - boxes[..., 1])


def instance_mask_loss(cfg, coef_pred, protos, ins_masks: Sequence[torch.Tensor], P: Positives, matched_idx, matched_box, output_dict=None):
    """Prototype x coefficient masks of the positive anchors, cropped to the matched box, BCE normalised by box area (ssg.py:398-452)."""
    ph, pw = protos.shape[1:3]
    n_pos = P.counts.sum()
    if P.pmax == 0:
        return cfg.alpha_ins * protos.sum() * 0
    idx, valid, weight = P.select()
    boxes, gidx, coef = _take(matched_box, idx), _take(matched_idx, idx), _take(coef_pred, idx)
    target_all = _resized_targets(ins_masks, ph, pw, protos.device).gt(0.5).float()            # [B, Gmax, h, w]
    target = _take(target_all, gidx)                                                           # [B, Pm, h, w]
    pred = _mask_maps(protos, coef) * _windows(ph, pw, boxes).to(protos.dtype)                 # crop (box_utils.py:150-169)
    if output_dict is not None and getattr(cfg, "intermidiate_output", False):
        output_dict["inter_mask_p"] = pred.detach()[valid]
        output_dict["inter_mask_gt"] = target[valid]
    per = F.binary_cross_entropy(torch.clamp(pred, 0, 1), target, reduction="none").sum(dim=(2, 3))
    per = per / torch.where(valid, _area(boxes), torch.ones_like(per)) * weight[:, None]
    return cfg.alpha_ins * torch.where(valid, per, torch.zeros_like(per)).sum() / ph / pw / n_pos


def grasp_mask_losses(cfg, gcoef_pred, protos, grasp_masks: Dict[str, Sequence[torch.Tensor]], P: Positives, matched_idx, matched_box):
    """Four grasp maps per positive anchor (quality, sin, cos, width) from the same prototypes, smooth-L1 against the bilinearly
    resized targets; the cos map is 1 outside the box (ssg.py:456-509).  The reference draws a fresh subset per image AND per map
    when an image exceeds `masks_to_train`; the draws below happen in the same (image-major, map-minor) order."""
    ph, pw = protos.shape[1:3]
    n_pos = P.counts.sum()
    keys = list(grasp_masks.keys())
    if P.pmax == 0:
        return {k: protos.sum() * 0 for k in keys}
    draws = {key: {} for key in keys}
    for i in P.over_limit():                         # reference loop order: for image: for map
        for key in keys:
            draws[key][i] = torch.randperm(P.counts_host[i])[:P.limit]
    out = {}
    for ch, key in enumerate(keys):
        idx, valid = P.base()
        weight = P.apply_draws(idx, draws[key])
        boxes, gidx = _take(matched_box, idx), _take(matched_idx, idx)
        coef = _take(gcoef_pred[:, :, ch, :], idx)
        target = _take(_resized_targets(grasp_masks[key], ph, pw, protos.device), gidx)
        win = _windows(ph, pw, boxes)
        pred = _mask_maps(protos, coef) * win.to(protos.dtype)
        if key == "cos":
            pred = pred + (~win).to(protos.dtype)      # ones_crop (box_utils.py:174-199)
        per = F.smooth_l1_loss(pred, target, reduction="none").sum(dim=(2, 3))
        per = per / torch.where(valid, _area(boxes), torch.ones_like(per)) * weight[:, None]
        out[key] = cfg.alpha_grasp * torch.where(valid, per, torch.zeros_like(per)).sum() / ph / pw / n_pos
    return out


def semantic_loss(cfg, seg_pred: torch.Tensor, sem_mask: torch.Tensor, labels: Sequence[torch.Tensor]) -> torch.Tensor:
    """ssg.py:512-530: the image's (single-channel) semantic mask, thresholded at 0.5 after bilinear resize, is the target of the
    FIRST label's class plane; every other plane's target is empty."""
    B, C, h, w = seg_pred.shape
    down = F.interpolate(sem_mask.to(seg_pred.device)[:, None].float(), (h, w), mode="bilinear", align_corners=False).gt(0.5).float()[:, 0]
    first = torch.stack([l[0] for l in labels]).to(seg_pred.device).long()
    target = torch.zeros_like(seg_pred, requires_grad=False)
    target[torch.arange(B, device=seg_pred.device), first] = down
    return cfg.alpha_sem * F.binary_cross_entropy_with_logits(seg_pred, target, reduction="sum") / h / w / B


def ssg_loss(cfg, anchors: torch.Tensor, raw: Dict[str, torch.Tensor], data: dict, output_dict: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """SSG.compute_loss (ssg.py:297-350).  `raw`: class_pred [B,A,C], box_pred [B,A,4], ins_coef_pred [B,A,P],
    grasp_coef_pred [B,A,4,P], protos [B,h,w,P], seg_pred [B,C,h',w'].  `data`: bboxes (list of [G,5] corner boxes + class),
    ins_masks (list of [G,H,W]), grasp_masks (dict key -> list of [G,H,W]), sem_mask [B,H,W], labels (list of [G])."""
    offsets, labels, m_box, m_idx = match_batch(cfg, data["bboxes"], anchors)
    pos = labels > 0
    inter = output_dict if getattr(cfg, "intermidiate_output", False) else None
    P = Positives(pos, cfg.masks_to_train)
    # evaluation order as the reference (the two mask terms may draw from the CPU generator)
    l_cls = category_loss(cfg, raw["class_pred"], labels, pos)
    l_box = box_loss(cfg, raw["box_pred"], offsets, pos)
    l_ins = instance_mask_loss(cfg, raw["ins_coef_pred"], raw["protos"], data["ins_masks"], P, m_idx, m_box, inter)
    g = grasp_mask_losses(cfg, raw["grasp_coef_pred"], raw["protos"], data["grasp_masks"], P, m_idx, m_box)
    l_sem = semantic_loss(cfg, raw["seg_pred"], data["sem_mask"], data["labels"])
    return {"loss_cls": l_cls, "loss_box": l_box, "loss_ins": l_ins, "loss_sem": l_sem,
            "loss_qua": g["qua"], "loss_sin": g["sin"], "loss_cos": g["cos"], "loss_wid": g["wid"]}


# ---- detection post-processing, tensor half (utils/grasp_eval.py:54-150; batch size 1 as in the reference) -------------------------------
def corner_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[..., N, 4] x [..., M, 4] corner boxes -> [..., N, M] IoU (box_utils.py:8-37)."""
    lo = torch.max(a[..., :, None, :2], b[..., None, :, :2])
    hi = torch.min(a[..., :, None, 2:], b[..., None, :, 2:])
    wh = (hi - lo).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area_a = ((a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]))[..., :, None]
    area_b = ((b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]))[..., None, :]
    return inter / (area_a + area_b - inter)


@torch.no_grad()
def fast_nms(cfg, boxes, scores, ins_coef, grasp_coef):
    """grasp_eval.py:54-94.  boxes [N,4], scores [classes,N], coefficients [N,P] / [N,4,P] of the score-filtered anchors ->
    (class ids, scores, boxes, instance coefficients, grasp coefficients) of at most cfg.max_detections detections."""
    scores, idx = scores.sort(1, descending=True)
    idx, scores = idx[:, :cfg.top_k], scores[:, :cfg.top_k]
    C, D = idx.shape
    boxes = boxes[idx.reshape(-1)].reshape(C, D, 4)
    ins_coef = ins_coef[idx.reshape(-1)].reshape(C, D, -1)
    grasp_coef = grasp_coef[idx.reshape(-1)].reshape(C, D, 4, -1)
    iou = corner_iou(boxes, boxes).triu_(diagonal=1)
    keep = iou.max(dim=1)[0] <= cfg.nms_iou_thre
    ids = torch.arange(C, device=boxes.device)[:, None].expand_as(keep)[keep]
    scores, boxes, ins_coef, grasp_coef = scores[keep], boxes[keep], ins_coef[keep], grasp_coef[keep]
    scores, order = scores.sort(0, descending=True)
    order, scores = order[:cfg.max_detections], scores[:cfg.max_detections]
    return ids[order], scores, boxes[order], ins_coef[order], grasp_coef[order]


@torch.no_grad()
def ssg_detections(cfg, output_dict: dict, score_floor: float = 0.3):
    """Tensor half of `ssg_post_processing` (grasp_eval.py:99-150,168-186) on the device, for one image: score filter, box decoding,
    fast NMS, final score floor, and the five cropped maps at prototype resolution.  The per-instance resize to the input
    size, Gaussian smoothing and `detect_grasps` that follow in the reference are host post-processing (skimage / numpy)."""
    protos = output_dict["protos"].squeeze(0)
    cls = output_dict["cls_pred"].squeeze(0).transpose(1, 0).contiguous()[1:]          # [classes - 1, A]: background dropped
    box = output_dict["box_pred"].squeeze(0)
    ins_coef = output_dict["ins_coef_pred"].squeeze(0)
    grasp_coef = output_dict["grasp_coef_pred"].squeeze(0)
    anchors = output_dict["anchors"]
    if not torch.is_tensor(anchors):
        anchors = torch.tensor(anchors, device=protos.device)
    anchors = anchors.reshape(-1, 4).to(protos.device)
    keep = cls.max(dim=0)[0] > cfg.nms_score_thre
    a, b = anchors[keep], box[keep]
    dec = torch.cat((a[:, :2] + b[:, :2] * 0.1 * a[:, 2:], a[:, 2:] * torch.exp(b[:, 2:] * 0.2)), 1)
    dec[:, :2] -= dec[:, 2:] / 2
    dec[:, 2:] += dec[:, :2]
    dec = torch.clip(dec, min=0., max=1.)
    ids, scores, boxes, icoef, gcoef = fast_nms(cfg, dec, cls[:, keep], ins_coef[keep], grasp_coef[keep])
    ok = scores > score_floor
    if bool(ok.any()):
        ids, scores, boxes, icoef, gcoef = ids[ok], scores[ok], boxes[ok], icoef[ok], gcoef[ok]
    h, w = protos.shape[:2]
    win = _windows(h, w, boxes[None])[0].to(protos.dtype)                                  # [D, h, w]
    lin = lambda c: torch.einsum("hwk,dk->dhw", protos, c)
    maps = {"ins": torch.sigmoid(lin(icoef)) * win, "qua": torch.sigmoid(lin(gcoef[:, 0])) * win, "sin": lin(gcoef[:, 1]) * win,
            "cos": lin(gcoef[:, 2]) * win, "wid": torch.sigmoid(lin(gcoef[:, 3])) * win}
    return {"cls": ids + 1, "scores": scores, "bboxes": boxes, "maps": maps}
