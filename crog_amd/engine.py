"""Training step / epoch loop of the HIP path — mirror of engine/crog_engine.py::train_with_grasp (:17-122).

Same call signature and per-step semantics (autocast forward, zero_grad, scaled backward, optional clipping,
optimizer step, scaler update, train IoU / Prec@50 metric, rank-averaged scalars), with two changes that do not
alter results: the three scalar all-reduces of crog_engine.py:88-90 are packed into one 3-float all-reduce, and
host reads of device scalars happen once per `print_freq` steps instead of ~8 times per step.
"""
from __future__ import annotations

import os
import time

import torch
import torch.distributed as dist

from . import functional as Fn
from . import kernels as K


class AverageMeter:
    def __init__(self, name, fmt=":f"):
        self.name, self.fmt = name, fmt
        self.val = self.avg = self.sum = self.count = 0.0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / max(self.count, 1)

    def __str__(self):
        return ("{name} {val" + self.fmt + "} ({avg" + self.fmt + "})").format(**self.__dict__)


def train_step(model, optimizer, scaler, batch, args=None, autocast_dtype=torch.bfloat16):
    """One optimisation step (crog_engine.py:60-90). `batch` holds device tensors img, word, mask, qua, sin, cos, wid with the
    masks already [B,1,H,W].  Returns a 3-element device tensor (loss, 100*IoU, 100*Prec@50), rank-averaged."""
    with torch.autocast("cuda", dtype=autocast_dtype, enabled=autocast_dtype is not None):
        pred, target, loss, loss_dict = model(batch["img"], batch["word"], batch["mask"], batch["qua"], batch["sin"], batch["cos"], batch["wid"])
    optimizer.zero_grad()
    if scaler is not None and scaler.is_enabled():
        scaler.scale(loss).backward()
    else:
        loss.backward()
    max_norm = getattr(args, "max_norm", 0.0) if args is not None else 0.0
    if max_norm:
        if scaler is not None and scaler.is_enabled():
            scaler.unscale_(optimizer)
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
    if scaler is not None and scaler.is_enabled():
        scaler.step(optimizer)
        scaler.update()
    else:
        optimizer.step()
    m = Fn.train_metric(pred[0], target[0], 0.35, 0.5)
    stats = torch.stack([loss.detach().float(), m[0], m[1]])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(stats)
        stats = stats / dist.get_world_size()
    return stats, loss_dict


EVAL_SIGMOID_GRASP = 0b10011   # crog_engine.py:182-184: sigmoid on (ins, qua, wid); sin / cos stay raw
EVAL_SIGMOID_MASK_ONLY = 0b1


@torch.no_grad()
def eval_maps(model, batch, autocast_dtype=None):
    """Device part of validate_with_grasp / validate_without_grasp (crog_engine.py:163-211, :348-361): eval forward, sigmoid on the
    probability maps and bicubic (align_corners=True) resize to the input resolution, in ONE kernel over all five maps.
    Returns (list of [B, H, W] fp32 prediction maps, tuple of targets as the model returns them); the per-image cv2 inverse warp,
    `detect_grasps` and the Jacquard index that follow in the reference are host post-processing and stay with the caller."""
    was_training = model.training
    model.eval()
    try:
        with torch.autocast("cuda", dtype=autocast_dtype or torch.bfloat16, enabled=autocast_dtype is not None):
            pred, target = model(batch["img"], batch["word"], batch.get("mask"), batch.get("qua"), batch.get("sin"), batch.get("cos"),
                                 batch.get("wid"))
    finally:
        model.train(was_training)
    H, W = batch["img"].shape[-2:]
    if isinstance(pred, tuple):
        logits = torch.cat([p.float() for p in pred], 1)
        mask = EVAL_SIGMOID_GRASP
    else:
        logits, mask = pred.float(), EVAL_SIGMOID_MASK_ONLY
    if tuple(logits.shape[-2:]) != (H, W):
        maps = K.eval_maps(logits, mask, H, W)
    else:                                   # crog_engine.py:186: no resize when the sizes already agree
        maps = logits.clone()
        for g in range(maps.shape[1]):
            if (mask >> g) & 1:
                maps[:, g].sigmoid_()
    return [maps[:, g] for g in range(maps.shape[1])], target


def mask_iou(pred_map, target, thr=0.35):
    """Per-image IoU of (pred > thr) against a {0,1} target, on the device (crog_engine.py:251-255 without the inverse warp)."""
    p = pred_map > thr
    t = target.reshape(p.shape) > 0.5
    inter = (p & t).flatten(1).sum(1).float()
    union = (p | t).flatten(1).sum(1).float()
    return inter / (union + 1e-6)


# Captured-step replay (crog_amd/graphs.py): "0" = issue every step from Python, "streams" / "1" = csrc/replay.hip on our own streams,
# "hipgraph" = hipGraphLaunch.  Default: on with one rank, off with several (a captured RCCL collective never ran with real peers here).
STEP_GRAPH = os.environ.get("CROG_STEP_GRAPH")
_GRAPHED = {}


def graphed_step_for(model, optimizer, scaler, args, autocast_dtype=torch.bfloat16):
    """The GraphedTrainStep of this (model, optimizer) pair, or None when the step cannot be one graph: a live GradScaler (its
    inf check steers the host), a stock torch optimizer (host-side step count), or CROG_STEP_GRAPH=0."""
    from .optim import FusedAdam
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    want = STEP_GRAPH if STEP_GRAPH is not None else ("0" if multi else "1")
    if want == "0" or not torch.cuda.is_available() or not isinstance(optimizer, FusedAdam):
        return None
    if scaler is not None and scaler.is_enabled():
        return None
    key = (id(model), id(optimizer))
    g = _GRAPHED.get(key)
    if g is None or g.model is not model or g.optimizer is not optimizer:
        from .graphs import GraphedTrainStep
        g = _GRAPHED[key] = GraphedTrainStep(model, optimizer, args, autocast_dtype)
    return g


def train_with_grasp(train_loader, model, optimizer, scheduler, scaler, epoch, args, log=print):
    """Epoch loop with the reference's meters (crog_engine.py:17-122).  With crog_amd's FusedAdam and no live GradScaler (bf16
    autocast needs none) the step is captured once and replayed as a hipGraph; otherwise it is issued eagerly."""
    batch_time, data_time = AverageMeter("Batch", ":2.2f"), AverageMeter("Data", ":2.2f")
    lr, loss_meter = AverageMeter("Lr", ":1.6f"), AverageMeter("Loss", ":2.4f")
    iou_meter, pr_meter = AverageMeter("IoU", ":2.2f"), AverageMeter("Prec@50", ":2.2f")
    model.train()
    end = time.time()
    pending = []
    for i, data in enumerate(train_loader):
        data_time.update(time.time() - end)
        gm = data["grasp_masks"]
        batch = dict(img=data["img"].cuda(non_blocking=True), word=data["word_vec"].cuda(non_blocking=True),
                     mask=data["mask"].cuda(non_blocking=True).unsqueeze(1), qua=gm["qua"].cuda(non_blocking=True).unsqueeze(1),
                     sin=gm["sin"].cuda(non_blocking=True).unsqueeze(1), cos=gm["cos"].cuda(non_blocking=True).unsqueeze(1),
                     wid=gm["wid"].cuda(non_blocking=True).unsqueeze(1))
        graphed = graphed_step_for(model, optimizer, scaler, args)
        if graphed is not None:
            stats, _ = graphed(batch)
        else:
            stats, _ = train_step(model, optimizer, scaler, batch, args)
        pending.append((stats, batch["img"].size(0)))
        lr.update(scheduler.get_last_lr()[-1])
        if (i + 1) % args.print_freq == 0 or i + 1 == len(train_loader):
            for st, n in pending:           # one host sync per print window
                v = st.tolist()
                loss_meter.update(v[0], n)
                iou_meter.update(v[1], n)
                pr_meter.update(v[2], n)
            pending.clear()
            batch_time.update((time.time() - end) / args.print_freq)
            end = time.time()
            log("Training: Epoch=[{}/{}] [{}/{}]  ".format(epoch, args.epochs, i + 1, len(train_loader)) +
                "  ".join(str(m) for m in (batch_time, data_time, lr, loss_meter, iou_meter, pr_meter)))
