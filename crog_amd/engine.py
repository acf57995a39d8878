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


# FusedAdam steps the chunks of the flat parameter buffer whose gradients are final INSIDE backward (optim.py: overlap_backward), when
# the step has no gradient clipping (max_norm: 0 in config/OCID-VLG/crog_multiple_r50.yaml:35) and no live GradScaler.  CROG_ADAM_OVERLAP=0: off.
ADAM_OVERLAP = os.environ.get("CROG_ADAM_OVERLAP", "1") != "0"


EARLY_ZERO = os.environ.get("CROG_EARLY_ZERO", "1") != "0"


def train_step(model, optimizer, scaler, batch, args=None, autocast_dtype=torch.bfloat16):
    """One optimisation step (crog_engine.py:60-90). `batch` holds device tensors img, word, mask, qua, sin, cos, wid with the
    masks already [B,1,H,W].  Returns a 3-element device tensor (loss, 100*IoU, 100*Prec@50), rank-averaged."""
    store = getattr(optimizer, "_store", None)
    if store is not None and EARLY_ZERO:
        # this step zeroes the gradients between forward and backward (below, crog_engine.py:77): a model that forks a side stream in its
        # forward may do the memset there, beside its MFMA-bound layers (588 MB of writes: 74 us of the main chain otherwise)
        store.request_zero()
    with torch.autocast("cuda", dtype=autocast_dtype, enabled=autocast_dtype is not None):
        pred, target, loss, loss_dict = model(batch["img"], batch["word"], batch["mask"], batch["qua"], batch["sin"], batch["cos"], batch["wid"])
    # (crog_engine.py:84 computes the batch metric after the optimizer step; it reads the forward's outputs only, so it is enqueued HERE,
    # beside the start of backward, instead of behind the last weight gradient at the very end of the step: 65 us of the step's tail)
    m = Fn.train_metric_beside(pred[0], target[0], 0.35, 0.5)
    optimizer.zero_grad()
    max_norm = getattr(args, "max_norm", 0.0) if args is not None else 0.0
    if scaler is not None and scaler.is_enabled():
        scaler.scale(loss).backward()
    else:
        if ADAM_OVERLAP and not max_norm and hasattr(optimizer, "overlap_backward"):
            optimizer.overlap_backward()      # nothing stands between backward and step: finished chunks are stepped inside backward
        loss.backward()
    if max_norm:
        if scaler is not None and scaler.is_enabled():
            scaler.unscale_(optimizer)
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
    if scaler is not None and scaler.is_enabled():
        scaler.step(optimizer)
        scaler.update()
    else:
        optimizer.step()
    stats = torch.stack([loss.detach().float(), m[0], m[1]])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        from .parallel import all_reduce_mean_
        all_reduce_mean_(stats, model)
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


def _reference_host_half():
    """The host post-processing of validate_with_grasp is the reference's own code (out of scope here): cv2's
    inverse warp and utils/grasp_eval.py's detect_grasps / calculate_jacquard_index.  In a deployment they are importable - the
    reference's `utils` package is on sys.path next to train_crog.py, cv2 and skimage are its dependencies; this image has neither."""
    try:
        import cv2
        from utils.grasp_eval import calculate_jacquard_index, detect_grasps
    except ImportError as e:
        raise ImportError("crog_amd.engine.validate_with_grasp: the host half (cv2.warpAffine, utils.grasp_eval.detect_grasps / "
                          "calculate_jacquard_index of the reference) is not importable here; pass inverse= / detect= / jacquard= callables") from e

    def inverse(img, mat, w, h):          # crog_engine.py:127-131
        return cv2.warpAffine(img, mat, (w, h), flags=cv2.INTER_CUBIC, borderValue=0.)
    return inverse, detect_grasps, calculate_jacquard_index


@torch.no_grad()
def validate_with_grasp(val_loader, model, epoch, args, inverse=None, detect=None, jacquard=None, log=print, autocast_dtype=None):
    """Drop-in for engine/crog_engine.py:125-285 (same arguments, same return `(iou, prec, J_index)`, same log line).
    Device half on the HIP path: eval forward + sigmoid + bicubic align_corners resize of all five maps in one kernel (`eval_maps`,
    crog_engine.py:163-211).  Host half, per sample as in the reference (:214-261): inverse warp to the original size, IoU of the
    thresholded instance mask, grasp detection and Jacquard index for 1 and 5 grasps - `inverse(img, mat, w, h)`,
    `detect(qua, sin, cos, wid, n) -> (grasps, _)`, `jacquard(grasps, targets) -> 0/1` default to the reference's own cv2 /
    utils.grasp_eval functions.  The per-sample IoUs are gathered over the ranks (utils/misc.py:46-59) before Pr@50..90."""
    import numpy as np
    if inverse is None or detect is None or jacquard is None:
        r_inv, r_det, r_jac = _reference_host_half()
        inverse, detect, jacquard = inverse or r_inv, detect or r_det, jacquard or r_jac
    model.eval()
    num_grasps = [1, 5]
    correct, total = [0, 0], [0, 0]
    iou_list = []
    dev = None
    for data in val_loader:
        gm = data["grasp_masks"]
        batch = dict(img=data["img"].cuda(non_blocking=True), word=data["word_vec"].cuda(non_blocking=True),
                     mask=data["mask"].cuda(non_blocking=True).unsqueeze(1), qua=gm["qua"].cuda(non_blocking=True).unsqueeze(1),
                     sin=gm["sin"].cuda(non_blocking=True).unsqueeze(1), cos=gm["cos"].cuda(non_blocking=True).unsqueeze(1),
                     wid=gm["wid"].cuda(non_blocking=True).unsqueeze(1))
        dev = batch["img"].device
        maps, targets = eval_maps(model, batch, autocast_dtype=autocast_dtype)
        preds_np = [m.cpu().numpy() for m in maps]                             # one device -> host copy per map, not per sample
        tgts_np = [t.squeeze(1).float().cpu().numpy() for t in targets]
        for idx in range(preds_np[0].shape[0]):
            inv_mat = data["inverse"][idx]
            h, w = data["ori_size"][idx]
            h, w = int(h), int(w)
            p = [inverse(np.ascontiguousarray(m[idx]), inv_mat, w, h) for m in preds_np]
            t_ins = inverse(np.ascontiguousarray(tgts_np[0][idx]), inv_mat, w, h)
            ins = p[0] > 0.35
            inter, union = np.logical_and(ins, t_ins), np.logical_or(ins, t_ins)
            iou_list.append(np.sum(inter) / (np.sum(union) + 1e-6))
            for i, n in enumerate(num_grasps):
                grasps, _ = detect(p[1], p[2], p[3], p[4], n)
                correct[i] += jacquard(grasps, data["grasps"][idx])
                total[i] += 1
    J_index = [correct[i] / max(total[i], 1) for i in range(2)]
    ious = torch.from_numpy(np.stack(iou_list)).to(dev)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:       # concat_all_gather, utils/misc.py:46-59
        parts = [torch.ones_like(ious) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, ious, async_op=False)
        ious = torch.cat(parts, 0)
    prec = {}
    temp = "  "
    for i, thres in enumerate(torch.arange(0.5, 1.0, 0.1)):          # the reference's thresholds, float32 as it forms them (:271)
        value = (ious > thres).float().mean().item()
        prec["Pr@{}".format((5 + i) * 10)] = value
        temp += "{}: {:.2f}  ".format("Pr@{}".format((5 + i) * 10), 100. * value)
    iou = ious.mean().item()
    log("Evaluation: Epoch=[{}/{}]  IoU={:.2f}  J_index@1: {:.2f}  J_index@5: {:.2f}".format(
        epoch, args.epochs, 100. * iou, 100. * J_index[0], 100. * J_index[1]) + temp)
    return iou, prec, J_index


def mask_iou(pred_map, target, thr=0.35):
    """Per-image IoU of (pred > thr) against a {0,1} target, on the device (crog_engine.py:251-255 without the inverse warp)."""
    p = pred_map > thr
    t = target.reshape(p.shape) > 0.5
    inter = (p & t).flatten(1).sum(1).float()
    union = (p | t).flatten(1).sum(1).float()
    return inter / (union + 1e-6)


# Captured-step replay (crog_amd/graphs.py): "0" = issue every step from Python, "streams" / "1" = csrc/replay.hip on our own streams,
# "hipgraph" = hipGraphLaunch.  Default: on - with several ranks when every collective of the step goes through the C-ABI communicators
# (parallel.step_is_capturable: plain RCCL / mailbox launches on captured streams, no torch ProcessGroup inside the step), and then the
# first replay is checked against an eager step from the same state before it is trusted (GraphedTrainStep verify=).
STEP_GRAPH = os.environ.get("CROG_STEP_GRAPH")


def graphed_step_for(model, optimizer, scaler, args, autocast_dtype=torch.bfloat16):
    """The GraphedTrainStep of this (model, optimizer) pair, or None when the step cannot be one graph: a live GradScaler (its
    inf check steers the host), a stock torch optimizer (host-side step count), or CROG_STEP_GRAPH=0."""
    from .optim import FusedAdam
    from .parallel import step_is_capturable
    from .runtime import RT
    multi = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or RT.comm is not None or hasattr(model, "reducer"))
    want = STEP_GRAPH if STEP_GRAPH is not None else ("1" if (not multi or step_is_capturable(model)) else "0")
    if want == "0" or not torch.cuda.is_available() or not isinstance(optimizer, FusedAdam):
        return None
    if scaler is not None and scaler.is_enabled():
        return None
    # the captured step (its graph, replay handle and private memory pool) lives ON the optimizer object, one per model it has been
    # asked to step: it is released with the optimizer instead of staying in a module-level table for the life of the process
    table = optimizer.__dict__.setdefault("_crog_graphed", {})
    g = table.get(id(model))
    if g is None or g.model is not model:
        from .graphs import GraphedTrainStep
        g = table[id(model)] = GraphedTrainStep(model, optimizer, args, autocast_dtype, verify=multi)
    return g


def release_graphed_step(model, optimizer):
    """Drop the captured step of this pair now (its hipGraph, the replay object's events and the graph's memory pool)."""
    g = optimizer.__dict__.get("_crog_graphed", {}).pop(id(model), None)
    if g is not None:
        g.release()


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
            from .runtime import RT
            if RT.comm is not None:         # a SyncBatchNorm exchange that gave up on a peer must stop the run here, not train on
                RT.comm.check()
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
