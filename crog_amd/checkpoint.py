"""Checkpoint wire format of the reference (train_crog.py:206-226 resume, :245-267 save).

A `*.pth` written by the reference is a dict {epoch, cur_iou, best_iou, best_j_index, prec, j_index, state_dict, optimizer,
scheduler}: `state_dict` comes from the DDP-wrapped, SyncBatchNorm-converted model (keys `module.<name>`, 449 parameter
tensors + BatchNorm buffers), `optimizer` from torch.optim.Adam over build_crog's two parameter groups, `scheduler` from
MultiStepLR.  crog_amd keeps parameter names/shapes, group order and Adam's per-parameter state layout identical, so
the same file resumes either implementation; the only thing handled here is the `module.` prefix (present or absent on
either side).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

KEYS = ("epoch", "cur_iou", "best_iou", "best_j_index", "prec", "j_index", "state_dict", "optimizer", "scheduler")


def _match_prefix(state: Dict[str, torch.Tensor], model: torch.nn.Module) -> Dict[str, torch.Tensor]:
    want = next(iter(model.state_dict().keys()), "").startswith("module.")
    have = next(iter(state.keys()), "").startswith("module.")
    if want == have:
        return state
    if have:
        return {k[len("module."):]: v for k, v in state.items()}
    return {"module." + k: v for k, v in state.items()}


def save_checkpoint(path: str, model, optimizer, scheduler, epoch: int, cur_iou: float = 0.0, best_iou: float = 0.0,
                    best_j_index: float = 0.0, prec: Optional[dict] = None, j_index=None) -> Dict[str, Any]:
    """train_crog.py:247-258 (rank 0 calls it; `model` is the DDP wrapper there, so keys carry `module.`)."""
    ckpt = {"epoch": epoch, "cur_iou": cur_iou, "best_iou": best_iou, "best_j_index": best_j_index, "prec": prec or {},
            "j_index": j_index if j_index is not None else [0, 0], "state_dict": model.state_dict(),
            "optimizer": optimizer.state_dict(), "scheduler": scheduler.state_dict() if scheduler is not None else {}}
    torch.save(ckpt, path)
    return ckpt


def load_checkpoint(path: str, model, optimizer=None, scheduler=None, map_location=None, strict: bool = True) -> Dict[str, Any]:
    """train_crog.py:206-219: restores model / optimizer / scheduler in place and returns the bookkeeping entries
    (`epoch` is the next start epoch, as the reference uses it)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    missing = [k for k in ("epoch", "state_dict") if k not in ckpt]
    if missing:
        raise KeyError(f"{path}: not a CROG checkpoint (missing {missing}; expected keys {KEYS})")
    model.load_state_dict(_match_prefix(ckpt["state_dict"], model), strict=strict)
    if optimizer is not None and "optimizer" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
    if scheduler is not None and ckpt.get("scheduler"):
        scheduler.load_state_dict(ckpt["scheduler"])
    return {k: ckpt[k] for k in ("epoch", "cur_iou", "best_iou", "best_j_index", "prec", "j_index") if k in ckpt}
