"""`from crog_amd.model import build_crog` — the reference's model/__init__.py boundary."""
from .crog import CROG


def build_crog(args):
    """model/__init__.py:6-23: returns (model, [backbone group, head group]).  Parameters whose name starts with
    `backbone` and does not contain `positional_embedding` get `initial_lr = lr_multi * base_lr`, the rest `base_lr`."""
    model = CROG(args)
    backbone, head = [], []
    for k, v in model.named_parameters():
        if k.startswith("backbone") and "positional_embedding" not in k:
            backbone.append(v)
        else:
            head.append(v)
    param_list = [{"params": backbone, "initial_lr": args.lr_multi * args.base_lr},
                  {"params": head, "initial_lr": args.base_lr}]
    return model, param_list


def build_ssg(args):
    """model/__init__.py:26-29: returns (model, model.parameters()) - one parameter group, no learning-rate split."""
    from .ssg import SSG
    model = SSG(args)
    return model, model.parameters()


__all__ = ["CROG", "build_crog", "build_ssg"]
