"""CROG module on the HIP path — drop-in for the reference's model/crog.py (same constructor keys, same
forward signature and return contract).

Compute dtype: bf16 when called under torch autocast (the reference's training step runs under
`amp.autocast()`, engine/crog_engine.py:72), fp32 otherwise (its eval path, crog_engine.py:166);
`model.compute_dtype = torch.float32 | torch.bfloat16` pins it explicitly.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .. import functional as Fn
from ..runtime import RT, ParamStore
from .blocks import BatchNorm, bind_all
from .clip import arch_from_state_dict, build_model, load_pretrained_clip
from .layers import FPN, MultiTaskProjector, Projector, TransformerDecoder

# Text tower forward/backward as two hipGraph replays (crog_amd/graphs.py).  Opt-in: it removes ~650 launches (~8 ms of host
# time) per step, but the step is GPU-bound on one MI355X (measured 39.8 vs 39.7 ms/step, 43.2 vs 43.4 under DDP), so the eager
# path stays the default until the host becomes the limiter.
TEXT_GRAPH = os.environ.get("CROG_TEXT_GRAPH", "0") == "1"
TEXT_AFTER = os.environ.get("CROG_TEXT_AFTER", "0") == "1"
TEXT_FREE = os.environ.get("CROG_PROBE_TEXT_FREE") == "1"      # TIMING PROBE, wrong training: the text tower runs once, its outputs are reused as constants
GATE_ON_TEXT = os.environ.get("CROG_GATE_ON_TEXT", "1") != "0"      # the neck's sentence gate runs on the text stream (single-GPU step)
DGW_LATE = os.environ.get("CROG_DGW_LATE", "1") != "0"      # the data-gradient weight copies are refreshed beside the neck, not beside the stem

RN50_ARCH = dict(embed_dim=1024, image_resolution=224, vision_layers=(3, 4, 6, 3), vision_width=64, vision_patch_size=None,
                 context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12)


class LossDict(dict):
    """loss_dict of crog.py:101-106 with ONE device->host sync on first access instead of five `.item()` calls."""

    KEYS = ("m_ins", "m_qua", "m_sin", "m_cos", "m_wid")

    def __init__(self, sums: torch.Tensor, heads: int):
        super().__init__()
        self._sums, self._heads = sums, heads
        self._done = False

    def _fill(self):
        if not self._done:
            vals = self._sums.tolist()
            for i, k in enumerate(self.KEYS):
                dict.__setitem__(self, k, vals[i] if i < self._heads else 0)
            self._done = True

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        return 5

    def __repr__(self):
        self._fill()
        return dict.__repr__(self)


class CROG(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.use_contrastive = cfg.use_contrastive
        self.use_pretrained_clip = cfg.use_pretrained_clip
        self.use_grasp_masks = cfg.use_grasp_masks
        arch, sd = self._clip_arch(cfg)
        self._clip_source = str(getattr(cfg, "clip_pretrain", "")) if sd is not None else "cfg.clip_arch"
        self.backbone = build_model(arch, cfg.word_len)
        if sd is not None and self.use_pretrained_clip:
            load_pretrained_clip(self.backbone, sd)      # clip.py:551-554
        self.neck = FPN(in_channels=cfg.fpn_in, out_channels=cfg.fpn_out)
        if self.use_contrastive:
            self.decoder = TransformerDecoder(num_layers=cfg.num_layers, d_model=cfg.vis_dim, nhead=cfg.num_head, dim_ffn=cfg.dim_ffn,
                                              dropout=cfg.dropout, return_intermediate=cfg.intermediate)
        if self.use_grasp_masks:
            self.proj = MultiTaskProjector(cfg.word_dim, cfg.vis_dim // 2, 3)
        else:
            self.proj = Projector(cfg.word_dim, cfg.vis_dim // 2, 3)
        self.compute_dtype = None
        self._store = None
        self._bns = None
        self.overlap_text = True
        self._side = None
        self._graphs = {}
        self.explicit_grad_ready = True  # kernels write gradients in place and call RT.reducer.mark_ready themselves

    @staticmethod
    def _clip_arch(cfg):
        """cfg.clip_pretrain names a TorchScript CLIP archive (crog.py:20-23).  When it exists, the architecture is
        inferred from its tensor shapes; otherwise `cfg.clip_arch` (or RN50) describes it and weights are random-init."""
        import os
        path = getattr(cfg, "clip_pretrain", None)
        if path and os.path.exists(str(path)):
            sd = torch.jit.load(path, map_location="cpu").eval().state_dict()
            return arch_from_state_dict(sd), sd
        return dict(getattr(cfg, "clip_arch", None) or RN50_ARCH), None

    # ---- storage ---------------------------------------------------------------------------------
    def prepare(self, device=None):
        """Move parameters into the flat store on `device` and bind the kernels' weight references."""
        if device is None:
            device = next(self.parameters()).device
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("crog_amd.CROG runs on an MI355X only: there is no CPU path (move the module with .cuda())")
        if hasattr(self.backbone.visual, "check_supported"):
            self.backbone.visual.check_supported(getattr(self, "_clip_source", ""))
        for b_name, buf in list(self.named_buffers()):
            if buf.device != device:
                mod = self
                *path, leaf = b_name.split(".")
                for p in path:
                    mod = getattr(mod, p)
                mod._buffers[leaf] = buf.to(device)
        RT.ensure_streams(device)        # before any communicator creates streams of its own (crog_amd/runtime.py)
        self._store = ParamStore(self, device)
        self._store.explicit = True
        bind_all(self, self._store)
        self._bns = [m for m in self.modules() if isinstance(m, BatchNorm)]
        return self

    def _ensure(self, device):
        if self._store is None or not self._store.valid() or self._store.device != device:
            self.prepare(device)

    @property
    def store(self) -> ParamStore:
        return self._store

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._store = None  # .cuda()/.float()/.to() re-materialise parameters: rebuild the flat store lazily
        return out

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        if self._store is not None:
            self._store.invalidate_shadow()
        return out

    def _text_graph(self, word, dtype):
        key = (tuple(word.shape), dtype, id(self._store))
        if self._graphs.get("key") != key:
            from ..graphs import GraphedTower
            self._graphs = {"key": key, "text": GraphedTower(lambda w: self.backbone.text_features(w, dtype), word, self._side)}
        return self._graphs["text"]

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, img, word, mask=None, grasp_qua_mask=None, grasp_sin_mask=None, grasp_cos_mask=None, grasp_wid_mask=None):
        """img: [b,3,h,w] fp32; word: [b,L] int64; masks: [b,1,h,w] fp32 (crog.py:47)."""
        dev = img.device
        self._ensure(dev)
        dtype = self.compute_dtype or (torch.bfloat16 if torch.is_autocast_enabled() else torch.float32)
        store = self._store
        store.forward_begins()              # re-cast the bf16 shadow unless FusedAdam just wrote it
        if self.training and torch.is_grad_enabled():
            RT.join_streams()               # e.g. a previous backward's weight-gradient stream when no fused optimizer joined it
            store.fresh_grads_if_dropped()   # gradients ACCUMULATE across forward/backward pairs until zero_grad(), as in torch
            RT.begin_step(dev)
        with torch.autocast("cuda", enabled=False):
            pad_mask = (word == 0).contiguous()
            # The text tower (640 token rows: ~100 latency-bound launches) is independent of the image tower until the
            # neck, so it runs on a second HIP stream and overlaps the 208x208 / 104x104 convolutions; autograd replays each
            # node on its forward stream, which overlaps the two backward passes the same way.
            main = torch.cuda.current_stream()
            store.weights(dtype)               # refresh the bf16 shadow on the main stream BEFORE the streams fork
            t_side = refresh_t = None
            if self.training and torch.is_grad_enabled():
                # ... and the data-gradient copies of the weights (one launch over all of them, ~0.3 ms of HBM traffic).  Backward reads
                # them on both streams, the forward not at all: the launch goes to the weight-gradient stream, idle during the forward,
                # and the main stream joins it at the end of this forward - ahead of every backward node on any stream
                # (issued AFTER the image tower below: the side stream then waits for the tower, and the refresh - pure HBM traffic - runs
                # beside the neck's MFMA-bound 3x3 convolutions instead of beside the HBM-bound stem, whose kernels it slowed threefold)
                t_side = RT.wgrad_stream()
                store.ensure_t(dtype)
                # (not with the text tower as hipGraphs: their capture reads the copies, and a refresh captured INTO the text graph would re-run every step)
                refresh_t = (lambda: RT._issue_wgrad(lambda: store.weights_t(dtype), ())) if (DGW_LATE and not TEXT_GRAPH) else None
                if refresh_t is None:
                    RT._issue_wgrad(lambda: store.weights_t(dtype), ())
            # (deterministic mode keeps the side streams since round 5: the run-to-run differences of rounds 3-4 were packed-fp32 VALU results
            # going wrong beside another stream's MFMA kernel - runtime.set_deterministic, LAB_NOTES section 10; RT.det_streams = "0" restores one stream)
            overlap_text = self.overlap_text and (not RT.deterministic or RT.det_streams in ("all", "text"))
            graphed = gate = hword = dtext = None
            if overlap_text:
                if self._side is None:
                    RT.ensure_streams(dev)      # creation ORDER of the side streams decides which hardware queues they share
                    self._side = RT.text_stream if RT.text_stream is not None else torch.cuda.Stream(device=dev)
                RT.streams = [main, self._side]
                self._side.wait_stream(main)
                graphed = self._text_graph(word, dtype) if (TEXT_GRAPH and self.training and torch.is_grad_enabled()) else None
            if overlap_text and graphed is not None:
                # both passes of the text tower are one hipGraph replay each (crog_amd/graphs.py): issued up front, it runs
                # beside the image stem without costing the host ~650 launches per step
                with torch.cuda.stream(self._side):
                    wfeat, state = graphed(self.backbone.tok.param, word)
                vis = self.backbone.image_features(img, dtype)
                main.wait_stream(self._side)
            elif overlap_text:
                txt = []
                steps = self.backbone.text_features_steps(word, dtype, parts=3)

                def issue_text():   # called by the image tower after layer1 / layer2 / layer3 are enqueued (host issue order only)
                    with torch.cuda.stream(self._side):
                        r = next(steps, None)
                        if r is not None:
                            txt.extend(r)
                # (CROG_TEXT_AFTER=1, A/B: the whole text tower issued AFTER the image tower - its autograd nodes then run, host-side, before
                # the image tower's backward, so its Adam chunks ripen early in the weight-gradient / aux stream's queue)
                frozen = getattr(self, "_probe_text", None) if TEXT_FREE else None
                vis = self.backbone.image_features(img, dtype, None if (TEXT_AFTER or frozen is not None) else issue_text)
                if frozen is not None:
                    txt = list(frozen)
                while not txt:          # whatever the image tower's hooks did not get to
                    issue_text()
                wfeat, state = txt
                if TEXT_FREE and frozen is None:
                    self._probe_text = (wfeat.detach().clone(), state.detach().clone())
                # the neck's sentence gate (layers.py:376) on the text stream, behind the tower: it needs nothing of the image side.  Not
                # under SyncBatchNorm: its BatchNorm1d would exchange statistics on a second stream, and the mailbox exchanges of one
                # communicator are sequenced on ONE stream (csrc/comm.hip).
                if GATE_ON_TEXT and RT.comm is None and hasattr(self.neck, "text_gate"):
                    with torch.cuda.stream(self._side):
                        gate = self.neck.text_gate(state)
                # ... and the dynamic head's per-sample kernel txt(state) (layers.py:90-91; a plain linear: also under SyncBatchNorm)
                if GATE_ON_TEXT and hasattr(self.proj, "text_word"):
                    with torch.cuda.stream(self._side):
                        hword = self.proj.text_word(state)
                # ... and the decoder's text side: word features + positions, every layer's cross-attention key / value projections
                if GATE_ON_TEXT and self.use_contrastive and self.training and torch.is_grad_enabled() and hasattr(self.decoder, "text_kv"):
                    with torch.cuda.stream(self._side):
                        dtext = self.decoder.text_kv(wfeat)
                main.wait_stream(self._side)
                wfeat.record_stream(main)
                state.record_stream(main)
                if gate is not None:
                    gate.record_stream(main)
                if hword is not None:
                    hword.record_stream(main)
                if dtext is not None:
                    for t_ in (dtext[0], dtext[1]) + tuple(kv[0] for kv in dtext[2]):
                        t_.record_stream(main)
            else:
                vis = self.backbone.image_features(img, dtype)
                wfeat, state = self.backbone.text_features(word, dtype)
            if refresh_t is not None:
                refresh_t()
            if t_side is not None:
                store.zero_early()      # (engine.train_step: the gradient memset, also beside the neck)
            if not isinstance(vis, tuple):
                # layers.py:373 unpacks three pyramid levels; a ViT tower returns one token tensor, so the reference fails here
                # too: ViT parity is encoder-level (encode_image / encode_text).
                raise ValueError("too many values to unpack (expected 3): the FPN neck needs the (C3, C4, C5) maps of the "
                                 "ModifiedResNet tower; with a ViT tower use backbone.encode_image / encode_text")
            fan = getattr(self.backbone.visual, "fan", None)
            self.backbone.visual.fan = None
            fq = self.neck(vis, state, fan=fan, gate=gate)
            if self.use_contrastive:
                fq = self.decoder(fq, wfeat, pad_mask, text=dtext)
                if isinstance(fq, list):
                    # cfg.intermediate=True: the reference's decoder returns a list and crog.py:69 then calls .reshape on it
                    raise AttributeError("'list' object has no attribute 'reshape' (TransformerDecoder(return_intermediate=True) "
                                         "returns per-layer outputs; CROG.forward consumes a single map, as in the reference)")
            pred = self.proj(fq, state, word=hword, word_stream=self._side if hword is not None else None)      # fp32 logits [b, groups, H, W]
            if t_side is not None:
                main.wait_stream(t_side)
            if self.training and self._bns:
                torch._foreach_add_([m.num_batches_tracked for m in self._bns], 1)
            n = 5 if self.use_grasp_masks else 1
            preds = tuple(pred[:, i:i + 1].detach() for i in range(n))
            targets = (mask, grasp_qua_mask, grasp_sin_mask, grasp_cos_mask, grasp_wid_mask)[:n]
            if self.training:
                total, sums, small = Fn.head_loss(pred, [t.float() for t in targets], weighted=self.use_grasp_masks)
                total = Fn.backward_begin(store, total)
                loss_dict = LossDict(sums, n)
                tgt = tuple(small[i] for i in range(n))
                if self.use_grasp_masks:
                    return preds, tgt, total, loss_dict
                return (preds[0], None, None, None, None), (tgt[0], None, None, None, None), total, loss_dict
            if self.use_grasp_masks:
                return preds, tuple(targets)
            return preds[0], mask
