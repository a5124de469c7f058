"""``RetinaNetLosses`` with the reference's surface (``retinanet/losses.py:11-145``).

``forward`` / ``calc_loss`` run as two HIP launches for the whole batch -- K2
``rn_iou_match`` and K3 ``rn_loss_fwd_bwd`` -- instead of the reference's per-image
Python loop of ~40 torch ops with 4 host syncs each (losses.py:126, :66-97).  K3
writes the gradients in the same pass that computes the loss values, so
``backward`` only has to apply the upstream scalar (a device-side no-op when it
is 1, the ``loss.backward()`` case).
"""
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import ops
from .config import (BBOX_REG_WEIGHTS, ENCODE_LOG_EPS, FOCAL_LOSS_ALPHA, FOCAL_LOSS_GAMMA,
                     IOU_THRESHOLDS_BACKGROUND, IOU_THRESHOLDS_FOREGROUND, LOGIT_SHIFT, SMOOTH_L1_LOSS_BETA)


class _FusedDenseHeadLoss(torch.autograd.Function):
    """(cls [B,A,K], box [B,A,4]) -> f32[2] = (classification_loss, regression_loss)."""

    @staticmethod
    def forward(ctx, cls, box, anchors, gt_boxes, gt_labels, gt_off, params, fg_thr, bg_thr):
        B = cls.shape[0]
        matches, num_fg = ops.iou_match(anchors, gt_boxes, gt_off, B, fg_thr, bg_thr)
        want_grad = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        loss, gcls, gbox = ops.loss_fwd_bwd(cls, box, anchors, gt_boxes, gt_labels, gt_off, matches, num_fg,
                                            params, want_grad)
        ctx.gcls, ctx.gbox = gcls, gbox
        ctx.box_dtype, ctx.cls_shape, ctx.box_shape = box.dtype, cls.shape, box.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        if ctx.gcls is None:
            raise RuntimeError("fused RetinaNet loss: backward called twice (or without gradients recorded); "
                               "re-run the forward pass instead of retain_graph=True")
        gcls, gbox = ctx.gcls, ctx.gbox
        ctx.gcls = ctx.gbox = None
        g = g.to(torch.float32)
        ops.scale_inplace(gcls, g[0:1])
        ops.scale_inplace(gbox, g[1:2])
        if gbox.dtype != ctx.box_dtype:
            gbox = gbox.to(ctx.box_dtype)
        return gcls.view(ctx.cls_shape), gbox.view(ctx.box_shape), None, None, None, None, None, None, None


class GtPack:
    "What ``ops.gt_pack`` prepared beside the GT arrays: the zeroed ``num_fg`` K2 adds into (None: K2 clears its own)."
    __slots__ = ("num_fg",)

    def __init__(self, num_fg):
        self.num_fg = num_fg


class MatchAhead:
    """K2's results, launched ahead of the head convolutions on a side stream (``RetinaNetLosses.match_ahead``): the matcher needs
    only the anchors and the GT boxes, so it does not have to sit on the critical path between the class-output conv and K3."""
    __slots__ = ("gt_boxes", "gt_labels", "gt_off", "matches", "num_fg", "special", "done", "anchors")


_SIDE_STREAMS = {}
# K2 inside K3 (rn_loss_match_fwd_bwd_levels) is OPT-IN: bit-identical results, but measured SLOWER than the two launches on MI355X
# (B = 8, A = 201 600, T = 8, isolated: 201 us against 154 us for K2 + K3).  num_fg[b] normalises every gradient of image b, so the
# fused kernel needs a grid barrier between its matching prologue and the stream; ablations on one box: the barrier alone +23 us (it
# exposes the launch ramp of the 1 536 workgroups, which the plain kernel hides under the early workgroups' streaming), the matching
# pass alone +15 us when other waves' streaming covers it and ~45 us when everything waits behind the barrier.
FUSE_MATCH = os.environ.get("RN_FUSE_MATCH", "0") == "1"
# The two batch losses from the streaming kernel itself instead of a one-block finalize launch behind it (rn_loss_fwd_bwd_levels_fin:
# fixed-point partial sums through device-scope atomics, workgroup 0 waits for the last arrival): same bits, one dependent launch less
IN_KERNEL_FINALIZE = True
# How K3 repairs its special rows (``ops.LOSS_FORM_*``, rn_loss_fwd_bwd_levels_rp).  "auto": the chunk form at the train shape, the
# list form from K3_LIST_MIN_GT boxes per image on (same-box A/B on MI355X, round 6, isolated graph replays, one-launch forms: T = 8
# 116.8 us chunks / 119.0 list; T = 64 131.7 / 127.9; T = 500 fp16 149.3 / 143.2).  The two-launch form (a pure background stream + a
# repair kernel over K2's flag words, one special row per lane) is correct and slower -- 145 / 170 / 198 us at T = 8 / 64 / 500: the
# repair's dependent loads have nothing to hide under once they leave the streaming kernel -- and stays for A/B: K3_FORM = 1.
K3_FORM = "auto"
K3_LIST_MIN_GT = 32


def k3_form(total_gt: int, B: int) -> int:
    if K3_FORM != "auto":
        return int(K3_FORM)
    return ops.LOSS_FORM_LIST if total_gt >= K3_LIST_MIN_GT * max(B, 1) else ops.LOSS_FORM_CHUNKS


# Gradient pre-scale (fp16 training): a device f32 scalar -- a torch.amp.GradScaler's ``_scale`` -- that K3 multiplies into every
# gradient BEFORE rounding it to fp16.  K3 writes d loss / d logits in the forward pass; a background element at the prior has
# 0.25 p^3 / (num_fg B) ~ 4e-10, below fp16's smallest subnormal (6e-8): stored unscaled it is zero whatever backward multiplies in
# later.  The reference's native-AMP run scales the fp32 loss first and keeps them (~2.6e-5 at a scale of 65 536).  backward then
# multiplies by upstream / prescale (exactly 1 under ``scaler.scale(loss).backward()``).
_GRAD_PRESCALE = None


class grad_prescale:
    "``with grad_prescale(scaler_scale_tensor):`` -- fused loss calls inside the block pre-scale their gradients by it."

    def __init__(self, scale: Optional[Tensor]):
        self.scale = scale

    def __enter__(self):
        global _GRAD_PRESCALE
        self.prev, _GRAD_PRESCALE = _GRAD_PRESCALE, self.scale
        return self

    def __exit__(self, *exc):
        global _GRAD_PRESCALE
        _GRAD_PRESCALE = self.prev
        return False


def scaler_prescale(scaler, dev) -> Optional[Tensor]:
    "The device scalar of a ``torch.amp.GradScaler`` (created on first use), or None when there is no enabled scaler."
    if scaler is None or not scaler.is_enabled():
        return None
    if getattr(scaler, "_scale", None) is None:
        scaler._lazy_init_scale_growth_tracker(torch.device(dev))
    return scaler._scale


def _side_stream(dev: torch.device) -> "torch.cuda.Stream":
    s = _SIDE_STREAMS.get(dev.index)
    if s is None:
        s = _SIDE_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
    return s


class _FusedDenseHeadLossLevels(torch.autograd.Function):
    """Per-level head outputs (cls_0..cls_{L-1}, box_0..box_{L-1}) -> f32[2]; no concatenation."""

    @staticmethod
    def forward(ctx, anchors, gt_boxes, gt_labels, gt_off, params, fg_thr, bg_thr, L, ahead, *levels):
        cls_levels, box_levels = levels[:L], levels[L:]
        B = cls_levels[0].shape[0]
        want_grad = any(ctx.needs_input_grad[9:])
        fused = None
        pack = None
        if isinstance(ahead, GtPack):
            pack, ahead = ahead, None
        if ahead is not None and not isinstance(ahead, MatchAhead):
            # ``ahead`` = the largest per-image GT count: K2 runs INSIDE the loss kernel (one launch; box_utils.py:51-80 in the
            # prologue of rn_loss_match_fwd_bwd_levels) unless the library declines the shape
            fused = ops.loss_match_fwd_bwd_levels(cls_levels, box_levels, anchors, gt_boxes, gt_labels, gt_off, int(ahead), fg_thr,
                                                  bg_thr, params, want_grad)
            ahead = None
        ctx.prescale = None
        if fused is not None:
            loss, gcls, gbox = fused[0], fused[1], fused[2]
        else:
            if ahead is not None:
                torch.cuda.current_stream(cls_levels[0].device).wait_event(ahead.done)        # K2 ran beside the head convolutions
                matches, num_fg, special = ahead.matches, ahead.num_fg, ahead.special
            else:
                # `matches` goes nowhere but into the loss kernel, which reads it through the flag words: K2 writes the flagged rows only
                matches, num_fg, special = ops.iou_match(anchors, gt_boxes, gt_off, B, fg_thr, bg_thr, want_special=True, flagged_only=True,
                                                         zeroed_num_fg=pack.num_fg if pack is not None else None)
            pre = _GRAD_PRESCALE if (want_grad and _GRAD_PRESCALE is not None and _GRAD_PRESCALE.device == cls_levels[0].device) else None
            loss, gcls, gbox = ops.loss_fwd_bwd_levels(cls_levels, box_levels, anchors, gt_boxes, gt_labels, gt_off, matches,
                                                       num_fg, params, want_grad, special=special, in_kernel_finalize=IN_KERNEL_FINALIZE,
                                                       grad_prescale=pre, form=k3_form(int(gt_boxes.shape[0]), B) if special is not None else None)
            ctx.prescale = pre
        ctx.grads = (gcls, gbox)
        ctx.meta = [(c.shape, c.dtype) for c in cls_levels] + [(b.shape, b.dtype) for b in box_levels]
        # two scalar outputs (views of the kernel's f32[2]): backward then receives the two upstream scalars directly, without
        # autograd's select-backward zeros / copies / add in front of the first gradient kernel
        return loss[0], loss[1]

    @staticmethod
    def backward(ctx, g0, g1):
        if ctx.grads is None or ctx.grads[0] is None:
            raise RuntimeError("fused RetinaNet loss: backward called twice (or without gradients recorded); "
                               "re-run the forward pass instead of retain_graph=True")
        gcls, gbox = ctx.grads
        ctx.grads = None
        dev = gcls[0].device
        g0 = torch.full((1,), 0.0, device=dev) if g0 is None else g0.reshape(1)
        g1 = torch.full((1,), 0.0, device=dev) if g1 is None else g1.reshape(1)
        pre = getattr(ctx, "prescale", None)
        if pre is not None:                 # the gradients already carry the pre-scale: what is left is upstream / prescale (1 under a GradScaler)
            g0, g1 = g0.float() / pre.reshape(1), g1.float() / pre.reshape(1)
        ops.scale_inplace_batched(list(gcls) + list(gbox), [g0] * len(gcls) + [g1] * len(gbox))      # one launch
        outs = [t.view(shape) if t.dtype == dt else t.to(dt).view(shape) for t, (shape, dt) in zip(list(gcls) + list(gbox), ctx.meta)]
        return (None,) * 9 + tuple(outs)


def _stack_anchors(anchors) -> Tensor:
    """List of per-image [A,4] tensors -> one shared [A,4] (the usual case: the
    generator hands out the same cached tensor per image) or a stacked [B,A,4]."""
    if isinstance(anchors, Tensor):
        return anchors
    first = anchors[0]
    if all(a is first or a.data_ptr() == first.data_ptr() for a in anchors):
        return first
    return torch.stack(list(anchors))


class RetinaNetLosses(nn.Module):
    def __init__(self, num_classes: int) -> None:
        super().__init__()
        self.n_c = num_classes
        self.alpha = FOCAL_LOSS_ALPHA
        self.gamma = FOCAL_LOSS_GAMMA
        self.beta = SMOOTH_L1_LOSS_BETA

    # -- stand-alone helpers (surface parity; the fused kernel does not call them) ----
    def smooth_l1_loss(self, input: Tensor, target: Tensor) -> Tensor:
        "Summed smooth-L1 with threshold `beta` (losses.py:19-27)."
        n = torch.abs(input - target)
        if self.beta < 1e-5:
            return n.sum()
        return torch.where(n < self.beta, 0.5 * n ** 2 / self.beta, n - 0.5 * self.beta).sum()

    def focal_loss(self, clas_pred: Tensor, clas_tgt: Tensor) -> Tensor:
        """Summed focal loss of logits vs {0,1} targets exactly as the reference
        defines it (losses.py:29-47): constant (detached) modulating weight, and
        alpha applied to the NEGATIVES' complement (positives get 1 - alpha)."""
        p = torch.sigmoid(clas_pred.detach())
        w = clas_tgt * (1 - p) + (1 - clas_tgt) * p
        a = (1 - clas_tgt) * self.alpha + clas_tgt * (1 - self.alpha)
        w = w.pow(self.gamma).mul(a)
        return F.binary_cross_entropy_with_logits(clas_pred, clas_tgt, w, reduction="sum")

    # -- fused path ----------------------------------------------------------------------
    def _params(self):
        return ops.make_loss_params(self.alpha, self.gamma, self.beta, LOGIT_SHIFT, ENCODE_LOG_EPS, BBOX_REG_WEIGHTS)

    def _fused(self, cls: Tensor, box: Tensor, anchors, boxes: Sequence[Tensor], labels: Sequence[Tensor]) -> Tensor:
        dev = cls.device
        counts = [int(b.reshape(-1, 4).shape[0]) for b in boxes]
        gt_boxes = torch.cat([b.reshape(-1, 4).to(device=dev, dtype=torch.float32) for b in boxes]) \
            if counts else torch.zeros((0, 4), device=dev)
        gt_labels = torch.cat([l.reshape(-1).to(device=dev, dtype=torch.int64) for l in labels]) \
            if counts else torch.zeros((0,), dtype=torch.int64, device=dev)
        gt_off = ops.gt_offsets(counts, dev)
        return _FusedDenseHeadLoss.apply(cls, box, _stack_anchors(anchors), gt_boxes, gt_labels, gt_off,
                                         self._params(), IOU_THRESHOLDS_FOREGROUND, IOU_THRESHOLDS_BACKGROUND)

    @staticmethod
    def _gt_arrays(targets, dev):
        return ops.gt_pack([t["boxes"] for t in targets], [t["labels"] for t in targets], dev)[:3]

    @staticmethod
    def fuses_match(targets) -> bool:
        """``RN_FUSE_MATCH=1``: the matcher runs inside the loss kernel (one launch, ``rn_loss_match_fwd_bwd_levels``) when no image has
        more than 64 GT boxes.  Off by default: measured slower than K2 + K3 (see ``FUSE_MATCH``)."""
        return FUSE_MATCH and max([int(t["boxes"].reshape(-1, 4).shape[0]) for t in targets] or [0]) <= 64

    def match_ahead(self, targets: List[Dict[str, Tensor]], anchors) -> MatchAhead:
        """Launch K2 (IoU + matcher, box_utils.py:51-80) NOW, on a side stream: it depends only on the anchors and the GT boxes,
        both known as soon as the feature-map shapes are, and then runs beside the head convolutions instead of between the
        class-output conv and K3.  Pass the result to ``forward_levels(..., ahead=...)``, which makes the loss kernel wait
        for it.  The outputs belong to the CALLING stream (allocated before the fork, consumed after the join)."""
        anchors = _stack_anchors(anchors)
        dev = anchors.device
        h = MatchAhead()
        h.anchors = anchors
        h.gt_boxes, h.gt_labels, h.gt_off = self._gt_arrays(targets, dev)
        B = len(targets)
        h.matches, h.num_fg, h.special = ops.iou_match_outputs(B, int(anchors.shape[-2]), dev)
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.iou_match(anchors, h.gt_boxes, h.gt_off, B, IOU_THRESHOLDS_FOREGROUND, IOU_THRESHOLDS_BACKGROUND,
                          out=(h.matches, h.num_fg, h.special))
            h.done = torch.cuda.Event()
            h.done.record(side)
        # the tensors were allocated on the calling stream but are written / read on the side stream: if the head raises before
        # the loss joins the streams and the handle is dropped, the allocator must not hand the blocks out while K2 still runs
        for t in (h.matches, h.num_fg, h.special, h.gt_boxes, h.gt_off, anchors):
            if t is not None:
                t.record_stream(side)
        return h

    def forward_levels(self, targets: List[Dict[str, Tensor]], cls_levels: Sequence[Tensor], box_levels: Sequence[Tensor],
                       anchors, ahead: Optional[MatchAhead] = None) -> Dict[str, Tensor]:
        """Same result as ``forward`` on ``torch.cat(levels, dim=1)``, without materialising the cat
        (the loss kernel reads the per-level conv outputs where they are).  ``ahead``: ``match_ahead``'s
        handle for these targets / anchors (K2 already in flight on a side stream)."""
        dev = cls_levels[0].device
        if ahead is not None:
            gt_boxes, gt_labels, gt_off, anchors_t = ahead.gt_boxes, ahead.gt_labels, ahead.gt_off, ahead.anchors
        else:
            gt_boxes, gt_labels, gt_off, nfg0 = ops.gt_pack([t["boxes"] for t in targets], [t["labels"] for t in targets], dev)
            anchors_t = _stack_anchors(anchors)
            if cls_levels[0].is_cuda and self.fuses_match(targets):
                ahead = max([int(t["boxes"].reshape(-1, 4).shape[0]) for t in targets] or [0])      # (an int: see the Function)
            elif nfg0 is not None:
                ahead = GtPack(nfg0)
        out = _FusedDenseHeadLossLevels.apply(anchors_t, gt_boxes, gt_labels, gt_off, self._params(),
                                              IOU_THRESHOLDS_FOREGROUND, IOU_THRESHOLDS_BACKGROUND, len(cls_levels), ahead,
                                              *cls_levels, *box_levels)
        return {"classification_loss": out[0], "regression_loss": out[1]}

    def calc_loss(self, anchors: Tensor, clas_pred: Tensor, bbox_pred: Tensor, clas_tgt: Tensor,
                  bbox_tgt: Tensor) -> Tuple[Tensor, Tensor]:
        "One image: returns (bb_loss, clas_loss), each already / clamp(num_fg, 1) (losses.py:49-111)."
        out = self._fused(clas_pred[None], bbox_pred[None], anchors, [bbox_tgt], [clas_tgt])
        return out[1], out[0]

    def forward(self, targets: List[Dict[str, Tensor]], head_outputs: Dict[str, Tensor],
                anchors: List[Tensor]) -> Dict[str, Tensor]:
        "Batch means of the per-image normalised losses (losses.py:113-145)."
        clas_preds, bbox_preds = head_outputs["cls_preds"], head_outputs["bbox_preds"]
        if len(targets) != clas_preds.shape[0]:
            raise ValueError(f"{len(targets)} targets for a batch of {clas_preds.shape[0]}")
        out = self._fused(clas_preds, bbox_preds, anchors, [t["boxes"] for t in targets], [t["labels"] for t in targets])
        return {"classification_loss": out[0], "regression_loss": out[1]}
