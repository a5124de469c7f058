"""Box helpers with the reference's names (``retinanet/box_utils.py:11-80``).

``matcher`` and ``activ_2_bbox`` are the HIP kernels K2 / K4.  ``bbox_2_activ`` and
the two ``convert_*`` helpers are small torch expressions kept for surface parity:
on the training path the encode step is fused into the loss kernel (K3) and never
runs as a separate op.
"""
from typing import Optional

import torch
from torch import Tensor

from . import ops
from .config import BBOX_REG_WEIGHTS, ENCODE_LOG_EPS, IOU_THRESHOLDS_BACKGROUND, IOU_THRESHOLDS_FOREGROUND
from .utilities import ifnone


def convert_xywh(boxes: Tensor) -> Tensor:
    "XYXY -> (cx, cy, w, h)  (box_utils.py:11-15)"
    return torch.cat([(boxes[:, :2] + boxes[:, 2:]) / 2, boxes[:, 2:] - boxes[:, :2]], 1)


def convert_x1y1x2y2(boxes: Tensor) -> Tensor:
    "(cx, cy, w, h) -> XYXY  (box_utils.py:18-22)"
    half = boxes[:, 2:] / 2
    return torch.cat([boxes[:, :2] - half, boxes[:, :2] + half], 1)


def bbox_2_activ(bboxes: Tensor, anchors: Tensor) -> Tensor:
    "Regression targets of `bboxes` w.r.t. `anchors`, both XYXY (box_utils.py:25-34)."
    b, a = convert_xywh(bboxes), convert_xywh(anchors)
    centers = (b[..., :2] - a[..., :2]) / a[..., 2:]
    sizes = torch.log(b[..., 2:] / a[..., 2:] + ENCODE_LOG_EPS)
    return torch.cat([centers, sizes], -1).mul_(b.new_tensor([BBOX_REG_WEIGHTS]))


def activ_2_bbox(activations: Tensor, anchors: Tensor) -> Tensor:
    """Model activations -> XYXY boxes (box_utils.py:37-48), HIP kernel K4.

    Keeps the reference's behaviour: sizes come from ``exp(activations[..., :2])``
    (SURVEY Q4), and the activations are divided IN PLACE by ``BBOX_REG_WEIGHTS``
    (Q5; a value no-op with the default unit weights)."""
    if any(w != 1.0 for w in BBOX_REG_WEIGHTS):
        activations.div_(activations.new_tensor([BBOX_REG_WEIGHTS]))
    return ops.decode_clip(activations, anchors, None)


def matcher(anchors: Tensor, targets: Tensor, match_thr: Optional[float] = None, back_thr: Optional[float] = None) -> Tensor:
    """Match `anchors` [A,4] to `targets` [T,4]: -2 ignore, -1 background, else the
    target index (box_utils.py:51-80), HIP kernel K2 (fused IoU + arg-max + thresholds)."""
    match_thr = ifnone(match_thr, IOU_THRESHOLDS_FOREGROUND)
    back_thr = ifnone(back_thr, IOU_THRESHOLDS_BACKGROUND)
    assert match_thr > back_thr
    targets = targets.reshape(-1, 4)
    off = ops.gt_offsets([targets.shape[0]], anchors.device)
    matches, _ = ops.iou_match(anchors, targets, off, 1, match_thr, back_thr, want_num_fg=False)
    return matches[0]
