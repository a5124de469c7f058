"""TEST-ONLY stand-in for the handful of torchvision symbols the reference imports.

This is NOT torchvision and NOT product code.  torchvision is absent from this
image, and the reference (``/root/reference/retinanet``) cannot be imported
without it (``models.py:7-8``, ``anchors.py:7``, ``box_utils.py:5``,
``backbone.py:6``).  This module restates the published semantics of the
seven symbols the reference touches (torchvision 0.7/0.8 era, as pinned by
``README.md:17`` of the reference) in plain torch so that the reference can be
imported *in the build container only* to generate golden vectors
(``gen_golden.py``).  It never travels into the product path.

``install()`` registers fake ``torchvision.*`` modules in ``sys.modules``.
"""
import math
import sys
import types
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn


# --------------------------------------------------------------------------- #
# torchvision.ops.boxes
# --------------------------------------------------------------------------- #
def box_area(boxes: Tensor) -> Tensor:
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def box_iou(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    a1 = box_area(boxes1)
    a2 = box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (a1[:, None] + a2 - inter)


def clip_boxes_to_image(boxes: Tensor, size: Tuple[int, int]) -> Tensor:
    dim = boxes.dim()
    bx = boxes[..., 0::2]
    by = boxes[..., 1::2]
    h, w = size
    bx = bx.clamp(min=0, max=w)
    by = by.clamp(min=0, max=h)
    return torch.stack((bx, by), dim=dim).reshape(boxes.shape)


def remove_small_boxes(boxes: Tensor, min_size: float) -> Tensor:
    ws = boxes[:, 2] - boxes[:, 0]
    hs = boxes[:, 3] - boxes[:, 1]
    keep = (ws >= min_size) & (hs >= min_size)
    return torch.where(keep)[0]


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """Greedy NMS with the arithmetic of torchvision's CPU kernel.

    Sort by score descending; a box is kept if no earlier kept box overlaps it
    with IoU > threshold; ``ovr = inter / (area_i + area_j - inter)`` with
    ``w = max(0, xx2 - xx1)``, ``h = max(0, yy2 - yy1)``, all in the boxes'
    dtype.  Returns kept indices in score order (int64).
    """
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    x1, y1, x2, y2 = boxes.unbind(1)
    areas = (x2 - x1) * (y2 - y1)
    order = torch.sort(scores, dim=0, descending=True, stable=True)[1]
    n = boxes.shape[0]
    suppressed = torch.zeros(n, dtype=torch.bool)
    keep: List[int] = []
    zero = boxes.new_zeros(())
    for _i in range(n):
        i = int(order[_i])
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        if rest.numel() == 0:
            continue
        xx1 = torch.max(x1[i], x1[rest])
        yy1 = torch.max(y1[i], y1[rest])
        xx2 = torch.min(x2[i], x2[rest])
        yy2 = torch.min(y2[i], y2[rest])
        w = torch.max(zero, xx2 - xx1)
        h = torch.max(zero, yy2 - yy1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > iou_threshold]] = True
    return torch.as_tensor(keep, dtype=torch.int64, device=boxes.device)


# --------------------------------------------------------------------------- #
# torchvision.models.detection.image_list / transform
# --------------------------------------------------------------------------- #
class ImageList(object):
    def __init__(self, tensors: Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, device):
        return ImageList(self.tensors.to(device), self.image_sizes)


def _scale_boxes(boxes: Tensor, old_hw, new_hw) -> Tensor:
    rh = torch.tensor(new_hw[0], dtype=torch.float32) / torch.tensor(old_hw[0], dtype=torch.float32)
    rw = torch.tensor(new_hw[1], dtype=torch.float32) / torch.tensor(old_hw[1], dtype=torch.float32)
    x1, y1, x2, y2 = boxes.unbind(1)
    return torch.stack((x1 * rw, y1 * rh, x2 * rw, y2 * rh), dim=1)


class GeneralizedRCNNTransform(nn.Module):
    def __init__(self, min_size, max_size, image_mean, image_std):
        super().__init__()
        if not isinstance(min_size, (list, tuple)):
            min_size = (min_size,)
        self.min_size = min_size
        self.max_size = max_size
        self.image_mean = image_mean
        self.image_std = image_std

    def forward(self, images, targets=None):
        images = [im for im in images]
        if targets is not None:
            targets = [{k: v for k, v in t.items()} for t in targets]
        for i, im in enumerate(images):
            if im.dim() != 3:
                raise ValueError("images must be a list of [C,H,W] tensors")
            tgt = targets[i] if targets is not None else None
            im = self.normalize(im)
            im, tgt = self.resize(im, tgt)
            images[i] = im
            if targets is not None and tgt is not None:
                targets[i] = tgt
        sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
        return ImageList(self.batch_images(images), sizes), targets

    def normalize(self, image):
        mean = torch.as_tensor(self.image_mean, dtype=image.dtype, device=image.device)
        std = torch.as_tensor(self.image_std, dtype=image.dtype, device=image.device)
        return (image - mean[:, None, None]) / std[:, None, None]

    def resize(self, image, target):
        h, w = image.shape[-2:]
        if self.training:
            k = self.min_size
            size = float(k[int(torch.empty(1).uniform_(0.0, float(len(k))).item())])
        else:
            size = float(self.min_size[-1])
        lo = float(min(h, w))
        hi = float(max(h, w))
        scale = size / lo
        if hi * scale > float(self.max_size):
            scale = float(self.max_size) / hi
        image = torch.nn.functional.interpolate(
            image[None], scale_factor=scale, mode="bilinear",
            recompute_scale_factor=True, align_corners=False)[0]
        if target is None:
            return image, target
        target["boxes"] = _scale_boxes(target["boxes"], (h, w), image.shape[-2:])
        return image, target

    def batch_images(self, images, size_divisible=32):
        c = max(im.shape[0] for im in images)
        hh = max(im.shape[1] for im in images)
        ww = max(im.shape[2] for im in images)
        s = float(size_divisible)
        hh = int(math.ceil(float(hh) / s) * s)
        ww = int(math.ceil(float(ww) / s) * s)
        out = images[0].new_full((len(images), c, hh, ww), 0)
        for im, dst in zip(images, out):
            dst[: im.shape[0], : im.shape[1], : im.shape[2]].copy_(im)
        return out

    def postprocess(self, result, image_shapes, original_image_sizes):
        if self.training:
            return result
        for i, (pred, s, o) in enumerate(zip(result, image_shapes, original_image_sizes)):
            result[i]["boxes"] = _scale_boxes(pred["boxes"], s, o)
        return result


def load_state_dict_from_url(*_a, **_k):
    raise RuntimeError("stand-in: no network; use pretrained=False")


def install() -> None:
    """Register the fake ``torchvision`` package tree in ``sys.modules``."""
    if "torchvision" in sys.modules and not getattr(sys.modules["torchvision"], "_RN_STANDIN", False):
        return  # a real torchvision is present: use it

    def mod(name):
        m = types.ModuleType(name)
        m._RN_STANDIN = True
        sys.modules[name] = m
        return m

    tv = mod("torchvision")
    ops = mod("torchvision.ops")
    boxes = mod("torchvision.ops.boxes")
    models = mod("torchvision.models")
    det = mod("torchvision.models.detection")
    il = mod("torchvision.models.detection.image_list")
    tr = mod("torchvision.models.detection.transform")
    mu = mod("torchvision.models.utils")
    for f in (box_area, box_iou, clip_boxes_to_image, remove_small_boxes, nms):
        setattr(boxes, f.__name__, f)
        setattr(ops, f.__name__, f)
    ops.boxes = boxes
    il.ImageList = ImageList
    tr.GeneralizedRCNNTransform = GeneralizedRCNNTransform
    mu.load_state_dict_from_url = load_state_dict_from_url
    tv.ops = ops
    tv.models = models
    models.detection = det
    models.utils = mu
    det.image_list = il
    det.transform = tr


def import_reference(path: str = "/root/reference"):
    """Import the reference ``retinanet`` package (build container only)."""
    import importlib
    import os
    sys.dont_write_bytecode = True  # the reference tree is read-only
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    install()
    if path not in sys.path:
        sys.path.insert(0, path)
    return importlib.import_module("retinanet")
