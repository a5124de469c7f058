"""Input transform used by ``Retinanet`` (reference call sites ``retinanet/models.py:116,
:262, :271, :279``; the reference takes it from torchvision's detection package,
which this framework does not depend on).

Semantics (torchvision 0.7/0.8 ``GeneralizedRCNNTransform``): per image
``(x - mean) / std`` -> bilinear resize so the short side is ``min_size`` unless the
long side would exceed ``max_size`` (``align_corners=False``, scale recomputed from
the integer output size) -> GT boxes scaled by the per-axis size ratio -> images
zero-padded into one batch whose H, W are rounded up to a multiple of 32.
``postprocess`` maps detections back to the original image sizes (eval mode only).

For CUDA fp32 images the normalise / resize / pad / batch sequence is ONE HIP launch
(``rn_transform_batch``, ``csrc/transform.hip``) that can also write the batch
directly in the layout and dtype the conv stack consumes (channels-last, autocast dtype), so the
per-image elementwise kernels, the batch memset + copies, the ``contiguous(channels_last)`` pass
and autocast's cast of the conv1 input all disappear.  Anything else (CPU tensors, other dtypes)
takes the PyTorch ops below, which implement the same arithmetic.
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn


class ImageList(object):
    """A padded batch plus each image's (h, w) before padding."""

    def __init__(self, tensors: Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, device) -> "ImageList":
        return ImageList(self.tensors.to(device), self.image_sizes)


def resize_boxes(boxes: Tensor, original_size: Sequence[int], new_size: Sequence[int]) -> Tensor:
    # fp32 ratios (as torchvision computes them), applied as host scalars: no H2D copy, no sync
    rh = float(np.float32(new_size[0]) / np.float32(original_size[0]))
    rw = float(np.float32(new_size[1]) / np.float32(original_size[1]))
    if rh == 1.0 and rw == 1.0:
        return boxes
    x1, y1, x2, y2 = boxes.unbind(1)
    return torch.stack((x1 * rw, y1 * rh, x2 * rw, y2 * rh), dim=1)


class GeneralizedRCNNTransform(nn.Module):
    def __init__(self, min_size, max_size: int, image_mean: Sequence[float], image_std: Sequence[float],
                 size_divisible: int = 32):
        super().__init__()
        self.min_size = tuple(min_size) if isinstance(min_size, (list, tuple)) else (min_size,)
        self.max_size = max_size
        self.image_mean = list(image_mean)
        self.image_std = list(image_std)
        self.size_divisible = size_divisible
        self._stats = {}

    # -- pieces ------------------------------------------------------------------------
    def normalize(self, image: Tensor) -> Tensor:
        key = (image.device, image.dtype, tuple(self.image_mean), tuple(self.image_std))
        stats = self._stats.get(key)
        if stats is None:       # uploaded once per (device, dtype): no per-image H2D copies
            mean = torch.as_tensor(self.image_mean, dtype=image.dtype, device=image.device)
            std = torch.as_tensor(self.image_std, dtype=image.dtype, device=image.device)
            stats = self._stats[key] = (mean[:, None, None], std[:, None, None])
        return (image - stats[0]) / stats[1]

    def _target_short_side(self) -> float:
        if self.training and len(self.min_size) > 1:
            return float(self.min_size[int(torch.empty(1).uniform_(0.0, float(len(self.min_size))).item())])
        return float(self.min_size[-1])

    def _scale_for(self, h: int, w: int, short: float) -> float:
        lo, hi = float(min(h, w)), float(max(h, w))
        scale = short / lo
        if hi * scale > float(self.max_size):
            scale = float(self.max_size) / hi
        return scale

    def resize(self, image: Tensor, target: Optional[Dict[str, Tensor]]):
        h, w = int(image.shape[-2]), int(image.shape[-1])
        scale = self._scale_for(h, w, self._target_short_side())
        nh, nw = int(math.floor(h * scale)), int(math.floor(w * scale))
        if (nh, nw) != (h, w):
            image = F.interpolate(image[None], scale_factor=scale, mode="bilinear",
                                  recompute_scale_factor=True, align_corners=False)[0]
        # (same size => bilinear resampling with align_corners=False is the identity: skip the launch)
        if target is not None:
            target["boxes"] = resize_boxes(target["boxes"], (h, w), image.shape[-2:])
        return image, target

    def batch_images(self, images: List[Tensor]) -> Tensor:
        d = float(self.size_divisible)
        c = max(im.shape[0] for im in images)
        hh = int(math.ceil(max(im.shape[1] for im in images) / d) * d)
        ww = int(math.ceil(max(im.shape[2] for im in images) / d) * d)
        out = images[0].new_zeros((len(images), c, hh, ww))
        for im, dst in zip(images, out):
            dst[: im.shape[0], : im.shape[1], : im.shape[2]].copy_(im)
        return out

    # -- fused path ----------------------------------------------------------------------
    def _fusable(self, images: List[Tensor]) -> bool:
        return (len(self.image_mean) == 3 and len(self.image_std) == 3 and
                all(im.is_cuda and im.dim() == 3 and im.shape[0] == 3 and im.dtype == torch.float32 for im in images)
                and self.size_divisible % 4 == 0)

    def _forward_fused(self, images: List[Tensor], targets, out_dtype: torch.dtype, channels_last: bool):
        from . import ops                       # the HIP library is only needed once a CUDA image shows up
        sizes = []
        for i, im in enumerate(images):
            h, w = int(im.shape[-2]), int(im.shape[-1])
            scale = self._scale_for(h, w, self._target_short_side())       # drawn per image, like torchvision
            new = (int(math.floor(h * scale)), int(math.floor(w * scale)))
            sizes.append(new)
            if targets is not None:
                targets[i]["boxes"] = resize_boxes(targets[i]["boxes"], (h, w), new)
        d = float(self.size_divisible)
        hp = int(math.ceil(max(s[0] for s in sizes) / d) * d)
        wp = int(math.ceil(max(s[1] for s in sizes) / d) * d)
        batch = ops.transform_batch(images, sizes, self.image_mean, self.image_std, hp, wp, out_dtype, channels_last)
        return ImageList(batch, sizes), targets

    # -- whole transform -----------------------------------------------------------------
    def forward(self, images: List[Tensor], targets: Optional[List[Dict[str, Tensor]]] = None,
                out_dtype: Optional[torch.dtype] = None, channels_last: bool = False):
        """``out_dtype`` / ``channels_last``: layout hints for the fused CUDA path (defaults: fp32, NCHW --
        what torchvision's transform returns); ignored by the PyTorch fallback."""
        images = list(images)
        if targets is not None:
            targets = [dict(t) for t in targets]
        for im in images:
            if im.dim() != 3:
                raise ValueError(f"images is expected to be a list of 3d tensors of shape [C, H, W], got {tuple(im.shape)}")
        if self._fusable(images):
            return self._forward_fused(images, targets, out_dtype or torch.float32, channels_last)
        for i, im in enumerate(images):
            if im.dim() != 3:
                raise ValueError(f"images is expected to be a list of 3d tensors of shape [C, H, W], got {tuple(im.shape)}")
            tgt = targets[i] if targets is not None else None
            im, tgt = self.resize(self.normalize(im), tgt)
            images[i] = im
            if tgt is not None:
                targets[i] = tgt
        sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
        return ImageList(self.batch_images(images), sizes), targets

    def postprocess(self, result: List[Dict[str, Tensor]], image_shapes: List[Tuple[int, int]],
                    original_image_sizes: List[Tuple[int, int]]) -> List[Dict[str, Tensor]]:
        if self.training:
            return result
        for i, (pred, s, o) in enumerate(zip(result, image_shapes, original_image_sizes)):
            result[i]["boxes"] = resize_boxes(pred["boxes"], s, o)
        return result
