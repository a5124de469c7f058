"""Anchor generator with the reference's surface (``retinanet/anchors.py:55-228``);
the grid emission itself is the HIP kernel ``rn_anchors_emit`` (K1).

Differences in *how*, not *what*: the reference re-derives identical anchors for
every image of every batch with ~6 tiny torch ops per level (anchors.py:223-228);
here one launch writes all levels, and the result is cached per (grid shapes,
device, cell-anchor contents) and shared by every image of the batch.
"""
import math
from typing import Dict, List, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import ops
from .config import ANCHOR_ASPECT_RATIOS, ANCHOR_OFFSET, ANCHOR_SIZES, ANCHOR_STRIDES
from .utilities import ifnone


class BufferList(nn.Module):
    """Buffers registered as "0", "1", ... so state-dict keys read
    ``anchor_generator.cell_anchors.{i}`` like the reference (anchors.py:13-27)."""

    def __init__(self, buffers: Sequence[Tensor]):
        super().__init__()
        for i, b in enumerate(buffers):
            self.register_buffer(str(i), b)

    def __len__(self) -> int:
        return len(self._buffers)

    def __iter__(self):
        return iter(self._buffers.values())


def _broadcast_params(params, num_features: int, name: str) -> List[List[float]]:
    """One list for all levels, or one list per level (anchors.py:30-52)."""
    assert isinstance(params, (list, tuple)), f"{name} in anchor generator has to be a list! Got {params}."
    assert len(params), f"{name} in anchor generator cannot be empty!"
    if not isinstance(params[0], (list, tuple)):
        return [list(params) for _ in range(num_features)]
    if len(params) == 1:
        return [list(params[0]) for _ in range(num_features)]
    assert len(params) == num_features, (
        f"Got {name} of length {len(params)} in anchor generator, "
        f"but the number of input features is {num_features}!")
    return [list(p) for p in params]


class AnchorGenerator(nn.Module):
    """``AnchorGenerator(sizes, aspect_ratios, strides, offset)``; every argument
    defaults to ``config.py`` (anchors.py:69-99)."""

    def __init__(self, sizes=None, aspect_ratios=None, strides=None, offset: float = None) -> None:
        super().__init__()
        strides = ifnone(strides, ANCHOR_STRIDES)
        sizes = ifnone(sizes, ANCHOR_SIZES)
        aspect_ratios = ifnone(aspect_ratios, ANCHOR_ASPECT_RATIOS)
        offset = ifnone(offset, ANCHOR_OFFSET)
        self.strides = strides
        self.num_features = len(strides)
        self.sizes = _broadcast_params(sizes, self.num_features, "sizes")
        self.aspect_ratios = _broadcast_params(aspect_ratios, self.num_features, "aspect_ratios")
        self.offset = offset
        self.cell_anchors = self._calculate_cell_anchors(self.sizes, self.aspect_ratios)
        self._cache: Dict[tuple, Tensor] = {}

    # -- A1: cell anchors, double arithmetic then fp32 (anchors.py:102-135) ----------
    def _calculate_cell_anchors(self, sizes, ratios) -> BufferList:
        return self._calculate_anchors(sizes, ratios)

    def _calculate_anchors(self, sizes, aspect_ratios) -> BufferList:
        return BufferList([self.generate_cell_anchors(s, a).float() for s, a in zip(sizes, aspect_ratios)])

    @staticmethod
    def generate_cell_anchors(sizes, aspect_ratios) -> Tensor:
        """[len(sizes)*len(aspect_ratios), 4] XYXY boxes centred on (0,0); size-major;
        aspect ratio = h / w."""
        rows = []
        for size in sizes:
            area = size ** 2.0
            for ar in aspect_ratios:
                w = math.sqrt(area / ar)
                h = ar * w
                rows.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        return torch.tensor(rows)

    @property
    def num_cell_anchors(self) -> List[int]:
        return self.num_anchors

    @property
    def num_anchors(self) -> List[int]:
        """Anchors per feature-map location, per level."""
        return [len(c) for c in self.cell_anchors]

    @staticmethod
    def _compute_grid_offsets(size: List[int], stride: int, offset: float, device: torch.device):
        """Flattened (x, y) shifts of a grid (anchors.py:151-170).  Kept for surface
        parity; the HIP kernel computes the same shifts in-register."""
        H, W = size
        sx = torch.arange(offset * stride, W * stride, step=stride, dtype=torch.float32, device=device)
        sy = torch.arange(offset * stride, H * stride, step=stride, dtype=torch.float32, device=device)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        return xx.reshape(-1), yy.reshape(-1)

    # -- A2-A4: one launch for all levels ---------------------------------------------
    def _levels(self, grid_sizes) -> List[Tuple[int, int, int]]:
        return [(int(g[0]), int(g[1]), int(s)) for g, s in zip(grid_sizes, self.strides)]

    def _all_levels(self, grid_sizes, device: torch.device) -> Tensor:
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        cells = [b if b.device == device else b.to(device) for b in self.cell_anchors]
        levels = self._levels(grid_sizes)
        key = (tuple(levels), str(device), float(self.offset), tuple((c.data_ptr(), c._version) for c in cells))
        hit = self._cache.get(key)
        if hit is None:
            if len(self._cache) > 16:
                self._cache.clear()
            hit = ops.anchors_emit(levels, cells, self.offset)
            self._cache[key] = hit
        return hit

    def grid_anchors(self, grid_sizes: List[List[int]], device: torch.device) -> List[Tensor]:
        """Per-level anchors, each [(H*W*num_cell), 4] (anchors.py:172-197).  The
        returned tensors are views of one cached buffer: treat them as read-only."""
        flat = self._all_levels(grid_sizes, device)
        counts = [h * w * n for (h, w, _), n in zip(self._levels(grid_sizes), self.num_anchors)]
        return list(torch.split(flat, counts))

    def forward(self, images, feature_maps: List[Tensor]) -> List[Tensor]:
        """One ``[A,4]`` tensor per image (anchors.py:199-228).  All entries are the
        SAME cached tensor (anchors depend only on the feature-map shapes, Q12)."""
        grid_sizes = [fm.shape[-2:] for fm in feature_maps]
        flat = self._all_levels(grid_sizes, feature_maps[0].device)
        return [flat for _ in images.image_sizes]
