"""The reference's CSV detection-dataset format (``README.md:103-112``; reader
``utils/pascal/pascal_utils.py:98-142``):

    filename,width,height,class,xmin,ymin,xmax,ymax,labels
    Images/007826.jpg,500,375,diningtable,80,217,320,273,11

one row per box, absolute xyxy pixel coordinates, integer ``labels`` (1..K), ``width`` / ``height`` /
``class`` optional.  Items are ``(image f32 [3,H,W] in 0..1, target, image_idx)`` with the reference's
target keys (``image_id, boxes, labels, area, iscrowd``).  Images are decoded with PIL (the reference
uses cv2 + albumentations, neither of which this framework depends on); ``transforms`` is an optional
callable ``(image uint8 [H,W,3] ndarray, boxes [n,4] ndarray, labels list) -> (image tensor, boxes, labels)``.
"""
import csv
import os
from collections import OrderedDict
from typing import Callable, List, Optional

import numpy as np
import torch
from torch.utils.data import Dataset

__all__ = ["CSVDetectionDataset"]


class CSVDetectionDataset(Dataset):
    def __init__(self, csv_path: str, transforms: Optional[Callable] = None, root: Optional[str] = None):
        self.root = root if root is not None else os.path.dirname(os.path.abspath(csv_path))
        self.tfms = transforms
        self.records: "OrderedDict[str, List[dict]]" = OrderedDict()          # first-appearance order, like df.unique()
        with open(csv_path, newline="") as f:
            reader = csv.DictReader(f)
            missing = {"filename", "xmin", "ymin", "xmax", "ymax", "labels"} - set(reader.fieldnames or [])
            if missing:
                raise ValueError(f"{csv_path}: missing column(s) {sorted(missing)}")
            for row in reader:
                self.records.setdefault(row["filename"], []).append(row)
        self.image_ids = list(self.records)

    def __len__(self) -> int:
        return len(self.image_ids)

    def _path(self, name: str) -> str:
        return name if os.path.isabs(name) or os.path.exists(name) else os.path.join(self.root, name)

    def __getitem__(self, index: int):
        from PIL import Image
        name = self.image_ids[index]
        im = np.asarray(Image.open(self._path(name)).convert("RGB"))
        rows = self.records[name]
        boxes = np.array([[float(r["xmin"]), float(r["ymin"]), float(r["xmax"]), float(r["ymax"])] for r in rows], dtype=np.float32)
        labels = [int(r["labels"]) for r in rows]
        area = torch.as_tensor((boxes[:, 3] - boxes[:, 1]) * (boxes[:, 2] - boxes[:, 0]), dtype=torch.float32)
        iscrowd = torch.zeros((len(rows),), dtype=torch.int64)
        if self.tfms is not None:
            image, boxes, labels = self.tfms(im, boxes, labels)
        else:
            image = torch.from_numpy(im.copy()).permute(2, 0, 1).to(torch.float32) / 255.0
        image_idx = torch.tensor([index])
        target = {"image_id": image_idx, "boxes": torch.as_tensor(np.asarray(boxes), dtype=torch.float32).reshape(-1, 4),
                  "labels": torch.as_tensor(labels, dtype=torch.int64), "area": area, "iscrowd": iscrowd}
        return image, target, image_idx
