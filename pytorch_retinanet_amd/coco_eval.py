"""COCO-style bbox evaluation without pycocotools (reference: ``utils/coco/coco_eval.py:14-93,159-161``
-- ``CocoEvaluator.update / synchronize_between_processes / accumulate / summarize`` and the wire format
of ``prepare_for_coco_detection`` -- driven from ``model.py:136-146`` ``test_step`` / ``test_epoch_end``).

The reference delegates the metric to ``pycocotools.cocoeval.COCOeval`` (a pip dependency, unpinned in
``requirements.txt``; absent from this image and from ``/root/reference``).  ``BBoxEval`` below restates
that published algorithm for ``iouType="bbox"``: per (image, category) greedy matching of score-sorted
detections to ground truth at IoU thresholds .50:.05:.95 (crowd / out-of-area ground truth is "ignore"),
101-point interpolated precision, area ranges all / small / medium / large, maxDets 1 / 10 / 100, and
the 12-number ``stats`` vector whose element 0 (AP@[.5:.95]) the reference logs as ``AP``.
Parity status: known-answer tests only (``tests/test_coco_eval.py``); there is no pycocotools here to
pin the numbers against.
"""
from collections import defaultdict
from typing import Dict, Iterable, List, Mapping, Optional, Sequence

import numpy as np
import torch
from torch import Tensor

__all__ = ["convert_to_xywh", "prepare_for_coco_detection", "BBoxEval", "CocoEvaluator", "gt_from_dataset"]


def convert_to_xywh(boxes: Tensor) -> Tensor:
    "xyxy -> xywh (coco_eval.py:159-161)."
    xmin, ymin, xmax, ymax = boxes.unbind(1)
    return torch.stack((xmin, ymin, xmax - xmin, ymax - ymin), dim=1)


def prepare_for_coco_detection(predictions: Mapping[int, Mapping[str, Tensor]]) -> List[dict]:
    """``{image_id: {"boxes" xyxy, "scores", "labels"}}`` -> COCO results list
    ``[{"image_id", "category_id", "bbox": [x, y, w, h], "score"}]`` (coco_eval.py:71-93)."""
    coco_results = []
    for original_id, prediction in predictions.items():
        if len(prediction) == 0:
            continue
        boxes = convert_to_xywh(prediction["boxes"]).tolist()
        scores = prediction["scores"].tolist()
        labels = prediction["labels"].tolist()
        coco_results.extend({"image_id": original_id, "category_id": labels[k], "bbox": box, "score": scores[k]}
                            for k, box in enumerate(boxes))
    return coco_results


def _iou_xywh(dt: np.ndarray, gt: np.ndarray, crowd: np.ndarray) -> np.ndarray:
    "IoU matrix [D, G]; against a crowd region the union is the detection's own area."
    if len(dt) == 0 or len(gt) == 0:
        return np.zeros((len(dt), len(gt)))
    dx1, dy1, dx2, dy2 = dt[:, 0:1], dt[:, 1:2], dt[:, 0:1] + dt[:, 2:3], dt[:, 1:2] + dt[:, 3:4]
    gx1, gy1, gx2, gy2 = gt[:, 0], gt[:, 1], gt[:, 0] + gt[:, 2], gt[:, 1] + gt[:, 3]
    w = np.clip(np.minimum(dx2, gx2) - np.maximum(dx1, gx1), 0, None)
    h = np.clip(np.minimum(dy2, gy2) - np.maximum(dy1, gy1), 0, None)
    inter = w * h
    da, ga = dt[:, 2:3] * dt[:, 3:4], gt[:, 2] * gt[:, 3]
    union = np.where(crowd[None, :], da, da + ga[None, :] - inter)
    return np.where(union > 0, inter / np.where(union > 0, union, 1), 0.0)


class BBoxEval:
    """COCOeval(iouType="bbox") restated.  ``gt`` / ``dt``: lists of annotation dicts (COCO json rows):
    gt ``{"image_id", "category_id", "bbox" xywh, optional "area", "iscrowd"}``, dt adds ``"score"``."""

    iou_thrs = np.linspace(0.5, 0.95, 10)
    rec_thrs = np.linspace(0.0, 1.0, 101)
    max_dets = (1, 10, 100)
    area_rng = ((0.0, 1e10), (0.0, 32.0 ** 2), (32.0 ** 2, 96.0 ** 2), (96.0 ** 2, 1e10))
    area_lbl = ("all", "small", "medium", "large")

    def __init__(self, gt: Iterable[dict], dt: Iterable[dict] = (), img_ids: Optional[Sequence[int]] = None):
        self.gt, self.dt = list(gt), list(dt)
        self.img_ids = img_ids
        self.stats = np.zeros(12)
        self.eval: Dict[str, np.ndarray] = {}

    # -- per image / category -----------------------------------------------------------------
    def _evaluate_img(self, g: List[dict], d: List[dict], rng) -> Optional[dict]:
        if not g and not d:
            return None
        T = len(self.iou_thrs)
        g_ig = np.array([bool(x.get("iscrowd", 0)) or not (rng[0] <= x["_area"] <= rng[1]) for x in g], dtype=bool)
        gorder = np.argsort(g_ig, kind="mergesort")                    # cared-for ground truth first
        g = [g[i] for i in gorder]
        g_ig = g_ig[gorder]
        crowd = np.array([bool(x.get("iscrowd", 0)) for x in g], dtype=bool)
        d = sorted(d, key=lambda x: -x["score"])[: self.max_dets[-1]]   # sorted() is stable, like mergesort
        ious = _iou_xywh(np.array([x["bbox"] for x in d], dtype=np.float64).reshape(-1, 4),
                         np.array([x["bbox"] for x in g], dtype=np.float64).reshape(-1, 4), crowd)
        gtm = -np.ones((T, len(g)), dtype=np.int64)
        dtm = -np.ones((T, len(d)), dtype=np.int64)
        dt_ig = np.zeros((T, len(d)), dtype=bool)
        for ti, t in enumerate(self.iou_thrs):
            for di in range(len(d)):
                best, m = min(t, 1 - 1e-10), -1
                for gi in range(len(g)):
                    if gtm[ti, gi] >= 0 and not crowd[gi]:
                        continue
                    if m > -1 and not g_ig[m] and g_ig[gi]:
                        break                                           # only ignored gt from here on
                    if ious[di, gi] < best:
                        continue
                    best, m = ious[di, gi], gi
                if m == -1:
                    continue
                dt_ig[ti, di] = g_ig[m]
                dtm[ti, di] = m
                gtm[ti, m] = di
        d_area = np.array([x["bbox"][2] * x["bbox"][3] for x in d], dtype=np.float64)
        out_rng = (d_area < rng[0]) | (d_area > rng[1])
        dt_ig |= (dtm < 0) & out_rng[None, :]
        return {"scores": np.array([x["score"] for x in d], dtype=np.float64), "dtm": dtm >= 0, "dt_ig": dt_ig, "g_ig": g_ig}

    # -- whole protocol ----------------------------------------------------------------------------
    def evaluate(self) -> "BBoxEval":
        for x in self.gt:
            x["_area"] = float(x["area"]) if "area" in x else float(x["bbox"][2] * x["bbox"][3])
        imgs = sorted(set(self.img_ids) if self.img_ids is not None else {x["image_id"] for x in self.gt} | {x["image_id"] for x in self.dt})
        cats = sorted({x["category_id"] for x in self.gt} | {x["category_id"] for x in self.dt})
        keep = set(imgs)
        gts, dts = defaultdict(list), defaultdict(list)
        for x in self.gt:
            if x["image_id"] in keep:
                gts[x["image_id"], x["category_id"]].append(x)
        for x in self.dt:
            if x["image_id"] in keep:
                dts[x["image_id"], x["category_id"]].append(x)
        T, R, K, A, M = len(self.iou_thrs), len(self.rec_thrs), len(cats), len(self.area_rng), len(self.max_dets)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        eps = np.spacing(1)
        for ki, cat in enumerate(cats):
            for ai, rng in enumerate(self.area_rng):
                per_img = [e for e in (self._evaluate_img(gts.get((i, cat), []), dts.get((i, cat), []), rng) for i in imgs) if e]
                if not per_img:
                    continue
                for mi, md in enumerate(self.max_dets):
                    scores = np.concatenate([e["scores"][:md] for e in per_img])
                    order = np.argsort(-scores, kind="mergesort")
                    dtm = np.concatenate([e["dtm"][:, :md] for e in per_img], axis=1)[:, order]
                    dt_ig = np.concatenate([e["dt_ig"][:, :md] for e in per_img], axis=1)[:, order]
                    npig = int(sum((~e["g_ig"]).sum() for e in per_img))
                    if npig == 0:
                        continue
                    tps = np.cumsum(dtm & ~dt_ig, axis=1).astype(np.float64)
                    fps = np.cumsum(~dtm & ~dt_ig, axis=1).astype(np.float64)
                    for ti in range(T):
                        tp, fp = tps[ti], fps[ti]
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + eps)
                        recall[ti, ki, ai, mi] = rc[-1] if nd else 0.0
                        for i in range(nd - 1, 0, -1):                  # precision envelope
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        q = np.zeros(R)
                        inds = np.searchsorted(rc, self.rec_thrs, side="left")
                        ok = inds < nd
                        q[ok] = pr[inds[ok]]
                        precision[ti, :, ki, ai, mi] = q
        self.eval = {"precision": precision, "recall": recall}
        return self

    accumulate = evaluate        # COCOeval splits the work in two calls; here it is one pass

    def _summ(self, ap: bool, iou: Optional[float] = None, area: str = "all", max_det: int = 100) -> float:
        ai, mi = self.area_lbl.index(area), self.max_dets.index(max_det)
        s = self.eval["precision"][:, :, :, ai, mi] if ap else self.eval["recall"][:, :, ai, mi]
        if iou is not None:
            s = s[np.where(np.isclose(self.iou_thrs, iou))[0]]
        s = s[s > -1]
        return float(s.mean()) if s.size else -1.0

    def summarize(self, verbose: bool = True) -> np.ndarray:
        if not self.eval:
            self.evaluate()
        spec = [(True, None, "all", 100), (True, 0.5, "all", 100), (True, 0.75, "all", 100), (True, None, "small", 100),
                (True, None, "medium", 100), (True, None, "large", 100), (False, None, "all", 1), (False, None, "all", 10),
                (False, None, "all", 100), (False, None, "small", 100), (False, None, "medium", 100), (False, None, "large", 100)]
        self.stats = np.array([self._summ(*s) for s in spec])
        if verbose:
            for (ap, iou, area, md), v in zip(spec, self.stats):
                name = "Average Precision  (AP)" if ap else "Average Recall     (AR)"
                rng = "0.50:0.95" if iou is None else f"{iou:0.2f}"
                print(f" {name} @[ IoU={rng:<9} | area={area:>6s} | maxDets={md:>3d} ] = {v:0.3f}")
        return self.stats


def gt_from_dataset(dataset) -> List[dict]:
    """Ground-truth annotation rows from a detection dataset yielding ``(image, target, image_id)`` or
    ``(image, target)`` with ``target = {"boxes" xyxy, "labels", optional "area", "iscrowd", "image_id"}``
    (what the reference's ``get_coco_api_from_dataset`` walks, utils/coco/coco_utils.py)."""
    rows = []
    for idx in range(len(dataset)):
        item = dataset[idx]
        target = item[1]
        image_id = int(target["image_id"]) if "image_id" in target else (int(item[2]) if len(item) > 2 else idx)
        boxes = convert_to_xywh(torch.as_tensor(target["boxes"], dtype=torch.float32).reshape(-1, 4)).tolist()
        labels = torch.as_tensor(target["labels"]).reshape(-1).tolist()
        areas = torch.as_tensor(target["area"]).reshape(-1).tolist() if "area" in target else [b[2] * b[3] for b in boxes]
        crowd = torch.as_tensor(target["iscrowd"]).reshape(-1).tolist() if "iscrowd" in target else [0] * len(boxes)
        rows.extend({"image_id": image_id, "category_id": int(l), "bbox": b, "area": float(a), "iscrowd": int(c)}
                    for b, l, a, c in zip(boxes, labels, areas, crowd))
    return rows


class CocoEvaluator:
    """Same call sequence as the reference's (coco_eval.py:14-58): ``update({image_id: detection dict})``
    per batch, ``synchronize_between_processes()``, ``accumulate()``, ``summarize()``; the result is read from
    ``coco_eval["bbox"].stats[0]`` (model.py:150-157).  ``gt``: annotation rows (``gt_from_dataset``)."""

    def __init__(self, gt: Iterable[dict], iou_types: Sequence[str] = ("bbox",)):
        assert isinstance(iou_types, (list, tuple))
        if any(t != "bbox" for t in iou_types):
            raise ValueError(f"only iou_type 'bbox' is supported (RetinaNet has no masks / keypoints), got {iou_types}")
        self.iou_types = list(iou_types)
        self.coco_eval = {"bbox": BBoxEval(gt)}
        self.img_ids: List[int] = []
        self.results: List[dict] = []

    def update(self, predictions: Mapping[int, Mapping[str, Tensor]]) -> None:
        self.img_ids.extend(int(i) for i in np.unique(list(predictions.keys())))
        self.results.extend(prepare_for_coco_detection(predictions))

    def synchronize_between_processes(self) -> None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            gathered: List = [None] * dist.get_world_size()
            dist.all_gather_object(gathered, (self.img_ids, self.results))
            self.img_ids = [i for ids, _ in gathered for i in ids]
            self.results = [r for _, res in gathered for r in res]

    def accumulate(self) -> None:
        ev = self.coco_eval["bbox"]
        ev.dt, ev.img_ids = self.results, sorted(set(self.img_ids))
        ev.evaluate()

    def summarize(self, verbose: bool = True) -> None:
        for iou_type, ev in self.coco_eval.items():
            if verbose:
                print("IoU metric: {}".format(iou_type))
            ev.summarize(verbose)
