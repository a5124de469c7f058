"""ctypes loader for the CPU oracle (``librn_oracle.so`` built from ``rn_oracle.c``).

TEST INFRASTRUCTURE ONLY: importable from ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg.  The product package
(``pytorch_retinanet_amd``) never imports this module.

All entry points take/return numpy arrays (C-contiguous).  See ``rn_oracle.c``
for the reference file:line each function restates.
"""
import ctypes as C
import math
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librn_oracle.so")
_lib = None


class Level(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("stride", C.c_int32), ("num_cell", C.c_int32)]


class LossParams(C.Structure):
    _fields_ = [("alpha", C.c_float), ("gamma", C.c_float), ("beta", C.c_float),
                ("logit_shift", C.c_float), ("log_eps", C.c_float), ("reg_w", C.c_float * 4)]


class DetectParams(C.Structure):
    _fields_ = [("score_thr", C.c_float), ("min_box", C.c_float), ("nms_thr", C.c_float),
                ("max_det", C.c_int32), ("reg_w", C.c_float * 4)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "rn_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "all"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.rno_anchors_count.restype = C.c_int64
        _lib.rno_nms.restype = C.c_int64
    return _lib


def _p(a: Optional[np.ndarray], t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def default_loss_params(alpha=0.25, gamma=2.0, beta=0.1, logit_shift=1.0, log_eps=1e-8,
                        reg_w=(1.0, 1.0, 1.0, 1.0)) -> LossParams:
    return LossParams(alpha, gamma, beta, logit_shift, log_eps, (C.c_float * 4)(*reg_w))


def default_detect_params(score_thr=0.05, min_box=1e-2, nms_thr=0.5, max_det=100,
                          reg_w=(1.0, 1.0, 1.0, 1.0)) -> DetectParams:
    return DetectParams(score_thr, min_box, nms_thr, max_det, (C.c_float * 4)(*reg_w))


def num_threads() -> int:
    return int(lib().rno_num_threads())


def set_num_threads(n: int) -> None:
    lib().rno_set_num_threads(int(n))


def cell_anchors(sizes: Sequence[float], ratios: Sequence[float]) -> np.ndarray:
    s = np.asarray(sizes, dtype=np.float64)
    r = np.asarray(ratios, dtype=np.float64)
    out = np.empty((len(s) * len(r), 4), dtype=np.float32)
    lib().rno_cell_anchors(_p(s, C.c_double), len(s), _p(r, C.c_double), len(r), _p(out, C.c_float))
    return out


def anchors_emit(levels: Sequence[Tuple[int, int, int]], cells: List[np.ndarray], offset: float = 0.0) -> np.ndarray:
    """levels: [(H, W, stride)], cells: per-level [num_cell,4] fp32."""
    L = len(levels)
    cells = [_f32(c) for c in cells]
    lv = (Level * L)(*[Level(h, w, s, c.shape[0]) for (h, w, s), c in zip(levels, cells)])
    n = lib().rno_anchors_count(lv, L)
    out = np.empty((n, 4), dtype=np.float32)
    ptrs = (C.POINTER(C.c_float) * L)(*[_p(c, C.c_float) for c in cells])
    lib().rno_anchors_emit(lv, L, ptrs, C.c_double(offset), _p(out, C.c_float))
    return out


def _gt_off(counts: Sequence[int]) -> np.ndarray:
    return np.concatenate([[0], np.cumsum(np.asarray(counts, dtype=np.int64))]).astype(np.int32)


def iou_match(anchors: np.ndarray, gt_list: List[np.ndarray], fg_thr=0.5, bg_thr=0.4):
    """anchors [A,4] (shared) or [B,A,4]; gt_list: per-image [T_b,4].  -> (matches i64[B,A], num_fg i32[B])."""
    anchors = _f32(anchors)
    B = len(gt_list)
    shared = anchors.ndim == 2
    A = anchors.shape[-2]
    gt = _f32(np.concatenate([np.asarray(g, dtype=np.float32).reshape(-1, 4) for g in gt_list], 0)) if B else np.zeros((0, 4), np.float32)
    off = _gt_off([np.asarray(g).reshape(-1, 4).shape[0] for g in gt_list])
    matches = np.empty((B, A), dtype=np.int64)
    nfg = np.zeros((B,), dtype=np.int32)
    lib().rno_iou_match(_p(anchors, C.c_float), C.c_int64(0 if shared else A * 4), _p(gt, C.c_float),
                        _p(off, C.c_int32), B, C.c_int64(A), C.c_float(fg_thr), C.c_float(bg_thr),
                        _p(matches, C.c_int64), _p(nfg, C.c_int32))
    return matches, nfg


def encode(gt: np.ndarray, anchors: np.ndarray, reg_w=(1.0, 1.0, 1.0, 1.0), log_eps=1e-8) -> np.ndarray:
    gt, anchors = _f32(gt), _f32(anchors)
    out = np.empty_like(gt)
    rw = np.asarray(reg_w, dtype=np.float32)
    lib().rno_encode(_p(gt, C.c_float), _p(anchors, C.c_float), C.c_int64(gt.shape[0]), _p(rw, C.c_float),
                     C.c_float(log_eps), _p(out, C.c_float))
    return out


def loss_fwd_bwd(cls: np.ndarray, box: np.ndarray, anchors: np.ndarray, gt_boxes: List[np.ndarray],
                 gt_labels: List[np.ndarray], matches: np.ndarray, params: Optional[LossParams] = None,
                 want_grads: bool = True):
    """-> dict(loss=[cls,reg], per_image=[B,2] (bb,clas), gcls, gbox)."""
    cls, box, anchors = _f32(cls), _f32(box), _f32(anchors)
    B, A, K = cls.shape
    shared = anchors.ndim == 2
    gtb = _f32(np.concatenate([np.asarray(g, dtype=np.float32).reshape(-1, 4) for g in gt_boxes], 0))
    gtl = np.ascontiguousarray(np.concatenate([np.asarray(l, dtype=np.int64).reshape(-1) for l in gt_labels], 0))
    off = _gt_off([np.asarray(g).reshape(-1, 4).shape[0] for g in gt_boxes])
    matches = np.ascontiguousarray(matches, dtype=np.int64)
    params = params or default_loss_params()
    out = np.zeros(2, dtype=np.float32)
    per = np.zeros((B, 2), dtype=np.float32)
    gcls = np.empty_like(cls) if want_grads else None
    gbox = np.empty_like(box) if want_grads else None
    lib().rno_loss_fwd_bwd(_p(cls, C.c_float), _p(box, C.c_float), B, C.c_int64(A), K,
                           _p(anchors, C.c_float), C.c_int64(0 if shared else A * 4),
                           _p(gtb, C.c_float), _p(gtl, C.c_int64), _p(off, C.c_int32),
                           _p(matches, C.c_int64), C.byref(params), _p(out, C.c_float), _p(per, C.c_float),
                           _p(gcls, C.c_float), _p(gbox, C.c_float))
    return {"loss": out, "per_image": per, "gcls": gcls, "gbox": gbox}


def decode_clip(deltas: np.ndarray, anchors: np.ndarray, image_hw: Optional[Sequence[Tuple[int, int]]] = None,
                reg_w=(1.0, 1.0, 1.0, 1.0)) -> np.ndarray:
    deltas, anchors = _f32(deltas), _f32(anchors)
    squeeze = deltas.ndim == 2
    if squeeze:
        deltas = deltas[None]
    B, A, _ = deltas.shape
    shared = anchors.ndim == 2
    hw = None if image_hw is None else np.ascontiguousarray(np.asarray(image_hw, dtype=np.int32).reshape(B, 2))
    rw = np.asarray(reg_w, dtype=np.float32)
    out = np.empty_like(deltas)
    lib().rno_decode_clip(_p(deltas, C.c_float), B, C.c_int64(A), _p(anchors, C.c_float),
                          C.c_int64(0 if shared else A * 4), _p(hw, C.c_int32), _p(rw, C.c_float), _p(out, C.c_float))
    return out[0] if squeeze else out


def nms(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    boxes, scores = _f32(boxes).reshape(-1, 4), _f32(scores).reshape(-1)
    n = boxes.shape[0]
    keep = np.empty((max(n, 1),), dtype=np.int64)
    k = lib().rno_nms(_p(boxes, C.c_float), _p(scores, C.c_float), C.c_int64(n), C.c_float(thr), _p(keep, C.c_int64))
    return keep[:k].copy()


def detect(cls: np.ndarray, deltas: np.ndarray, anchors: np.ndarray, image_hw: Sequence[Tuple[int, int]],
           params: Optional[DetectParams] = None):
    """-> list of dict(boxes [n,4], scores [n], labels i64[n]) per image."""
    cls, deltas, anchors = _f32(cls), _f32(deltas), _f32(anchors)
    B, A, K = cls.shape
    shared = anchors.ndim == 2
    params = params or default_detect_params()
    hw = np.ascontiguousarray(np.asarray(image_hw, dtype=np.int32).reshape(B, 2))
    md = params.max_det
    ob = np.zeros((B, md, 4), np.float32)
    os_ = np.zeros((B, md), np.float32)
    ol = np.zeros((B, md), np.int64)
    oc = np.zeros((B,), np.int32)
    lib().rno_detect(_p(cls, C.c_float), _p(deltas, C.c_float), B, C.c_int64(A), K, _p(anchors, C.c_float),
                     C.c_int64(0 if shared else A * 4), _p(hw, C.c_int32), C.byref(params),
                     _p(ob, C.c_float), _p(os_, C.c_float), _p(ol, C.c_int64), _p(oc, C.c_int32))
    return [{"boxes": ob[b, :oc[b]].copy(), "scores": os_[b, :oc[b]].copy(), "labels": ol[b, :oc[b]].copy()}
            for b in range(B)]


def transform_sizes(h: int, w: int, min_size: float, max_size: float) -> Tuple[int, int]:
    """Resized (h, w) as torchvision's GeneralizedRCNNTransform.resize computes them: scale =
    min(min_size / short, max_size / long) in double, size = floor(side * scale)."""
    lo, hi = float(min(h, w)), float(max(h, w))
    scale = float(min_size) / lo
    if hi * scale > float(max_size):
        scale = float(max_size) / hi
    return int(math.floor(h * scale)), int(math.floor(w * scale))


def transform_batch(images: Sequence[np.ndarray], min_size: float, max_size: float, mean: Sequence[float],
                    std: Sequence[float], size_divisible: int = 32):
    """T1 -> (batch f32 [B,3,Hp,Wp], [(h, w) after the resize])."""
    imgs = [_f32(im) for im in images]
    B = len(imgs)
    in_hw = np.asarray([[im.shape[1], im.shape[2]] for im in imgs], dtype=np.int32)
    sizes = [transform_sizes(int(h), int(w), min_size, max_size) for h, w in in_hw]
    out_hw = np.asarray(sizes, dtype=np.int32).reshape(B, 2)
    d = float(size_divisible)
    Hp = int(math.ceil(max(s[0] for s in sizes) / d) * d)
    Wp = int(math.ceil(max(s[1] for s in sizes) / d) * d)
    out = np.empty((B, 3, Hp, Wp), np.float32)
    ptrs = (C.c_void_p * B)(*[im.ctypes.data for im in imgs])
    lib().rno_transform_batch(ptrs, _p(in_hw, C.c_int32), _p(out_hw, C.c_int32), B, _p(_f32(np.asarray(mean)), C.c_float),
                              _p(_f32(np.asarray(std)), C.c_float), Hp, Wp, _p(out, C.c_float))
    return out, sizes
