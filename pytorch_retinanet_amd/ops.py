"""Tensor-level wrappers over the C ABI (``include/retinanet_hip.h``).

PyTorch is plumbing here: it owns device memory and the stream.  Every wrapper
passes ``tensor.data_ptr()`` and ``torch.cuda.current_stream().cuda_stream`` to
the HIP library, never synchronises with the host, and REFUSES CPU tensors --
there is no CPU implementation of this path in the product.
"""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import RN_BF16, RN_F16, RN_F32, RnDetectParams, RnLevel, RnLossParams, check, lib

RN_MATCH_NUM_FG_ZEROED, RN_MATCH_FLAGGED_ONLY = 1, 2          # include/retinanet_hip.h
LOSS_FORM_CHUNKS, LOSS_FORM_REPAIR_PASS, LOSS_FORM_LIST = 0, 1, 2   # RN_LOSS_FORM_*

_DT = {torch.float32: RN_F32, torch.bfloat16: RN_BF16, torch.float16: RN_F16}


# Optional per-op device timing (bench.py): name -> list of (start_event, end_event) recorded on the
# stream the kernels are launched on (torch's current stream).  None = disabled (default).
_TIMERS = None


def enable_timing(on: bool = True) -> None:
    global _TIMERS
    _TIMERS = {} if on else None


def timing_events():
    return _TIMERS


class _timed:
    def __init__(self, name: str, dev: torch.device):
        self.name, self.dev = name, dev

    def __enter__(self):
        if _TIMERS is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream(self.dev))
        return self

    def __exit__(self, *exc):
        if _TIMERS is not None:
            self.e1.record(torch.cuda.current_stream(self.dev))
            _TIMERS.setdefault(self.name, []).append((self.e0, self.e1))
        return False


def _need_dev(*tensors: Tensor) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "pytorch_retinanet_amd: the dense-head path runs only as HIP kernels on an MI355X; "
                f"got a {t.device} tensor (there is no CPU fallback).")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def _stream(dev: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _ptr(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None or t.numel() == 0 else t.data_ptr())


def _dtype_code(t: Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}; expected float32, bfloat16 or float16")


def _c(t: Tensor) -> Tensor:
    """Dense and 16-byte aligned (the kernels use 16-byte vector accesses): views into the middle of a
    tensor (e.g. ``cls_preds[1]``) are copied to a fresh allocation."""
    if not t.is_contiguous():
        return t.contiguous()
    if t.numel() and t.data_ptr() % 16:
        return t.clone()
    return t


def _anchor_args(anchors: Tensor, B: int, A: int) -> Tuple[Tensor, int]:
    """anchors [A,4] shared or [B,A,4] per image -> (contiguous fp32 tensor, batch stride in elements)."""
    anchors = _c(anchors)
    if anchors.dtype != torch.float32:
        anchors = anchors.float()
    if anchors.dim() == 2:
        if anchors.shape != (A, 4):
            raise ValueError(f"anchors shape {tuple(anchors.shape)} != ({A}, 4)")
        return anchors, 0
    if anchors.shape != (B, A, 4):
        raise ValueError(f"anchors shape {tuple(anchors.shape)} != ({B}, {A}, 4)")
    return anchors, A * 4


# --------------------------------------------------------------------------- #
def anchors_emit(levels: Sequence[Tuple[int, int, int]], cells: Sequence[Tensor], offset: float) -> Tensor:
    """K1.  levels: [(H, W, stride)], cells: per-level device f32 [num_cell,4].  -> f32 [A,4]."""
    L = len(levels)
    if L == 0 or L > _lib.RN_MAX_LEVELS or len(cells) != L:
        raise ValueError("need 1..8 levels and one cell-anchor tensor per level")
    dev = _need_dev(*cells)
    cells = [_c(c.float()) for c in cells]
    lv = (RnLevel * L)(*[RnLevel(int(h), int(w), int(s), int(c.shape[0])) for (h, w, s), c in zip(levels, cells)])
    total = lib.rn_anchors_count(lv, L)
    out = torch.empty((total, 4), dtype=torch.float32, device=dev)
    ptrs = (C.c_void_p * L)(*[c.data_ptr() for c in cells])
    with torch.cuda.device(dev):
        check(lib.rn_anchors_emit(lv, L, ptrs, float(offset), _ptr(out), _stream(dev)), "rn_anchors_emit")
    return out


_GT_OFF_CACHE = {}          # (counts, device) -> device int32[B+1]; bounded (see gt_offsets)


def gt_offsets(counts: Sequence[int], device: torch.device) -> Tensor:
    """Prefix offsets of the per-image GT rows as a device int32[B+1].  Cached per (counts, device): training data repeats
    a handful of count tuples, the upload disappears from the step, and a step captured in a hipGraph (``graph.py``) never
    copies from temporary host memory.  Treat the result as read-only."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("pytorch_retinanet_amd: gt offsets are a device array (there is no CPU fallback)")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (tuple(int(c) for c in counts), device.index)
    hit = _GT_OFF_CACHE.get(key)
    if hit is not None:
        return hit
    off = [0]
    for c in key[0]:
        off.append(off[-1] + c)
    host = torch.tensor(off, dtype=torch.int32)
    # pinned + non_blocking: the copy is stream-ordered and does not drain the launch queue
    dev_t = host.pin_memory().to(device, non_blocking=True)
    if len(_GT_OFF_CACHE) >= 4096:
        _GT_OFF_CACHE.clear()
    _GT_OFF_CACHE[key] = dev_t
    return dev_t


def gt_pack(boxes: Sequence[Tensor], labels: Sequence[Tensor], dev: torch.device):
    """The ragged GT of a batch as the library takes it -- (gt_boxes f32 [sum T, 4], gt_labels i64 [sum T], gt_off i32 [B + 1]) -- plus a
    ZEROED ``num_fg`` i32 [B] for K2, all written by ONE launch (``rn_copy_many``: 2 B + 1 small copies, the last from the zero page)
    when the per-image tensors already are f32 / i64 CUDA tensors; else two ``torch.cat`` and num_fg = None (K2 then clears it).
    Reference: the per-image ``targets[i]["boxes"]`` / ``["labels"]`` of ``RetinaNetLosses.forward`` (losses.py:113-145)."""
    counts = [int(b.reshape(-1, 4).shape[0]) for b in boxes]
    gt_off = gt_offsets(counts, dev)
    B, total = len(counts), sum(counts)
    bs = [b.reshape(-1, 4) for b in boxes]
    ls = [l.reshape(-1) for l in labels]
    fast = total > 0 and 2 * B + 1 <= 64 and 4 * B <= 256 and all(b.is_cuda and b.device == dev and b.dtype == torch.float32 and b.is_contiguous() for b in bs) \
        and all(l.is_cuda and l.device == dev and l.dtype == torch.int64 and l.is_contiguous() for l in ls)
    if not fast:
        gt_boxes = torch.cat([b.to(device=dev, dtype=torch.float32) for b in bs]) if B else torch.zeros((0, 4), device=dev)
        gt_labels = torch.cat([l.to(device=dev, dtype=torch.int64) for l in ls]) if B else torch.zeros((0,), dtype=torch.int64, device=dev)
        return gt_boxes, gt_labels, gt_off, None
    gt_boxes = torch.empty((total, 4), dtype=torch.float32, device=dev)
    gt_labels = torch.empty((total,), dtype=torch.int64, device=dev)
    num_fg = torch.empty((B,), dtype=torch.int32, device=dev)
    from .biasact import _zero_page
    srcs, dsts, nb, o = [], [], [], 0
    for b, l, c in zip(bs, ls, counts):
        if c:
            srcs += [b.data_ptr(), l.data_ptr()]
            dsts += [gt_boxes.data_ptr() + 16 * o, gt_labels.data_ptr() + 8 * o]
            nb += [16 * c, 8 * c]
            o += c
    srcs.append(_zero_page(dev).data_ptr()); dsts.append(num_fg.data_ptr()); nb.append(4 * B)
    n = len(srcs)
    with torch.cuda.device(dev), _timed("gt_pack", dev):
        check(lib.rn_copy_many((C.c_void_p * n)(*srcs), (C.c_void_p * n)(*dsts), (C.c_int64 * n)(*nb), n, _stream(dev)), "rn_copy_many")
    return gt_boxes, gt_labels, gt_off, num_fg


def iou_match(anchors: Tensor, gt_boxes: Tensor, gt_off: Tensor, B: int, fg_thr: float, bg_thr: float,
              want_num_fg: bool = True, want_special: bool = False, out: Optional[tuple] = None, flagged_only: bool = False,
              zeroed_num_fg: Optional[Tensor] = None):
    """K2.  anchors [A,4] or [B,A,4]; gt_boxes f32 [sum T,4]; gt_off i32 [B+1] (device).
    -> (matches i64 [B,A], num_fg i32 [B]) and, with ``want_special``, a third tensor ``special`` i64 [B, ceil(A/64)]:
    bit (a & 63) of word a >> 6 is set where ``matches[b, a] != -1`` (what ``loss_fwd_bwd_levels(special=...)`` reads instead of
    streaming ``matches``).  ``out``: pre-allocated (matches, num_fg, special) -- for a caller that launches K2 on a side
    stream and wants the outputs to belong to its main stream (``losses.RetinaNetLosses.match_ahead``).
    ``flagged_only`` (needs ``want_special``): ``matches`` is written only where a flag bit is set -- every other entry is
    UNINITIALISED; for results that go straight into ``loss_fwd_bwd_levels(special=...)`` (RN_MATCH_FLAGGED_ONLY).
    ``zeroed_num_fg``: an i32 [B] tensor that already holds zeros (``gt_pack``): used as ``num_fg``, no clear launch."""
    dev = _need_dev(anchors, gt_boxes, gt_off)
    A = anchors.shape[-2]
    anchors, bstride = _anchor_args(anchors, B, A)
    gt_boxes = _c(gt_boxes.float()).reshape(-1, 4)
    if not fg_thr > bg_thr:
        raise AssertionError("match_thr must be greater than back_thr")   # box_utils.py:66
    if out is not None:
        matches, num_fg, special = out
    else:
        matches, num_fg, special = iou_match_outputs(B, A, dev, want_num_fg and zeroed_num_fg is None, want_special)
    flags = 0
    if zeroed_num_fg is not None:
        num_fg, flags = zeroed_num_fg, flags | RN_MATCH_NUM_FG_ZEROED
    if flagged_only:
        if special is None:
            raise ValueError("flagged_only needs the flag words (want_special=True)")
        flags |= RN_MATCH_FLAGGED_ONLY
    with torch.cuda.device(dev), _timed("iou_match", dev):
        # gt_off[B] - gt_off[0] == the row count of gt_boxes (host-known): lets the library pick the batch-shaped kernel
        check(lib.rn_iou_match_special_ex(_ptr(anchors), bstride, _ptr(gt_boxes), _ptr(gt_off), B, A, fg_thr, bg_thr,
                                          _ptr(matches), _ptr(num_fg), _ptr(special), int(gt_boxes.shape[0]), flags, _stream(dev)),
              "rn_iou_match_special_ex")
    return (matches, num_fg, special) if (want_special or out is not None) else (matches, num_fg)


def iou_match_outputs(B: int, A: int, dev: torch.device, want_num_fg: bool = True, want_special: bool = True):
    "Uninitialised output tensors of ``iou_match`` (matches, num_fg, special), allocated on the current stream."
    matches = torch.empty((B, A), dtype=torch.int64, device=dev)
    num_fg = torch.empty((B,), dtype=torch.int32, device=dev) if want_num_fg else None
    special = torch.empty((B, (A + 63) // 64), dtype=torch.int64, device=dev) if want_special else None
    return matches, num_fg, special


def make_loss_params(alpha: float, gamma: float, beta: float, logit_shift: float = 1.0, log_eps: float = 1e-8,
                     reg_w: Sequence[float] = (1.0, 1.0, 1.0, 1.0)) -> RnLossParams:
    return RnLossParams(alpha, gamma, beta, logit_shift, log_eps, (C.c_float * 4)(*reg_w))


def loss_fwd_bwd(cls: Tensor, box: Tensor, anchors: Tensor, gt_boxes: Tensor, gt_labels: Tensor, gt_off: Tensor,
                 matches: Tensor, num_fg: Tensor, params: RnLossParams, want_grad: bool = True):
    """K3.  cls [B,A,K], box [B,A,4] (same dtype) -> (loss f32[2] = (cls, reg), grad_cls, grad_box)."""
    dev = _need_dev(cls, box, anchors, gt_boxes, gt_labels, gt_off, matches, num_fg)
    if cls.dim() != 3 or box.dim() != 3 or box.shape[-1] != 4 or cls.shape[:2] != box.shape[:2]:
        raise ValueError(f"bad head output shapes {tuple(cls.shape)} / {tuple(box.shape)}")
    if box.dtype != cls.dtype:
        box = box.to(cls.dtype)
    B, A, K = cls.shape
    cls, box = _c(cls), _c(box)
    anchors, bstride = _anchor_args(anchors, B, A)
    gt_boxes = _c(gt_boxes.float()).reshape(-1, 4)
    gt_labels = _c(gt_labels.to(torch.int64)).reshape(-1)
    code = _dtype_code(cls)
    out = torch.empty((2,), dtype=torch.float32, device=dev)
    gcls = torch.empty_like(cls) if want_grad else None
    gbox = torch.empty_like(box) if want_grad else None
    ws_bytes = lib.rn_loss_workspace_bytes(B, A, K)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev), _timed("loss_fwd_bwd" if want_grad else "loss_fwd", dev):
        check(lib.rn_loss_fwd_bwd(_ptr(cls), _ptr(box), code, B, A, K, _ptr(anchors), bstride, _ptr(gt_boxes),
                                  _ptr(gt_labels), _ptr(gt_off), _ptr(matches), _ptr(num_fg), C.byref(params),
                                  _ptr(out), _ptr(gcls), _ptr(gbox), _ptr(ws), ws_bytes, _stream(dev)),
              "rn_loss_fwd_bwd")
    return out, gcls, gbox


def loss_fwd_bwd_levels(cls_levels: Sequence[Tensor], box_levels: Sequence[Tensor], anchors: Tensor, gt_boxes: Tensor,
                        gt_labels: Tensor, gt_off: Tensor, matches: Tensor, num_fg: Tensor, params: RnLossParams,
                        want_grad: bool = True, special: Optional[Tensor] = None, in_kernel_finalize: bool = False,
                        grad_prescale: Optional[Tensor] = None, repair_pass: bool = False, form: Optional[int] = None):
    """K3 on per-level head outputs (no concatenation): cls_levels[l] [B,A_l,K], box_levels[l] [B,A_l,4].
    -> (loss f32[2], [grad_cls_l], [grad_box_l]).  ``special``: the third output of ``iou_match(want_special=True)`` --
    the kernel then reads ``matches`` only at the rows flagged there instead of streaming all of it.
    ``in_kernel_finalize``: no finalize launch -- the streaming kernel's workgroup 0 writes the two losses
    (``rn_loss_fwd_bwd_levels_fin``; same bits; uses this device's / capture's state words, ``_match_state``).
    ``grad_prescale``: device f32 scalar every gradient is multiplied by BEFORE its rounding to the I/O dtype (a GradScaler's
    scale: fp16 class gradients of ~4e-10 would otherwise flush to zero).  ``form`` (``rn_loss_fwd_bwd_levels_rp``):
    ``LOSS_FORM_CHUNKS`` (default; the one-launch kernel repairing its special rows chunk by chunk), ``LOSS_FORM_LIST`` (one launch, the
    special rows of a wave through one compact list: faster from ~32 GT boxes per image on) or ``LOSS_FORM_REPAIR_PASS`` (= ``repair_pass``;
    needs ``special``: a pure background stream + a repair kernel over the flag words; slower, kept for A/B)."""
    if form is None:
        form = LOSS_FORM_REPAIR_PASS if repair_pass else LOSS_FORM_CHUNKS
    repair_pass = form == LOSS_FORM_REPAIR_PASS
    L = len(cls_levels)
    if L == 0 or L > _lib.RN_MAX_LEVELS or len(box_levels) != L:
        raise ValueError("need 1..8 levels of (cls, box) outputs")
    dev = _need_dev(*cls_levels, *box_levels, anchors, gt_boxes, gt_labels, gt_off, matches, num_fg)
    dt = cls_levels[0].dtype
    cls_levels = [_c(c) for c in cls_levels]
    box_levels = [_c(b if b.dtype == dt else b.to(dt)) for b in box_levels]
    B, _, K = cls_levels[0].shape
    counts = [int(c.shape[1]) for c in cls_levels]
    for c, b in zip(cls_levels, box_levels):
        if c.dim() != 3 or b.shape != (B, c.shape[1], 4) or c.shape[0] != B or c.shape[2] != K or c.dtype != dt:
            raise ValueError(f"bad level shapes {tuple(c.shape)} / {tuple(b.shape)}")
    A = sum(counts)
    anchors, bstride = _anchor_args(anchors, B, A)
    gt_boxes = _c(gt_boxes.float()).reshape(-1, 4)
    gt_labels = _c(gt_labels.to(torch.int64)).reshape(-1)
    out = torch.empty((2,), dtype=torch.float32, device=dev)
    gcls = [torch.empty_like(c) for c in cls_levels] if want_grad else None
    gbox = [torch.empty_like(b) for b in box_levels] if want_grad else None
    ws_bytes = lib.rn_loss_workspace_bytes(B, A, K)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    arr = lambda ts: (C.c_void_p * L)(*[t.data_ptr() for t in ts])
    # with timing on, the library itself records a pair of events right around the streaming kernel (the dominant kernel:
    # what bench.py's roofline is quoted on); the outer pair also covers the one-block finalize
    k0 = k1 = None
    if _TIMERS is not None:
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record(torch.cuda.current_stream(dev)); k1.record(torch.cuda.current_stream(dev))      # creates the handles
    with torch.cuda.device(dev), _timed("loss_fwd_bwd" if want_grad else "loss_fwd", dev):
        if special is not None and (special.dtype != torch.int64 or tuple(special.shape) != (B, (A + 63) // 64) or not special.is_contiguous()):
            raise ValueError(f"special-row words: expected int64 [{B}, {(A + 63) // 64}], got {special.dtype} {tuple(special.shape)}")
        if repair_pass and special is None:
            raise ValueError("repair_pass needs the flag words of iou_match(want_special=True)")
        if grad_prescale is not None:
            if not (grad_prescale.is_cuda and grad_prescale.dtype == torch.float32 and grad_prescale.numel() == 1):
                raise TypeError("grad_prescale must be a CUDA fp32 scalar")
            _need_dev(grad_prescale, cls_levels[0])
        if form != LOSS_FORM_CHUNKS or grad_prescale is not None:
            check(lib.rn_loss_fwd_bwd_levels_rp(arr(cls_levels), arr(box_levels), (C.c_int64 * L)(*counts), L, _dtype_code(cls_levels[0]),
                                                 B, K, _ptr(anchors), bstride, _ptr(gt_boxes), _ptr(gt_labels), _ptr(gt_off),
                                                 _ptr(matches), _ptr(special), _ptr(num_fg), C.byref(params), _ptr(grad_prescale),
                                                 int(form), _ptr(out),
                                                 arr(gcls) if want_grad else None, arr(gbox) if want_grad else None,
                                                 _ptr(ws), ws_bytes, _ptr(_match_state(dev)), _stream(dev), k0.cuda_event if k0 else None,
                                                 k1.cuda_event if k1 else None), "rn_loss_fwd_bwd_levels_rp")
        elif in_kernel_finalize:
            check(lib.rn_loss_fwd_bwd_levels_fin(arr(cls_levels), arr(box_levels), (C.c_int64 * L)(*counts), L, _dtype_code(cls_levels[0]),
                                                  B, K, _ptr(anchors), bstride, _ptr(gt_boxes), _ptr(gt_labels), _ptr(gt_off),
                                                  _ptr(matches), _ptr(special), _ptr(num_fg), C.byref(params), _ptr(out),
                                                  arr(gcls) if want_grad else None, arr(gbox) if want_grad else None,
                                                  _ptr(ws), ws_bytes, _ptr(_match_state(dev)), _stream(dev), k0.cuda_event if k0 else None,
                                                  k1.cuda_event if k1 else None), "rn_loss_fwd_bwd_levels_fin")
        else:
            check(lib.rn_loss_fwd_bwd_levels_ex(arr(cls_levels), arr(box_levels), (C.c_int64 * L)(*counts), L, _dtype_code(cls_levels[0]),
                                                 B, K, _ptr(anchors), bstride, _ptr(gt_boxes), _ptr(gt_labels), _ptr(gt_off),
                                                 _ptr(matches), _ptr(special), _ptr(num_fg), C.byref(params), _ptr(out),
                                                 arr(gcls) if want_grad else None, arr(gbox) if want_grad else None,
                                                 _ptr(ws), ws_bytes, _stream(dev), k0.cuda_event if k0 else None,
                                                 k1.cuda_event if k1 else None), "rn_loss_fwd_bwd_levels_ex")
    if k0 is not None:
        _TIMERS.setdefault("loss_stream_kernel" if want_grad else "loss_stream_kernel_fwd", []).append((k0, k1))
    return out, gcls, gbox


# ---- K2 + K3 in one launch -------------------------------------------------------------------------------------------------
# State words of rn_loss_match_fwd_bwd_levels (grid-barrier counter + per-image foreground counters): zero-filled ONCE, left
# zero-filled by every completed call, never shared by two streams (one buffer per device and stream).  A hipGraph capture gets
# its own buffer, allocated BEFORE the capture begins (``new_match_state`` / ``use_match_state``: a buffer allocated inside the
# capture would come with a zero-fill node that replays every step).
MATCH_STATE_IMAGES = 4096
RN_EUNSUPPORTED = -4
_MATCH_STATE = {}
_MATCH_STATE_OVERRIDE = None


def new_match_state(dev: torch.device) -> Tensor:
    return torch.zeros((16 + MATCH_STATE_IMAGES,), dtype=torch.int32, device=dev)


class use_match_state:
    "``with use_match_state(buf):`` -- fused loss calls on ``buf.device`` inside the block use ``buf`` (graph capture)."

    def __init__(self, buf: Optional[Tensor]):
        self.buf = buf

    def __enter__(self):
        global _MATCH_STATE_OVERRIDE
        self.prev, _MATCH_STATE_OVERRIDE = _MATCH_STATE_OVERRIDE, self.buf
        return self

    def __exit__(self, *exc):
        global _MATCH_STATE_OVERRIDE
        _MATCH_STATE_OVERRIDE = self.prev
        return False


def reset_match_state(dev: Optional[torch.device] = None) -> None:
    """Drop the cached state buffers (all devices, or ``dev``'s): the next loss call allocates zero-filled ones.  For after a POISONED call
    -- the in-kernel finalize met a launch whose workgroups did not all arrive (a faulted launch) and set the sticky word: every later
    call on that state returns NaN losses, loudly, instead of finishing early on stale counters."""
    for key in [k for k in _MATCH_STATE if dev is None or k[0] == torch.device(dev).index]:
        del _MATCH_STATE[key]


def _match_state(dev: torch.device) -> Tensor:
    if _MATCH_STATE_OVERRIDE is not None and _MATCH_STATE_OVERRIDE.device == dev:
        return _MATCH_STATE_OVERRIDE
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    st = _MATCH_STATE.get(key)
    if st is None:
        st = _MATCH_STATE[key] = new_match_state(dev)
    return st


def loss_match_fwd_bwd_levels(cls_levels: Sequence[Tensor], box_levels: Sequence[Tensor], anchors: Tensor, gt_boxes: Tensor,
                              gt_labels: Tensor, gt_off: Tensor, max_gt: int, fg_thr: float, bg_thr: float, params: RnLossParams,
                              want_grad: bool = True, want_matches: bool = False):
    """K2 + K3 in ONE launch (``rn_loss_match_fwd_bwd_levels``): the matcher runs in the loss kernel's prologue, ``matches`` is
    written only when ``want_matches``.  -> (loss f32[2], [grad_cls_l], [grad_box_l], num_fg i32[B], matches or None), or
    ``None`` when the library declines the shape (more than 64 GT boxes in an image, ranges that do not fit its lists): the
    caller then runs ``iou_match`` + ``loss_fwd_bwd_levels``.  ``max_gt``: the largest per-image GT count (host-known)."""
    L = len(cls_levels)
    if L == 0 or L > _lib.RN_MAX_LEVELS or len(box_levels) != L:
        raise ValueError("need 1..8 levels of (cls, box) outputs")
    dev = _need_dev(*cls_levels, *box_levels, anchors, gt_boxes, gt_labels, gt_off)
    if not fg_thr > bg_thr:
        raise AssertionError("match_thr must be greater than back_thr")   # box_utils.py:66
    B, _, K = cls_levels[0].shape
    if max_gt > 64 or B > MATCH_STATE_IMAGES:
        return None
    dt = cls_levels[0].dtype
    cls_levels = [_c(c) for c in cls_levels]
    box_levels = [_c(b if b.dtype == dt else b.to(dt)) for b in box_levels]
    counts = [int(c.shape[1]) for c in cls_levels]
    for c, b in zip(cls_levels, box_levels):
        if c.dim() != 3 or b.shape != (B, c.shape[1], 4) or c.shape[0] != B or c.shape[2] != K or c.dtype != dt:
            raise ValueError(f"bad level shapes {tuple(c.shape)} / {tuple(b.shape)}")
    A = sum(counts)
    anchors, bstride = _anchor_args(anchors, B, A)
    gt_boxes = _c(gt_boxes.float()).reshape(-1, 4)
    gt_labels = _c(gt_labels.to(torch.int64)).reshape(-1)
    out = torch.empty((2,), dtype=torch.float32, device=dev)
    gcls = [torch.empty_like(c) for c in cls_levels] if want_grad else None
    gbox = [torch.empty_like(b) for b in box_levels] if want_grad else None
    num_fg = torch.empty((B,), dtype=torch.int32, device=dev)
    matches = torch.empty((B, A), dtype=torch.int64, device=dev) if want_matches else None
    ws_bytes = lib.rn_loss_workspace_bytes(B, A, K)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    state = _match_state(dev)
    arr = lambda ts: (C.c_void_p * L)(*[t.data_ptr() for t in ts])
    k0 = k1 = None
    if _TIMERS is not None:
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record(torch.cuda.current_stream(dev)); k1.record(torch.cuda.current_stream(dev))      # creates the handles
    with torch.cuda.device(dev), _timed("loss_fwd_bwd" if want_grad else "loss_fwd", dev):
        rc = lib.rn_loss_match_fwd_bwd_levels(arr(cls_levels), arr(box_levels), (C.c_int64 * L)(*counts), L, _dtype_code(cls_levels[0]),
                                              B, K, _ptr(anchors), bstride, _ptr(gt_boxes), _ptr(gt_labels), _ptr(gt_off), int(max_gt),
                                              fg_thr, bg_thr, _ptr(matches), _ptr(num_fg), C.byref(params), _ptr(out),
                                              arr(gcls) if want_grad else None, arr(gbox) if want_grad else None, _ptr(ws), ws_bytes,
                                              _ptr(state), state.numel() * 4, _stream(dev), k0.cuda_event if k0 else None,
                                              k1.cuda_event if k1 else None)
        if rc == RN_EUNSUPPORTED:
            return None
        check(rc, "rn_loss_match_fwd_bwd_levels")
    if k0 is not None:
        _TIMERS.setdefault("loss_stream_kernel" if want_grad else "loss_stream_kernel_fwd", []).append((k0, k1))
        _TIMERS.setdefault("loss_stream_kernel_fused_match", []).append((k0, k1))
    return out, gcls, gbox, num_fg, matches


def scale_inplace(t: Tensor, scale: Tensor) -> Tensor:
    """t *= scale (device scalar, f32); a no-op on the device when scale == 1."""
    dev = _need_dev(t, scale)
    if not t.is_contiguous():
        raise ValueError("scale_inplace needs a contiguous tensor")
    scale = _c(scale.detach().to(torch.float32)).reshape(1)
    with torch.cuda.device(dev):
        check(lib.rn_scale_inplace(_ptr(t), _dtype_code(t), t.numel(), _ptr(scale), _stream(dev)), "rn_scale_inplace")
    return t


def scale_inplace_batched(tensors: Sequence[Tensor], scales: Sequence[Tensor]) -> None:
    """tensors[k] *= scales[k] (device scalars, f32[1]) for up to 16 contiguous tensors per launch, grouped by dtype; a no-op on
    the device where a scale is 1."""
    if not tensors:
        return
    dev = _need_dev(*tensors, *scales)
    scales = [_c(s.detach().to(torch.float32)).reshape(1) for s in scales]
    by_dtype = {}
    for t, s in zip(tensors, scales):
        if not t.is_contiguous():
            raise ValueError("scale_inplace_batched needs contiguous tensors")
        if t.numel():
            by_dtype.setdefault(t.dtype, []).append((t, s))
    with torch.cuda.device(dev):
        for dt, items in by_dtype.items():
            for i in range(0, len(items), 16):
                part = items[i:i + 16]
                n = len(part)
                check(lib.rn_scale_inplace_batched((C.c_void_p * n)(*[_ptr(t) for t, _ in part]), (C.c_int64 * n)(*[t.numel() for t, _ in part]),
                                                   (C.c_void_p * n)(*[_ptr(s) for _, s in part]), n, _dtype_code(part[0][0]), _stream(dev)),
                      "rn_scale_inplace_batched")


def image_hw_tensor(sizes: Sequence[Tuple[int, int]], device: torch.device) -> Tensor:
    host = torch.tensor([[int(h), int(w)] for h, w in sizes], dtype=torch.int32)
    return host.pin_memory().to(device, non_blocking=True)


def transform_batch(images: Sequence[Tensor], out_sizes: Sequence[Tuple[int, int]], mean: Sequence[float],
                    std: Sequence[float], Hp: int, Wp: int, out_dtype: torch.dtype = torch.float32,
                    channels_last: bool = False) -> Tensor:
    """T1: (x - mean) / std -> bilinear resize of image b to out_sizes[b] -> zero-padded batch
    ``[B, 3, Hp, Wp]`` (``out_dtype``; channels_last memory format on request) in one launch.
    images: CUDA f32 ``[3, h, w]`` tensors; Wp % 4 == 0."""
    dev = _need_dev(*images)
    B = len(images)
    if B == 0 or len(out_sizes) != B:
        raise ValueError("need one output size per image")
    imgs = []
    for im in images:
        if im.dim() != 3 or im.shape[0] != 3 or im.dtype != torch.float32:
            raise ValueError(f"transform_batch expects f32 [3, h, w] images, got {tuple(im.shape)} {im.dtype}")
        imgs.append(im if im.is_contiguous() else im.contiguous())
    if out_dtype not in _DT:
        raise ValueError(f"unsupported output dtype {out_dtype}")
    out = torch.empty((B, 3, Hp, Wp), dtype=out_dtype, device=dev,
                      memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    ptrs = (C.c_void_p * B)(*[im.data_ptr() for im in imgs])
    in_hw = (C.c_int32 * (2 * B))(*[int(v) for im in imgs for v in im.shape[1:]])
    out_hw = (C.c_int32 * (2 * B))(*[int(v) for s in out_sizes for v in s])
    with torch.cuda.device(dev), _timed("transform_batch", dev):
        check(lib.rn_transform_batch(ptrs, in_hw, out_hw, B, (C.c_float * 3)(*mean), (C.c_float * 3)(*std), int(Hp), int(Wp),
                                     _ptr(out), _DT[out_dtype], int(bool(channels_last)), _stream(dev)), "rn_transform_batch")
    return out


def decode_clip(deltas: Tensor, anchors: Tensor, image_hw: Optional[Tensor],
                reg_w: Sequence[float] = (1.0, 1.0, 1.0, 1.0)) -> Tensor:
    """K4.  deltas [B,A,4] or [A,4]; image_hw i32 [B,2] (device) or None.  -> f32 boxes, same leading shape."""
    dev = _need_dev(deltas, anchors, image_hw)
    squeeze = deltas.dim() == 2
    d = _c(deltas[None] if squeeze else deltas)
    B, A, _ = d.shape
    anchors, bstride = _anchor_args(anchors, B, A)
    out = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
    rw = (C.c_float * 4)(*reg_w)
    with torch.cuda.device(dev):
        check(lib.rn_decode_clip(_ptr(d), _dtype_code(d), B, A, _ptr(anchors), bstride, _ptr(image_hw), rw,
                                 _ptr(out), _stream(dev)), "rn_decode_clip")
    return out[0] if squeeze else out


def nms_segments(boxes: Tensor, scores: Tensor, seg_off: Tensor, iou_thr: float) -> Tuple[Tensor, Tensor]:
    """K6 at the op boundary (torchvision.ops.nms batched over segments).
    boxes f32 [N,4], scores f32 [N], seg_off i32 [S+1] (device).
    -> (keep i64 [N] (per segment: indices relative to the segment start, score order), keep_count i32 [S])."""
    dev = _need_dev(boxes, scores, seg_off)
    boxes = _c(boxes.float()).reshape(-1, 4)
    scores = _c(scores.float()).reshape(-1)
    N, S = boxes.shape[0], seg_off.numel() - 1
    keep = torch.empty((max(N, 1),), dtype=torch.int64, device=dev)
    count = torch.empty((S,), dtype=torch.int32, device=dev)
    ws_bytes = lib.rn_nms_workspace_bytes(N, S)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.rn_nms_segments(_ptr(boxes), _ptr(scores), _ptr(seg_off), S, N, iou_thr, _ptr(keep), _ptr(count),
                                  _ptr(ws), ws_bytes, _stream(dev)), "rn_nms_segments")
    return keep[:N], count


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """Drop-in for ``torchvision.ops.nms(boxes, scores, iou_threshold) -> int64[k]`` (one segment).
    Reading the kept count is the one host sync, exactly as in the op it replaces."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    off = torch.tensor([0, n], dtype=torch.int32, device=boxes.device)
    keep, count = nms_segments(boxes, scores, off, iou_threshold)
    return keep[: int(count.item())]


def detect(cls: Tensor, deltas: Tensor, anchors, image_sizes: Sequence[Tuple[int, int]], score_thr: float,
           min_box: float, nms_thr: float, max_det: int, reg_w: Sequence[float] = (1.0, 1.0, 1.0, 1.0),
           max_candidates: Optional[int] = None) -> List[dict]:
    """K4-K7 for a batch: -> [{"boxes" f32[n,4], "scores" f32[n], "labels" i64[n]}], n <= max_det.

    One host sync at the end (the ragged result sizes have to reach Python, as in the
    reference's list-of-dicts contract).  If an image has more candidates than the workspace
    was sized for, the call is repeated once with the exact worst case (A*K)."""
    return detect_levels([cls], [deltas], anchors, image_sizes, score_thr, min_box, nms_thr, max_det, reg_w, max_candidates)


def detect_levels(cls_levels: Sequence[Tensor], box_levels: Sequence[Tensor], anchors, image_sizes: Sequence[Tuple[int, int]],
                  score_thr: float, min_box: float, nms_thr: float, max_det: int,
                  reg_w: Sequence[float] = (1.0, 1.0, 1.0, 1.0), max_candidates: Optional[int] = None,
                  enqueue_only: Optional[list] = None) -> List[dict]:
    """``detect`` on per-level head outputs (cls_levels[l] [B,A_l,K], box_levels[l] [B,A_l,4]) without
    concatenating them; identical results to ``detect(torch.cat(cls_levels, 1), torch.cat(box_levels, 1), ...)``."""
    L = len(cls_levels)
    if L == 0 or L > _lib.RN_MAX_LEVELS or len(box_levels) != L:
        raise ValueError("need 1..8 levels of (cls, box) outputs")
    dev = _need_dev(*cls_levels, *box_levels)
    dt = cls_levels[0].dtype
    for c, d in zip(cls_levels, box_levels):
        if c.dim() != 3 or d.dim() != 3 or c.shape[:2] != d.shape[:2] or d.shape[-1] != 4 or c.dtype != dt \
                or c.shape[0] != cls_levels[0].shape[0] or c.shape[2] != cls_levels[0].shape[2]:
            raise ValueError(f"bad head output shapes {tuple(c.shape)} / {tuple(d.shape)}")
    cls_levels = [_c(c) for c in cls_levels]
    box_levels = [_c(d if d.dtype == dt else d.to(dt)) for d in box_levels]
    B, _, K = cls_levels[0].shape
    counts = [int(c.shape[1]) for c in cls_levels]
    A = sum(counts)
    if not isinstance(anchors, Tensor):
        first = anchors[0]
        anchors = first if all(a is first or a.data_ptr() == first.data_ptr() for a in anchors) else torch.stack(list(anchors))
    anchors, bstride = _anchor_args(anchors.to(dev), B, A)
    hw = image_hw_tensor(image_sizes, dev)
    params = RnDetectParams(score_thr, min_box, nms_thr, int(max_det), (C.c_float * 4)(*reg_w))
    cap = int(max_candidates) if max_candidates else min(A * K, 1 << 18)
    out_boxes = torch.empty((B, max_det, 4), dtype=torch.float32, device=dev)
    out_scores = torch.empty((B, max_det), dtype=torch.float32, device=dev)
    out_labels = torch.empty((B, max_det), dtype=torch.int64, device=dev)
    meta = torch.empty((2, B), dtype=torch.int32, device=dev)          # [0] = count, [1] = status
    arr = lambda ts: (C.c_void_p * L)(*[t.data_ptr() for t in ts])
    while True:
        ws_bytes = lib.rn_detect_workspace_bytes(B, A, K, cap)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)

        def enqueue():
            check(lib.rn_detect_levels(arr(cls_levels), arr(box_levels), (C.c_int64 * L)(*counts), L, _dtype_code(cls_levels[0]),
                                       B, K, _ptr(anchors), bstride, _ptr(hw), C.byref(params), cap, _ptr(out_boxes),
                                       _ptr(out_scores), _ptr(out_labels), _ptr(meta[0]), _ptr(meta[1]), _ptr(ws), ws_bytes,
                                       _stream(dev)), "rn_detect_levels")
        if enqueue_only is not None:          # (bench.py: the chain's launches as a closure -- for a hipGraph capture -- instead of running them)
            enqueue_only.append(enqueue)
            enqueue_only.append((out_boxes, out_scores, out_labels, meta, ws, hw, anchors, cls_levels, box_levels))     # keep-alive
            return []
        with torch.cuda.device(dev), _timed("detect", dev):
            enqueue()
        meta_h = meta.cpu()                                              # the one sync
        if not bool(meta_h[1].any()) or cap >= A * K:
            break
        cap = A * K
    counts = meta_h[0].tolist()
    return [{"boxes": out_boxes[b, :n], "scores": out_scores[b, :n], "labels": out_labels[b, :n]}
            for b, n in enumerate(counts)]
