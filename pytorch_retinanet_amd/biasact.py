"""Conv epilogues for the head: ``bias_act`` (bias + ReLU + optional position mask, one HIP launch each
way) and the packed level canvas that lets the shared head towers run ONE convolution per layer over
all five pyramid levels (reference: ``retinanet/layers.py:143-171``, ``:213-241`` -- four 3x3 conv +
ReLU pairs per tower, applied level by level).

Why a canvas: MIOpen runs the 3x3/256-channel tower conv of the R50 config at 657 TFLOP/s on P3 but at
170 / 60 / 18 TFLOP/s on P5 / P6 / P7 (tiny grids), and every level costs its own bias-add, ReLU and
weight-gradient accumulation kernels.  Packing the levels into one ``[N, C, Hc, Wc]`` canvas -- P3 on
top, P4..P7 side by side below it, one zero row/column between neighbours -- makes it one conv per
layer (fwd+bwd 1.10 ms instead of 1.58 ms per layer, ``tools/head_conv_probe.py``).  A 3x3 conv with
padding 1 never mixes levels as long as the gaps hold zeros, which the epilogue's mask re-establishes
after every layer.
"""
import ctypes as C
import os
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

from ._lib import RN_BF16, RN_F16, RN_F32, check, lib
from .ops import _timed          # event pairs around the MFMA conv launches when ops.enable_timing(True) (bench.py)

_DT = {torch.float32: RN_F32, torch.bfloat16: RN_BF16, torch.float16: RN_F16}
H16 = (torch.bfloat16, torch.float16)          # the element types of the MFMA conv kernels (v_mfma_*_bf16 / _f16: same rate, same kernels)
_WS: Dict[tuple, Tensor] = {}
PAIR_SHEETS = True     # two images per canvas sheet (Canvas.of) when that takes fewer positions
MFMA_FLOP: Dict[str, float] = {}      # USEFUL flop per call of every timed MFMA launch (bench.py: achieved TFLOP/s of the own conv kernels)
_REAL_PER_SHEET: Dict[tuple, int] = {}  # (Hp, Wp) of a canvas sheet -> feature positions on it (gaps / borders are not useful work)


def _real_positions(N: int, Hp: int, Wp: int) -> int:
    "Feature positions of N canvas sheets (what the flop figures count; a conv that is not on a known canvas counts all)."
    return N * _REAL_PER_SHEET.get((Hp, Wp), Hp * Wp)


def _mfma_call(tag: str, dev: torch.device, flop: float, status_thunk, what: str) -> None:
    "Launch one hand-written MFMA conv; with ops.enable_timing(True) the launch is bracketed by events and its flop recorded."
    MFMA_FLOP[tag] = flop
    with _timed(tag, dev):
        check(status_thunk(), what)


def _workspace(dev: torch.device, stream: int, channels: int):
    need = lib.rn_bn_workspace_bytes(channels)
    key = (dev.index, stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _WS[key] = torch.empty((max(need, lib.rn_bn_workspace_bytes(1024)),), dtype=torch.uint8, device=dev)
    return ws.data_ptr(), ws.numel()


def _cl(t: Tensor) -> bool:
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def fusable(x: Tensor, bias: Optional[Tensor]) -> bool:
    return (x.is_cuda and x.dtype in _DT and _cl(x) and x.shape[1] % 8 == 0 and x.numel() > 0 and bias is not None
            and bias.dtype == torch.float32)


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, mask, relu):
        N, Cc, H, W = x.shape
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        y = torch.empty_like(x)
        check(lib.rn_bias_act_forward(x.data_ptr(), bias.data_ptr(), mask.data_ptr() if mask is not None else 0, y.data_ptr(),
                                      _DT[x.dtype], N * H * W, Cc, H * W, int(relu), stream), "rn_bias_act_forward")
        ctx.save_for_backward(y if relu else None, mask)
        ctx.cfg = (bool(relu), N * H * W, Cc, H * W, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, mask = ctx.saved_tensors
        relu, M, Cc, HW, dt = ctx.cfg
        dev = dy.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        if dy.dtype != dt or not _cl(dy):
            dy = dy.to(dt).contiguous(memory_format=torch.channels_last)
        need_dx = relu or mask is not None
        dx = torch.empty_like(dy) if need_dx else dy
        dbias = torch.empty((Cc,), dtype=torch.float32, device=dev)
        wp, wn = _workspace(dev, stream, Cc)
        check(lib.rn_bias_act_backward(dy.data_ptr(), y.data_ptr() if y is not None else 0, mask.data_ptr() if mask is not None else 0,
                                       dx.data_ptr() if need_dx else 0, dbias.data_ptr(), _DT[dt], M, Cc, HW, int(relu), wp, wn,
                                       stream), "rn_bias_act_backward")
        return dx, dbias, None, None


_WG_WS: Dict[tuple, Tensor] = {}
MFMA_WGRAD = True      # weight gradient of the canvas convs on the MFMA kernel (False: MIOpen)


def _canvas_wgrad(gs, xs, ws, Wp: int, stream: int):
    """Weight gradients of P canvas convs (256 -> 256, bf16) by ``rn_conv3x3_canvas_wgrad_batched``; None when the
    shapes are outside the kernel's range (the caller then asks MIOpen)."""
    x0, w0 = xs[0], ws[0]
    if not (MFMA_WGRAD and x0.dtype in H16 and tuple(w0.shape) == (256, 256, 3, 3)):
        return None
    dev = x0.device
    N, _, Hp, _ = x0.shape
    M = N * Hp * Wp
    P = len(gs)
    need = lib.rn_conv3x3_wgrad_workspace_bytes(P, M)
    key = (dev.index, stream)
    wsb = _WG_WS.get(key)
    if wsb is None or wsb.numel() < need:
        wsb = _WG_WS[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
    dws = [torch.empty((256, 256, 3, 3), dtype=x0.dtype, device=dev, memory_format=torch.channels_last) for _ in range(P)]
    _mfma_call(f"mfma_tower_wgrad_x{P}", dev, P * 2.0 * _real_positions(N, Hp, Wp) * 256 * 2304,
               lambda: lib.rn_conv3x3_canvas_wgrad_batched(_ptr_array(gs), _ptr_array(xs), _ptr_array(dws), P, _DT[x0.dtype], M, Wp, 256, 256,
                                                           _zero_page(dev).data_ptr(), wsb.data_ptr(), wsb.numel(), stream),
               "rn_conv3x3_canvas_wgrad_batched")
    return dws


class _TowerConv(torch.autograd.Function):
    """relu(conv3x3(x, w) + bias) * mask on a zero-bordered canvas: forward and data gradient are the hand-written
    MFMA implicit GEMM (``rn_conv3x3_canvas``), the weight gradient the MFMA position-contraction GEMM
    (``rn_conv3x3_canvas_wgrad_batched``; MIOpen for shapes outside its range)."""

    @staticmethod
    def forward(ctx, x, w, bias, mask):
        N, Cin, Hp, Wp = x.shape
        Cout = w.shape[0]
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        if not _cl(w):
            w = w.contiguous(memory_format=torch.channels_last)
        y = torch.empty((N, Cout, Hp, Wp), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        _mfma_call("mfma_tower_fwd_x1", dev, 2.0 * _real_positions(N, Hp, Wp) * Cout * 9 * Cin,
                   lambda: lib.rn_conv3x3_canvas(x.data_ptr(), w.data_ptr(), bias.data_ptr(), mask.data_ptr(), y.data_ptr(), _DT[x.dtype],
                                                 N * Hp * Wp, Hp * Wp, Wp, Cin, Cout, 1, stream), "rn_conv3x3_canvas")
        ctx.save_for_backward(x, w, y, mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, mask = ctx.saved_tensors
        N, Cin, Hp, Wp = x.shape
        Cout = w.shape[0]
        dev = dy.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        if dy.dtype != x.dtype or not _cl(dy):
            dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        M = N * Hp * Wp
        g = torch.empty_like(dy)                                   # gradient at the conv output: ReLU + canvas mask
        dbias = torch.empty((Cout,), dtype=torch.float32, device=dev)
        wp, wn = _workspace(dev, stream, Cout)
        check(lib.rn_bias_act_backward(dy.data_ptr(), y.data_ptr(), mask.data_ptr(), g.data_ptr(), dbias.data_ptr(), _DT[x.dtype],
                                       M, Cout, Hp * Wp, 1, wp, wn, stream), "rn_bias_act_backward")
        dx = dw = None
        if ctx.needs_input_grad[0] and Cin % 256 == 0 and Cout % 64 == 0:
            wt = w.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)     # [Cin, Cout, 3, 3], taps reversed
            dx = torch.empty_like(x)
            _mfma_call("mfma_tower_dgrad_x1", dev, 2.0 * _real_positions(N, Hp, Wp) * Cout * 9 * Cin,
                       lambda: lib.rn_conv3x3_canvas(g.data_ptr(), wt.data_ptr(), 0, mask.data_ptr(), dx.data_ptr(), _DT[x.dtype],
                                                     M, Hp * Wp, Wp, Cout, Cin, 0, stream), "rn_conv3x3_canvas")
        if ctx.needs_input_grad[1]:
            r = _canvas_wgrad([g], [x], [w], Wp, stream)
            dw = r[0] if r is not None else None
        need = [ctx.needs_input_grad[0] and dx is None, ctx.needs_input_grad[1] and dw is None, False]
        if need[0] or need[1]:
            r = torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, need)
            dx = r[0] if need[0] else dx
            dw = r[1] if need[1] else dw
        return dx, dw, dbias, None


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() if t is not None else 0 for t in ts])


TOWER_SUM2 = True        # first tower layer: the shared input's gradient as ONE two-source data-gradient launch (rn_conv3x3_canvas_sum2)


class TowerLink:
    """Hand-over between consecutive ``_TowerConvPair`` layers of one tower pair in backward.  When layer l's inputs are the
    ReLU outputs of layer l - 1 and feed nothing else (``tower_conv_pair(..., prev=link)``), layer l's data-gradient kernel
    also applies layer l - 1's ReLU mask and sums the columns (``rn_conv3x3_canvas_dgrad_relu_batched``): what it returns as
    the input gradients IS layer l - 1's pre-activation gradient, and the bias gradients wait here.  Layer l - 1 uses them
    only if the gradient tensors autograd hands it are exactly the ones layer l returned -- same storage AND same version
    counter: when the output has another consumer autograd may add that gradient IN PLACE into the returned buffer -- ;
    anything else takes the ordinary ``rn_bias_act_backward`` pass (masking a pre-masked gradient again is harmless)."""
    __slots__ = ("ptrs", "dbias", "relu_masks", "half")

    def __init__(self):
        self.ptrs, self.dbias, self.relu_masks = None, None, None     # relu_masks: the ReLU bits of the layer's outputs (forward)
        # the LAST tower layer: its two outputs feed two different convs (class- / box-output), each of which deposits its half
        # here -- (gradient data_ptr, version, bias gradient) -- after applying this layer's ReLU mask in its own data-gradient kernel
        self.half = [None, None]


_CS_WS: Dict[tuple, Tensor] = {}
FUSE_TOWER_RELU_BWD = True
BOX_OUTPUT_WGRAD_MFMA = True   # ... and its weight gradient on the narrow gathering kernel
BOX_OUTPUT_WGRAD_NARROW = True  # ... or (round 5, preferred) on csrc/wgrad3x3.hip over the 64-channel canvas gradient
BOX_OUTPUT_FWD_MFMA = True     # box-output conv forward on the narrow MFMA level-mode kernel


class _TowerConvPair(torch.autograd.Function):
    """``_TowerConv`` for two towers of identical geometry (cls and box) in one batched launch each way: their tiles
    together fill the chip's workgroup waves.  ``prev`` / ``link``: see ``TowerLink``."""

    @staticmethod
    def forward(ctx, x0, x1, w0, w1, b0, b1, mask, prev, link):
        N, Cin, Hp, Wp = x0.shape
        Cout = w0.shape[0]
        dev = x0.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        w0 = w0 if _cl(w0) else w0.contiguous(memory_format=torch.channels_last)
        w1 = w1 if _cl(w1) else w1.contiguous(memory_format=torch.channels_last)
        ys = [torch.empty((N, Cout, Hp, Wp), dtype=x0.dtype, device=dev, memory_format=torch.channels_last) for _ in range(2)]
        rms = None
        if link is not None and FUSE_TOWER_RELU_BWD and any(ctx.needs_input_grad[:6]):
            rms = [torch.empty((N * Hp * Wp * (Cout // 8),), dtype=torch.uint8, device=dev) for _ in range(2)]
        _mfma_call("mfma_tower_fwd_x2", dev, 2 * 2.0 * _real_positions(N, Hp, Wp) * Cout * 9 * Cin,
                   lambda: lib.rn_conv3x3_canvas_batched_ex(_ptr_array([x0, x1]), _ptr_array([w0, w1]), _ptr_array([b0, b1]), mask.data_ptr(),
                                                            _ptr_array(ys), _ptr_array(rms) if rms else None, 2, _DT[x0.dtype], N * Hp * Wp,
                                                            Hp * Wp, Wp, Cin, Cout, 1, stream),
                   "rn_conv3x3_canvas_batched_ex")
        if link is not None:
            link.relu_masks = rms
        ctx.save_for_backward(x0, x1, w0, w1, ys[0], ys[1], mask)
        ctx.prev, ctx.link = prev, link
        return ys[0], ys[1]

    @staticmethod
    def backward(ctx, dy0, dy1):
        x0, x1, w0, w1, y0, y1, mask = ctx.saved_tensors
        N, Cin, Hp, Wp = x0.shape
        Cout = w0.shape[0]
        dev = x0.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        M = N * Hp * Wp
        link, prev = ctx.link, ctx.prev
        gs, dbs = [], []
        if (link is not None and link.ptrs is not None
                and link.ptrs == (dy0.data_ptr(), dy0._version, dy1.data_ptr(), dy1._version)
                and dy0.dtype == x0.dtype and _cl(dy0) and _cl(dy1)):
            gs, dbs = [dy0, dy1], list(link.dbias)            # the layer above already applied this layer's ReLU mask
        else:
            wp, wn = _workspace(dev, stream, Cout)
            for i, (dy, y) in enumerate(((dy0, y0), (dy1, y1))):
                h = link.half[i] if link is not None else None
                if h is not None and h[:2] == (dy.data_ptr(), dy._version) and dy.dtype == x0.dtype and _cl(dy):
                    gs.append(dy); dbs.append(h[2])              # the output conv above already applied this layer's ReLU mask
                    continue
                if dy.dtype != x0.dtype or not _cl(dy):
                    dy = dy.to(x0.dtype).contiguous(memory_format=torch.channels_last)
                g = torch.empty_like(dy)
                db = torch.empty((Cout,), dtype=torch.float32, device=dev)
                check(lib.rn_bias_act_backward(dy.data_ptr(), y.data_ptr(), mask.data_ptr(), g.data_ptr(), db.data_ptr(), _DT[x0.dtype],
                                               M, Cout, Hp * Wp, 1, wp, wn, stream), "rn_bias_act_backward")
                gs.append(g); dbs.append(db)
        if link is not None:
            link.ptrs, link.dbias, link.half = None, None, [None, None]
        dxs = [None, None]
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            wts = dgrad_weights([w0, w1], stream)                      # [Cin, Cout, 3, 3], taps reversed (the step's table, or one launch for both)
            same_input = (TOWER_SUM2 and x0.data_ptr() == x1.data_ptr() and tuple(x0.shape) == tuple(x1.shape) and x0.stride() == x1.stride()
                          and not (prev is not None and prev.relu_masks is not None))
            dxs = [torch.empty_like(x0), None if same_input else torch.empty_like(x1)]
            if same_input:
                # both towers read the SAME canvas (their first layer): its gradient is the sum of the two data gradients -- one launch whose
                # contraction walks gs[0]'s channels, then gs[1]'s, against the two weights side by side; no second output, no add pass
                wcat = torch.cat([wts[0], wts[1]], dim=1).contiguous(memory_format=torch.channels_last)
                _mfma_call("mfma_tower_dgrad_x2", dev, 2 * 2.0 * _real_positions(N, Hp, Wp) * Cout * 9 * Cin,
                           lambda: lib.rn_conv3x3_canvas_sum2(gs[0].data_ptr(), gs[1].data_ptr(), wcat.data_ptr(), mask.data_ptr(), dxs[0].data_ptr(),
                                                              _DT[x0.dtype], M, Hp * Wp, Wp, Cout, Cin, stream), "rn_conv3x3_canvas_sum2")
            elif prev is not None and prev.relu_masks is not None and Cin == Cout:
                need = lib.rn_conv3x3_colsum_workspace_bytes(2, M, Cin)
                key = (dev.index, stream)
                wsb = _CS_WS.get(key)
                if wsb is None or wsb.numel() < need:
                    wsb = _CS_WS[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
                dbp = [torch.empty((Cin,), dtype=torch.float32, device=dev) for _ in range(2)]
                _mfma_call("mfma_tower_dgrad_x2", dev, 2 * 2.0 * _real_positions(N, Hp, Wp) * Cout * 9 * Cin,
                           lambda: lib.rn_conv3x3_canvas_dgrad_relu_batched(_ptr_array(gs), _ptr_array(wts), _ptr_array(prev.relu_masks), mask.data_ptr(),
                                                                            _ptr_array(dxs), _ptr_array(dbp), 2, _DT[x0.dtype], M, Hp * Wp, Wp,
                                                                            Cout, Cin, wsb.data_ptr(), wsb.numel(), stream),
                           "rn_conv3x3_canvas_dgrad_relu_batched")
                prev.ptrs, prev.dbias = (dxs[0].data_ptr(), dxs[0]._version, dxs[1].data_ptr(), dxs[1]._version), dbp
            else:
                _mfma_call("mfma_tower_dgrad_x2", dev, 2 * 2.0 * _real_positions(N, Hp, Wp) * Cout * 9 * Cin,
                           lambda: lib.rn_conv3x3_canvas_batched(_ptr_array(gs), _ptr_array(wts), None, mask.data_ptr(), _ptr_array(dxs), 2,
                                                                 _DT[x0.dtype], M, Hp * Wp, Wp, Cout, Cin, 0, stream), "rn_conv3x3_canvas_batched")
        dws = _canvas_wgrad(gs, [x0, x1], [w0, w1], Wp, stream)
        if dws is None:
            dws = [torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
                   for g, x, w in ((gs[0], x0, w0), (gs[1], x1, w1))]
        return dxs[0], dxs[1], dws[0], dws[1], dbs[0], dbs[1], None, None, None


def tower_conv_pair(x0: Tensor, x1: Tensor, w0: Tensor, w1: Tensor, b0: Tensor, b1: Tensor, mask: Tensor,
                    prev: Optional[TowerLink] = None, link: Optional[TowerLink] = None):
    """Two ``tower_conv`` of identical geometry (Cin == Cout % 256 == 0) in one launch each way.  Chaining: pass the
    ``link`` given to the layer below as ``prev`` here when (x0, x1) are that layer's outputs and feed NOTHING else; its
    ReLU backward and bias gradient then ride in this layer's data-gradient kernel (``TowerLink``)."""
    return _TowerConvPair.apply(x0, x1, w0.to(x0.dtype), w1.to(x0.dtype), b0, b1, mask, prev, link)


def tower_conv_fusable(x: Tensor, conv) -> bool:
    "The MFMA canvas conv covers bf16, 3x3 / stride 1 / pad 1, Cin % 64 == 0 and Cout % 256 == 0 (head towers: 256 -> 256)."
    # (its data gradient runs on the same kernel when Cin % 256 == 0 as well, else on MIOpen)
    return (x.is_cuda and x.dtype in H16 and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and
            conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is not None and
            conv.in_channels % 64 == 0 and conv.out_channels % 256 == 0)


def tower_conv(x: Tensor, weight: Tensor, bias: Tensor, mask: Tensor) -> Tensor:
    """``relu(conv3x3(x, weight) + bias) * mask`` on a canvas built with ``Canvas(..., pad=1)``; ``weight`` is cast to
    the activation dtype here (what autocast would do)."""
    return _TowerConv.apply(x, weight.to(x.dtype), bias, mask)


def bias_act(x: Tensor, bias: Tensor, mask: Optional[Tensor] = None, relu: bool = True) -> Tensor:
    """``mask * act(x + bias[None, :, None, None])``; ``mask``: u8 ``[H*W]`` (1 = keep) or None."""
    if fusable(x, bias):
        return _BiasAct.apply(x, bias, mask, relu)
    y = x + bias.to(x.dtype)[None, :, None, None]
    if relu:
        y = F.relu(y)
    if mask is not None:
        y = y * mask.view(1, 1, x.shape[2], x.shape[3]).to(y.dtype)
    return y


# ---------------------------------------------------------------------------------------------------
class Canvas:
    """Placement of the L feature maps of ``slots`` images on one canvas SHEET, one empty row / column between neighbours and
    ``pad`` extra empty rows / columns all around (the MFMA canvas conv wants a one-pixel zero border instead of bounds
    checks).  The copies of level 0 are stacked; below them the other rectangles -- level 1 of every slot, then level 2 of
    every slot, ... -- are laid left to right on shelves as wide as the sheet.  With one slot that is "level 0 on top, the
    others side by side below it"; with two, the second image's P4 fills the space beside the first one's (the standard
    800 x 1344 pyramid: 23 940 positions per image instead of 26 010, which takes the head's 256 x 256 conv tiles of a
    batch of 8 from 7 rounds of the chip's 256 CUs to 6).  Image n lies on sheet n // slots in slot n % slots.

    ``regions``: (level, slot, row, col, h, w) of every rectangle; ``origin``: the rectangles of slot 0 (one per level);
    ``mask``: uint8 [H*W], 1 on feature positions; ``map``: int32 [H*W] for the level-mode kernels (``rn_canvas_layout``):
    -1 on gaps, else slot << 28 | level << 24 | (y * w + x)."""

    def __init__(self, shapes: Sequence[Tuple[int, int]], device: torch.device, pad: int = 0, slots: int = 1):
        self.shapes = [(int(h), int(w)) for h, w in shapes]
        self.pad, self.slots = pad, int(slots)
        S = self.slots
        assert 1 <= S <= 8 and len(self.shapes) <= 16
        h0, w0 = self.shapes[0]
        rest = [(l, s_) for l in range(1, len(self.shapes)) for s_ in range(S)]
        inner_w = max([w0] + ([S * (self.shapes[1][1] + 1) - 1] if len(self.shapes) > 1 else []))
        self.regions: List[Tuple[int, int, int, int, int, int]] = []
        row = pad
        for s_ in range(S):
            self.regions.append((0, s_, row, pad, h0, w0))
            row += h0 + 1
        col, shelf_h = 0, 0
        for l, s_ in rest:
            h, w = self.shapes[l]
            if col and col + w > inner_w:                       # next shelf
                row += shelf_h + 1
                col, shelf_h = 0, 0
            self.regions.append((l, s_, row, pad + col, h, w))
            col += w + 1
            shelf_h = max(shelf_h, h)
        self.H = (row + shelf_h if rest else row - 1) + pad
        self.W = inner_w + 2 * pad
        self.origin: List[Tuple[int, int]] = [(r, c) for l, s_, r, c, h, w in sorted(self.regions) if s_ == 0]
        m = np.zeros((self.H, self.W), dtype=np.uint8)
        mp = np.full((self.H, self.W), -1, dtype=np.int32)
        for l, s_, r, c, h, w in self.regions:
            assert not m[max(r - 1, 0):r + h + 1, max(c - 1, 0):c + w + 1].any(), "canvas rectangles touch"
            m[r:r + h, c:c + w] = 1
            mp[r:r + h, c:c + w] = (s_ << 28) | (l << 24) | (np.arange(h)[:, None] * w + np.arange(w)[None, :])
        self.mask = torch.from_numpy(m.reshape(-1)).to(device)
        self.map = torch.from_numpy(mp.reshape(-1)).to(device)
        self.fill = S * sum(h * w for h, w in self.shapes) / float(self.H * self.W)
        _REAL_PER_SHEET[(self.H, self.W)] = S * sum(h * w for h, w in self.shapes)

    def sheets(self, n_images: int) -> int:
        return (n_images + self.slots - 1) // self.slots

    _cache: Dict[tuple, "Canvas"] = {}

    @classmethod
    def of(cls, feature_maps: Sequence[Tensor], pad: int = 0, slots: Optional[int] = None) -> "Canvas":
        """The canvas for these feature maps; ``slots`` None: two images per sheet when that takes fewer positions (zero-
        bordered canvases of the MFMA kernels only), else one."""
        shapes = tuple(tuple(int(v) for v in f.shape[-2:]) for f in feature_maps)
        dev, N = feature_maps[0].device, int(feature_maps[0].shape[0])
        if slots is None:
            slots = 1
            if pad and N > 1 and PAIR_SHEETS:
                one, two = cls._get(shapes, dev, pad, 1), cls._get(shapes, dev, pad, 2)
                if two.sheets(N) * two.H * two.W < N * one.H * one.W:
                    slots = 2
        return cls._get(shapes, dev, pad, slots)

    @classmethod
    def _get(cls, shapes, dev, pad, slots) -> "Canvas":
        key = (shapes, dev, pad, slots)
        c = cls._cache.get(key)
        if c is None:
            c = cls._cache[key] = Canvas(shapes, dev, pad, slots)
        return c


def _pack_native(cv: "Canvas", levels: Sequence[Tensor], canvas_t: Tensor, n_images: int, to_canvas: bool) -> bool:
    """``rn_canvas_pack``: all levels <-> the canvas sheets in one launch (False: shapes / dtypes the kernel does not take)."""
    x = canvas_t
    if not (x.is_cuda and x.dtype in H16 and x.dim() == 4 and _cl(x) and x.shape[1] % 2 == 0
            and len(levels) <= 6 and x.shape[0] * cv.H * cv.W < (1 << 22)):
        return False
    for t in levels:
        if not (t.is_cuda and t.dtype == x.dtype and t.dim() == 4 and t.shape[1] == x.shape[1] and (_cl(t) or t.shape[1] == 1)
                and t.data_ptr() % 16 == 0):
            return False
    dev = x.device
    if dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)
    check(lib.rn_canvas_pack(_ptr_array(levels), _layout(cv, n_images), x.data_ptr(), _DT[x.dtype], x.shape[0], cv.H, cv.W, x.shape[1],
                             1 if to_canvas else 0, torch.cuda.current_stream().cuda_stream), "rn_canvas_pack")
    return True


class _Pack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, canvas: Canvas, *feats):
        f0 = feats[0]
        N, S = f0.shape[0], canvas.slots
        out = torch.empty((canvas.sheets(N), f0.shape[1], canvas.H, canvas.W), dtype=f0.dtype, device=f0.device,
                          memory_format=torch.channels_last)
        if not _pack_native(canvas, feats, out, N, True):
            out.fill_(0)            # (a fill KERNEL: zero_() is a hipMemsetAsync, a memset node under capture -- layers._conv_own_bias)
            for l, s_, r, c, h, w in canvas.regions:
                src = feats[l][s_::S]
                if src.shape[0]:
                    out[:src.shape[0], :, r:r + h, c:c + w].copy_(src)
        ctx.canvas, ctx.n = canvas, N
        return out

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(_gather_levels(ctx.canvas, g, ctx.n))


def _gather_levels(cv: "Canvas", x: Tensor, n_images: int) -> List[Tensor]:
    "The per-level tensors [n_images, C, h, w] (channels-last) cut out of canvas sheets."
    S = cv.slots
    outs = [torch.empty((n_images, x.shape[1], h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last) for h, w in cv.shapes]
    if _pack_native(cv, outs, x if _cl(x) else x.contiguous(memory_format=torch.channels_last), n_images, False):
        return outs
    if S == 1:
        return [x[:, :, r:r + h, c:c + w].contiguous(memory_format=torch.channels_last) for (r, c), (h, w) in zip(cv.origin, cv.shapes)]
    for l, s_, r, c, h, w in cv.regions:
        dst = outs[l][s_::S]
        if dst.shape[0]:
            dst.copy_(x[:dst.shape[0], :, r:r + h, c:c + w])
    return outs


class _Unpack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, canvas: Canvas, x, n_images):
        ctx.canvas = canvas
        ctx.meta = (x.shape, x.dtype, x.device)
        ctx.n = int(n_images)
        return tuple(_gather_levels(canvas, x, n_images))

    @staticmethod
    def backward(ctx, *grads):
        cv = ctx.canvas
        shape, dt, dev = ctx.meta
        return None, _scatter_levels(cv, grads, shape, dt, dev, ctx.n), None


def _scatter_levels(cv: "Canvas", grads, shape, dt, dev, n_images: int) -> Tensor:
    "The canvas sheets holding the per-level tensors ``grads`` (None = zeros), zeros in the gaps."
    S = cv.slots
    g = torch.empty(shape, dtype=dt, device=dev, memory_format=torch.channels_last)
    if all(gl is not None for gl in grads):
        lv = [gl if (gl.dtype == dt and _cl(gl)) else gl.to(dt).contiguous(memory_format=torch.channels_last) for gl in grads]
        if _pack_native(cv, lv, g, n_images, True):
            return g
    g.fill_(0)
    for l, s_, r, c, h, w in cv.regions:
        gl = grads[l]
        if gl is not None:
            src = gl[s_::S]
            if src.shape[0]:
                g[:src.shape[0], :, r:r + h, c:c + w].copy_(src)
    return g


def pack_levels(canvas: Canvas, feature_maps: Sequence[Tensor]) -> Tensor:
    return _Pack.apply(canvas, *feature_maps)


def unpack_levels(canvas: Canvas, x: Tensor, n_images: Optional[int] = None) -> List[Tensor]:
    "``n_images``: the batch size (default: every slot of every sheet holds an image)."
    return list(_Unpack.apply(canvas, x, x.shape[0] * canvas.slots if n_images is None else int(n_images)))


# ---------------------------------------------------------------------------------------------------
# The class-output conv (retinanet/layers.py:163-167) on the canvas, dense per-level logits out
_ZEROS: Dict[int, Tensor] = {}


def _zero_page(dev: torch.device) -> Tensor:
    z = _ZEROS.get(dev.index)
    if z is None:
        z = _ZEROS[dev.index] = torch.empty((256,), dtype=torch.uint8, device=dev).fill_(0)
    return z


class RnCanvasLayout(C.Structure):
    _fields_ = [("map", C.c_void_p), ("slots", C.c_int32), ("n_images", C.c_int32), ("T", C.c_int32), ("hw", C.c_int32 * 6)]


def _layout(cv: "Canvas", n_images: int):
    lay = RnCanvasLayout()
    lay.map, lay.slots, lay.n_images, lay.T = cv.map.data_ptr(), cv.slots, int(n_images), len(cv.shapes)
    for i, (h, w) in enumerate(cv.shapes):
        lay.hw[i] = h * w
    return C.byref(lay)


# ---- the data-gradient weights of all 3x3 convolutions of a step in ONE launch -------------------------------------------------
# A data gradient issued as a forward convolution needs the weight with its taps reversed and its channel roles swapped.  Flipping it
# where it is needed costs a launch per convolution (18 per R50 step, ~4.7 us each on the step's critical path).  Instead every site
# asks ``dgrad_weights``: a weight seen once is REGISTERED, and ``refresh_dgrad_weights()`` -- called by the model at the start of a
# training forward -- flips all registered weights in one launch into persistent buffers.  An entry is used only while it is provably
# current: same tensor object, same ``_version`` as when it was flipped, and no ``invalidate_dgrad_weights()`` since (this package's
# optimizer writes parameters through raw pointers and calls it).
DGRAD_WEIGHT_TABLE = True


class _FlippedWeight:
    __slots__ = ("ref", "flipped", "version", "epoch")

    def __init__(self, w: Tensor):
        self.ref = weakref.ref(w)
        self.flipped = torch.empty((int(w.shape[1]), int(w.shape[0]), 3, 3), dtype=w.dtype, device=w.device, memory_format=torch.channels_last)
        self.version, self.epoch = -1, -1


_DW_TABLE: Dict[int, _FlippedWeight] = {}
_DW_EPOCH = 0          # bumped by invalidate_dgrad_weights(): entries stamped with an older epoch are stale


def invalidate_dgrad_weights() -> None:
    "The parameters changed behind autograd's back (an optimizer that writes through raw pointers): nothing flipped so far is current."
    global _DW_EPOCH
    _DW_EPOCH += 1


def _dw_eligible(w: Tensor) -> bool:
    return (DGRAD_WEIGHT_TABLE and w.is_cuda and w.dtype in H16 and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3)
            and _cl(w) and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0)


def refresh_dgrad_weights(device=None) -> int:
    "Flip every registered weight (that still exists, on ``device`` if given) in one launch; -> how many."
    live = []
    for key in list(_DW_TABLE):
        e = _DW_TABLE[key]
        w = e.ref()
        if w is None or not _dw_eligible(w) or w.device != e.flipped.device or tuple(e.flipped.shape[:2]) != (w.shape[1], w.shape[0]):
            del _DW_TABLE[key]
            continue
        if device is None or w.device == torch.device(device):
            live.append((w, e))
    if not live:
        return 0
    dev = live[0][0].device
    live = [(w, e) for w, e in live if w.device == dev]
    if dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)
    n = len(live)
    check(lib.rn_conv3x3_dgrad_weight_many(_ptr_array([w for w, _ in live]), _ptr_array([e.flipped for _, e in live]),
                                           _int_array([int(w.shape[0]) for w, _ in live]), _int_array([int(w.shape[1]) for w, _ in live]), n,
                                           torch.cuda.current_stream(dev).cuda_stream), "rn_conv3x3_dgrad_weight_many")
    for w, e in live:
        e.version, e.epoch = w._version, _DW_EPOCH
    return n


def dgrad_weights(ws: Sequence[Tensor], stream: int) -> List[Tensor]:
    """[Cin, Cout, 3, 3] data-gradient weights (taps reversed, roles swapped) of ``ws`` (all the same [Cout, Cin, 3, 3]): the step's table
    where it is current, else flipped here in one launch (and registered for the next ``refresh_dgrad_weights``)."""
    out: List[Optional[Tensor]] = []
    for w in ws:
        e = _DW_TABLE.get(id(w)) if DGRAD_WEIGHT_TABLE else None
        ok = e is not None and e.ref() is w and e.version == w._version and e.epoch == _DW_EPOCH
        out.append(e.flipped if ok else None)
    todo = [i for i, t in enumerate(out) if t is None]
    if todo:
        Cout, Cin = int(ws[0].shape[0]), int(ws[0].shape[1])
        srcs = [ws[i] if _cl(ws[i]) else ws[i].contiguous(memory_format=torch.channels_last) for i in todo]
        dsts = [torch.empty((Cin, Cout, 3, 3), dtype=ws[i].dtype, device=ws[i].device, memory_format=torch.channels_last) for i in todo]
        check(lib.rn_conv3x3_dgrad_weight_batched(_ptr_array(srcs), _ptr_array(dsts), len(todo), Cout, Cin, stream), "rn_conv3x3_dgrad_weight_batched")
        for i, d in zip(todo, dsts):
            out[i] = d
            w = ws[i]
            stale = _DW_TABLE.get(id(w))                         # (an id is reused once its tensor is gone: replace such an entry)
            if _dw_eligible(w) and w.is_leaf and (stale is None or stale.ref() is not w):
                _DW_TABLE[id(w)] = _FlippedWeight(w)              # (a parameter: flipped with the others from the next step on)
    return out


def _dgrad_weight(w: Tensor) -> Tensor:
    """Forward weight [Cout, Cin, 3, 3] -> the data gradient's weight [Cin, Kpad, 3, 3] (channels-last memory
    [Cin][3][3][Kpad]): taps reversed, channel roles swapped, and the contraction axis laid out as the kernel walks it
    (``rn_conv3x3_levels_to_canvas``): slot k = channel k below e = Cout - Cout % 8; if Cout % 8 != 0 the slots
    e .. e+7 carry channels Cout-8 .. Cout-1 with zero weight on the repeated ones; zeros up to Kpad."""
    Cout, Cin = w.shape[0], w.shape[1]
    Kpad = (Cout + 63) // 64 * 64
    if w.is_cuda and w.dtype in H16 and _cl(w) and Cout >= 8:
        # one launch (the torch form below is a flip, a fill and two or three strided copies: ~35 us of 5-us kernels per conv)
        out = torch.empty((Cin, Kpad, 3, 3), dtype=w.dtype, device=w.device, memory_format=torch.channels_last)
        check(lib.rn_conv3x3_levels_dgrad_weight(w.data_ptr(), out.data_ptr(), Cout, Cin, Kpad, torch.cuda.current_stream(w.device).cuda_stream),
              "rn_conv3x3_levels_dgrad_weight")
        return out
    wt = w.flip(2, 3).transpose(0, 1)                                   # [Cin, Cout, 3, 3]
    out = torch.empty((Cin, Kpad, 3, 3), dtype=w.dtype, device=w.device).fill_(0).contiguous(memory_format=torch.channels_last)
    e = Cout - Cout % 8
    out[:, :e] = wt[:, :e]
    if Cout % 8:
        s = 8 - Cout % 8                                                # repeated channels at the head of the last piece
        out[:, e + s: e + 8] = wt[:, e:]
    return out


class _ClsOutputConv(torch.autograd.Function):
    """``class_subnet_output`` on the zero-bordered canvas: the result is written per pyramid level as the dense
    ``[N, h*w*A, K]`` logits the loss / detection kernels stream (exactly Cout = A*K channels: no dead classes, no unpack
    copy); data and weight gradients gather the dense per-level gradients back (``csrc/conv.hip``, level modes)."""

    @staticmethod
    def forward(ctx, x, w, bias, canvas, num_classes, n_images, relu_link=None):
        ctx.relu_link = relu_link
        sheets, Cin, Hp, Wp = x.shape
        N = int(n_images)
        Cout = w.shape[0]
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        if not _cl(w):
            w = w.contiguous(memory_format=torch.channels_last)
        ys = [torch.empty((N, h * wd * (Cout // num_classes), num_classes), dtype=x.dtype, device=dev) for h, wd in canvas.shapes]
        # flop counted on the positions and channels that exist (the kernel also walks canvas gaps and pads Cout to 256s)
        real = N * sum(h * wd for h, wd in canvas.shapes)
        _mfma_call("mfma_cls_output_fwd", dev, 2.0 * real * Cout * 9 * Cin,
                   lambda: lib.rn_conv3x3_canvas_to_levels(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else 0,
                                                           _layout(canvas, N), _ptr_array(ys), _DT[x.dtype], sheets, Hp, Wp, Cin, Cout,
                                                           _zero_page(dev).data_ptr(), stream), "rn_conv3x3_canvas_to_levels")
        ctx.save_for_backward(x, w)
        ctx.canvas, ctx.has_bias, ctx.n_images = canvas, bias is not None, N
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        x, w = ctx.saved_tensors
        cv = ctx.canvas
        sheets, Cin, Hp, Wp = x.shape
        N = ctx.n_images
        Cout = w.shape[0]
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        gs = []
        for dy, (h, wd) in zip(dys, cv.shapes):
            if dy is None:
                dy = torch.empty((N, h * wd * Cout), dtype=x.dtype, device=dev).fill_(0)
            gs.append(dy.to(x.dtype).contiguous())
        lv = _layout(cv, N)
        real = N * sum(h * wd for h, wd in cv.shapes)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            wt = _dgrad_weight(w)
            _mfma_call("mfma_cls_output_dgrad", dev, 2.0 * real * Cout * 9 * Cin,
                       lambda: _levels_dgrad(ctx.relu_link, gs, lv, Cout, wt, cv, dx, x, sheets, Hp, Wp, Cin, stream),
                       "rn_conv3x3_levels_to_canvas")
        if ctx.needs_input_grad[1]:
            need = lib.rn_conv3x3_wgrad_workspace_bytes((Cout + 255) // 256, sheets * Hp * Wp)
            key = (dev.index, stream)
            wsb = _WG_WS.get(key)
            if wsb is None or wsb.numel() < need:
                wsb = _WG_WS[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
            dw = torch.empty((Cout, Cin, 3, 3), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
            _mfma_call("mfma_cls_output_wgrad", dev, 2.0 * real * Cout * 9 * Cin,
                       lambda: lib.rn_conv3x3_levels_wgrad(_ptr_array(gs), lv, Cout, x.data_ptr(), dw.data_ptr(), _DT[x.dtype], sheets, Hp, Wp,
                                                           Cin, _zero_page(dev).data_ptr(), wsb.data_ptr(), wsb.numel(), stream),
                       "rn_conv3x3_levels_wgrad")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum_levels(gs, Cout)
        return dx, dw, db, None, None, None, None


def _colsum_levels(gs: Sequence[Tensor], C_: int) -> Tensor:
    "f32[C] = column sums over all rows of the dense per-level tensors ``[N, rows * C]`` (bias gradient): one read, two launches."
    dev = gs[0].device
    if gs[0].dtype not in H16 or C_ % 2 or len(gs) > 6:
        return sum(g.reshape(-1, C_).sum(0, dtype=torch.float32) for g in gs)
    stream = torch.cuda.current_stream().cuda_stream
    need = lib.rn_colsum_rows_workspace_bytes(len(gs), C_)
    key = (dev.index, stream, "rows")
    wsb = _CS_WS.get(key)
    if wsb is None or wsb.numel() < need:
        wsb = _CS_WS[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
    out = torch.empty((C_,), dtype=torch.float32, device=dev)
    n = len(gs)
    check(lib.rn_colsum_rows(_ptr_array(gs), (C.c_int64 * n)(*[g.numel() // C_ for g in gs]), n, C_, _DT[gs[0].dtype], out.data_ptr(),
                             wsb.data_ptr(), wsb.numel(), stream), "rn_colsum_rows")
    return out


class _BoxOutputConv(torch.autograd.Function):
    """``box_subnet_output`` (256 -> A*4 = 36 channels) on the canvas: MIOpen forward + unpack to the dense per-level
    ``[N, h*w*A, 4]`` deltas; the DATA gradient gathers the per-level gradients straight from those dense tensors with the
    MFMA level-mode kernel (``rn_conv3x3_levels_to_canvas``: 9 K-tiles per tile instead of a 36-channel MIOpen igemm, 55 vs
    165 us); the weight gradient stays with MIOpen on the re-assembled canvas gradient."""

    @staticmethod
    def forward(ctx, x, w, bias, canvas, n_images, relu_link=None):
        ctx.relu_link = relu_link
        ctx.save_for_backward(x, w)
        ctx.canvas, ctx.n_images, ctx.has_bias = canvas, int(n_images), bias is not None
        N = ctx.n_images
        sheets, Cin, Hp, Wp = x.shape
        Cout = w.shape[0]
        if BOX_OUTPUT_FWD_MFMA and (bias is None or bias.dtype == torch.float32):
            # the narrow (<= 64 columns) variant of the MFMA level-mode kernel writes the dense per-level deltas directly
            dev = x.device
            if dev.index != torch.cuda.current_device():
                torch.cuda.set_device(dev)
            stream = torch.cuda.current_stream().cuda_stream
            wc = w if _cl(w) else w.contiguous(memory_format=torch.channels_last)
            ys = [torch.empty((N, h * wd * (Cout // 4), 4), dtype=x.dtype, device=dev) for h, wd in canvas.shapes]
            _mfma_call("mfma_box_output_fwd", dev, 2.0 * N * sum(h * wd for h, wd in canvas.shapes) * Cout * 9 * Cin,
                       lambda: lib.rn_conv3x3_canvas_to_levels(x.data_ptr(), wc.data_ptr(), bias.data_ptr() if bias is not None else 0,
                                                               _layout(canvas, N), _ptr_array(ys), _DT[x.dtype], sheets, Hp, Wp, Cin, Cout,
                                                               _zero_page(dev).data_ptr(), stream), "rn_conv3x3_canvas_to_levels")
            return tuple(ys)
        y = F.conv2d(x, w, bias.to(x.dtype) if bias is not None else None, stride=1, padding=1)
        outs = _gather_levels(canvas, y, N)
        return tuple(t.permute(0, 2, 3, 1).reshape(N, -1, 4) for t in outs)        # layers.py:189-191, zero-copy

    @staticmethod
    def backward(ctx, *dys):
        x, w = ctx.saved_tensors
        cv, N = ctx.canvas, ctx.n_images
        sheets, Cin, Hp, Wp = x.shape
        Cout = w.shape[0]
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        gs = []
        for dy, (h, wd) in zip(dys, cv.shapes):
            if dy is None:
                dy = torch.empty((N, h * wd * Cout), dtype=x.dtype, device=dev).fill_(0)
            gs.append(dy.to(x.dtype).contiguous())
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            wt = _dgrad_weight(w)
            _mfma_call("mfma_box_output_dgrad", dev, 2.0 * N * sum(h * wd for h, wd in cv.shapes) * Cout * 9 * Cin,
                       lambda: _levels_dgrad(ctx.relu_link, gs, _layout(cv, N), Cout, wt, cv, dx, x, sheets, Hp, Wp, Cin, stream),
                       "rn_conv3x3_levels_to_canvas")
        if (ctx.needs_input_grad[1] and BOX_OUTPUT_WGRAD_NARROW and Cout <= 64 and Cin % 64 == 0 and x.dtype in H16
                and wgrad_narrow_ok(torch.empty((64, Cin, 3, 3), dtype=x.dtype, device="meta"), (1, 1), x)):
            # Round 5: the canvas IS a zero-bordered channels-last tensor and the gradient is zero on its gaps, so the weight gradient of
            # the conv on the canvas is the weight gradient of a plain 3x3 / pad-1 conv over the sheets: csrc/wgrad3x3.hip (all nine
            # taps of a 64 x 64 block of dW per team of waves: x is read once per 64 output channels instead of once per tap) on the
            # per-level gradients scattered into a 64-channel canvas (Cout = 36 padded with zero channels).  110 us against 198 for the
            # narrow gathering variant of the position-contraction kernel (which stages the 256-channel x nine times).
            g36 = _scatter_levels(cv, [gl.view(N, h, wd, Cout).permute(0, 3, 1, 2) for gl, (h, wd) in zip(gs, cv.shapes)],
                                  (sheets, Cout, Hp, Wp), x.dtype, dev, N)
            g64 = F.pad(g36, (0, 0, 0, 0, 0, 64 - Cout)).contiguous(memory_format=torch.channels_last)
            dw64 = conv3x3_wgrad_narrow(g64, x, torch.empty((64, Cin, 3, 3), dtype=x.dtype, device="meta"), tag="mfma_box_output_wgrad",
                                        flop=2.0 * N * sum(h * wd for h, wd in cv.shapes) * Cout * 9 * Cin)
            dw = dw64[:Cout].contiguous(memory_format=torch.channels_last)
        elif ctx.needs_input_grad[1] and BOX_OUTPUT_WGRAD_MFMA and Cin == 256:
            # the narrow (<= 64 rows) variant of the gathering MFMA weight-gradient kernel
            need = lib.rn_conv3x3_wgrad_workspace_bytes(1, sheets * Hp * Wp)
            key = (dev.index, stream)
            wsb = _WG_WS.get(key)
            if wsb is None or wsb.numel() < need:
                wsb = _WG_WS[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
            dw = torch.empty((Cout, Cin, 3, 3), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
            _mfma_call("mfma_box_output_wgrad", dev, 2.0 * N * sum(h * wd for h, wd in cv.shapes) * Cout * 9 * Cin,
                       lambda: lib.rn_conv3x3_levels_wgrad(_ptr_array(gs), _layout(cv, N), Cout, x.data_ptr(), dw.data_ptr(), _DT[x.dtype], sheets, Hp, Wp,
                                                           Cin, _zero_page(dev).data_ptr(), wsb.data_ptr(), wsb.numel(), stream),
                       "rn_conv3x3_levels_wgrad")
        elif ctx.needs_input_grad[1]:
            g = _scatter_levels(cv, [gl.view(N, h, wd, Cout).permute(0, 3, 1, 2) for gl, (h, wd) in zip(gs, cv.shapes)],
                                (sheets, Cout, Hp, Wp), x.dtype, dev, N)
            dw = torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _colsum_levels(gs, Cout)
        return dx, dw, db, None, None, None


def box_output_conv_fusable(x: Tensor, conv, canvas: "Canvas") -> bool:
    return cls_output_conv_fusable(x, conv, canvas) and conv.out_channels % 4 == 0


def _levels_dgrad(relu_link, gs, lv, Cout, wt, cv, dx, x, sheets, Hp, Wp, Cin, stream) -> int:
    """Data gradient of a level-mode output conv.  ``relu_link = (TowerLink, i)``: x was output i of the last tower layer and feeds
    nothing else -- that layer's ReLU backward and bias gradient ride in this kernel's epilogue (``rn_conv3x3_levels_to_canvas_relu``)
    and the result is deposited in the link for ``_TowerConvPair.backward`` to pick up."""
    dev = x.device
    link, i = relu_link if relu_link is not None else (None, 0)
    if link is None or link.relu_masks is None or not FUSE_TOWER_RELU_BWD:
        return lib.rn_conv3x3_levels_to_canvas(_ptr_array(gs), lv, Cout, wt.data_ptr(), cv.mask.data_ptr(), dx.data_ptr(), _DT[x.dtype], sheets,
                                               Hp, Wp, wt.shape[1], Cin, _zero_page(dev).data_ptr(), stream)
    need = lib.rn_conv3x3_colsum_workspace_bytes(1, sheets * Hp * Wp, Cin)
    key = (dev.index, stream, "lv")
    wsb = _CS_WS.get(key)
    if wsb is None or wsb.numel() < need:
        wsb = _CS_WS[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
    db = torch.empty((Cin,), dtype=torch.float32, device=dev)
    rc = lib.rn_conv3x3_levels_to_canvas_relu(_ptr_array(gs), lv, Cout, wt.data_ptr(), link.relu_masks[i].data_ptr(), cv.mask.data_ptr(),
                                              dx.data_ptr(), db.data_ptr(), _DT[x.dtype], sheets, Hp, Wp, wt.shape[1], Cin,
                                              _zero_page(dev).data_ptr(), wsb.data_ptr(), wsb.numel(), stream)
    if rc == 0:
        link.half[i] = (dx.data_ptr(), dx._version, db)
    return rc


def box_output_conv(x: Tensor, conv, canvas: "Canvas", n_images: int, relu_link=None) -> List[Tensor]:
    "``conv(x)`` for the box-output conv on a canvas -> per-level deltas ``[n_images, h*w*A, 4]`` (dense)."
    return list(_BoxOutputConv.apply(x, conv.weight.to(x.dtype), conv.bias, canvas, n_images, relu_link))


def cls_output_conv_fusable(x: Tensor, conv, canvas: "Canvas") -> bool:
    "bf16 canvas with a zero border, 3x3 / stride 1 / pad 1, Cin == 256, an even number of output channels (>= 8), <= 6 levels."
    return (x.is_cuda and x.dtype in H16 and _cl(x) and canvas.pad == 1 and conv.kernel_size == (3, 3)
            and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.in_channels == 256 and conv.out_channels % 2 == 0 and 8 <= conv.out_channels <= 1024
            and len(canvas.shapes) <= 6 and canvas.slots <= 8 and x.shape[0] * canvas.H * canvas.W < (1 << 22)
            and (conv.bias is None or conv.bias.dtype == torch.float32))


def cls_output_conv(x: Tensor, conv, canvas: "Canvas", num_classes: int, n_images: Optional[int] = None, relu_link=None) -> List[Tensor]:
    """``conv(x)`` for the class-output conv on a canvas -> per-level logits ``[n_images, h*w*A, num_classes]`` (dense);
    ``n_images`` defaults to every slot of every sheet."""
    n = x.shape[0] * canvas.slots if n_images is None else int(n_images)
    return list(_ClsOutputConv.apply(x, conv.weight.to(x.dtype), conv.bias, canvas, num_classes, n, relu_link))


# ---------------------------------------------------------------------------------------------------
# The FPN's 3x3 output convs (retinanet/layers.py:34-38, applied at :62-64) on the dense MFMA kernels
DENSE_GROUP = True     # False: every level through its nn.Conv2d (MIOpen)
_DENSE_WS: Dict[tuple, Tensor] = {}


def dense_group_fusable(xs: Sequence[Tensor], convs) -> bool:
    "bf16 channels-last CUDA activations, 3x3 / stride 1 / pad 1, 256 -> 256 with a bias, at most 4 levels of one batch size."
    if not (DENSE_GROUP and 0 < len(xs) <= 4 and len(xs) == len(convs)):
        return False
    N = xs[0].shape[0]
    for x, conv in zip(xs, convs):
        if not (x.is_cuda and x.dtype in H16 and _cl(x) and x.shape[0] == N and x.shape[1] == 256 and
                N * x.shape[2] * x.shape[3] < (1 << 22) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and
                conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is not None and
                conv.in_channels == 256 and conv.out_channels == 256 and conv.bias.dtype == torch.float32):
            return False
    return True


def _int_array(v):
    return (C.c_int * len(v))(*[int(i) for i in v])


class _DenseConvGroup(torch.autograd.Function):
    """``[conv_p(x_p) for p]`` -- P <= 4 convolutions 3x3 / pad 1, 256 -> 256, each with its own weights and its own
    ``[N, 256, h_p, w_p]`` channels-last bf16 input -- as ONE launch of the MFMA implicit GEMM each way
    (``rn_conv3x3_dense_batched``: the levels' row tiles share the grid, taps that leave the image read zeros), one launch of
    the position-contraction weight-gradient kernel (``rn_conv3x3_dense_wgrad_batched``) and the bias fused into the
    forward epilogue.  Arguments: x_0 .. x_{P-1}, w_0 .. (bf16, channels-last), b_0 .. (f32)."""

    @staticmethod
    def forward(ctx, *args):
        P = len(args) // 3
        xs, ws, bs = args[:P], [w if _cl(w) else w.contiguous(memory_format=torch.channels_last) for w in args[P:2 * P]], args[2 * P:]
        dev = xs[0].device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        N = xs[0].shape[0]
        hs, wds = [x.shape[2] for x in xs], [x.shape[3] for x in xs]
        ys = [torch.empty_like(x) for x in xs]
        flop = sum(2.0 * N * h * w * 256 * 2304 for h, w in zip(hs, wds))
        _mfma_call(f"mfma_fpn_output_fwd_x{P}", dev, flop,
                   lambda: lib.rn_conv3x3_dense_batched(_ptr_array(xs), _ptr_array(ws), _ptr_array(bs), _ptr_array(ys), P, _DT[xs[0].dtype], N,
                                                        _int_array(hs), _int_array(wds), 256, 256, _zero_page(dev).data_ptr(), stream),
                   "rn_conv3x3_dense_batched")
        ctx.save_for_backward(*xs, *ws)
        ctx.geom = (P, N, hs, wds, flop)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        P, N, hs, wds, flop = ctx.geom
        xs, ws = ctx.saved_tensors[:P], ctx.saved_tensors[P:]
        dev = xs[0].device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        gs = [dy if (dy.dtype == xs[0].dtype and _cl(dy)) else dy.to(xs[0].dtype).contiguous(memory_format=torch.channels_last) for dy in dys]
        dxs = [None] * P
        if any(ctx.needs_input_grad[:P]):
            wts = dgrad_weights(list(ws), stream)
            dxs = [torch.empty_like(x) for x in xs]
            _mfma_call(f"mfma_fpn_output_dgrad_x{P}", dev, flop,
                       lambda: lib.rn_conv3x3_dense_batched(_ptr_array(gs), _ptr_array(wts), None, _ptr_array(dxs), P, _DT[xs[0].dtype], N,
                                                            _int_array(hs), _int_array(wds), 256, 256, _zero_page(dev).data_ptr(), stream),
                       "rn_conv3x3_dense_batched")
        dws = [None] * P
        if any(ctx.needs_input_grad[P:2 * P]):
            need = lib.rn_conv3x3_dense_wgrad_workspace_bytes(P)
            key = (dev.index, stream)
            wsb = _DENSE_WS.get(key)
            if wsb is None or wsb.numel() < need:
                wsb = _DENSE_WS[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
            dws = [torch.empty((256, 256, 3, 3), dtype=xs[0].dtype, device=dev, memory_format=torch.channels_last) for _ in range(P)]
            _mfma_call(f"mfma_fpn_output_wgrad_x{P}", dev, flop,
                       lambda: lib.rn_conv3x3_dense_wgrad_batched(_ptr_array(gs), _ptr_array(xs), _ptr_array(dws), P, _DT[xs[0].dtype], N, _int_array(hs),
                                                                  _int_array(wds), 256, 256, _zero_page(dev).data_ptr(), wsb.data_ptr(),
                                                                  wsb.numel(), stream),
                       "rn_conv3x3_dense_wgrad_batched")
        dbs = [(_colsum_levels([g.permute(0, 2, 3, 1).reshape(N, -1)], 256) if ctx.needs_input_grad[2 * P + p] else None) for p, g in enumerate(gs)]
        return (*dxs, *dws, *dbs)


def dense_conv_group(xs: Sequence[Tensor], convs) -> List[Tensor]:
    "``[conv(x) for x, conv in zip(xs, convs)]`` for ``dense_group_fusable`` inputs; weights are cast to the activations' dtype here (autocast's cast)."
    return list(_DenseConvGroup.apply(*xs, *[c.weight.to(xs[0].dtype) for c in convs], *[c.bias for c in convs]))


# ---------------------------------------------------------------------------------------------------
# Bottleneck conv2 of layer3 (3x3 / stride 1, 256 -> 256, 33 600 positions at the bench shape): forward stays on MIOpen (CK's
# kernel runs it at 1.2 PFLOP/s, a half-empty round of 256 x 256 tiles cannot match that), the two gradients run on the dense
# MFMA kernels: MIOpen's data gradient 95 us + weight gradient 87 us + 15 us of zero / cast helpers against 62 + 50 us.
CONV3X3_BWD = True
CONV3X3_FWD = True      # ... and the forward too (in the step CK's kernel takes 73-76 us on this shape, the dense kernel 59)


def conv3x3_bwd_fusable(conv, x: Tensor) -> bool:
    return (CONV3X3_BWD and x.is_cuda and x.dtype in H16 and conv.weight.dtype == x.dtype and _cl(x) and
            conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and
            conv.groups == 1 and conv.bias is None and conv.in_channels == 256 and conv.out_channels == 256 and
            torch.is_grad_enabled() and x.shape[0] * x.shape[2] * x.shape[3] < (1 << 22))


class _Conv3x3MfmaBwd(torch.autograd.Function):
    "``F.conv2d(x, w, None, 1, 1)`` (256 -> 256, bf16 channels-last) with both gradients on ``rn_conv3x3_dense_*`` (P = 1)."

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        if not CONV3X3_FWD:
            return F.conv2d(x, w, None, 1, 1)
        N, _, h, wd = x.shape
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        wc = w if _cl(w) else w.contiguous(memory_format=torch.channels_last)
        y = torch.empty_like(x)
        _mfma_call("mfma_conv2_fwd", dev, 2.0 * N * h * wd * 256 * 2304,
                   lambda: lib.rn_conv3x3_dense_batched(_ptr_array([x]), _ptr_array([wc]), None, _ptr_array([y]), 1, _DT[x.dtype], N,
                                                        _int_array([h]), _int_array([wd]), 256, 256, _zero_page(dev).data_ptr(), stream),
                   "rn_conv3x3_dense_batched")
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, _, h, wd = x.shape
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        g = dy if (dy.dtype == x.dtype and _cl(dy)) else dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        wc = w if _cl(w) else w.contiguous(memory_format=torch.channels_last)
        flop = 2.0 * N * h * wd * 256 * 2304
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt = dgrad_weights([w], stream)[0]
            dx = torch.empty_like(x)
            _mfma_call("mfma_conv2_dgrad", dev, flop,
                       lambda: lib.rn_conv3x3_dense_batched(_ptr_array([g]), _ptr_array([wt]), None, _ptr_array([dx]), 1, _DT[x.dtype], N,
                                                            _int_array([h]), _int_array([wd]), 256, 256, _zero_page(dev).data_ptr(), stream),
                       "rn_conv3x3_dense_batched")
        if ctx.needs_input_grad[1]:
            need = lib.rn_conv3x3_dense_wgrad_workspace_bytes(1)
            key = (dev.index, stream)
            wsb = _DENSE_WS.get(key)
            if wsb is None or wsb.numel() < need:
                wsb = _DENSE_WS[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
            dw = torch.empty((256, 256, 3, 3), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
            _mfma_call("mfma_conv2_wgrad", dev, flop,
                       lambda: lib.rn_conv3x3_dense_wgrad_batched(_ptr_array([g]), _ptr_array([x]), _ptr_array([dw]), 1, _DT[x.dtype], N, _int_array([h]),
                                                                  _int_array([wd]), 256, 256, _zero_page(dev).data_ptr(), wsb.data_ptr(),
                                                                  wsb.numel(), stream),
                       "rn_conv3x3_dense_wgrad_batched")
        return dx, dw


def conv3x3_mfma_bwd(conv, x: Tensor) -> Tensor:
    return _Conv3x3MfmaBwd.apply(x, conv.weight)


# ---------------------------------------------------------------------------------------------------
# The other 3x3 / stride-1 convolutions that stay on MIOpen (conv2 of layer1 / layer2 / layer4: 64, 128, 512 channels): the data
# gradient of a stride-1 / pad-1 3x3 convolution IS a 3x3 / pad-1 convolution of the output gradient with the flipped, transposed
# weights.  MIOpen's forward kernels (CK) beat its backward-data kernels on every one of these shapes and need no zero fill of the
# result first (step timeline, layer1: 78 us against 93 + 18; layer2: 70 against 77 + 11; layer4: 101 against 119 + 8), so the data
# gradient is issued as a forward convolution; the weight flip is one launch of the kernel the MFMA data gradients already use.
DGRAD_AS_FWD = True

# The forward product at 64 channels on this library's own kernel (csrc/narrow3x3.hip: weights in registers, input rows in an LDS ring)
# instead of the CK grouped-convolution kernel MIOpen picks for it (46 against 72-76 us per launch in the R50 step; False keeps MIOpen's).
NARROW_FWD = True


def narrow_fwd_ok(x: Tensor, w: Tensor) -> bool:
    return (NARROW_FWD and x.is_cuda and x.dtype in H16 and w.dtype == x.dtype and x.dim() == 4 and _cl(x) and
            tuple(w.shape) == (64, 64, 3, 3) and x.shape[1] == 64 and x.numel() // 64 < (1 << 31))


def conv3x3_narrow_forward(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, relu: bool = False) -> Tensor:
    """``F.conv2d(x, w, None, 1, 1)`` for bf16 channels-last x [N, 64, H, W] and w [64, 64, 3, 3] on ``rn_conv3x3_narrow_forward``;
    with ``bias`` (f32 [64]): ``act(conv + bias)``, act = ReLU when ``relu`` (the folded-BatchNorm inference path)."""
    dev = x.device
    if dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream().cuda_stream
    wc = w if _cl(w) else w.contiguous(memory_format=torch.channels_last)
    N, C, H, W = x.shape
    y = torch.empty_like(x, memory_format=torch.channels_last)
    _mfma_call("mfma_conv2_narrow_fwd", dev, 2.0 * N * H * W * C * C * 9,
               lambda: lib.rn_conv3x3_narrow_forward(x.data_ptr(), wc.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), _DT[x.dtype], N, H, W, C,
                                                     int(bool(relu)), _zero_page(dev).data_ptr(), stream), "rn_conv3x3_narrow_forward")
    return y


# Inference (frozen BatchNorm folded into weight + bias): conv2 of the layer3 / layer4 bottlenecks on the dense mode of the head's MFMA
# kernel with the bias and the ReLU in its epilogue, instead of CK's kernel + an epilogue pass over the output.
DENSE_EVAL = True


def dense_eval_ok(x: Tensor, w: Tensor) -> bool:
    return (DENSE_EVAL and x.is_cuda and x.dtype in H16 and w.dtype == x.dtype and x.dim() == 4 and _cl(x) and _cl(w) and
            tuple(w.shape[2:]) == (3, 3) and w.shape[1] == x.shape[1] and w.shape[1] % 64 == 0 and w.shape[0] % 256 == 0 and
            x.shape[0] * x.shape[2] * x.shape[3] < (1 << 22))


def conv3x3_dense_bias_act(x: Tensor, w: Tensor, bias: Optional[Tensor], relu: bool) -> Tensor:
    "``act(F.conv2d(x, w, bias, 1, 1))`` (bf16 channels-last, Cin % 64 == 0, Cout % 256 == 0; bias f32) on ``rn_conv3x3_dense_batched_act``."
    dev = x.device
    if dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream().cuda_stream
    N, Cin, h, wd = x.shape
    Cout = int(w.shape[0])
    y = torch.empty((N, Cout, h, wd), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
    bs = (C.c_void_p * 1)(bias.data_ptr()) if bias is not None else None
    _mfma_call("mfma_conv2_eval", dev, 2.0 * N * h * wd * Cout * 9 * Cin,
               lambda: lib.rn_conv3x3_dense_batched_act(_ptr_array([x]), _ptr_array([w]), bs, _ptr_array([y]), 1, _DT[x.dtype], N, _int_array([h]),
                                                        _int_array([wd]), Cin, Cout, _zero_page(dev).data_ptr(), int(bool(relu)), stream),
               "rn_conv3x3_dense_batched_act")
    return y


DENSE_SPLITK = True     # few-row-tile convolutions (conv2 of layer4: forward and, with flipped weights, data gradient) on the dense MFMA kernel, K split
_SPLITK_WS: Dict[tuple, Tensor] = {}


def dense_splitk_bytes(x: Tensor, w: Tensor) -> int:
    if not (DENSE_SPLITK and x.is_cuda and x.dtype in H16 and w.dtype == x.dtype and x.dim() == 4 and _cl(x) and _cl(w) and
            tuple(w.shape[2:]) == (3, 3) and w.shape[1] == x.shape[1] and w.shape[1] % 64 == 0 and w.shape[0] % 256 == 0 and
            x.shape[0] * x.shape[2] * x.shape[3] < (1 << 22)):
        return 0
    return int(lib.rn_conv3x3_dense_splitk_workspace_bytes(x.shape[0], x.shape[2], x.shape[3], int(w.shape[0])))


def conv3x3_dense_splitk(x: Tensor, w: Tensor, need: int, tag: str = "mfma_conv2_splitk") -> Tensor:
    "``F.conv2d(x, w, None, 1, 1)`` on ``rn_conv3x3_dense_splitk`` (``need`` = ``dense_splitk_bytes(x, w)`` > 0); no autograd."
    dev = x.device
    if dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream().cuda_stream
    N, Cin, h, wd = x.shape
    Cout = int(w.shape[0])
    key = (dev.index, stream)
    ws = _SPLITK_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _SPLITK_WS[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
    y = torch.empty((N, Cout, h, wd), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
    _mfma_call(tag, dev, 2.0 * N * h * wd * Cout * 9 * Cin,
               lambda: lib.rn_conv3x3_dense_splitk(x.data_ptr(), w.data_ptr(), y.data_ptr(), _DT[x.dtype], N, h, wd, Cin, Cout, _zero_page(dev).data_ptr(),
                                                   ws.data_ptr(), ws.numel(), stream), "rn_conv3x3_dense_splitk")
    return y


DENSE_BAND = True       # 128-channel 3x3 convolutions (conv2 of layer2: forward and data gradient) on the band-staged dense kernel (False: MIOpen / CK)
DENSE_BAND_MAX_COUT = 128


def dense_band_ok(x: Tensor, w: Tensor) -> bool:
    return (DENSE_BAND and x.is_cuda and x.dtype in H16 and w.dtype == x.dtype and x.dim() == 4 and _cl(x) and _cl(w) and
            tuple(w.shape[2:]) == (3, 3) and w.shape[1] == x.shape[1] and w.shape[1] % 64 == 0 and w.shape[0] % 128 == 0 and
            w.shape[0] <= DENSE_BAND_MAX_COUT and x.shape[0] * x.shape[2] * x.shape[3] * max(int(w.shape[0]), int(w.shape[1])) < (1 << 31))


def conv3x3_dense_band(x: Tensor, w: Tensor, tag: str = "mfma_conv2_band") -> Tensor:
    "``F.conv2d(x, w, None, 1, 1)`` on ``rn_conv3x3_dense_band``; no autograd."
    dev = x.device
    if dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream().cuda_stream
    N, Cin, h, wd = x.shape
    Cout = int(w.shape[0])
    y = torch.empty((N, Cout, h, wd), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
    _mfma_call(tag, dev, 2.0 * N * h * wd * Cout * 9 * Cin,
               lambda: lib.rn_conv3x3_dense_band(x.data_ptr(), w.data_ptr(), y.data_ptr(), _DT[x.dtype], N, h, wd, Cin, Cout, _zero_page(dev).data_ptr(),
                                                 stream), "rn_conv3x3_dense_band")
    return y


DENSE_BAND_STATS = True   # ... with the statistics of the BatchNorm that follows (bn2 of the layer2 blocks) in its epilogue


def conv3x3_dense_band_stats(x: Tensor, w: Tensor, tag: str = "mfma_conv2_band"):
    """``conv3x3_dense_band`` + (sum y, sum y^2) per channel of the stored output, no second pass over it -> (y, partial f32
    [tiles, 2, Cout], tiles) for ``rn_bn_stats_finalize``."""
    dev = x.device
    if dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream().cuda_stream
    N, Cin, h, wd = x.shape
    Cout = int(w.shape[0])
    tiles = int(lib.rn_conv3x3_dense_band_tiles(N, h, wd))
    y = torch.empty((N, Cout, h, wd), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
    partial = torch.empty((tiles * 2 * Cout,), dtype=torch.float32, device=dev)
    _mfma_call(tag, dev, 2.0 * N * h * wd * Cout * 9 * Cin,
               lambda: lib.rn_conv3x3_dense_band_stats(x.data_ptr(), w.data_ptr(), y.data_ptr(), partial.data_ptr(), _DT[x.dtype], N, h, wd, Cin, Cout,
                                                       _zero_page(dev).data_ptr(), stream), "rn_conv3x3_dense_band_stats")
    return y, partial, tiles


def conv3x3_same(x: Tensor, w: Tensor) -> Tensor:
    "``F.conv2d(x, w, None, 1, 1)`` (no autograd): the narrow / band / K-split dense MFMA kernels where they apply, else MIOpen."
    if narrow_fwd_ok(x, w):
        return conv3x3_narrow_forward(x, w)
    if dense_band_ok(x, w):
        return conv3x3_dense_band(x, w)
    need = dense_splitk_bytes(x, w)
    if need > 0:
        return conv3x3_dense_splitk(x, w if _cl(w) else w.contiguous(memory_format=torch.channels_last), need)
    return F.conv2d(x, w, None, 1, 1)


def dgrad_as_fwd_ok(w: Tensor, stride, x: Tensor) -> bool:
    return (DGRAD_AS_FWD and x.is_cuda and x.dtype in H16 and w.dtype == x.dtype and _cl(x) and w.dim() == 4 and
            tuple(w.shape[2:]) == (3, 3) and tuple(stride) == (1, 1) and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0)


def conv3x3_dgrad_as_fwd(g: Tensor, w: Tensor) -> Tensor:
    "Data gradient of ``F.conv2d(x, w, None, 1, 1)`` (w [Cout, Cin, 3, 3] bf16) for the output gradient ``g``: a forward convolution."
    dev = g.device
    if dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream().cuda_stream
    wt = dgrad_weights([w], stream)[0]
    gc = g if (g.dtype == w.dtype and _cl(g)) else g.to(w.dtype).contiguous(memory_format=torch.channels_last)
    return conv3x3_same(gc, wt)


# Their weight gradient: one team of waves holds all nine taps of a 64 x 64 block of dW (csrc/wgrad3x3.hip) instead of one
# workgroup per tap re-reading both operands (MIOpen: 123 us + a zero fill + a cast per layer1 block).
NARROW_WGRAD = True
_NARROW_WS: Dict[tuple, Tensor] = {}


def wgrad_narrow_ok(w: Tensor, stride, x: Tensor) -> bool:
    return (NARROW_WGRAD and x.is_cuda and x.dtype in H16 and w.dtype == x.dtype and _cl(x) and w.dim() == 4 and
            tuple(w.shape[2:]) == (3, 3) and tuple(stride) == (1, 1) and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0 and
            w.shape[0] * w.shape[1] <= 512 * 512)


def conv3x3_wgrad_narrow(g: Tensor, x: Tensor, w: Tensor, tag: str = "mfma_conv2_narrow_wgrad", flop: Optional[float] = None) -> Tensor:
    "Weight gradient of ``F.conv2d(x, w, None, 1, 1)`` (bf16 channels-last, Cout / Cin multiples of 64) -> like ``w``, channels-last."
    dev = x.device
    if dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream().cuda_stream
    N, Cin, H, W = x.shape
    Cout = int(w.shape[0])
    gc = g if (g.dtype == x.dtype and _cl(g)) else g.to(x.dtype).contiguous(memory_format=torch.channels_last)
    need = lib.rn_conv3x3_wgrad_narrow_workspace_bytes(Cout, Cin)
    key = (dev.index, stream)
    ws = _NARROW_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _NARROW_WS[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
    dw = torch.empty((Cout, Cin, 3, 3), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
    _mfma_call(tag, dev, flop if flop is not None else 2.0 * N * H * W * Cout * Cin * 9,
               lambda: lib.rn_conv3x3_wgrad_narrow(gc.data_ptr(), x.data_ptr(), dw.data_ptr(), _DT[x.dtype], N, H, W, Cout, Cin,
                                                   _zero_page(dev).data_ptr(), ws.data_ptr(), ws.numel(), stream), "rn_conv3x3_wgrad_narrow")
    return dw


def conv3x3_weight_gradient(g: Tensor, x: Tensor, w: Tensor) -> Tensor:
    "Weight gradient of a 3x3 / stride-1 / pad-1 convolution: the narrow MFMA kernel where it applies, else MIOpen's."
    if wgrad_narrow_ok(w, (1, 1), x):
        return conv3x3_wgrad_narrow(g, x, w)
    return torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]


class _Conv3x3DgradAsFwd(torch.autograd.Function):
    "``F.conv2d(x, w, None, 1, 1)``; backward: data gradient as a forward convolution, weight gradient on csrc/wgrad3x3.hip."

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return conv3x3_same(x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        g = dy if (dy.dtype == x.dtype and _cl(dy)) else dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dx = conv3x3_dgrad_as_fwd(g, w) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dw = conv3x3_weight_gradient(g, x, w)
        return dx, dw


def conv3x3_dgrad_fwd_fusable(conv, x: Tensor) -> bool:
    return (dgrad_as_fwd_ok(conv.weight, conv.stride, x) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and
            conv.bias is None and torch.is_grad_enabled())


def conv3x3_dgrad_fwd(conv, x: Tensor) -> Tensor:
    return _Conv3x3DgradAsFwd.apply(x, conv.weight)
