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
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from ._lib import RN_BF16, RN_F16, RN_F32, check, lib

_DT = {torch.float32: RN_F32, torch.bfloat16: RN_BF16, torch.float16: RN_F16}
_WS: Dict[tuple, Tensor] = {}


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
MFMA_WGRAD = os.environ.get("RN_MFMA_WGRAD", "1") != "0"      # weight gradient of the canvas convs on the MFMA kernel (0: MIOpen)


def _canvas_wgrad(gs, xs, ws, Wp: int, stream: int):
    """Weight gradients of P canvas convs (256 -> 256, bf16) by ``rn_conv3x3_canvas_wgrad_batched``; None when the
    shapes are outside the kernel's range (the caller then asks MIOpen)."""
    x0, w0 = xs[0], ws[0]
    if not (MFMA_WGRAD and x0.dtype == torch.bfloat16 and tuple(w0.shape) == (256, 256, 3, 3)):
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
    check(lib.rn_conv3x3_canvas_wgrad_batched(_ptr_array(gs), _ptr_array(xs), _ptr_array(dws), P, _DT[x0.dtype], M, Wp, 256, 256,
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
        check(lib.rn_conv3x3_canvas(x.data_ptr(), w.data_ptr(), bias.data_ptr(), mask.data_ptr(), y.data_ptr(), _DT[x.dtype],
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
            check(lib.rn_conv3x3_canvas(g.data_ptr(), wt.data_ptr(), 0, mask.data_ptr(), dx.data_ptr(), _DT[x.dtype],
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


class _TowerConvPair(torch.autograd.Function):
    """``_TowerConv`` for two towers of identical geometry (cls and box) in one batched launch each way: their tiles
    together fill the chip's workgroup waves (2 x 813 tiles: 7 waves of 256 instead of 2 x 4)."""

    @staticmethod
    def forward(ctx, x0, x1, w0, w1, b0, b1, mask):
        N, Cin, Hp, Wp = x0.shape
        Cout = w0.shape[0]
        dev = x0.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        w0 = w0 if _cl(w0) else w0.contiguous(memory_format=torch.channels_last)
        w1 = w1 if _cl(w1) else w1.contiguous(memory_format=torch.channels_last)
        ys = [torch.empty((N, Cout, Hp, Wp), dtype=x0.dtype, device=dev, memory_format=torch.channels_last) for _ in range(2)]
        check(lib.rn_conv3x3_canvas_batched(_ptr_array([x0, x1]), _ptr_array([w0, w1]), _ptr_array([b0, b1]), mask.data_ptr(),
                                            _ptr_array(ys), 2, _DT[x0.dtype], N * Hp * Wp, Hp * Wp, Wp, Cin, Cout, 1, stream),
              "rn_conv3x3_canvas_batched")
        ctx.save_for_backward(x0, x1, w0, w1, ys[0], ys[1], mask)
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
        gs, dbs = [], []
        wp, wn = _workspace(dev, stream, Cout)
        for dy, y in ((dy0, y0), (dy1, y1)):
            if dy.dtype != x0.dtype or not _cl(dy):
                dy = dy.to(x0.dtype).contiguous(memory_format=torch.channels_last)
            g = torch.empty_like(dy)
            db = torch.empty((Cout,), dtype=torch.float32, device=dev)
            check(lib.rn_bias_act_backward(dy.data_ptr(), y.data_ptr(), mask.data_ptr(), g.data_ptr(), db.data_ptr(), _DT[x0.dtype],
                                           M, Cout, Hp * Wp, 1, wp, wn, stream), "rn_bias_act_backward")
            gs.append(g); dbs.append(db)
        dxs = [None, None]
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            wts = [w.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last) for w in (w0, w1)]
            dxs = [torch.empty_like(x0), torch.empty_like(x1)]
            check(lib.rn_conv3x3_canvas_batched(_ptr_array(gs), _ptr_array(wts), None, mask.data_ptr(), _ptr_array(dxs), 2,
                                                _DT[x0.dtype], M, Hp * Wp, Wp, Cout, Cin, 0, stream), "rn_conv3x3_canvas_batched")
        dws = _canvas_wgrad(gs, [x0, x1], [w0, w1], Wp, stream)
        if dws is None:
            dws = [torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
                   for g, x, w in ((gs[0], x0, w0), (gs[1], x1, w1))]
        return dxs[0], dxs[1], dws[0], dws[1], dbs[0], dbs[1], None


def tower_conv_pair(x0: Tensor, x1: Tensor, w0: Tensor, w1: Tensor, b0: Tensor, b1: Tensor, mask: Tensor):
    "Two ``tower_conv`` of identical geometry (Cin == Cout % 256 == 0) in one launch each way."
    return _TowerConvPair.apply(x0, x1, w0.to(x0.dtype), w1.to(x0.dtype), b0, b1, mask)


def tower_conv_fusable(x: Tensor, conv) -> bool:
    "The MFMA canvas conv covers bf16, 3x3 / stride 1 / pad 1, Cin % 64 == 0 and Cout % 256 == 0 (head towers: 256 -> 256)."
    # (its data gradient runs on the same kernel when Cin % 256 == 0 as well, else on MIOpen)
    return (x.is_cuda and x.dtype == torch.bfloat16 and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and
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
    """Placement of L feature maps in one canvas: level 0 on top, the others side by side below it,
    one empty row / column between neighbours; ``pad`` extra empty rows / columns all around (the MFMA canvas
    conv wants a one-pixel zero border instead of bounds checks)."""

    def __init__(self, shapes: Sequence[Tuple[int, int]], device: torch.device, pad: int = 0):
        self.shapes = [(int(h), int(w)) for h, w in shapes]
        self.pad = pad
        h0, w0 = self.shapes[0]
        self.origin: List[Tuple[int, int]] = [(pad, pad)]
        col, below = 0, 0
        for h, w in self.shapes[1:]:
            self.origin.append((pad + h0 + 1, pad + col))
            col += w + 1
            below = max(below, h)
        self.H = h0 + (1 + below if len(self.shapes) > 1 else 0) + 2 * pad
        self.W = max(w0, col - 1) + 2 * pad
        m = torch.zeros((self.H, self.W), dtype=torch.uint8)
        for (r, c), (h, w) in zip(self.origin, self.shapes):
            m[r:r + h, c:c + w] = 1
        self.mask = m.reshape(-1).to(device)
        self.fill = sum(h * w for h, w in self.shapes) / float(self.H * self.W)

    _cache: Dict[tuple, "Canvas"] = {}

    @classmethod
    def of(cls, feature_maps: Sequence[Tensor], pad: int = 0) -> "Canvas":
        key = (tuple(tuple(f.shape[-2:]) for f in feature_maps), feature_maps[0].device, pad)
        c = cls._cache.get(key)
        if c is None:
            c = cls._cache[key] = Canvas(key[0], key[1], pad)
        return c


class _Pack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, canvas: Canvas, *feats):
        f0 = feats[0]
        out = torch.empty((f0.shape[0], f0.shape[1], canvas.H, canvas.W), dtype=f0.dtype, device=f0.device,
                          memory_format=torch.channels_last).zero_()
        for f, (r, c), (h, w) in zip(feats, canvas.origin, canvas.shapes):
            out[:, :, r:r + h, c:c + w].copy_(f)
        ctx.canvas = canvas
        return out

    @staticmethod
    def backward(ctx, g):
        cv = ctx.canvas
        return (None,) + tuple(g[:, :, r:r + h, c:c + w].contiguous(memory_format=torch.channels_last)
                               for (r, c), (h, w) in zip(cv.origin, cv.shapes))


class _Unpack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, canvas: Canvas, x):
        ctx.canvas = canvas
        ctx.meta = (x.shape, x.dtype, x.device)
        return tuple(x[:, :, r:r + h, c:c + w].contiguous(memory_format=torch.channels_last)
                     for (r, c), (h, w) in zip(canvas.origin, canvas.shapes))

    @staticmethod
    def backward(ctx, *grads):
        cv = ctx.canvas
        shape, dt, dev = ctx.meta
        g = torch.empty(shape, dtype=dt, device=dev, memory_format=torch.channels_last).zero_()
        for gl, (r, c), (h, w) in zip(grads, cv.origin, cv.shapes):
            if gl is not None:
                g[:, :, r:r + h, c:c + w].copy_(gl)
        return None, g


def pack_levels(canvas: Canvas, feature_maps: Sequence[Tensor]) -> Tensor:
    return _Pack.apply(canvas, *feature_maps)


def unpack_levels(canvas: Canvas, x: Tensor) -> List[Tensor]:
    return list(_Unpack.apply(canvas, x))


# ---------------------------------------------------------------------------------------------------
# The MFMA conv on ordinary dense tensors (stride-1 3x3 convs of layer3 and the FPN smoothing convs)
_ZEROS: Dict[int, Tensor] = {}


def _zero_page(dev: torch.device) -> Tensor:
    z = _ZEROS.get(dev.index)
    if z is None:
        z = _ZEROS[dev.index] = torch.zeros((256,), dtype=torch.uint8, device=dev)
    return z


class _Conv3x3Dense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        if not _cl(w):
            w = w.contiguous(memory_format=torch.channels_last)
        y = torch.empty((N, Cout, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        check(lib.rn_conv3x3_nhwc(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else 0, y.data_ptr(), _DT[x.dtype],
                                  N, H, W, Cin, Cout, 0, _zero_page(dev).data_ptr(), stream), "rn_conv3x3_nhwc")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        if dy.dtype != x.dtype or not _cl(dy):
            dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dx = dw = db = None
        if ctx.needs_input_grad[0] and Cin % 256 == 0:
            wt = w.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
            dx = torch.empty_like(x)
            check(lib.rn_conv3x3_nhwc(dy.data_ptr(), wt.data_ptr(), 0, dx.data_ptr(), _DT[x.dtype], N, H, W, Cout, Cin, 0,
                                      _zero_page(dev).data_ptr(), stream), "rn_conv3x3_nhwc")
        need = [ctx.needs_input_grad[0] and dx is None, ctx.needs_input_grad[1], False]
        if need[0] or need[1]:
            r = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, need)
            dx = r[0] if need[0] else dx
            dw = r[1] if need[1] else None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty((Cout,), dtype=torch.float32, device=dev)
            wp, wn = _workspace(dev, stream, Cout)
            check(lib.rn_bias_act_backward(dy.data_ptr(), 0, 0, 0, db.data_ptr(), _DT[x.dtype], N * H * W, Cout, 1, 0, wp, wn, stream),
                  "rn_bias_act_backward")
        return dx, dw, db


class _Conv3x3MfmaWgrad(torch.autograd.Function):
    """A 3x3 / stride-1 / pad-1, 256 -> 256 bf16 conv whose forward and data gradient stay on MIOpen and whose WEIGHT
    gradient runs on the MFMA position-contraction GEMM (``rn_conv3x3_nhwc_wgrad``) -- the one piece where the
    hand-written kernel is clearly ahead on these shapes (MIOpen's split-K wrw + zero / cast kernels)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        y = F.conv2d(x, w, None if bias is None else bias.to(x.dtype), padding=1)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, Cin, H, W = x.shape
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        if dy.dtype != x.dtype or not _cl(dy):
            dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            need = lib.rn_conv3x3_wgrad_workspace_bytes(1, N * H * W)
            key = (dev.index, stream)
            wsb = _WG_WS.get(key)
            if wsb is None or wsb.numel() < need:
                wsb = _WG_WS[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
            dw = torch.empty((256, 256, 3, 3), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
            check(lib.rn_conv3x3_nhwc_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), _DT[x.dtype], N, H, W, 256, 256,
                                            _zero_page(dev).data_ptr(), wsb.data_ptr(), wsb.numel(), stream), "rn_conv3x3_nhwc_wgrad")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty((256,), dtype=torch.float32, device=dev)
            wp, wn = _workspace(dev, stream, 256)
            check(lib.rn_bias_act_backward(dy.data_ptr(), 0, 0, 0, db.data_ptr(), _DT[x.dtype], N * H * W, 256, 1, 0, wp, wn, stream),
                  "rn_bias_act_backward")
        return dx, dw, db


# off by default: a tie with MIOpen inside the step on the layer3 / FPN shapes (216.2 / 217.3 vs 215.8 / 217.3 images/s)
MFMA_DENSE_WGRAD = os.environ.get("RN_MFMA_DENSE_WGRAD", "0") == "1"
MFMA_CONV_MIN_POSITIONS = 30000       # below this the kernel's 256-row tiles leave most CUs idle and MIOpen is as fast
# Off by default: in isolation the kernel beats MIOpen on these shapes (layer3 conv2 78 vs 122 us, FPN P3 232 vs 272 us),
# but inside the train step the A/B is a tie (211.3 vs 211.4 images/s on one box), so the stock path stays.  RN_MFMA_CONV=1.
MFMA_DENSE_CONV = os.environ.get("RN_MFMA_CONV", "0") == "1"


def conv3x3(conv, x: Tensor) -> Tensor:
    """``conv(x)`` for a 3x3 / stride-1 / pad-1 ``nn.Conv2d``: on the MFMA kernel when it is the faster one (bf16 CUDA
    channels-last input, Cin % 64 == 0, Cout % 256 == 0, >= 30 000 output positions), else the module itself."""
    if (x.is_cuda and x.dtype == torch.bfloat16 and _cl(x) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels % 64 == 0
            and conv.out_channels % 256 == 0 and x.shape[0] * x.shape[2] * x.shape[3] >= MFMA_CONV_MIN_POSITIONS
            and (conv.bias is None or conv.bias.dtype == torch.float32) and MFMA_DENSE_CONV):
        return _Conv3x3Dense.apply(x, conv.weight.to(x.dtype), conv.bias)
    if (MFMA_DENSE_WGRAD and x.is_cuda and x.dtype == torch.bfloat16 and _cl(x) and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels == 256
            and conv.out_channels == 256 and x.shape[0] * x.shape[2] * x.shape[3] >= MFMA_CONV_MIN_POSITIONS
            and (conv.bias is None or conv.bias.dtype == torch.float32) and torch.is_grad_enabled() and conv.weight.requires_grad):
        w = conv.weight.to(x.dtype)
        return _Conv3x3MfmaWgrad.apply(x, w if _cl(w) else w.contiguous(memory_format=torch.channels_last), conv.bias)
    return conv(x)


class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        y = torch.empty((N, Cout, H, W), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        check(lib.rn_conv1x1_nhwc(x.data_ptr(), w.data_ptr(), 0, y.data_ptr(), _DT[x.dtype], N * H * W, Cin, Cout,
                                  torch.cuda.current_stream().cuda_stream), "rn_conv1x1_nhwc")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        if dy.dtype != x.dtype or not _cl(dy):
            dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dx = dw = None
        if ctx.needs_input_grad[0] and Cin % 256 == 0 and Cout >= 128:
            wt = w.reshape(Cout, Cin).t().contiguous()                      # [Cin][Cout]
            dx = torch.empty_like(x)
            check(lib.rn_conv1x1_nhwc(dy.data_ptr(), wt.data_ptr(), 0, dx.data_ptr(), _DT[x.dtype], N * H * W, Cout, Cin,
                                      torch.cuda.current_stream().cuda_stream), "rn_conv1x1_nhwc")
        need = [ctx.needs_input_grad[0] and dx is None, ctx.needs_input_grad[1], False]
        if need[0] or need[1]:
            r = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, need)
            dx = r[0] if need[0] else dx
            dw = r[1] if need[1] else None
        return dx, dw


MFMA_CONV1X1 = os.environ.get("RN_MFMA_1X1", "0") == "1"


def conv1x1(conv, x: Tensor) -> Tensor:
    "``conv(x)`` for a bias-free 1x1 / stride-1 ``nn.Conv2d``; opt-in (RN_MFMA_1X1=1) MFMA GEMM path for A/B measurements."
    if (MFMA_CONV1X1 and x.is_cuda and x.dtype == torch.bfloat16 and _cl(x) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None and conv.in_channels % 64 == 0
            and conv.in_channels >= 128 and conv.out_channels % 256 == 0):
        w = conv.weight.to(x.dtype)
        return _Conv1x1.apply(x, w if w.is_contiguous() or _cl(w) else w.contiguous())
    return conv(x)
