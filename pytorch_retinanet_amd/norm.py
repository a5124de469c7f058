"""``FusedBatchNorm2d`` -- ``nn.BatchNorm2d`` (same parameters, buffers and state-dict keys) whose
forward can also take the residual and the ReLU that follow it in a ResNet block:

    y = bn(x)                          ->  bn(x)
    y = relu(bn(x))                    ->  bn(x, relu=True)
    y = relu(bn(x) + identity)         ->  bn(x, relu=True, residual=identity)

(reference: ``retinanet/backbone.py:70-80``, ``:118-136``, ``:248-250``).  For channels-last CUDA
activations this runs the fused HIP kernels of ``csrc/norm.hip`` (3 launches forward, 3 backward,
instead of MIOpen's 7 batch-norm kernels plus separate add / ReLU / ReLU-backward kernels); anything
else (CPU tensors, NCHW layout, odd channel counts) takes the ordinary PyTorch ops, so the conv stack
still runs anywhere -- the *dense head* is the part that has no CPU path.

The wrapper is on the host's critical path (53 layers x forward/backward per step), so it keeps the
Python work per call small: one scratch allocation, one shared reduction workspace per device
(kernels of one stream are ordered, the workspace is dead when a call's last kernel has run), raw
pointers, no device-context switch when the tensor already lives on the current device.
"""
import ctypes as C
from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from ._lib import RN_BF16, RN_F16, RN_F32, check, lib

_DT = {torch.float32: RN_F32, torch.bfloat16: RN_BF16, torch.float16: RN_F16}
_WS: Dict[tuple, Tensor] = {}          # (device index, stream) -> reduction workspace
# Bumped whenever the library writes parameters or BN buffers THROUGH RAW POINTERS (the fused training forward updates the
# running statistics, ``optim.MasterSGD.step`` the weights, a captured step replays both): torch's ``_version`` counters do
# not see those writes, so anything derived from such tensors (``backbone._folded``) stamps itself with this counter too.
RAW_WRITES = [0]


def note_raw_write() -> None:
    RAW_WRITES[0] += 1

_fwd, _bwd = lib.rn_bn_act_forward, lib.rn_bn_act_backward


def _workspace(dev: torch.device, stream: int, C_: int):
    need = lib.rn_bn_workspace_bytes(C_)
    key = (dev.index, stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _WS[key] = torch.empty((max(need, lib.rn_bn_workspace_bytes(2048)),), dtype=torch.uint8, device=dev)
    return ws.data_ptr(), ws.numel()


def _cl(t: Tensor) -> bool:
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, num_batches_tracked, training, momentum, eps, relu, link=None):
        N, Cc, H, W = x.shape
        M = N * H * W
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        y = torch.empty_like(x)                                   # keeps the channels_last strides
        stats = torch.empty((4 * Cc,), dtype=torch.float32, device=dev)   # save_mean | save_invstd | coef a | coef b
        sp = stats.data_ptr()
        wp, wn = _workspace(dev, stream, Cc) if training else (0, 0)
        # ReLU mask for backward: recomputed from x when there is no residual; with a residual the forward writes one bit
        # per element (1/16 of the bytes of y) and backward reads that instead of y
        bits = torch.empty((M * Cc // 8,), dtype=torch.uint8, device=dev) if (relu and residual is not None and any(ctx.needs_input_grad[0:3])) else None
        check(_fwd(x.data_ptr(), residual.data_ptr() if residual is not None else 0, y.data_ptr(), _DT[x.dtype], M, Cc,
                   weight.data_ptr(), bias.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(),
                   num_batches_tracked.data_ptr() if num_batches_tracked is not None else 0, int(training), momentum, eps,
                   int(relu), sp, sp + 4 * Cc, sp + 8 * Cc, bits.data_ptr() if bits is not None else 0, wp, wn, stream),
              "rn_bn_act_forward")
        if training and running_mean is not None:
            RAW_WRITES[0] += 1                                    # running statistics updated behind torch's back
        ctx.save_for_backward(x, bits, weight, stats)
        ctx.cfg = (bool(training), bool(relu), residual is not None, M, Cc)
        ctx.link = link
        return y

    @staticmethod
    def backward(ctx, dy):
        x, bits, weight, stats = ctx.saved_tensors
        training, relu, has_res, M, Cc = ctx.cfg
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream
        if dy.dtype != x.dtype or not _cl(dy):
            dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        grads = torch.empty((5 * Cc,), dtype=torch.float32, device=dev)     # dgamma | dbeta | coef a | k0 | k1
        gp, sp = grads.data_ptr(), stats.data_ptr()
        wp, wn = _workspace(dev, stream, Cc)
        check(_bwd(dy.data_ptr(), bits.data_ptr() if bits is not None else 0, x.data_ptr(), dx.data_ptr(),
                   dres.data_ptr() if dres is not None else 0, _DT[x.dtype], M, Cc, weight.data_ptr(), sp, sp + 4 * Cc,
                   sp + 8 * Cc, int(training), (2 if bits is not None else 1) if relu else 0, gp, gp + 4 * Cc, gp + 8 * Cc, wp, wn,
                   stream), "rn_bn_act_backward")
        if ctx.link is not None and dres is not None:
            # the residual branch's gradient is handed to the consumer that adds it inside its own kernel (backbone.Bottleneck:
            # conv1's data gradient is one GEMM with this tensor as its accumulator input) instead of to autograd's add
            ctx.link.dres = dres
            dres = None
        return dx, dres, grads[:Cc], grads[Cc:2 * Cc], None, None, None, None, None, None, None, None


class FusedBatchNorm2d(nn.BatchNorm2d):
    def _fusable(self, x: Tensor, residual: Optional[Tensor]) -> bool:
        if not (x.is_cuda and x.dtype in _DT and x.shape[1] % 8 == 0 and _cl(x) and x.numel()):
            return False
        if not (self.affine and self.track_running_stats and self.momentum is not None and self.weight.dtype == torch.float32):
            return False
        if residual is not None and not (residual.shape == x.shape and residual.dtype == x.dtype and _cl(residual)):
            return False
        return True

    def forward(self, x: Tensor, relu: bool = False, residual: Optional[Tensor] = None, link=None) -> Tensor:
        """``link`` (an object with a ``dres`` attribute, fused path only): backward leaves the residual branch's gradient there
        instead of returning it -- for a caller that adds it elsewhere (``backbone.Bottleneck``)."""
        if self._fusable(x, residual):
            return _BNAct.apply(x, residual, self.weight, self.bias, self.running_mean, self.running_var,
                                self.num_batches_tracked if self.training else None, self.training, self.momentum,
                                self.eps, relu, link)
        assert link is None, "the residual-gradient hand-over needs the fused BN path"
        y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
