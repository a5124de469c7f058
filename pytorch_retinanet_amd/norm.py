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
"""
import ctypes as C
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from ._lib import check, lib
from .ops import _dtype_code, _ptr, _stream


def _cl(t: Tensor) -> bool:
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, num_batches_tracked, training, momentum, eps, relu):
        N, Cc, H, W = x.shape
        M = N * H * W
        dev = x.device
        y = torch.empty_like(x)                                   # keeps the channels_last strides
        stats = torch.empty((4, Cc), dtype=torch.float32, device=dev)   # save_mean, save_invstd, coef a, coef b
        ws_bytes = lib.rn_bn_workspace_bytes(Cc)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev) if training else None
        with torch.cuda.device(dev):
            check(lib.rn_bn_act_forward(_ptr(x), _ptr(residual), _ptr(y), _dtype_code(x), M, Cc, _ptr(weight), _ptr(bias),
                                        _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked), int(training),
                                        float(momentum), float(eps), int(relu), _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]),
                                        _ptr(ws), ws_bytes if training else 0, _stream(dev)), "rn_bn_act_forward")
        ctx.save_for_backward(x, y if relu else None, weight, stats)
        ctx.cfg = (bool(training), bool(relu), residual is not None, M, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, stats = ctx.saved_tensors
        training, relu, has_res, M, Cc = ctx.cfg
        dev = x.device
        if not _cl(dy) or dy.dtype != x.dtype:
            dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        grads = torch.empty((5, Cc), dtype=torch.float32, device=dev)     # dgamma, dbeta, coef a, k0, k1
        ws_bytes = lib.rn_bn_workspace_bytes(Cc)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            check(lib.rn_bn_act_backward(_ptr(dy), _ptr(y), _ptr(x), _ptr(dx), _ptr(dres), _dtype_code(x), M, Cc, _ptr(weight),
                                         _ptr(stats[0]), _ptr(stats[1]), int(training), int(relu), _ptr(grads[0]), _ptr(grads[1]),
                                         _ptr(grads[2]), _ptr(ws), ws_bytes, _stream(dev)), "rn_bn_act_backward")
        dgamma = grads[0].to(weight.dtype) if weight is not None else None
        dbeta = grads[1].to(weight.dtype) if weight is not None else None
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None, None


class FusedBatchNorm2d(nn.BatchNorm2d):
    def _fusable(self, x: Tensor, residual: Optional[Tensor]) -> bool:
        if not (x.is_cuda and _cl(x) and x.shape[1] % 8 == 0 and x.dtype in (torch.float32, torch.bfloat16, torch.float16)):
            return False
        if not (self.affine and self.track_running_stats and self.momentum is not None):
            return False
        if self.weight.dtype != torch.float32 or x.numel() == 0:
            return False
        if residual is not None and not (residual.shape == x.shape and residual.dtype == x.dtype and _cl(residual)):
            return False
        return True

    def forward(self, x: Tensor, relu: bool = False, residual: Optional[Tensor] = None) -> Tensor:
        if self._fusable(x, residual):
            return _BNAct.apply(x, residual, self.weight, self.bias, self.running_mean, self.running_var,
                                self.num_batches_tracked if self.training else None, self.training, self.momentum,
                                self.eps, relu)
        y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
