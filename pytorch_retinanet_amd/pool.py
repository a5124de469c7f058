"""``FusedMaxPool2d`` -- the stem's ``nn.MaxPool2d(3, 2, 1)`` (reference ``retinanet/backbone.py:251``) on the HIP
kernels of ``csrc/pool.hip`` for channels-last CUDA activations: the arg-max is kept as one byte per output element
instead of PyTorch's int64 index (whose NHWC kernels take 176 + 463 us on the R50 stem), and the backward gathers
from it.  Anything else takes ``F.max_pool2d``."""
import torch
import torch.nn.functional as F
from torch import Tensor, nn

from ._lib import RN_BF16, RN_F16, RN_F32, check, lib

_DT = {torch.float32: RN_F32, torch.bfloat16: RN_BF16, torch.float16: RN_F16}


class _MaxPool3x3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        dev = x.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        y = torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        arg = torch.empty(y.shape, dtype=torch.uint8, device=dev, memory_format=torch.channels_last) if x.requires_grad else None
        check(lib.rn_maxpool3x3s2_forward(x.data_ptr(), y.data_ptr(), arg.data_ptr() if arg is not None else 0, _DT[x.dtype],
                                          N, H, W, C, torch.cuda.current_stream().cuda_stream), "rn_maxpool3x3s2_forward")
        ctx.save_for_backward(arg)
        ctx.meta = (x.shape, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        (N, C, H, W), dt = ctx.meta
        dev = dy.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        if dy.dtype != dt or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(dt).contiguous(memory_format=torch.channels_last)
        dx = torch.empty((N, C, H, W), dtype=dt, device=dev, memory_format=torch.channels_last)
        check(lib.rn_maxpool3x3s2_backward(arg.data_ptr(), dy.data_ptr(), dx.data_ptr(), _DT[dt], N, H, W, C,
                                           torch.cuda.current_stream().cuda_stream), "rn_maxpool3x3s2_backward")
        return dx


class FusedMaxPool2d(nn.MaxPool2d):
    def forward(self, x: Tensor) -> Tensor:
        if (x.is_cuda and x.dim() == 4 and x.dtype in _DT and x.shape[1] % 8 == 0 and x.numel() > 0
                and x.is_contiguous(memory_format=torch.channels_last)
                and self.kernel_size == 3 and self.stride == 2 and self.padding == 1 and self.dilation == 1
                and not self.ceil_mode and not self.return_indices):
            return _MaxPool3x3s2.apply(x)
        return super().forward(x)


class _AddUpsample2x(torch.autograd.Function):
    """``lat + nearest_upsample_2x(top)`` of the FPN's top-down pathway in one pass (``rn_fpn_add_upsample2x``); backward:
    the lateral's gradient is the incoming one, the top's is its 2 x 2 block sums (``rn_fpn_upsample2x_backward``)."""

    @staticmethod
    def forward(ctx, lat, top):
        N, C, H, W = lat.shape
        dev = lat.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        out = torch.empty_like(lat)
        check(lib.rn_fpn_add_upsample2x(lat.data_ptr(), top.data_ptr(), out.data_ptr(), _DT[lat.dtype], N, H, W, C,
                                        torch.cuda.current_stream().cuda_stream), "rn_fpn_add_upsample2x")
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = g.shape
        dev = g.device
        if dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        dtop = None
        if ctx.needs_input_grad[1]:
            dtop = torch.empty((N, C, H // 2, W // 2), dtype=g.dtype, device=dev, memory_format=torch.channels_last)
            check(lib.rn_fpn_upsample2x_backward(g.data_ptr(), dtop.data_ptr(), _DT[g.dtype], N, H // 2, W // 2, C,
                                                 torch.cuda.current_stream().cuda_stream), "rn_fpn_upsample2x_backward")
        return (g if ctx.needs_input_grad[0] else None), dtop


def add_upsample2x(lat: Tensor, top: Tensor):
    "``lat + nearest_upsample_2x(top)`` on the fused kernel, or None when the tensors are not what it takes."
    if (lat.is_cuda and lat.dim() == 4 and lat.dtype in _DT and top.dtype == lat.dtype and lat.shape[1] % 8 == 0
            and lat.shape[0] == top.shape[0] and lat.shape[1] == top.shape[1] and lat.shape[2] == 2 * top.shape[2]
            and lat.shape[3] == 2 * top.shape[3] and lat.is_contiguous(memory_format=torch.channels_last)
            and top.is_contiguous(memory_format=torch.channels_last) and lat.numel() > 0):
        return _AddUpsample2x.apply(lat, top)
    return None
