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
