"""``MasterSGD`` -- ``torch.optim.SGD`` (the reference's optimizer, ``hparams.yaml:63-68``) on fp32 master weights,
with the convolution weights of the model held in bf16 for the autocast forward.

Why: under bf16 autocast PyTorch re-casts every fp32 conv weight to bf16 in each forward and every bf16 weight
gradient back to fp32 in each backward (94 + 92 small kernels per R50-FPN step), then runs the foreach SGD kernels.
``use_bf16_conv_weights(model)`` swaps every 4-D fp32 conv weight for its bf16 rounding (what autocast would have
fed the convolution anyway) and parks the fp32 tensor as the parameter's master; ``MasterSGD.step()`` does the
whole update in fp32 on the masters -- same arithmetic and order as ``torch.optim.SGD`` -- and refreshes the bf16
copies, for all parameters in one HIP launch per 48 tensors (``rn_sgd_master_step``, ``csrc/optim.hip``).
The trajectory is the autocast + SGD one (fp32 masters, bf16-rounded weights in the forward, bf16 weight gradients
promoted exactly); a converted model must run under autocast.  ``master_state_dict`` / ``load_master_state_dict``
give and take fp32 checkpoints with the reference's keys.
"""
import ctypes as C
from typing import Dict, Iterable, List, Optional

import torch
from torch import Tensor, nn

from ._lib import RN_BF16, RN_F16, check, lib
from .norm import note_raw_write


def use_16bit_conv_weights(model: nn.Module, dtype: torch.dtype = torch.bfloat16) -> int:
    """Convert every 4-D fp32 parameter (conv weights) to ``dtype`` (bf16 or fp16: the autocast dtype of the run) in place, keeping the
    fp32 values as ``p.master``.  Returns the number of converted parameters.  BatchNorm parameters and biases stay fp32."""
    if dtype not in (torch.bfloat16, torch.float16):
        raise TypeError(f"working copies are bf16 or fp16, not {dtype}")
    n = 0
    for p in model.parameters():
        if p.dim() == 4 and p.dtype == torch.float32 and p.is_cuda:
            master = p.data
            p.data = master.to(dtype)                     # preserves the memory format (channels_last stays)
            p.master = master
            if p.grad is not None:
                p.grad = None
            n += 1
    return n


def use_bf16_conv_weights(model: nn.Module) -> int:
    return use_16bit_conv_weights(model, torch.bfloat16)


def master_state_dict(model: nn.Module) -> Dict[str, Tensor]:
    "``model.state_dict()`` with every converted weight replaced by its fp32 master (checkpoint format of the reference)."
    sd = model.state_dict()
    for name, p in model.named_parameters():
        if hasattr(p, "master"):
            sd[name] = p.master.detach().clone()
    return sd


def load_master_state_dict(model: nn.Module, state: Dict[str, Tensor], strict: bool = True):
    "Load an fp32 checkpoint into a converted model: masters take the fp32 values, the bf16 copies their rounding."
    out = model.load_state_dict({k: v for k, v in state.items()}, strict=strict)      # copies (rounding) into the bf16 params
    with torch.no_grad():
        for name, p in model.named_parameters():
            if hasattr(p, "master") and name in state:
                p.master.copy_(state[name])
                p.data.copy_(p.master)
    return out


class MasterSGD(torch.optim.Optimizer):
    # torch.amp.GradScaler.step() hands such an optimizer `grad_scale` / `found_inf` (device scalars) instead of unscaling the gradients
    # and reading found_inf back on the host: the kernel divides and skips on the device (rn_sgd_master_step_ex), nothing synchronises
    _step_supports_amp_scaling = True

    def __init__(self, params: Iterable, lr: float = 1e-3, momentum: float = 0.0, dampening: float = 0.0,
                 weight_decay: float = 0.0, nesterov: bool = False):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov))

    @torch.no_grad()
    def step(self, closure=None, grads: Optional[Dict[Tensor, Tensor]] = None):
        """``grads``: optional ``{param: fp32 gradient}`` overriding ``param.grad`` (the fp32 views of
        ``parallel.BucketedGradAllReduce`` after the exchange)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        note_raw_write()                   # masters, bf16 copies and BN affine parameters change without a _version bump
        for group in self.param_groups:
            masters, moms, gptrs, p16s, ns = [], [], [], [], []
            grads16 = None
            dt16 = None
            first = None
            keep: List[Tensor] = []
            for p in group["params"]:
                g = grads.get(p) if grads is not None else None
                if g is None:
                    g = p.grad
                if g is None:
                    continue
                has16 = hasattr(p, "master")
                w = p.master if has16 else p.data
                if w.dtype != torch.float32 or not p.is_cuda:
                    raise TypeError("MasterSGD handles CUDA fp32 parameters and bf16 parameters converted by use_bf16_conv_weights")
                st = self.state[p]
                if "momentum_buffer" not in st:
                    # (under a GradScaler the very first step may be SKIPPED by found_inf: the buffer then has to hold zeros, with
                    # which the next step's momentum * buf + (1 - dampening) * g is torch's first-step buf = g -- for dampening == 0
                    # only; a fill kernel, not a memset: graph.py)
                    amp = getattr(self, "found_inf", None) is not None
                    if amp and group["momentum"] != 0 and group["dampening"] != 0:
                        raise ValueError("MasterSGD under loss scaling needs dampening == 0: a first step skipped by found_inf leaves a zero "
                                         "momentum buffer, and the next step's momentum * 0 + (1 - dampening) * g is not torch.optim.SGD's first-step buf = g")
                    st["momentum_buffer"] = (torch.empty_like(w).fill_(0) if amp else torch.empty_like(w)) if group["momentum"] != 0 else None
                    st["steps"] = 0
                if first is None:
                    first = st["steps"] == 0
                elif first != (st["steps"] == 0):
                    raise RuntimeError("parameters of one group must have taken the same number of steps")
                st["steps"] += 1
                if has16:
                    if dt16 is None:
                        dt16 = p.dtype
                    elif dt16 != p.dtype:
                        raise RuntimeError("the 16-bit working copies of one group must share a dtype")
                    is16 = g.dtype == p.dtype
                    if not is16 and g.dtype != torch.float32:
                        raise TypeError(f"unsupported gradient dtype {g.dtype} for a {p.dtype} working copy")
                    if grads16 is None:
                        grads16 = is16
                    elif grads16 != is16:
                        raise RuntimeError("gradients of the 16-bit parameters must be all 16-bit or all fp32")
                elif g.dtype != torch.float32:
                    raise TypeError("fp32 parameters need fp32 gradients")
                # same memory order for master / momentum / gradient / bf16 copy: all carry the parameter's strides
                if g.stride() != w.stride():
                    g = g.contiguous(memory_format=torch.channels_last) if w.dim() == 4 and w.is_contiguous(memory_format=torch.channels_last) \
                        else g.contiguous()
                    keep.append(g)
                masters.append(w.data_ptr()); moms.append(st["momentum_buffer"].data_ptr() if st["momentum_buffer"] is not None else 0)
                gptrs.append(g.data_ptr()); p16s.append(p.data.data_ptr() if has16 else 0); ns.append(w.numel())
            n = len(masters)
            if n == 0:
                continue
            dev = group["params"][0].device
            scale, found = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)       # (set by GradScaler.step around this call)
            for t in (scale, found):
                if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.numel() == 1):
                    raise TypeError("grad_scale / found_inf must be CUDA fp32 scalars (torch.amp.GradScaler)")
            with torch.cuda.device(dev):
                check(lib.rn_sgd_master_step_ex((C.c_void_p * n)(*masters), (C.c_void_p * n)(*moms), (C.c_void_p * n)(*gptrs),
                                                (C.c_void_p * n)(*p16s), (C.c_int64 * n)(*ns), n, int(bool(grads16)),
                                                RN_F16 if dt16 == torch.float16 else RN_BF16, float(group["lr"]),
                                                float(group["momentum"]), float(group["dampening"]), float(group["weight_decay"]),
                                                int(group["nesterov"]), int(bool(first)), scale.data_ptr() if scale is not None else None,
                                                found.data_ptr() if found is not None else None, torch.cuda.current_stream().cuda_stream),
                      "rn_sgd_master_step_ex")
        from . import biasact
        biasact.invalidate_dgrad_weights()           # (the kernel wrote the parameters through raw pointers: no version counter moved)
        return loss
