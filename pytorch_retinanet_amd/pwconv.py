"""Backbone convolutions on the hand-written MFMA GEMMs of ``csrc/pw.hip`` with the BatchNorm work fused into their operand
loads and epilogues -- the training path of ``backbone.Bottleneck`` (reference ``retinanet/backbone.py:105-136``:
conv1x1 -> bn -> relu -> conv3x3 -> bn -> relu -> conv1x1 -> bn -> (+ identity) -> relu, and its autograd backward).

Round 2 ran every one of those layers as its own pass over the activation (MIOpen conv, BN statistics, BN apply, and in
backward BN sums, BN apply, data gradient, weight gradient, residual add).  At the R50 shapes the bottleneck convolutions of
layer1 / layer2 are HBM-bound (51 - 102 flop per byte), so the passes, not the flops, were the cost.  One
``_BottleneckFn`` per block now runs

  forward   conv1  [+ column sums of z1]                    -> bn1 finalize -> a1 = relu(bn1(z1))      (materialised for conv2)
            conv2  (MIOpen 3x3)                              -> bn2 statistics
            conv3  [relu(bn2(z2)) applied in the operand load, + column sums of z3]   -> bn3 finalize
            out = relu(bn3(z3) + identity)                   (one pass, ReLU bits kept for backward)
  backward  bn3 sums -> conv3 data gradient [bn3-backward applied to (g, z3) in the operand load; ReLU mask of a2 and the two
            bn2-backward sums in the epilogue] and conv3 weight gradient [same operand transform; a2 recomputed from z2]
            -> bn2 finalize + apply -> conv2 backward (MIOpen) -> bn1 backward
            -> conv1 data gradient [+ the identity branch's gradient g * bits in the epilogue: no add pass, no copy]
            and conv1 weight gradient.

Neither a2 = relu(bn2(z2)), nor dz3, nor the identity branch's gradient is ever written to memory.  Everything else
(eval-mode BN, fp32 / fp16 models, NCHW tensors, CPU) takes the layer-by-layer path of ``backbone.py``.
"""
import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import norm
from ._lib import (RN_BF16, RN_F16, RN_PW_EPI_BIAS, RN_PW_EPI_RELU_BWD, RN_PW_EPI_RESID, RN_PW_EPI_STATS, RN_PW_PRO_AFFINE_RELU, RN_PW_PRO_BN_BWD,
                   RnPwConv, RnPwEpilogue, RnPwPrologue, check, lib)
from .ops import _timed

FUSED_BOTTLENECK = os.environ.get("RN_FUSED_BOTTLENECK", "1") != "0"      # 0: the layer-by-layer path of round 2 (A/B)
# Widest bottleneck (mid channels) that takes the fused block: layer1 / layer2 of a ResNet-50 are HBM-bound and gain from the
# fusion; at layer3 / layer4 (256 / 512 mid channels, K up to 2048) the GEMMs are compute-bound and the 128 x 128 register-staged
# tiles of csrc/pw.hip run at a third of MIOpen's rate (measured: conv3 data gradient 99 us against 36 + 22 us)
FUSED_MAX_MID = int(os.environ.get("RN_FUSED_MAX_MID", "128"))
FUSE_BWD_CHAIN = True          # ... and, in backward, the previous block's bn3-backward sums with its own conv1 data gradient: rn_pw_dgrad_resid_sums
_BWD_CHAIN: Dict[int, tuple] = {}      # data_ptr of a block's input gradient -> (partials, rows, shape, dtype); emptied by every trunk forward
CONV3_FWD_WALKER = True        # conv3 forward (+ bn2 apply in the operand load, bn3 statistics) on the row-tile walker kernel instead of pw_gemm_kernel
FUSE_CHAIN = True              # a fused block forms its output together with the NEXT fused block's conv1 (+ bn1 statistics): rn_pw_block_out_conv1
FUSE_CONV3_BWD = True          # conv3's data and weight gradients in one pass over the block-output gradient (layer1 / layer2 shapes)
DEFER_WGRAD_REDUCE = True      # a fused block sums the splits of its 1x1 weight gradients in one launch at the end of its backward
_WG_WS: Dict[tuple, Tensor] = {}
H16 = (torch.bfloat16, torch.float16)          # element types of csrc/pw.hip / stem.hip (the same kernels on v_mfma_*_bf16 / _f16)
_DT16 = {torch.bfloat16: RN_BF16, torch.float16: RN_F16}
PW_FLOP: Dict[str, float] = {}        # useful flop per call of the timed pw launches (bench.py)


def _cl(t: Tensor) -> bool:
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)


def _desc(x: Tensor, n_out: int, taps: int, stride: int) -> Tuple[RnPwConv, Tuple[int, int, int, int]]:
    "Geometry of a conv over channels-last ``x`` [Nimg, Cin, H, W] -> ([Nimg, n_out, Ho, Wo])."
    Nimg, Cin, H, W = x.shape
    pad = 1 if taps == 9 else 0
    k = 3 if taps == 9 else 1
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    return RnPwConv(Nimg * Ho * Wo, Cin, n_out, taps, stride, pad, Ho, Wo, H, W, _DT16[x.dtype]), (Nimg, n_out, Ho, Wo)


def _stream(dev: torch.device) -> int:
    if dev.index != torch.cuda.current_device():
        torch.cuda.set_device(dev)
    return torch.cuda.current_stream().cuda_stream


def affine_relu(coef: Tensor) -> RnPwPrologue:
    "x' = relu(x * a + b); coef = f32 [2 C] = (a | b), the forward coefficients of a BatchNorm."
    Cc = coef.numel() // 2
    p = coef.data_ptr()
    return RnPwPrologue(RN_PW_PRO_AFFINE_RELU, 0, p, p + 4 * Cc, 0, 0, 0, 0, 0)


def bn_bwd(coef3: Tensor, z: Tensor, relu_mode: int = 0, fwd_coef: Optional[Tensor] = None, bits: Optional[Tensor] = None) -> RnPwPrologue:
    "x' = a g' + k1 z + k0 with coef3 = f32 [3 C] = (a | k0 | k1); g' = g masked (2: recomputed from z and fwd_coef, 3: bits)."
    Cc = coef3.numel() // 3
    p = coef3.data_ptr()
    fa = fwd_coef.data_ptr() if fwd_coef is not None else 0
    return RnPwPrologue(RN_PW_PRO_BN_BWD, relu_mode, p, p + 4 * Cc, p + 8 * Cc, fa, fa + 4 * Cc if fa else 0, z.data_ptr(),
                        bits.data_ptr() if bits is not None else 0)


def pw_forward(x: Tensor, w: Tensor, stride: int = 1, pro: Optional[RnPwPrologue] = None, epi: Optional[RnPwEpilogue] = None,
               tag: str = "pw_fwd") -> Tensor:
    """``conv2d(pro(x), w, stride, padding = k // 2)`` on channels-last bf16 tensors; w [N, Cin, k, k] channels-last, k in {1, 3}."""
    taps = int(w.shape[2] * w.shape[3])
    d, oshape = _desc(x, int(w.shape[0]), taps, stride)
    y = torch.empty(oshape, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    st = _stream(x.device)
    PW_FLOP[tag] = 2.0 * d.M * d.N * taps * d.Cin
    with _timed(tag, x.device):
        check(lib.rn_pw_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(pro) if pro is not None else None,
                                     C.byref(epi) if epi is not None else None, st), "rn_pw_conv_forward")
    return y


def bias_act_epilogue(bias: Tensor, relu: bool, residual: Optional[Tensor] = None) -> RnPwEpilogue:
    "y = act(conv + bias[n] (+ residual)), act = ReLU when ``relu`` -- the GEMM's epilogue (inference: folded BatchNorm + identity + ReLU)."
    e = RnPwEpilogue(RN_PW_EPI_BIAS | (RN_PW_EPI_RESID if residual is not None else 0), 0, residual.data_ptr() if residual is not None else 0)
    e.relu, e.bias = int(bool(relu)), bias.data_ptr()
    return e


# Inference with frozen BatchNorm (backbone.conv_bn): a 1x1 convolution with its folded bias, the identity branch and the ReLU as ONE
# GEMM -- csrc/pw.hip where its contraction is short, hipBLASLt's bias / bias + ReLU epilogue where it is long (pw_gemm is 2 x behind
# hipBLASLt at 1024 - 2048 input channels, tools/pw_gemm_probe.py) -- instead of MIOpen's convolution + an epilogue pass over its output
# (R101 predict at 16 x 1344^2: 9 ms of 37 in those passes, 6.5 of them behind conv3).
EVAL_1X1_FUSED = True


def eval_conv1x1_ok(conv, x: Tensor, w: Tensor, residual: Optional[Tensor]) -> bool:
    if not (EVAL_1X1_FUSED and x.is_cuda and x.dtype in H16 and w.dtype == x.dtype and x.dim() == 4 and _cl(x) and _cl(w)):
        return False
    if conv.kernel_size != (1, 1) or conv.padding != (0, 0) or conv.dilation != (1, 1) or conv.groups != 1 or conv.stride not in ((1, 1), (2, 2)):
        return False
    Cout, Cin = int(w.shape[0]), int(w.shape[1])
    if Cin % 64 or Cout % 64 or x.shape[0] * x.shape[2] * x.shape[3] >= (1 << 31):
        return False
    return residual is None or (residual.dtype in H16 and _cl(residual))


def eval_conv1x1(conv, x: Tensor, w: Tensor, bias: Tensor, relu: bool, residual: Optional[Tensor]) -> Tensor:
    "``act(conv2d(x, w, stride) + bias (+ residual))`` for a 1x1 convolution: one GEMM with the rest in its epilogue."
    s = int(conv.stride[0])
    N, Cin, H, W = x.shape
    Cout = int(w.shape[0])
    if s == 1 and residual is None and MM_1X1 and _fwd_by_mm(N * H * W, Cin, Cout):
        x2, w2 = x.permute(0, 2, 3, 1).reshape(-1, Cin), w.reshape(Cout, Cin)
        b16 = getattr(bias, "_rn_b16", None)                     # (the folded bias lives as long as its fold-cache entry: cast it once, not per call)
        if b16 is None or b16.dtype != x.dtype:
            b16 = bias.to(x.dtype)
            bias._rn_b16 = b16
        y2 = torch._addmm_activation(b16, x2, w2.t()) if relu else torch.addmm(b16, x2, w2.t())
        return y2.view(N, H, W, Cout).permute(0, 3, 1, 2)
    return pw_forward(x, w, stride=s, epi=bias_act_epilogue(bias, relu, residual), tag="pw_eval_1x1")


def stats_epilogue(M: int, n_out: int, dev: torch.device) -> Tuple[RnPwEpilogue, Tensor, int]:
    nb = lib.rn_pw_walkers(M)
    partial = torch.empty((nb * 2 * n_out,), dtype=torch.float32, device=dev)
    return RnPwEpilogue(RN_PW_EPI_STATS, partial.data_ptr(), 0, 0, 0, 0, 0, 0, 0), partial, nb


def pw_wgrad(g: Tensor, x: Tensor, w_like: Tensor, stride: int = 1, gpro: Optional[RnPwPrologue] = None,
             xpro: Optional[RnPwPrologue] = None, tag: str = "pw_wgrad", defer: Optional[list] = None) -> Tensor:
    """Weight gradient of ``conv2d(xpro(x), w, stride)`` for the output gradient ``gpro(g)``; returns a tensor like ``w_like``.
    ``defer``: a list -- only the position-contraction kernel runs now, into a private partial buffer; the returned tensor is filled by
    ``pw_wgrad_flush(defer)``, which sums the splits of everything on the list in one launch."""
    taps = int(w_like.shape[2] * w_like.shape[3])
    d, oshape = _desc(x, int(w_like.shape[0]), taps, stride)
    assert tuple(g.shape) == oshape, (tuple(g.shape), oshape)
    dw = torch.empty_like(w_like)
    dev = x.device
    st = _stream(dev)
    need = lib.rn_pw_wgrad_workspace_bytes(C.byref(d))
    if defer is not None and DEFER_WGRAD_REDUCE and len(defer) < 8:
        ws = torch.empty((need,), dtype=torch.uint8, device=dev)
        S = C.c_int(0)
        PW_FLOP[tag] = 2.0 * d.M * d.N * taps * d.Cin
        with _timed(tag, dev):
            check(lib.rn_pw_conv_wgrad_partial(C.byref(d), g.data_ptr(), x.data_ptr(), C.byref(gpro) if gpro is not None else None,
                                               C.byref(xpro) if xpro is not None else None, ws.data_ptr(), ws.numel(), C.byref(S), st),
                  "rn_pw_conv_wgrad_partial")
        defer.append((ws, int(S.value), dw.numel(), dw))
        return dw
    key = (dev.index, st)
    ws = _WG_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = _WG_WS[key] = torch.empty((max(need, 32 << 20),), dtype=torch.uint8, device=dev)
    PW_FLOP[tag] = 2.0 * d.M * d.N * taps * d.Cin
    with _timed(tag, dev):
        check(lib.rn_pw_conv_wgrad(C.byref(d), g.data_ptr(), x.data_ptr(), dw.data_ptr(), C.byref(gpro) if gpro is not None else None,
                                   C.byref(xpro) if xpro is not None else None, ws.data_ptr(), ws.numel(), st), "rn_pw_conv_wgrad")
    return dw


def pw_wgrad_flush(defer: list) -> None:
    "Sum the splits of every weight gradient ``pw_wgrad(..., defer=defer)`` left on the list (one launch) and clear it."
    if not defer:
        return
    n = len(defer)
    assert all(e[3].dtype == defer[0][3].dtype for e in defer)
    check(lib.rn_pw_wgrad_reduce_many_dt((C.c_void_p * n)(*[e[0].data_ptr() for e in defer]), (C.c_int * n)(*[e[1] for e in defer]),
                                         (C.c_int64 * n)(*[e[2] for e in defer]), (C.c_void_p * n)(*[e[3].data_ptr() for e in defer]), n,
                                         _DT16[defer[0][3].dtype], _stream(defer[0][3].device)), "rn_pw_wgrad_reduce_many_dt")
    defer.clear()


# ---- BatchNorm pieces (csrc/norm.hip) --------------------------------------------------------------------------------------
def _bn_buffers(bn):
    return (bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr() if bn.num_batches_tracked is not None else 0)


def bn_finalize(partial: Tensor, nb: int, M: int, bn) -> Tensor:
    "Per-channel statistics from GEMM-epilogue partial sums -> f32 [4 C] = (mean | invstd | a | b); running statistics updated."
    Cc = bn.num_features
    stats = torch.empty((4 * Cc,), dtype=torch.float32, device=partial.device)
    sp = stats.data_ptr()
    rm, rv, nbt = _bn_buffers(bn)
    check(lib.rn_bn_stats_finalize(partial.data_ptr(), nb, M, Cc, bn.weight.data_ptr(), bn.bias.data_ptr(), rm, rv, nbt, bn.momentum, bn.eps,
                                   sp, sp + 4 * Cc, sp + 8 * Cc, _stream(partial.device)), "rn_bn_stats_finalize")
    norm.note_raw_write()
    return stats


def bn_stats(x: Tensor, bn) -> Tensor:
    "Statistics pass over channels-last ``x`` (no apply) -> f32 [4 C]."
    Nimg, Cc, H, W = x.shape
    M = Nimg * H * W
    dev = x.device
    st = _stream(dev)
    stats = torch.empty((4 * Cc,), dtype=torch.float32, device=dev)
    sp = stats.data_ptr()
    wp, wn = norm._workspace(dev, st, Cc)
    rm, rv, nbt = _bn_buffers(bn)
    check(lib.rn_bn_stats(x.data_ptr(), _DT16[x.dtype], M, Cc, bn.weight.data_ptr(), bn.bias.data_ptr(), rm, rv, nbt, bn.momentum, bn.eps,
                          sp, sp + 4 * Cc, sp + 8 * Cc, wp, wn, st), "rn_bn_stats")
    norm.note_raw_write()
    return stats


def bn_apply(x: Tensor, stats: Tensor, relu: bool, residual: Optional[Tensor] = None, want_bits: bool = False,
             res_stats: Optional[Tensor] = None):
    "``res_stats``: the residual is the INPUT of another BatchNorm whose statistics these are (its output is formed on the fly)."
    Nimg, Cc, H, W = x.shape
    M = Nimg * H * W
    y = torch.empty_like(x)
    bits = torch.empty((M * Cc // 8,), dtype=torch.uint8, device=x.device) if want_bits else None
    if res_stats is not None:
        assert relu and residual is not None
        check(lib.rn_bn_apply_res_affine(x.data_ptr(), residual.data_ptr(), res_stats.data_ptr() + 8 * Cc, y.data_ptr(), _DT16[x.dtype], M, Cc,
                                         stats.data_ptr() + 8 * Cc, bits.data_ptr() if bits is not None else 0, _stream(x.device)),
              "rn_bn_apply_res_affine")
        return y, bits
    check(lib.rn_bn_apply(x.data_ptr(), residual.data_ptr() if residual is not None else 0, y.data_ptr(), _DT16[x.dtype], M, Cc,
                          stats.data_ptr() + 8 * Cc, int(relu), bits.data_ptr() if bits is not None else 0, _stream(x.device)), "rn_bn_apply")
    return y, bits


class _BottleneckFn(torch.autograd.Function):
    """One ResNet bottleneck, BatchNorm in training mode, bf16 channels-last.  Tensor arguments (all receive gradients):
    x, conv1.weight, bn1.weight, bn1.bias, conv2.weight, bn2.weight, bn2.bias, conv3.weight, bn3.weight, bn3.bias and, for a
    block with a downsample branch, its conv weight and BN weight / bias (else three ``None``)."""

    @staticmethod
    def forward(ctx, blk, x, w1, g1, b1, w2, g2, b2, w3, g3, b3, wd, gd, bd):
        s = blk.conv2.stride[0]
        dev = x.device
        Nimg, Cin, H, W = x.shape
        M0 = Nimg * H * W
        # conv1 + bn1 statistics in its epilogue -- unless the block before has already formed them with its output (FUSE_CHAIN)
        cin = blk.__dict__.pop("_chain_in", None)
        if (cin is not None and cin[1] == x.data_ptr() and cin[2] == x._version and cin[0][3] == w1.data_ptr() and cin[0][4] == w1._version
                and tuple(cin[0][0].shape) == (Nimg, w1.shape[0], H, W) and cin[0][0].dtype == x.dtype):
            z1, p1, nb1 = cin[0][:3]
            prev = cin[0][5:8]             # the producer's z3, ReLU bits and bn3 statistics: its bn3-backward sums ride in this block's backward
        else:
            prev = (None, None, None)
            e1, p1, nb1 = stats_epilogue(M0, w1.shape[0], dev)
            z1 = pw_forward(x, w1, epi=e1, tag="pw_conv1_fwd")
        st1 = bn_finalize(p1, nb1, M0, blk.bn1)
        a1, _ = bn_apply(z1, st1, relu=True)                                 # conv2 is MIOpen's: it needs the activation
        from . import biasact
        st2 = None
        if s == 1 and tuple(blk.conv2.padding) == (1, 1) and biasact.DENSE_BAND_STATS and not biasact.narrow_fwd_ok(a1, w2) and biasact.dense_band_ok(a1, w2):
            # 128 channels: the band-staged dense kernel, bn2's statistics in its epilogue
            z2, pp2, nbb2 = biasact.conv3x3_dense_band_stats(a1, w2)
            st2 = bn_finalize(pp2, nbb2, z2.shape[0] * z2.shape[2] * z2.shape[3], blk.bn2)
        elif s == 1 and tuple(blk.conv2.padding) == (1, 1) and (biasact.narrow_fwd_ok(a1, w2) or biasact.dense_band_ok(a1, w2)):
            z2 = biasact.conv3x3_same(a1, w2)                                # 64 channels: csrc/narrow3x3.hip
        else:
            z2 = F.conv2d(a1, w2, None, blk.conv2.stride, blk.conv2.padding)
        if not _cl(z2):
            z2 = z2.contiguous(memory_format=torch.channels_last)
        if st2 is None:
            st2 = bn_stats(z2, blk.bn2)
        M1 = z2.shape[0] * z2.shape[2] * z2.shape[3]
        # conv3: relu(bn2(z2)) in the operand load, bn3 statistics in the epilogue
        Cm = w2.shape[0]
        nb3 = lib.rn_pw_conv3_forward_walkers(M1, Cm, int(w3.shape[0])) if CONV3_FWD_WALKER else 0
        if nb3 > 0:
            # the row-tile walker kernel (512 threads, one workgroup per CU, weight rows resident in LDS): same z3, statistics partials per walker
            C4o = int(w3.shape[0])
            z3 = torch.empty((z2.shape[0], C4o, z2.shape[2], z2.shape[3]), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
            p3 = torch.empty((nb3 * 2 * C4o,), dtype=torch.float32, device=dev)
            PW_FLOP["pw_conv3_fwd"] = 2.0 * M1 * Cm * C4o
            with _timed("pw_conv3_fwd", dev):
                check(lib.rn_pw_conv3_forward(M1, Cm, C4o, _DT16[x.dtype], z2.data_ptr(), st2.data_ptr() + 8 * Cm, w3.data_ptr(), z3.data_ptr(),
                                              p3.data_ptr(), _stream(dev)), "rn_pw_conv3_forward")
        else:
            e3, p3, nb3 = stats_epilogue(M1, w3.shape[0], dev)
            z3 = pw_forward(z2, w3, pro=affine_relu(st2[2 * Cm:]), epi=e3, tag="pw_conv3_fwd")
        st3 = bn_finalize(p3, nb3, M1, blk.bn3)
        zd = std = None
        if wd is not None:
            ed, pd, nbd = stats_epilogue(M1, wd.shape[0], dev)
            zd = pw_forward(x, wd, stride=blk.downsample[0].stride[0], epi=ed, tag="pw_down_fwd")
            std = bn_finalize(pd, nbd, M1, blk.downsample[1])
        C4 = int(w3.shape[0])
        nxt = blk.__dict__.get("_rn_next") if FUSE_CHAIN else None
        nbn = 0
        if nxt is not None and nxt[0].conv1.weight.dtype == x.dtype and nxt[0].conv1.weight.is_cuda and bottleneck_fusable(nxt[0], z3, in_forward=True):
            w1n = nxt[0].conv1.weight
            if tuple(w1n.shape[1:]) == (C4, 1, 1) and nxt[0].conv1.stride == (1, 1):
                nbn = lib.rn_pw_block_out_conv1_walkers(M1, C4, int(w1n.shape[0]))
        if nbn > 0:
            # the block output AND the next block's conv1 + bn1 statistics in one pass (y is written, not read back): csrc/pw.hip
            CN = int(w1n.shape[0])
            out = torch.empty_like(z3)
            bits = torch.empty((M1 * C4 // 8,), dtype=torch.uint8, device=dev)
            z1n = torch.empty((z3.shape[0], CN, z3.shape[2], z3.shape[3]), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
            partn = torch.empty((nbn * 2 * CN,), dtype=torch.float32, device=dev)
            p3o = st3.data_ptr()
            res = zd if wd is not None else x
            ra, rb = (std.data_ptr() + 8 * C4, std.data_ptr() + 12 * C4) if wd is not None else (0, 0)
            PW_FLOP["pw_block_out_conv1"] = 2.0 * M1 * C4 * CN
            with _timed("pw_block_out_conv1", dev):
                check(lib.rn_pw_block_out_conv1(M1, C4, CN, _DT16[x.dtype], z3.data_ptr(), res.data_ptr(), ra, rb, p3o + 8 * C4, p3o + 12 * C4,
                                                w1n.data_ptr(), out.data_ptr(), bits.data_ptr(), z1n.data_ptr(), partn.data_ptr(), _stream(dev)),
                      "rn_pw_block_out_conv1")
            blk.__dict__["_chain_tmp"] = (z1n, partn, nbn, w1n.data_ptr(), w1n._version, z3, bits, st3)
        elif wd is not None:
            # the branch's BatchNorm output is never written: the block-output pass forms it from zd and its coefficients
            out, bits = bn_apply(z3, st3, relu=True, residual=zd, res_stats=std, want_bits=True)
        else:
            out, bits = bn_apply(z3, st3, relu=True, residual=x, want_bits=True)
        ctx.save_for_backward(x, w1, g1, w2, g2, w3, g3, wd, gd, z1, a1, z2, z3, zd, bits, st1, st2, st3, std, *prev)
        ctx.blk = blk
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, w1, g1, w2, g2, w3, g3, wd, gd, z1, a1, z2, z3, zd, bits, st1, st2, st3, std, pz3, pbits, pst3 = ctx.saved_tensors
        blk = ctx.blk
        dev = x.device
        st = _stream(dev)
        if g_out.dtype != x.dtype or not _cl(g_out):
            g_out = g_out.to(x.dtype).contiguous(memory_format=torch.channels_last)
        C4, Cm = w3.shape[0], w2.shape[0]
        M1 = z3.shape[0] * z3.shape[2] * z3.shape[3]
        M0 = x.shape[0] * x.shape[2] * x.shape[3]
        # bn3 backward sums over (g_out * bits, z3) -> coefficients of dz3 = a g' + k1 z3 + k0
        gr3 = torch.empty((5 * C4,), dtype=torch.float32, device=dev)            # dgamma | dbeta | a | k0 | k1
        wp, wn = norm._workspace(dev, st, C4)
        p3 = st3.data_ptr()
        hit = _BWD_CHAIN.pop(g_out.data_ptr(), None)
        if (hit is not None and FUSE_BWD_CHAIN and hit[2] == tuple(g_out.shape) and hit[3] == g_out.dtype and hit[4].data_ptr() == g_out.data_ptr()
                and hit[4]._version == hit[5]):
            # the consumer of this block's output formed the two sums together with this very gradient (rn_pw_dgrad_resid_sums)
            check(lib.rn_bn_bwd_finalize(hit[0].data_ptr(), hit[1], M1, C4, g3.data_ptr(), p3, p3 + 4 * C4, 1, gr3.data_ptr(), gr3.data_ptr() + 4 * C4,
                                         gr3.data_ptr() + 8 * C4, st), "rn_bn_bwd_finalize")
        else:
            check(lib.rn_bn_bwd_reduce(g_out.data_ptr(), bits.data_ptr(), z3.data_ptr(), _DT16[x.dtype], M1, C4, g3.data_ptr(), p3, p3 + 4 * C4, 0, 1, 2,
                                       gr3.data_ptr(), gr3.data_ptr() + 4 * C4, gr3.data_ptr() + 8 * C4, wp, wn, st), "rn_bn_bwd_reduce")
        pro3 = bn_bwd(gr3[2 * C4:], z3, relu_mode=3, bits=bits)
        # conv3 data gradient with bn3-backward in the operand load; epilogue: ReLU mask of a2 + the two bn2-backward sums
        nb2 = lib.rn_pw_walkers(M1)
        part2 = torch.empty((nb2 * 2 * Cm,), dtype=torch.float32, device=dev)
        p2 = st2.data_ptr()
        epi = RnPwEpilogue(RN_PW_EPI_RELU_BWD, part2.data_ptr(), 0, 0, z2.data_ptr(), p2 + 8 * Cm, p2 + 12 * Cm, p2, p2 + 4 * Cm)
        # the three data-gradient weights of the block in one launch
        Cin = w1.shape[1]
        mats = [(w3, C4, Cm), (w1, Cm, Cin)] + ([(wd, C4, Cin)] if wd is not None else [])
        wts = [torch.empty((c, r, 1, 1), dtype=m.dtype, device=dev) for m, r, c in mats]
        nm = len(mats)
        check(lib.rn_transpose_many((C.c_void_p * nm)(*[m.data_ptr() for m, _, _ in mats]), (C.c_void_p * nm)(*[t.data_ptr() for t in wts]),
                                    (C.c_int * nm)(*[r for _, r, _ in mats]), (C.c_int * nm)(*[c for _, _, c in mats]), nm, st),
              "rn_transpose_many")
        w3t, w1t = wts[0], wts[1]
        wdt = wts[2] if wd is not None else None
        pending: list = []                  # the block's 1x1 weight gradients: kernels now, one reduction of their splits at the end
        nbp = lib.rn_pw_conv3_backward_walkers(M1, Cm, C4) if (FUSE_CONV3_BWD and DEFER_WGRAD_REDUCE) else 0
        if nbp > 0:
            # both gradients of conv3 in ONE pass over (g_out, z3, bits): csrc/pw.hip, pw_conv3_bwd_kernel
            nb2 = nbp
            part2 = torch.empty((nb2 * 2 * Cm,), dtype=torch.float32, device=dev)
            dy2 = torch.empty_like(z2)
            dw3 = torch.empty_like(w3)
            ws3 = torch.empty((lib.rn_pw_conv3_backward_workspace_bytes(M1, Cm, C4),), dtype=torch.uint8, device=dev)
            S3 = C.c_int(0)
            pg = gr3.data_ptr() + 8 * C4
            PW_FLOP["pw_conv3_bwd"] = 4.0 * M1 * C4 * Cm
            with _timed("pw_conv3_bwd", dev):
                check(lib.rn_pw_conv3_backward(M1, Cm, C4, _DT16[x.dtype], g_out.data_ptr(), z3.data_ptr(), bits.data_ptr(), pg, pg + 4 * C4,
                                               pg + 8 * C4, w3t.data_ptr(), z2.data_ptr(), p2 + 8 * Cm, p2 + 12 * Cm, p2, p2 + 4 * Cm,
                                               dy2.data_ptr(), part2.data_ptr(), ws3.data_ptr(), ws3.numel(), C.byref(S3), st),
                      "rn_pw_conv3_backward")
            pending.append((ws3, int(S3.value), dw3.numel(), dw3))
        else:
            dy2 = pw_forward(g_out, w3t, pro=pro3, epi=epi, tag="pw_conv3_dgrad")
            dw3 = pw_wgrad(g_out, z2, w3, gpro=pro3, xpro=affine_relu(st2[2 * Cm:]), tag="pw_conv3_wgrad", defer=pending)
        # bn2: finalize from the epilogue sums, apply (conv2's backward is MIOpen's and wants dz2 in memory)
        gr2 = torch.empty((5 * Cm,), dtype=torch.float32, device=dev)
        check(lib.rn_bn_bwd_finalize(part2.data_ptr(), nb2, M1, Cm, g2.data_ptr(), p2, p2 + 4 * Cm, 1, gr2.data_ptr(), gr2.data_ptr() + 4 * Cm,
                                     gr2.data_ptr() + 8 * Cm, st), "rn_bn_bwd_finalize")
        dz2 = torch.empty_like(z2)
        check(lib.rn_bn_bwd_apply(dy2.data_ptr(), 0, z2.data_ptr(), dz2.data_ptr(), 0, _DT16[x.dtype], M1, Cm, gr2.data_ptr() + 8 * Cm, 0, 0, st),
              "rn_bn_bwd_apply")
        from . import biasact
        if biasact.dgrad_as_fwd_ok(w2, blk.conv2.stride, dz2) and tuple(blk.conv2.padding) == (1, 1):
            # stride 1: the data gradient as a forward convolution with the flipped weights (CK's forward kernel, no zero fill)
            da1 = biasact.conv3x3_dgrad_as_fwd(dz2, w2)
            dw2 = biasact.conv3x3_weight_gradient(dz2, a1, w2)
        elif STRIDED_WGRAD_PW and tuple(blk.conv2.stride) == (2, 2) and tuple(blk.conv2.padding) == (1, 1) and _cl(w2):
            da1 = torch.ops.aten.convolution_backward(dz2, a1, w2, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
            dw2 = pw_wgrad(dz2, a1, w2, stride=2, tag="pw_conv2_s2_wgrad")
        else:
            da1, dw2 = torch.ops.aten.convolution_backward(dz2, a1, w2, None, list(blk.conv2.stride), list(blk.conv2.padding), [1, 1], False,
                                                           [0, 0], 1, [True, True, False])[:2]
        if not _cl(da1):
            da1 = da1.contiguous(memory_format=torch.channels_last)
        # bn1 backward (ReLU mask recomputed from z1 and the forward coefficients)
        gr1 = torch.empty((5 * Cm,), dtype=torch.float32, device=dev)
        dz1 = torch.empty_like(z1)
        p1 = st1.data_ptr()
        wp, wn = norm._workspace(dev, st, Cm)
        check(lib.rn_bn_act_backward(da1.data_ptr(), 0, z1.data_ptr(), dz1.data_ptr(), 0, _DT16[x.dtype], M0, Cm, g1.data_ptr(), p1, p1 + 4 * Cm,
                                     p1 + 8 * Cm, 1, 1, gr1.data_ptr(), gr1.data_ptr() + 4 * Cm, gr1.data_ptr() + 8 * Cm, wp, wn, st),
              "rn_bn_act_backward")
        dwd = dgd = dbd = None
        if wd is None:
            # the identity branch's gradient g_out * bits joins in the data-gradient GEMM's epilogue
            dx = _conv1_dgrad(dz1, w1t, x, g_out, bits, 1, pz3, pbits, pst3)
        else:
            dn = blk.downsample
            grd = torch.empty((5 * C4,), dtype=torch.float32, device=dev)
            dzd = torch.empty_like(zd)
            pd = std.data_ptr()
            wp, wn = norm._workspace(dev, st, C4)
            check(lib.rn_bn_act_backward(g_out.data_ptr(), bits.data_ptr(), zd.data_ptr(), dzd.data_ptr(), 0, _DT16[x.dtype], M1, C4, gd.data_ptr(), pd,
                                         pd + 4 * C4, pd + 8 * C4, 1, 2, grd.data_ptr(), grd.data_ptr() + 4 * C4, grd.data_ptr() + 8 * C4, wp,
                                         wn, st), "rn_bn_act_backward")
            dgd, dbd = grd[:C4], grd[C4:2 * C4]
            dwd = pw_wgrad(dzd, x, wd, stride=dn[0].stride[0], tag="pw_down_wgrad", defer=pending)
            # the downsample branch's data gradient is a GEMM on ITS grid (the stride-2 grid of x for layer2 .. layer4's first
            # blocks) and joins conv1's data gradient in that GEMM's epilogue -- no scatter into a zero-filled tensor (MIOpen's
            # strided data gradient) and no add pass over the block's largest tensor (0.14 + 0.03 ms per step at the bench shape)
            sd = dn[0].stride[0]
            dxd = pw_forward(dzd, wdt, tag="pw_down_dgrad")
            if sd in (1, 2):
                dx = _conv1_dgrad(dz1, w1t, x, dxd, None, sd, pz3, pbits, pst3)
            else:
                full = torch.empty_like(x).fill_(0)
                full[:, :, ::sd, ::sd] = dxd
                dx = pw_forward(dz1, w1t, tag="pw_conv1_dgrad") + full
        dw1 = pw_wgrad(dz1, x, w1, tag="pw_conv1_wgrad", defer=pending)
        pw_wgrad_flush(pending)
        return (None, dx, dw1, gr1[:Cm], gr1[Cm:2 * Cm], dw2, gr2[:Cm], gr2[Cm:2 * Cm], dw3, gr3[:C4], gr3[C4:2 * C4], dwd, dgd, dbd)


def _conv1_dgrad(dz1: Tensor, w1t: Tensor, x: Tensor, resid: Tensor, rbits: Optional[Tensor], rs: int, pz3, pbits, pst3) -> Tensor:
    """conv1's data gradient joined by the identity / downsample branch's (``RN_PW_EPI_RESID``).  With the producer of ``x`` known
    (``pz3``, ``pbits``, ``pst3``: its z3, ReLU bits, bn3 statistics) the same pass also takes that block's bn3-backward sums over the
    gradient it has just formed and leaves them in ``_BWD_CHAIN`` for its backward."""
    dev = x.device
    Cm, Cin = int(dz1.shape[1]), int(x.shape[1])
    M0 = x.shape[0] * x.shape[2] * x.shape[3]
    nbs = lib.rn_pw_dgrad_resid_sums_walkers(M0, Cm, Cin) if (FUSE_BWD_CHAIN and pz3 is not None and tuple(pz3.shape) == tuple(x.shape)) else 0
    if nbs > 0:
        dx = torch.empty_like(x)
        parts = torch.empty((nbs * 2 * Cin,), dtype=torch.float32, device=dev)
        pm = pst3.data_ptr()
        PW_FLOP["pw_conv1_dgrad_sums"] = 2.0 * M0 * Cm * Cin
        with _timed("pw_conv1_dgrad_sums", dev):
            check(lib.rn_pw_dgrad_resid_sums(M0, Cm, Cin, _DT16[x.dtype], dz1.data_ptr(), w1t.data_ptr(), resid.data_ptr(),
                                             rbits.data_ptr() if rbits is not None else 0, rs, x.shape[2], x.shape[3], pz3.data_ptr(),
                                             pbits.data_ptr(), pm, pm + 4 * Cin, dx.data_ptr(), parts.data_ptr(), _stream(dev)),
                  "rn_pw_dgrad_resid_sums")
        # the entry keeps dx itself: a gradient somebody else still references is never accumulated into in place by the autograd engine
        # (torch/csrc/autograd/input_buffer.cpp: only uniquely owned buffers are), so a second consumer of the producer's output shows up
        # as a NEW tensor at its backward -- another data_ptr, no hit -- and not as our buffer with a different content
        _BWD_CHAIN[dx.data_ptr()] = (parts, nbs, tuple(dx.shape), dx.dtype, dx, dx._version)
        return dx
    if rs == 1:
        epi1 = RnPwEpilogue(RN_PW_EPI_RESID, 0, resid.data_ptr(), rbits.data_ptr() if rbits is not None else 0, 0, 0, 0, 0, 0)
    else:
        epi1 = RnPwEpilogue(RN_PW_EPI_RESID, 0, resid.data_ptr(), 0, 0, 0, 0, 0, 0, rs, x.shape[2], x.shape[3])
    return pw_forward(dz1, w1t, epi=epi1, tag="pw_conv1_dgrad")


def bottleneck_fusable(blk, x: Tensor, in_forward: bool = False) -> bool:
    "``in_forward``: asked from inside an autograd Function's forward (grad mode is off there) about the block that will run next"
    if not (FUSED_BOTTLENECK and x.is_cuda and x.dtype in H16 and _cl(x) and (in_forward or torch.is_grad_enabled())):
        return False
    bns = [blk.bn1, blk.bn2, blk.bn3] + ([blk.downsample[1]] if blk.downsample is not None else [])
    convs = [blk.conv1, blk.conv2, blk.conv3] + ([blk.downsample[0]] if blk.downsample is not None else [])
    for bn in bns:
        if not (bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None and bn.weight.dtype == torch.float32):
            return False
    for cv in convs:
        if not (cv.weight.dtype in H16 and cv.bias is None and cv.groups == 1 and cv.in_channels % 64 == 0
                and cv.out_channels % 64 == 0 and cv.dilation == (1, 1)):
            return False
    if blk.conv2.out_channels > FUSED_MAX_MID:
        return False
    if blk.conv1.stride != (1, 1) or blk.conv3.stride != (1, 1) or blk.conv2.stride[0] != blk.conv2.stride[1] or blk.conv2.stride[0] not in (1, 2):
        return False
    if not _cl(blk.conv2.weight):
        return False
    if x.shape[0] * x.shape[2] * x.shape[3] >= (1 << 31) // max(blk.conv3.out_channels, 1) * 8:
        return False
    return True


def bottleneck(blk, x: Tensor) -> Tensor:
    dn = blk.downsample
    # FUSE_CHAIN hand-over: the block before left its successor's conv1 output and statistics partials ON the tensor it returned; the
    # consumer checks that this is that tensor (storage + version) and that conv1's weight is the one the products were formed with
    hit = x.__dict__.pop("_rn_chain", None)
    if hit is not None and FUSE_CHAIN:
        blk.__dict__["_chain_in"] = (hit, x.data_ptr(), x._version)
    try:
        out = _BottleneckFn.apply(blk, x, blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias,
                                  blk.conv3.weight, blk.bn3.weight, blk.bn3.bias,
                                  dn[0].weight if dn is not None else None, dn[1].weight if dn is not None else None,
                                  dn[1].bias if dn is not None else None)
    finally:
        blk.__dict__.pop("_chain_in", None)
        tmp = blk.__dict__.pop("_chain_tmp", None)
    if tmp is not None:
        out._rn_chain = tmp
    return out


def link_blocks(blocks) -> None:
    "Tell every bottleneck which block consumes its output (plain ``__dict__`` entries: not sub-modules, not in the state dict)."
    for a, b in zip(blocks[:-1], blocks[1:]):
        a.__dict__["_rn_next"] = [b]


# ---- 1x1 convolutions outside the fused blocks (layer3 / layer4, downsample branches, FPN laterals): the fastest of three ----------
# For a 1x1 / stride-1 convolution on channels-last activations all three products are plain GEMMs on the [M, C] views.  Measured
# on MI355X at the R50 shapes (tools: scratch probe of round 3, isolated, bf16): hipBLASLt beats MIOpen's convolution kernels for
# the forward product when the contraction is >= 1024 channels (l3.conv1 24 vs 35 us, l4.conv1 24 vs 41) and for the data
# gradient whenever the convolution is wide (l3.conv3 25 vs 50, l4.conv3 26 vs 61, l3.conv1 44 vs 59) -- MIOpen launches 132
# tiles of 256 x 256 on 256 CUs there --; for the weight gradient hipBLASLt is 2 - 5 x slower than MIOpen (k-strided operands),
# and the position-contraction kernel of csrc/pw.hip is the fastest (37 - 43 us against 49 - 52 + the zero / cast helpers).
MM_1X1 = os.environ.get("RN_MM_1X1", "1") != "0"


MANY_ROWS_MM = True       # 512-channel contraction on >= 100 000 rows without bias (conv1 of layer3's first block: CK 87 us, hipBLASLt 63 - 72)
BIAS_1X1_MM = True        # a 1x1 conv WITH bias and >= 512 input channels goes to hipBLASLt (bias in the GEMM's epilogue) at any size


def _fwd_by_mm(M: int, cin: int, cout: int, has_bias: bool = False) -> bool:
    # (the FPN's C3 lateral, 512 -> 256 on 134 400 positions with bias: MIOpen's kernel 66 us + a strided elementwise bias add over the
    # output 30 us in the step; isolated: conv2d + bias 136 us, hipBLASLt addmm 63, pw_gemm_kernel with a bias epilogue 84)
    return (cin >= 1024 and M <= 40000) or (BIAS_1X1_MM and has_bias and cin >= 512) or (MANY_ROWS_MM and cin >= 512 and M >= 100000)


def _dgrad_by_mm(M: int, cin: int, cout: int) -> bool:
    return cout >= 1024 or (cin >= 512 and not (M < 10000 and cout <= 256))


# ---- joining the gradients of ONE activation that feeds several 1x1 convolutions ------------------------------------------------
# C3 / C4 feed the next layer's conv1, its stride-2 downsample conv and an FPN lateral: autograd adds the three data gradients with two
# elementwise kernels over the 137 / 69 MB tensor (and MIOpen first scatters the downsample's gradient into a zero-filled one).  All
# three are GEMMs.  The first 1x1 / stride-1 consumer of a tensor (in forward order: conv1 of the next block) becomes the RECEIVER of a
# `_GradJoin` hung on the tensor; later consumers are DONORS: their backward runs first (autograd executes in reverse creation order),
# leaves its result in the join and returns no gradient; the receiver's data-gradient GEMM then accumulates INTO the donated tensor
# (`addmm_`) and adds the compact stride-2 gradients at their pixels.  A donor donates only when the engine reports that the receiver's
# node WILL execute in this backward pass and has not run yet (`torch._C._will_engine_execute_node`); otherwise it returns its gradient the
# ordinary way, so the result is the same sum whatever the execution order and whatever part of the graph the loss uses.
JOIN_GRADS = True
JOIN_STATS = {"full": 0, "compact": 0}     # donated gradients since import (tests)


class _GradJoin:
    __slots__ = ("recv_done", "full", "compact", "_has_receiver", "recv_node")

    def __init__(self):
        self.recv_done, self.full, self.compact, self._has_receiver, self.recv_node = False, None, [], False, None


def share_gradients(x: Tensor) -> Tensor:
    """Opt-in for ``x`` (called where the consumers are known: ResNetBackbone for C3 / C4): its 1x1 consumers join their data
    gradients."""
    if JOIN_GRADS and x.is_cuda and x.requires_grad and torch.is_grad_enabled() and getattr(x, "_rn_join", None) is None:
        x._rn_join = _GradJoin()
    return x


def _receiver_will_run(j: "_GradJoin") -> bool:
    """Inside a backward pass: is the receiver's autograd node part of THIS pass?  (A donor must not leave its gradient for a
    receiver that never executes -- e.g. a loss that uses the lateral's output but nothing behind the next layer.)"""
    node = j.recv_node
    if node is None or j.recv_done:
        return False
    try:
        return bool(torch._C._will_engine_execute_node(node))
    except Exception:                    # noqa: BLE001 -- not inside an engine run / API missing: take the ordinary path
        return False


def _join_of(x: Tensor):
    "-> (join, receiver?): the first 1x1 / stride-1 consumer of a tensor opted in by ``share_gradients`` is the receiver"
    j = getattr(x, "_rn_join", None)
    if j is None:
        return None, False
    if not j._has_receiver:
        j._has_receiver = True
        return j, True
    return j, False


class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, join=None, receiver=False):
        ctx.join, ctx.receiver = join, receiver
        Nimg, Cin, H, W = x.shape
        Cout = w.shape[0]
        M = Nimg * H * W
        if _fwd_by_mm(M, Cin, Cout, bias is not None):
            x2, w2 = x.permute(0, 2, 3, 1).reshape(M, Cin), w.reshape(Cout, Cin)
            y2 = torch.addmm(bias.to(x.dtype), x2, w2.t()) if bias is not None else x2 @ w2.t()
            y = y2.view(Nimg, H, W, Cout).permute(0, 3, 1, 2)
        else:
            y = F.conv2d(x, w, bias.to(x.dtype) if bias is not None else None)
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        Nimg, Cin, H, W = x.shape
        Cout = w.shape[0]
        M = Nimg * H * W
        if g.dtype != x.dtype or not _cl(g):
            g = g.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dx = dw = db = None
        g2 = g.permute(0, 2, 3, 1).reshape(M, Cout)
        join = ctx.join
        if ctx.needs_input_grad[0]:
            acc = None
            if join is not None and ctx.receiver:
                acc, join.full = join.full, None
                if acc is not None and not (acc.dtype == x.dtype and acc.shape == x.shape and _cl(acc)):
                    acc = acc.to(x.dtype).contiguous(memory_format=torch.channels_last)
            if _dgrad_by_mm(M, Cin, Cout):
                if acc is not None:
                    dx = acc.permute(0, 2, 3, 1).reshape(M, Cin).addmm_(g2, w.reshape(Cout, Cin)).view(Nimg, H, W, Cin).permute(0, 3, 1, 2)
                else:
                    dx = (g2 @ w.reshape(Cout, Cin)).view(Nimg, H, W, Cin).permute(0, 3, 1, 2)
            else:
                dx = torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
                if acc is not None:
                    dx = acc.add_(dx)
            if join is not None and ctx.receiver:
                for comp, sd in join.compact:                    # stride-2 consumers' gradients, on their own grid
                    dx[:, :, ::sd, ::sd] += comp
                join.compact, join.recv_done = [], True
            elif join is not None and join.full is None and _receiver_will_run(join):
                join.full, dx = dx, None                         # donor: the receiver's GEMM will accumulate into this tensor
                JOIN_STATS["full"] += 1
        if ctx.needs_input_grad[1]:
            dw = pw_wgrad(g, x, w, tag="pw_1x1_wgrad")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            from .biasact import _colsum_levels                 # column sums in two launches (torch: fill + reduce, slower per byte)
            db = _colsum_levels([g2.reshape(1, -1)], Cout) if g2.is_contiguous() else g2.sum(0, dtype=torch.float32)
        return dx, dw, db, None, None


STRIDED_WGRAD_PW = True   # weight gradient of the 3x3 / stride-2 conv2 of layer2's .. layer4's first blocks on pw_wgrad_kernel (taps = 9; False: MIOpen)
DOWN_WGRAD_PW = True      # weight gradient of the unfused blocks' 1x1 / stride-2 downsample convs on pw_wgrad_kernel (False: MIOpen + its zero fill + cast)


class _Conv1x1S2(torch.autograd.Function):
    """A 1x1 / stride-2 convolution without bias (the downsample branch of layer2 .. layer4's first blocks): MIOpen forward; the data
    gradient is a GEMM on the OUTPUT grid (hipBLASLt) that stays compact when a `_GradJoin` receiver will add it at its pixels; the
    weight gradient is csrc/pw.hip's (``DOWN_WGRAD_PW``)."""

    @staticmethod
    def forward(ctx, x, w, join):
        ctx.join = join
        ctx.save_for_backward(x, w)
        return F.conv2d(x, w, None, 2)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        Nimg, Cin, H, W = x.shape
        Cout = w.shape[0]
        if g.dtype != x.dtype or not _cl(g):
            g = g.to(x.dtype).contiguous(memory_format=torch.channels_last)
        Ho, Wo = g.shape[2], g.shape[3]
        dx = dw = None
        if ctx.needs_input_grad[0]:
            comp = (g.permute(0, 2, 3, 1).reshape(-1, Cout) @ w.reshape(Cout, Cin)).view(Nimg, Ho, Wo, Cin).permute(0, 3, 1, 2)
            join = ctx.join
            if join is not None and _receiver_will_run(join):
                join.compact.append((comp, 2))
                JOIN_STATS["compact"] += 1
            else:
                dx = torch.empty_like(x).fill_(0)
                dx[:, :, ::2, ::2] = comp
        if ctx.needs_input_grad[1]:
            if DOWN_WGRAD_PW and x.shape[1] % 64 == 0 and Cout % 64 == 0 and _cl(x):
                dw = pw_wgrad(g, x, w, stride=2, tag="pw_down_wgrad")       # csrc/pw.hip: rows of x picked at stride 2 in the operand load
            else:
                dw = torch.ops.aten.convolution_backward(g, x, w, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return dx, dw, None


class _Conv3x3S2(torch.autograd.Function):
    """A 3x3 / stride-2 / pad-1 convolution without bias (conv2 of layer3's / layer4's first blocks, retinanet/backbone.py:112,128):
    MIOpen forward and data gradient; the weight gradient on csrc/pw.hip's position-contraction kernel (taps = 9, the rows of x picked at
    stride 2 in the operand load) instead of MIOpen's kernel + zero fill + cast."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return F.conv2d(x, w, None, 2, 1)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        if g.dtype != x.dtype or not _cl(g):
            g = g.to(x.dtype).contiguous(memory_format=torch.channels_last)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.ops.aten.convolution_backward(g, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            dw = pw_wgrad(g, x, w if _cl(w) else w.contiguous(memory_format=torch.channels_last), stride=2, tag="pw_conv2_s2_wgrad")
        return dx, dw


def conv3x3_s2_ok(conv, x: Tensor, bias_ok: bool = False) -> bool:
    "``_Conv3x3S2`` applies (``bias_ok``: the caller adds the bias itself)."
    w = conv.weight
    return (STRIDED_WGRAD_PW and x.is_cuda and x.dtype in H16 and w.dtype == x.dtype and _cl(x) and conv.kernel_size == (3, 3) and conv.stride == (2, 2)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and (bias_ok or conv.bias is None) and conv.in_channels % 64 == 0
            and conv.out_channels % 64 == 0 and torch.is_grad_enabled() and x.shape[0] * x.shape[2] * x.shape[3] < (1 << 24))


def conv1x1(conv, x: Tensor) -> Tensor:
    """``conv(x)`` for a 1x1 / stride-1 ``nn.Conv2d`` on bf16 channels-last activations with every product on the fastest of
    MIOpen / hipBLASLt / csrc/pw.hip (``_Conv1x1``); anything else is ``conv(x)``."""
    w = conv.weight
    if (MM_1X1 and x.is_cuda and x.dtype in H16 and w.dtype == x.dtype and _cl(x) and conv.kernel_size == (1, 1)
            and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 and conv.in_channels % 64 == 0
            and conv.out_channels % 64 == 0 and torch.is_grad_enabled() and x.shape[0] * x.shape[2] * x.shape[3] < (1 << 24)):
        join, receiver = _join_of(x)
        y = _Conv1x1.apply(x, w, conv.bias, join, receiver)
        if receiver:
            join.recv_node = y.grad_fn
        return y
    if (MM_1X1 and JOIN_GRADS and x.is_cuda and x.dtype in H16 and w.dtype == x.dtype and _cl(x) and conv.kernel_size == (1, 1)
            and conv.stride == (2, 2) and conv.padding == (0, 0) and conv.groups == 1 and conv.bias is None and conv.in_channels % 64 == 0
            and conv.out_channels % 64 == 0 and torch.is_grad_enabled() and x.requires_grad and getattr(x, "_rn_join", None) is not None
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and x.shape[0] * x.shape[2] * x.shape[3] < (1 << 24)):
        return _Conv1x1S2.apply(x, w, x._rn_join)          # a stride-2 consumer of a tensor that already has a receiver
    if conv3x3_s2_ok(conv, x):
        return _Conv3x3S2.apply(x, w)
    from . import biasact
    if biasact.conv3x3_bwd_fusable(conv, x):        # 3x3 / 256 -> 256 (layer3's conv2): MIOpen forward, gradients on csrc/conv.hip
        return biasact.conv3x3_mfma_bwd(conv, x)
    if biasact.conv3x3_dgrad_fwd_fusable(conv, x):  # other 3x3 / stride-1 convs (layer4's conv2): data gradient as a forward convolution
        return biasact.conv3x3_dgrad_fwd(conv, x)
    return conv(x)


# ---- the stem: conv 7x7 / stride 2 (3 -> 64) + BatchNorm (batch statistics) + ReLU on csrc/stem.hip ---------------------------------
FUSED_STEM = True
STEM_POOL_BN_SUMS = True   # ... and bn1's two backward sums inside the max pooling's backward (rn_maxpool3x3s2_backward_bn)
STEM_WGRAD_BN = True   # ... with bn1's backward apply step in its operand load (rn_stem_conv_wgrad_bn)
STEM_WGRAD = True      # weight gradient on csrc/stem.hip as well (False: MIOpen)
_STEM_WS: Dict[tuple, Tensor] = {}


def stem_fusable(conv, bn, x: Tensor) -> bool:
    return (FUSED_STEM and x.is_cuda and x.dtype in H16 and _cl(x) and torch.is_grad_enabled() and x.shape[1] == 3 and
            conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.dilation == (1, 1) and
            conv.groups == 1 and conv.bias is None and conv.out_channels == 64 and conv.weight.dtype in H16 and
            bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None and bn.weight.dtype == torch.float32 and
            x.shape[0] * (x.shape[2] + 6) * (x.shape[3] + 8) < (1 << 31))


class _StemFn(torch.autograd.Function):
    """``relu(bn(conv7x7s2(x)))`` in training mode on bf16 channels-last tensors: the MFMA stem kernel with the BatchNorm statistics
    in its epilogue (``rn_stem_conv_forward``), the finalize step, one apply pass; backward: the BatchNorm backward pair, the weight
    gradient on the transposing-read MFMA kernel (``rn_stem_conv_wgrad``); the image gets no gradient."""

    @staticmethod
    def forward(ctx, bn, pool, x, w, gamma, beta):
        B, _, H, W = x.shape
        dev = x.device
        st = _stream(dev)
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        # xp: the zero-bordered NHWC4 copy of the image; the weight gradient reads it again, so it belongs to this call
        ws = (torch.empty((lib.rn_stem_padded_bytes(B, H, W),), dtype=torch.uint8, device=dev),
              torch.empty((64 * 7 * 32,), dtype=x.dtype, device=dev))
        wc = w if _cl(w) else w.contiguous(memory_format=torch.channels_last)
        z = torch.empty((B, 64, Ho, Wo), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
        nb = lib.rn_stem_partial_rows(B, H, W)
        partial = torch.empty((nb * 2 * 64,), dtype=torch.float32, device=dev)
        PW_FLOP["stem_fwd"] = 2.0 * B * Ho * Wo * 64 * 147
        with _timed("stem_fwd", dev):
            check(lib.rn_stem_conv_forward(x.data_ptr(), wc.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), z.data_ptr(), partial.data_ptr(),
                                           _DT16[x.dtype], B, H, W, st), "rn_stem_conv_forward")
        stats = bn_finalize(partial, nb, B * Ho * Wo, bn)
        if pool:
            # BatchNorm apply + ReLU inside the 3x3 / stride-2 max pooling: the 275 MB activation between them is never written
            Hq, Wq = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
            out = torch.empty((B, 64, Hq, Wq), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
            arg = torch.empty(out.shape, dtype=torch.uint8, device=dev, memory_format=torch.channels_last)
            check(lib.rn_bn_relu_maxpool3x3s2_forward(z.data_ptr(), stats.data_ptr() + 8 * 64, out.data_ptr(), arg.data_ptr(), _DT16[x.dtype], B, Ho, Wo,
                                                      64, st), "rn_bn_relu_maxpool3x3s2_forward")
        else:
            out, _ = bn_apply(z, stats, relu=True)
            arg = None
        ctx.pool = pool
        ctx.save_for_backward(x, wc, gamma, z, stats, ws[0], arg)
        return out

    @staticmethod
    def backward(ctx, da):
        x, w, gamma, z, stats, xp, arg = ctx.saved_tensors
        dev = z.device
        st = _stream(dev)
        Cc = 64
        M = z.shape[0] * z.shape[2] * z.shape[3]
        if not (da.dtype == z.dtype and _cl(da)):
            da = da.to(z.dtype).contiguous(memory_format=torch.channels_last)
        gr = torch.empty((5 * Cc,), dtype=torch.float32, device=dev)
        sp = stats.data_ptr()
        fused_apply = STEM_WGRAD and STEM_WGRAD_BN
        pool_rows = lib.rn_maxpool3x3s2_backward_bn_rows(z.shape[0], z.shape[2], z.shape[3], Cc) if (ctx.pool and fused_apply and STEM_POOL_BN_SUMS) else 0
        if ctx.pool:                                                 # `da` is the pooled gradient: back through the arg-max codes first
            dpool, da = da, torch.empty_like(z)
            if pool_rows > 0:
                # ... and bn1's two backward sums over the gradient just formed in the same pass (that pass then does not re-read it)
                part = torch.empty((pool_rows * 2 * Cc,), dtype=torch.float32, device=dev)
                check(lib.rn_maxpool3x3s2_backward_bn(arg.data_ptr(), dpool.data_ptr(), z.data_ptr(), sp + 8 * Cc, sp, sp + 4 * Cc, da.data_ptr(),
                                                      part.data_ptr(), _DT16[z.dtype], z.shape[0], z.shape[2], z.shape[3], Cc, st),
                      "rn_maxpool3x3s2_backward_bn")
            else:
                check(lib.rn_maxpool3x3s2_backward(arg.data_ptr(), dpool.data_ptr(), da.data_ptr(), _DT16[z.dtype], z.shape[0], z.shape[2], z.shape[3], Cc,
                                                   st), "rn_maxpool3x3s2_backward")
        wp, wn = norm._workspace(dev, st, Cc)
        B, _, H, W = x.shape
        if pool_rows > 0:
            check(lib.rn_bn_bwd_finalize(part.data_ptr(), pool_rows, M, Cc, gamma.data_ptr(), sp, sp + 4 * Cc, 1, gr.data_ptr(), gr.data_ptr() + 4 * Cc,
                                         gr.data_ptr() + 8 * Cc, st), "rn_bn_bwd_finalize")
        elif fused_apply:
            # the two sums and the coefficients only: the apply step rides in the weight gradient's operand load (the image needs no gradient,
            # so nothing else reads the conv-output gradient -- 275 MB that are neither written nor read back)
            check(lib.rn_bn_bwd_reduce(da.data_ptr(), 0, z.data_ptr(), _DT16[z.dtype], M, Cc, gamma.data_ptr(), sp, sp + 4 * Cc, sp + 8 * Cc, 1, 1,
                                       gr.data_ptr(), gr.data_ptr() + 4 * Cc, gr.data_ptr() + 8 * Cc, wp, wn, st), "rn_bn_bwd_reduce")
        else:
            dz = torch.empty_like(z)
            check(lib.rn_bn_act_backward(da.data_ptr(), 0, z.data_ptr(), dz.data_ptr(), 0, _DT16[z.dtype], M, Cc, gamma.data_ptr(), sp, sp + 4 * Cc,
                                         sp + 8 * Cc, 1, 1, gr.data_ptr(), gr.data_ptr() + 4 * Cc, gr.data_ptr() + 8 * Cc, wp, wn, st),
                  "rn_bn_act_backward")
        if STEM_WGRAD:
            need = lib.rn_stem_wgrad_workspace_bytes(B, H, W)
            key = (dev.index, st)
            wsb = _STEM_WS.get(key)
            if wsb is None or wsb.numel() < need:
                wsb = _STEM_WS[key] = torch.empty((need,), dtype=torch.uint8, device=dev)
            dw = torch.empty_like(w)
            PW_FLOP["stem_wgrad"] = 2.0 * M * 64 * 147
            with _timed("stem_wgrad", dev):
                if fused_apply:
                    check(lib.rn_stem_conv_wgrad_bn(da.data_ptr(), z.data_ptr(), gr.data_ptr() + 8 * Cc, sp + 8 * Cc, xp.data_ptr(), dw.data_ptr(),
                                                    _DT16[z.dtype], B, H, W, wsb.data_ptr(), wsb.numel(), st), "rn_stem_conv_wgrad_bn")
                else:
                    check(lib.rn_stem_conv_wgrad(dz.data_ptr(), xp.data_ptr(), dw.data_ptr(), _DT16[z.dtype], B, H, W, wsb.data_ptr(), wsb.numel(), st),
                          "rn_stem_conv_wgrad")
        else:
            dw = torch.ops.aten.convolution_backward(dz, x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return None, None, None, dw, gr[:Cc], gr[Cc:2 * Cc]


# Inference: the same stem kernel with the BatchNorm folded into its weights; the folded bias and the ReLU are applied inside the max
# pooling (coefficients a = 1, b = bias), so the 925 MB activation of a 16 x 1344^2 batch between them is written once and read once
# (MIOpen's implicit GEMM + an epilogue pass + the pooling pass before).
STEM_EVAL = True


def stem_eval_ok(conv, pool, x: Tensor, w: Tensor) -> bool:
    return (STEM_EVAL and x.is_cuda and x.dtype in H16 and w.dtype == x.dtype and x.dim() == 4 and _cl(x) and x.shape[1] == 3 and
            conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1 and
            conv.out_channels == 64 and pool is not None and pool.kernel_size == 3 and pool.stride == 2 and pool.padding == 1 and
            pool.dilation == 1 and not pool.ceil_mode and not pool.return_indices and
            x.shape[0] * (x.shape[2] + 6) * (x.shape[3] + 8) < (1 << 31))


def stem_eval(x: Tensor, w: Tensor, bias: Tensor) -> Tensor:
    "``maxpool3x3s2(relu(conv7x7s2(x, w) + bias))`` on csrc/stem.hip + the pooling kernel (w: the folded bf16 weight, bias f32 [64])."
    B, _, H, W = x.shape
    dev = x.device
    st = _stream(dev)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    xp = torch.empty((lib.rn_stem_padded_bytes(B, H, W),), dtype=torch.uint8, device=dev)
    wk = torch.empty((64 * 7 * 32,), dtype=x.dtype, device=dev)
    wc = w if _cl(w) else w.contiguous(memory_format=torch.channels_last)
    z = torch.empty((B, 64, Ho, Wo), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
    partial = torch.empty((lib.rn_stem_partial_rows(B, H, W) * 2 * 64,), dtype=torch.float32, device=dev)      # (the kernel's statistics: unused here)
    PW_FLOP["stem_fwd"] = 2.0 * B * Ho * Wo * 64 * 147
    with _timed("stem_fwd", dev):
        check(lib.rn_stem_conv_forward(x.data_ptr(), wc.data_ptr(), xp.data_ptr(), wk.data_ptr(), z.data_ptr(), partial.data_ptr(), _DT16[x.dtype], B, H, W, st),
              "rn_stem_conv_forward")
    coef = torch.cat([torch.ones(64, dtype=torch.float32, device=dev), bias.float()])
    Hq, Wq = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    out = torch.empty((B, 64, Hq, Wq), dtype=x.dtype, device=dev, memory_format=torch.channels_last)
    arg = torch.empty(out.shape, dtype=torch.uint8, device=dev, memory_format=torch.channels_last)
    check(lib.rn_bn_relu_maxpool3x3s2_forward(z.data_ptr(), coef.data_ptr(), out.data_ptr(), arg.data_ptr(), _DT16[x.dtype], B, Ho, Wo, 64, st),
          "rn_bn_relu_maxpool3x3s2_forward")
    return out


def stem(conv, bn, x: Tensor, pool=None) -> Tensor:
    "``relu(bn(conv(x)))``, or ``pool(relu(bn(conv(x))))`` when ``pool`` is the stem's 3x3 / stride-2 / pad-1 max pooling module."
    fuse = (pool is not None and pool.kernel_size == 3 and pool.stride == 2 and pool.padding == 1 and pool.dilation == 1
            and not pool.ceil_mode and not pool.return_indices)
    y = _StemFn.apply(bn, fuse, x, conv.weight, bn.weight, bn.bias)
    return y if fuse or pool is None else pool(y)
