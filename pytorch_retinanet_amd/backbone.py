"""ResNet feature extractors C3/C4/C5 (reference ``retinanet/backbone.py:20-381``).

Module/parameter names follow torchvision's ResNet (``conv1, bn1, layer{1..4}.{i}.
conv{1,2,3}/bn{1,2,3}/downsample.{0,1}``) under ``backbone.backbone.*`` so ImageNet
and reference checkpoints load unchanged.  Stride placement is ResNet v1.5 (stride
on the 3x3 of a bottleneck).  The convolutions are PyTorch-ROCm / MIOpen; run the
model in ``channels_last`` + bf16 autocast on MI355X (see ``models.Retinanet``).
"""
import os
from typing import Dict, List, Optional, Tuple, Type, Union

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import pwconv
from .norm import RAW_WRITES, FusedBatchNorm2d, _BNAct
from .pwconv import H16
from .pool import FusedMaxPool2d

__all__ = ["resnet18", "resnet34", "resnet50", "resnet101", "resnet152"]

model_urls = {
    "resnet18": "https://download.pytorch.org/models/resnet18-5c106cde.pth",
    "resnet34": "https://download.pytorch.org/models/resnet34-333f7ec4.pth",
    "resnet50": "https://download.pytorch.org/models/resnet50-19c8e357.pth",
    "resnet101": "https://download.pytorch.org/models/resnet101-5d3b4d8f.pth",
    "resnet152": "https://download.pytorch.org/models/resnet152-b121ed2d.pth",
}


# ---- frozen-BN folding for inference (reference backbone.py:348-351 freezes BN by eval()) --------------
# bn(conv(x)) with BN on its running statistics is one convolution with rescaled weights plus a per-channel bias:
#   w' = w * gamma / sqrt(var + eps),  b' = beta - mean * gamma / sqrt(var + eps).
# Under no_grad with the BN layer in eval mode the blocks below run conv(x, w') and hand b' (+ residual, + ReLU) to the
# same fused one-pass epilogue kernel the BN layers use (as an identity normalisation), so an inference step neither
# reads the BN statistics nor re-casts fp32 weights under autocast.  The folded tensors are cached on the BN module per
# (conv, dtype) and rebuilt when any of the five source tensors changes -- torch's version counters (load_state_dict, in-place
# torch ops) or the library's raw-pointer writes (norm.RAW_WRITES: training steps).
FOLD_FROZEN_BN = True


def _folded(conv: nn.Conv2d, bn: nn.BatchNorm2d, dtype: torch.dtype) -> Tuple[Tensor, ...]:
    """Folded (weight, bias, ones, zeros, var1) of ``bn(conv(.))``, cached ON the BN module (a plain attribute: not in the
    state dict, gone with the module) per (conv, dtype).  A hit needs the same source storages, the same torch version
    counters AND the same ``norm.RAW_WRITES`` epoch: the library's own training kernels (fused BN forward, ``MasterSGD.step``)
    write these tensors through raw pointers, which ``_version`` never sees."""
    src = (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    stamp = (RAW_WRITES[0], id(conv)) + tuple((t.data_ptr(), t._version) for t in src)
    cache = bn.__dict__.setdefault("_rn_fold", {})
    key = (dtype, conv.weight.device)
    hit = cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    with torch.no_grad():
        w32 = getattr(conv.weight, "master", conv.weight).float()          # fp32 master of a bf16 working copy (optim.py)
        scale = bn.weight.float() * torch.rsqrt(bn.running_var.float() + bn.eps)
        w = (w32 * scale[:, None, None, None]).to(dtype)
        if conv.weight.is_contiguous(memory_format=torch.channels_last):
            w = w.contiguous(memory_format=torch.channels_last)
        b = (bn.bias.float() - bn.running_mean.float() * scale).contiguous()
        C = b.numel()
        ones, zeros = torch.ones(C, device=b.device), torch.zeros(C, device=b.device)
        var1 = torch.full((C,), 1.0 - bn.eps, device=b.device)             # identity normalisation: (x - 0) / sqrt(var1 + eps) * 1 + b'
    out = (w, b, ones, zeros, var1)
    cache[key] = (stamp, out)
    return out


def conv_bn(conv: nn.Conv2d, bn: FusedBatchNorm2d, x: Tensor, relu: bool = False, residual: Optional[Tensor] = None) -> Tensor:
    "``[relu](bn(conv(x)) [+ residual])``; folded into the conv weights when BN is frozen and no gradient is recorded."
    if FOLD_FROZEN_BN and not bn.training and not torch.is_grad_enabled() and x.is_cuda and bn.track_running_stats:
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        w, b, ones, zeros, var1 = _folded(conv, bn, dt)
        from . import biasact
        xd = x.to(dt)
        if pwconv.eval_conv1x1_ok(conv, xd, w, residual):
            # 1x1: the folded bias, the identity branch and the ReLU ride in the GEMM's epilogue -- no pass over the output at all
            return pwconv.eval_conv1x1(conv, xd, w, b, relu, residual)
        plain3x3 = conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
        if plain3x3 and residual is None and biasact.dense_eval_ok(xd, w):
            # 256 / 512 channels (conv2 of layer3 / layer4): bias + ReLU in the epilogue of the head's dense MFMA kernel
            return biasact.conv3x3_dense_bias_act(xd, w, b, relu)
        if plain3x3 and residual is None and biasact.narrow_fwd_ok(xd, w):
            return biasact.conv3x3_narrow_forward(xd, w, b, relu)  # 64 channels (conv2 of layer1): csrc/narrow3x3.hip, bias + ReLU in its epilogue
        if plain3x3 and biasact.narrow_fwd_ok(xd, w):
            y = biasact.conv3x3_narrow_forward(xd, w)
        else:
            y = F.conv2d(xd, w, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        if bn._fusable(y, residual):
            return _BNAct.apply(y, residual, ones, b, zeros, var1, None, False, 0.0, bn.eps, relu)
        y = y + b.to(y.dtype)[None, :, None, None]
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
    return bn(pwconv.conv1x1(conv, x), relu=relu, residual=residual)


def _conv3x3(cin: int, cout: int, stride: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def _conv1x1(cin: int, cout: int, stride: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: Optional[nn.Module] = None):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = FusedBatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = FusedBatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x: Tensor) -> Tensor:
        identity = x if self.downsample is None else conv_bn(self.downsample[0], self.downsample[1], x)
        out = conv_bn(self.conv1, self.bn1, x, relu=True)
        return conv_bn(self.conv2, self.bn2, out, relu=True, residual=identity)      # relu(bn2(.) + identity), one kernel


class _SkipLink:
    "Carries the identity branch's gradient from bn3's backward to conv1's (``_Conv1x1Skip``)."
    __slots__ = ("dres",)

    def __init__(self):
        self.dres = None


class _Conv1x1Skip(torch.autograd.Function):
    """conv1 (1x1, stride 1) of an identity bottleneck.  Its data gradient is a plain GEMM [M, Cmid] x [Cmid, Cin] on the
    channels-last activations, and the block input's other gradient -- the identity branch's, which autograd would add with
    a separate elementwise kernel (3 passes over the block's largest tensor; 12 blocks, 0.6 ms per step) -- goes in as the
    GEMM's accumulator input: ``addmm(dres, g, w)``, one library GEMM (hipBLASLt) instead of MIOpen's data gradient + add."""

    @staticmethod
    def forward(ctx, x, w, link):
        ctx.save_for_backward(x, w)
        ctx.link = link
        N, Cin, H, W = x.shape
        Cmid = w.shape[0]
        if pwconv.MM_1X1 and x.dtype in H16 and pwconv._fwd_by_mm(N * H * W, Cin, Cmid):       # (hipBLASLt: pwconv._Conv1x1)
            return (x.permute(0, 2, 3, 1).reshape(-1, Cin) @ w.reshape(Cmid, Cin).t()).view(N, H, W, Cmid).permute(0, 3, 1, 2)
        return F.conv2d(x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        link = ctx.link
        dres, link.dres = link.dres, None
        N, Cmid, H, W = g.shape
        Cin = w.shape[1]
        if not g.is_contiguous(memory_format=torch.channels_last):
            g = g.contiguous(memory_format=torch.channels_last)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            g2 = g.permute(0, 2, 3, 1).reshape(-1, Cmid)
            w2 = w.reshape(Cmid, Cin)
            if dres is not None and dres.is_contiguous(memory_format=torch.channels_last):
                # accumulate INTO the identity gradient (bn3's backward wrote it for this block alone): the out-of-place addmm
                # first copies its 69 MB into the result (25 us per layer3 block)
                dx2 = dres.permute(0, 2, 3, 1).reshape(-1, Cin).addmm_(g2, w2)
            elif dres is not None:
                dx2 = torch.addmm(dres.permute(0, 2, 3, 1).reshape(-1, Cin), g2, w2)
            else:
                dx2 = g2 @ w2
            dx = dx2.view(N, H, W, Cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            if pwconv.MM_1X1 and x.dtype in H16 and w.dtype == x.dtype and Cin % 64 == 0 and Cmid % 64 == 0:
                dw = pwconv.pw_wgrad(g, x, w, tag="pw_1x1_wgrad")                # position-contraction kernel of csrc/pw.hip
            else:
                dw = torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return dx, dw, None


FUSE_SKIP_ADD = True
# layer1 / layer2 only (64 / 128 mid channels): 491 us for the five GEMMs against 285 us of MIOpen data gradients + 438 us of adds;
# layer3 is a tie (5 x 54 us vs 125 + 150), layer4 a loss (2 x 38 vs 24 + 30) -- step timeline, hipBLASLt's stream-K picks
SKIP_ADD_MAX_MID = 256


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: Optional[nn.Module] = None):
        super().__init__()
        self.conv1 = _conv1x1(inplanes, planes)
        self.bn1 = FusedBatchNorm2d(planes)
        self.conv2 = _conv3x3(planes, planes, stride)
        self.bn2 = FusedBatchNorm2d(planes)
        self.conv3 = _conv1x1(planes, planes * self.expansion)
        self.bn3 = FusedBatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x: Tensor) -> Tensor:
        if pwconv.bottleneck_fusable(self, x):
            # training, bf16 channels-last, BatchNorm on batch statistics: the whole block on the MFMA GEMMs of csrc/pw.hip with
            # the BatchNorm work fused into their operand loads / epilogues (pwconv._BottleneckFn)
            return pwconv.bottleneck(self, x)
        if (FUSE_SKIP_ADD and self.downsample is None and torch.is_grad_enabled() and x.requires_grad and x.is_cuda
                and x.dtype in (torch.bfloat16, torch.float16) and x.is_contiguous(memory_format=torch.channels_last)
                and self.conv1.stride == (1, 1) and self.conv1.groups == 1 and self.conv1.out_channels <= SKIP_ADD_MAX_MID):
            # identity block, training: the identity branch's gradient rides in conv1's data-gradient GEMM (_Conv1x1Skip)
            link = _SkipLink()
            out = self.bn1(_Conv1x1Skip.apply(x, self.conv1.weight.to(x.dtype), link), relu=True)
            out = conv_bn(self.conv2, self.bn2, out, relu=True)
            out3 = pwconv.conv1x1(self.conv3, out)
            if self.bn3._fusable(out3, x):
                return self.bn3(out3, relu=True, residual=x, link=link)
            return self.bn3(out3, relu=True, residual=x)
        # (conv1 before the downsample branch: the first 1x1 consumer of x receives the other consumers' gradients, pwconv._GradJoin)
        out = conv_bn(self.conv1, self.bn1, x, relu=True)
        identity = x if self.downsample is None else conv_bn(self.downsample[0], self.downsample[1], x)
        out = conv_bn(self.conv2, self.bn2, out, relu=True)
        return conv_bn(self.conv3, self.bn3, out, relu=True, residual=identity)      # relu(bn3(.) + identity), one kernel


class ResNetBackbone(nn.Module):
    "ResNet trunk without avgpool/fc; returns layer2/3/4 outputs (backbone.py:139-263)."

    def __init__(self, block: Type[Union[BasicBlock, Bottleneck]], layers: List[int], zero_init_residual: bool = False):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = FusedBatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = FusedMaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        if block is Bottleneck:        # who consumes whose output: a fused block forms the next one's conv1 with its own output (pwconv.FUSE_CHAIN)
            pwconv.link_blocks(list(self.layer1) + list(self.layer2) + list(self.layer3) + list(self.layer4))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):        # (FusedBatchNorm2d is a BatchNorm2d)
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes: int, blocks: int, stride: int = 1) -> nn.Sequential:
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_conv1x1(self.inplanes, planes * block.expansion, stride),
                                       FusedBatchNorm2d(planes * block.expansion))
        stack = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        stack += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*stack)

    # Stage cuts (graph.CapturedTrainStep with a gradient exchange): C3 / C4 / C5 are replaced by detached leaves, so that the backward
    # pass can run as separate autograd calls -- head + FPN | layer4, layer3 | layer2 .. stem -- each captured in its own linear hipGraph,
    # with the finished buckets' all-reduces issued eagerly between the replays (``StageCuts.pairs``: (producer output, leaf)).
    stage_cuts: Optional["StageCuts"] = None

    def _stem_inference(self, x: Tensor) -> Tensor:
        "``maxpool(relu(bn1(conv1(x))))`` outside training: frozen BatchNorm folded, on the MFMA stem kernel where it applies."
        bn = self.bn1
        if FOLD_FROZEN_BN and not bn.training and not torch.is_grad_enabled() and x.is_cuda and bn.track_running_stats:
            dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
            w, b = _folded(self.conv1, bn, dt)[:2]
            xd = x.to(dt)
            if pwconv.stem_eval_ok(self.conv1, self.maxpool, xd, w):
                return pwconv.stem_eval(xd, w, b)
        return self.maxpool(conv_bn(self.conv1, self.bn1, x, relu=True))

    def _forward_impl(self, x: Tensor) -> Dict[str, Tensor]:
        if pwconv.stem_fusable(self.conv1, self.bn1, x):     # training, bf16: the MFMA stem kernel with bn1's statistics in its epilogue
            x = pwconv.stem(self.conv1, self.bn1, x, pool=self.maxpool)      # ... and bn1's apply + ReLU inside the max pooling
        else:
            x = self._stem_inference(x)
        pwconv._BWD_CHAIN.clear()        # (hand-overs of a backward pass that never ran)
        x = self.layer1(x)
        # C3 / C4 feed the next layer's conv1 + stride-2 downsample conv and an FPN lateral: their data gradients join in one GEMM
        cuts = self.stage_cuts if (self.stage_cuts is not None and torch.is_grad_enabled()) else None
        c3 = self.layer2(x)
        if cuts is not None:
            c3 = cuts.cut(c3)
        c3 = pwconv.share_gradients(c3)       # (across a cut the consumers run in different backward passes: donors notice and return their own)
        c4 = self.layer3(c3)
        if cuts is not None:
            c4 = cuts.cut(c4)
        c4 = pwconv.share_gradients(c4)
        c5 = self.layer4(c4)
        if cuts is not None:
            c5 = cuts.cut(c5)
        return {"layer_2": c3, "layer_3": c4, "layer_4": c5}

    def forward(self, x: Tensor) -> Dict[str, Tensor]:
        return self._forward_impl(x)


class StageCuts:
    """Collector of the cut points of one forward pass: ``cut(t)`` returns a detached leaf that stands in for ``t`` downstream."""

    def __init__(self):
        self.pairs: List[Tuple[Tensor, Tensor]] = []

    def cut(self, t: Tensor) -> Tensor:
        if not t.requires_grad:
            return t
        leaf = t.detach().requires_grad_(True)
        self.pairs.append((t, leaf))
        return leaf

    def clear(self) -> None:
        self.pairs = []


_SPECS = {
    "resnet18": (BasicBlock, [2, 2, 2, 2]),
    "resnet34": (BasicBlock, [3, 4, 6, 3]),
    "resnet50": (Bottleneck, [3, 4, 6, 3]),
    "resnet101": (Bottleneck, [3, 4, 23, 3]),
    "resnet152": (Bottleneck, [3, 8, 36, 3]),
}


def _load_pretrained(model: nn.Module, arch: str, progress: bool) -> None:
    """ImageNet weights (backbone.py:269-274).  Uses ``torch.hub`` (the reference's
    ``torchvision.models.utils`` loader no longer exists); a local file named by
    ``RETINANET_PRETRAINED_DIR/<basename>`` is preferred so air-gapped nodes work."""
    url = model_urls[arch]
    local_dir = os.environ.get("RETINANET_PRETRAINED_DIR")
    if local_dir and os.path.exists(os.path.join(local_dir, os.path.basename(url))):
        state = torch.load(os.path.join(local_dir, os.path.basename(url)), map_location="cpu")
    else:
        state = torch.hub.load_state_dict_from_url(url, progress=progress)
    model.load_state_dict(state, strict=False)


def _resnet(arch: str, pretrained: bool, progress: bool, **kwargs) -> ResNetBackbone:
    block, layers = _SPECS[arch]
    model = ResNetBackbone(block, layers, **kwargs)
    if pretrained:
        _load_pretrained(model, arch, progress)
    return model


def resnet18(pretrained: bool = False, progress: bool = True, **kw): return _resnet("resnet18", pretrained, progress, **kw)
def resnet34(pretrained: bool = False, progress: bool = True, **kw): return _resnet("resnet34", pretrained, progress, **kw)
def resnet50(pretrained: bool = False, progress: bool = True, **kw): return _resnet("resnet50", pretrained, progress, **kw)
def resnet101(pretrained: bool = False, progress: bool = True, **kw): return _resnet("resnet101", pretrained, progress, **kw)
def resnet152(pretrained: bool = False, progress: bool = True, **kw): return _resnet("resnet152", pretrained, progress, **kw)


loaders = {"resnet18": resnet18, "resnet34": resnet34, "resnet50": resnet50, "resnet101": resnet101, "resnet152": resnet152}


class BackBone(nn.Module):
    """``BackBone(kind, pretrained, freeze_bn)`` -> [C3, C4, C5] (backbone.py:340-360).

    ``freeze_bn`` only puts the BN layers in eval mode at construction, exactly like
    the reference: a later ``.train()`` un-freezes them."""

    def __init__(self, kind: str = "resnet18", pretrained: bool = True, freeze_bn: bool = True, **kwargs):
        super().__init__()
        self.backbone = loaders[kind](pretrained=pretrained, **kwargs)
        if freeze_bn:
            for layer in self.modules():
                if isinstance(layer, nn.BatchNorm2d):
                    layer.eval()

    def forward(self, xb: Tensor) -> List[Tensor]:
        out = self.backbone(xb)
        return [out["layer_2"], out["layer_3"], out["layer_4"]]


def get_backbone(kind: str = "resnet50", pretrained: bool = True, freeze_bn: bool = True) -> nn.Module:
    if kind not in __all__:
        raise ValueError(f"`kind` must be one of {__all__} got {kind}")
    return BackBone(kind=kind, pretrained=pretrained, freeze_bn=freeze_bn)
