"""FPN and the RetinaNet head (reference ``retinanet/layers.py:12-260``).

Parameter names, shapes and initialisation match the reference so its
checkpoints load unchanged (``fpn.conv_c{3,4,5}_{1x1,3x3}``,
``fpn.conv_c6_3x3``, ``fpn.conv_c7_3x3``, ``retinanet_head.*_head.*_subnet.{0,2,4,6}``,
``*_subnet_output``).  The convolutions run on PyTorch-ROCm (MIOpen); what is
MI355X-specific here is the output layout: with ``channels_last`` activations the
conv result ``[N, A*K, H, W]`` is *already* ``[N, H, W, A, K]`` in memory, so the
reference's ``view -> permute -> contiguous`` (layers.py:189-191, :253-255) is a
zero-copy reshape and the loss / detection kernels stream it directly.
"""
import math
import os
from typing import Dict, List

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import biasact
from .losses import RetinaNetLosses
from .pool import add_upsample2x

FUSE_FPN_UPSAMPLE = True     # lateral + 2x nearest upsampling in one kernel


class FeaturePyramid(nn.Module):
    """P3-P7 from C3-C5 (layers.py:12-64): lateral 1x1 + top-down nearest 2x +
    3x3 smoothing; P6 = 3x3/s2 on **C5**; P7 = 3x3/s2 on relu(P6)."""

    def __init__(self, C_3_size: int, C_4_size: int, C_5_size: int, out_channels: int = 256) -> None:
        super().__init__()
        self.conv_c3_1x1 = nn.Conv2d(C_3_size, out_channels, 1, 1, padding=0)
        self.conv_c3_3x3 = nn.Conv2d(out_channels, out_channels, 3, 1, padding=1)
        self.conv_c4_1x1 = nn.Conv2d(C_4_size, out_channels, 1, 1, padding=0)
        self.conv_c4_3x3 = nn.Conv2d(out_channels, out_channels, 3, 1, padding=1)
        self.conv_c5_1x1 = nn.Conv2d(C_5_size, out_channels, 1, 1, padding=0)
        self.conv_c5_3x3 = nn.Conv2d(out_channels, out_channels, 3, 1, padding=1)
        self.conv_c6_3x3 = nn.Conv2d(C_5_size, out_channels, 3, stride=2, padding=1)
        self.conv_c7_3x3 = nn.Conv2d(out_channels, out_channels, 3, stride=2, padding=1)
        self.upsample_2x = nn.Upsample(scale_factor=2, mode="nearest")
        for m in self.children():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)

    def _up(self, x: Tensor) -> Tensor:
        # autocast runs nearest upsampling in fp32 and thereby promotes the whole top-down pathway (adds, their
        # backward, the casts in front of the 3x3 convs) to fp32; a 2x nearest upsampling is a copy, so keep the dtype
        if x.is_cuda and torch.is_autocast_enabled("cuda"):
            with torch.autocast("cuda", enabled=False):
                return self.upsample_2x(x)
        return self.upsample_2x(x)

    def _lateral_plus_up(self, lat: Tensor, top: Tensor) -> Tensor:
        "``lat + upsample_2x(top)`` (layers.py:36,52-53): one fused pass on channels-last CUDA tensors (pool.add_upsample2x)."
        if FUSE_FPN_UPSAMPLE and lat.is_cuda:
            out = add_upsample2x(lat, top)
            if out is not None:
                return out
        return lat + self._up(top)

    @staticmethod
    def _conv_own_bias(conv: nn.Conv2d, x: Tensor) -> Tensor:
        """``conv(x)`` with the bias added -- and its gradient summed -- by the library's kernels (``biasact.bias_act``).  Why: autograd's
        ``convolution_backward`` forms a bias gradient with ``aten::sum``, which clears its accumulator with ``hipMemsetAsync``; in a
        captured step that is a MEMSET NODE, and on ROCm 7.0 the memset nodes of a replayed hipGraph write garbage once the process has
        synchronised with the device and done other work (round 4: K2's 32-byte ``num_fg`` memset made every replay after the first
        ``torch.cuda.synchronize()`` scale the losses by 1 / garbage; ``tests/test_graph_gpu.py``).  The captured step holds no memset
        node any more."""
        if x.is_cuda and conv.bias is not None and conv.bias.dtype == torch.float32 and torch.is_grad_enabled() and conv.bias.requires_grad:
            from . import pwconv
            if pwconv.conv3x3_s2_ok(conv, x, bias_ok=True):
                y = pwconv._Conv3x3S2.apply(x, conv.weight)          # P6 / P7: weight gradient on csrc/pw.hip
            else:
                y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
            if biasact.fusable(y, conv.bias):
                return biasact.bias_act(y, conv.bias, None, relu=False)
            return y + conv.bias.to(y.dtype)[None, :, None, None]
        return conv(x)

    def forward(self, inps: List[Tensor]) -> List[Tensor]:
        c3, c4, c5 = inps
        from .pwconv import conv1x1            # laterals: 1x1 GEMMs on the fastest of MIOpen / hipBLASLt / csrc/pw.hip per product
        p5 = conv1x1(self.conv_c5_1x1, c5)
        p4 = self._lateral_plus_up(conv1x1(self.conv_c4_1x1, c4), p5)
        p3 = self._lateral_plus_up(conv1x1(self.conv_c3_1x1, c3), p4)
        p6 = self._conv_own_bias(self.conv_c6_3x3, c5)
        p7 = self._conv_own_bias(self.conv_c7_3x3, F.relu(p6))
        outs, convs = [p3, p4, p5], [self.conv_c3_3x3, self.conv_c4_3x3, self.conv_c5_3x3]
        if biasact.dense_group_fusable(outs, convs):       # one MFMA launch each way for the three levels (csrc/conv.hip, MODE_DENSE)
            return biasact.dense_conv_group(outs, convs) + [p6, p7]
        return [conv(p) for conv, p in zip(convs, outs)] + [p6, p7]


def _tower(in_channels: int, out_channels: int) -> nn.Sequential:
    "4 x [3x3 conv + ReLU]; module indices 0,2,4,6 hold the convs like the reference."
    mods = []
    for i in range(4):
        mods += [nn.Conv2d(in_channels if i == 0 else out_channels, out_channels, 3, stride=1, padding=1),
                 nn.ReLU(inplace=True)]
    return nn.Sequential(*mods)


def _tower_on_canvas(tower: nn.Sequential, x: Tensor, mask: Tensor, mfma: bool = False) -> Tensor:
    """The same four conv + ReLU pairs applied to the packed level canvas.  ``mfma`` (bf16, zero-bordered canvas):
    the hand-written MFMA implicit GEMM with the bias + ReLU + gap-mask epilogue fused (csrc/conv.hip);
    otherwise MIOpen's conv without bias followed by the fused epilogue kernel (biasact.py)."""
    for layer in tower:
        if isinstance(layer, nn.Conv2d):
            if mfma:
                x = biasact.tower_conv(x, layer.weight, layer.bias, mask)
            else:
                x = biasact.bias_act(F.conv2d(x, layer.weight, None, layer.stride, layer.padding), layer.bias, mask, relu=True)
    return x


def _init_head(*modules: nn.Module) -> None:
    for mod in modules:
        for layer in mod.modules():
            if isinstance(layer, nn.Conv2d):
                nn.init.normal_(layer.weight, mean=0, std=0.01)
                nn.init.constant_(layer.bias, 0)


def _to_anchor_major(x: Tensor, last: int) -> Tensor:
    """[N, A*last, H, W] -> [N, H*W*A, last], anchor index (h*W + w)*A + a
    (layers.py:189-191).  Zero-copy when `x` is channels_last."""
    n = x.shape[0]
    return x.permute(0, 2, 3, 1).reshape(n, -1, last)


class RetinaNetClassSubnet(nn.Module):
    """Classification tower -> ``[N, sum(H*W*A), num_classes]`` logits (layers.py:118-196)."""

    def __init__(self, in_channels: int, out_channels: int, num_anchors: int, num_classes: int, prior: float) -> None:
        super().__init__()
        self.num_classes = num_classes
        self.num_anchors = num_anchors
        self.class_subnet = _tower(in_channels, out_channels)
        self.class_subnet_output = nn.Conv2d(out_channels, num_anchors * num_classes, 3, stride=1, padding=1)
        _init_head(self.class_subnet, self.class_subnet_output)
        # prior-probability bias (layers.py:175-178)
        nn.init.constant_(self.class_subnet_output.bias, -math.log((1 - prior) / prior))

    # MIOpen's fast bf16 NHWC kernels need the output-channel count to be a multiple of 8: the reference's
    # 9 x 90 = 810 runs at 290 TFLOP/s on MI355X, 9 x 96 = 864 at 800 (tools/head_conv_probe.py: the
    # final cls conv of the R50 config is 7.7 ms fwd+bwd at 810 and 3.2 ms at 864).  The fast paths
    # therefore pad the CLASS dimension per anchor up to a multiple of 8 with dead classes: zero
    # weights and a bias of PAD_LOGIT, for which sigmoid(x + 1)^2 underflows to exactly 0 in fp32, so
    # they add exactly nothing to the focal loss and its gradients (K3), and never pass the score
    # threshold (K5).  The parameters keep the reference's shapes ([A*K, C, 3, 3]); the padded copies
    # are rebuilt from them every forward (2 MB).
    PAD_LOGIT = -80.0

    @property
    def padded_classes(self) -> int:
        return (self.num_classes + 7) // 8 * 8

    def _padded_output_params(self):
        conv, A, K, Kp = self.class_subnet_output, self.num_anchors, self.num_classes, self.padded_classes
        w = F.pad(conv.weight.reshape(A, K, *conv.weight.shape[1:]), (0, 0, 0, 0, 0, 0, 0, Kp - K))
        b = F.pad(conv.bias.reshape(A, K), (0, Kp - K), value=self.PAD_LOGIT)
        return w.reshape(A * Kp, *conv.weight.shape[1:]), b.reshape(A * Kp)

    def forward_levels(self, feature_maps: List[Tensor], pad_classes: bool = False) -> List[Tensor]:
        """Per-level logits [N, H*W*A, K]; views of the conv outputs when the activations are channels_last.
        ``pad_classes``: logits come back as [N, H*W*A, padded_classes], columns K.. are dead classes."""
        return self.output_levels([self.class_subnet(f) for f in feature_maps], pad_classes)

    def output_levels(self, tower_outputs: List[Tensor], pad_classes: bool = False) -> List[Tensor]:
        "Final 3x3 conv on every level's tower output -> logits [N, H*W*A, K or padded_classes]."
        if not pad_classes or self.padded_classes == self.num_classes:
            return [_to_anchor_major(self.class_subnet_output(t), self.num_classes) for t in tower_outputs]
        w, b = self._padded_output_params()
        return [_to_anchor_major(F.conv2d(t, w, b, stride=1, padding=1), self.padded_classes) for t in tower_outputs]

    def forward(self, feature_maps: List[Tensor]) -> Tensor:
        return torch.cat(self.forward_levels(feature_maps), dim=1)


class RetinaNetBoxSubnet(nn.Module):
    """Box tower -> ``[N, sum(H*W*A), 4]`` deltas (layers.py:199-260)."""

    def __init__(self, in_channels: int, out_channels: int, num_anchors: int) -> None:
        super().__init__()
        self.num_anchors = num_anchors
        self.box_subnet = _tower(in_channels, out_channels)
        self.box_subnet_output = nn.Conv2d(out_channels, num_anchors * 4, 3, padding=1, stride=1)
        _init_head(self.box_subnet, self.box_subnet_output)

    def forward_levels(self, feature_maps: List[Tensor]) -> List[Tensor]:
        return self.output_levels([self.box_subnet(f) for f in feature_maps])

    def output_levels(self, tower_outputs: List[Tensor]) -> List[Tensor]:
        return [_to_anchor_major(self.box_subnet_output(t), 4) for t in tower_outputs]

    def forward(self, feature_maps: List[Tensor]) -> Tensor:
        return torch.cat(self.forward_levels(feature_maps), dim=1)


class RetinaNetHead(nn.Module):
    """Both subnets + the loss module (layers.py:67-115)."""

    def __init__(self, in_channels: int, out_channels: int, num_anchors: int, num_classes: int, prior: float) -> None:
        super().__init__()
        self.classification_head = RetinaNetClassSubnet(in_channels, out_channels, num_anchors, num_classes, prior)
        self.regression_head = RetinaNetBoxSubnet(in_channels, out_channels, num_anchors)
        self.losses = RetinaNetLosses(num_classes)
        # bf16 canvas towers: "pair" = hand-written MFMA conv, cls + box tower batched per layer; "single" = the same kernel,
        # one launch per conv; "miopen" = MIOpen conv + fused epilogue kernel.  RN_TOWERS overrides (one of the three documented A/B switches, README).
        mode = os.environ.get("RN_TOWERS", "pair")
        self.mfma_towers = mode != "miopen"
        self.pair_towers = mode == "pair"
        # class-output conv on the hand-written MFMA kernel with dense 9*K-channel output (0: MIOpen on 9*ceil8(K) channels)
        self.mfma_cls_output = True          # (False: MIOpen on 9 * ceil8(K) channels with dead classes -- also the automatic path of fp32 / fp16 models)
        self.mfma_box_output = True      # box-output conv: MFMA data gradient (biasact._BoxOutputConv)

    def compute_loss(self, targets: List[Dict[str, Tensor]], outputs: Dict[str, Tensor],
                     anchors: List[Tensor]) -> Dict[str, Tensor]:
        return self.losses(targets, outputs, anchors)

    def forward(self, xb: List[Tensor]) -> Dict[str, Tensor]:
        return {"cls_preds": self.classification_head(xb), "bbox_preds": self.regression_head(xb)}

    def forward_levels(self, xb: List[Tensor], pad_classes: bool = True, canvas: bool = True) -> Dict[str, List[Tensor]]:
        """Head outputs left per pyramid level (no concatenation); consumed by ``compute_loss_levels`` and
        ``Retinanet.process_detections_levels``.  With ``pad_classes`` the logits carry dead classes up to a
        multiple of 8 (see ``RetinaNetClassSubnet``); they change neither losses, gradients nor detections.
        With ``canvas`` (CUDA, channels-last) the two towers run on all levels packed into one canvas --
        one conv per layer instead of five (biasact.py); the final convs run per level on the unpacked
        tower outputs so the loss / detection kernels keep reading dense per-level tensors."""
        ch, rh = self.classification_head, self.regression_head
        if canvas and len(xb) > 1 and all(biasact.fusable(f, ch.class_subnet[0].bias) for f in xb):
            convs = [m for m in list(ch.class_subnet) + list(rh.box_subnet) if isinstance(m, nn.Conv2d)]
            mfma = self.mfma_towers and all(biasact.tower_conv_fusable(xb[0], m) for m in convs)
            cv = biasact.Canvas.of(xb, pad=1 if mfma else 0)
            packed = biasact.pack_levels(cv, xb)
            top = None                             # TowerLink of the towers' last layer (paired towers only)
            cc = [m for m in ch.class_subnet if isinstance(m, nn.Conv2d)]
            bc = [m for m in rh.box_subnet if isinstance(m, nn.Conv2d)]
            if mfma and self.pair_towers and len(cc) == len(bc) and all(c.weight.shape == b.weight.shape and c.in_channels % 256 == 0
                                                    for c, b in zip(cc, bc)):
                xc = xb_ = packed                  # both towers layer by layer, one batched launch per layer and direction
                prev = None                        # (a layer's outputs feed only the next layer: its ReLU backward rides in
                for i, (c, b) in enumerate(zip(cc, bc)):       #  that layer's data-gradient kernel, biasact.TowerLink)
                    link = biasact.TowerLink()           # (the last layer's link goes to the two output convs below)
                    xc, xb_ = biasact.tower_conv_pair(xc, xb_, c.weight, b.weight, c.bias, b.bias, cv.mask, prev, link)
                    prev = link
                cls_c, box_t = xc, xb_
                top = prev
            else:
                cls_c = _tower_on_canvas(ch.class_subnet, packed, cv.mask, mfma)
                box_t = _tower_on_canvas(rh.box_subnet, packed, cv.mask, mfma)
            # the 36-channel box conv is tiny per level (2 TFLOP/s on P7): run it on the canvas too and unpack its
            # small output instead of the 256-channel tower output
            n_img = xb[0].shape[0]
            if mfma and self.mfma_box_output and biasact.box_output_conv_fusable(box_t, rh.box_subnet_output, cv):
                box_levels = biasact.box_output_conv(box_t, rh.box_subnet_output, cv, n_img, relu_link=(top, 1) if top is not None else None)
            else:
                box_c = rh.box_subnet_output(box_t)
                box_levels = [_to_anchor_major(t, 4) for t in biasact.unpack_levels(cv, box_c, n_img)]
            if mfma and self.mfma_cls_output and biasact.cls_output_conv_fusable(cls_c, ch.class_subnet_output, cv):
                # class-output conv straight from the canvas to dense per-level logits [N, h*w*A, K]: exactly A*K channels
                # (no dead classes for the loss kernel to stream) and no unpack copy of the 256-channel tower output
                return {"cls_levels": biasact.cls_output_conv(cls_c, ch.class_subnet_output, cv, ch.num_classes, n_img,
                                                              relu_link=(top, 0) if top is not None else None),
                        "bbox_levels": box_levels}
            cls_t = biasact.unpack_levels(cv, cls_c, n_img)
            return {"cls_levels": ch.output_levels(cls_t, pad_classes), "bbox_levels": box_levels}
        return {"cls_levels": ch.forward_levels(xb, pad_classes), "bbox_levels": rh.forward_levels(xb)}

    def compute_loss_levels(self, targets, outputs: Dict[str, List[Tensor]], anchors, ahead=None) -> Dict[str, Tensor]:
        return self.losses.forward_levels(targets, outputs["cls_levels"], outputs["bbox_levels"], anchors, ahead=ahead)
