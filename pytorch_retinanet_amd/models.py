"""``Retinanet`` -- the detector, with the reference's constructor, methods,
state-dict keys and output formats (``retinanet/models.py:21-288``).

Host side (this file, PyTorch-ROCm): transform -> ResNet -> FPN -> heads.
Device side (HIP, ``csrc/``): anchors (K1), matching + loss with gradients
(K2, K3) in ``forward``; decode, per-class NMS and top-k (K4-K7) in ``predict``.
"""
import logging
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from .anchors import AnchorGenerator
from .backbone import get_backbone
from .config import (BACKBONE, BBOX_REG_WEIGHTS, FREEZE_BN, MAX_DETECTIONS_PER_IMAGE, MAX_IMAGE_SIZE, MEAN,
                     MIN_BOX_SIZE, MIN_IMAGE_SIZE, NMS_THRES, NUM_CLASSES, PRETRAINED_BACKBONE, PRIOR, SCORE_THRES, STD)
from .layers import FeaturePyramid, RetinaNetHead
from .transform import GeneralizedRCNNTransform
from .utilities import ifnone

__small__ = ["resnet18", "resnet34"]
__big__ = ["resnet50", "resnet101", "resnet101", "resnet152"]


class Retinanet(nn.Module):
    """RetinaNet (Lin et al.) on a ResNet-FPN backbone.

    ``forward(images, targets)`` -> ``{"classification_loss", "regression_loss"}``;
    ``predict(images)`` -> ``[{"boxes" [n,4], "scores" [n], "labels" [n] in 1..K}]``, n <= 100.
    ``images``: list of ``[3,H,W]`` tensors in 0..1; ``targets``: list of
    ``{"boxes": F32[N,4] xyxy, "labels": I64[N] in 1..K}``.  Every constructor argument
    defaults to ``config.py`` (models.py:73-107).
    """

    def __init__(
        self,
        num_classes: Optional[int] = None,
        backbone_kind: Optional[str] = None,
        prior: Optional[float] = None,
        pretrained: Optional[bool] = None,
        nms_thres: Optional[float] = None,
        score_thres: Optional[float] = None,
        max_detections_per_images: Optional[int] = None,
        freeze_bn: Optional[bool] = None,
        min_size: Optional[int] = None,
        max_size: Optional[int] = None,
        image_mean: Optional[List[float]] = None,
        image_std: Optional[List[float]] = None,
        anchor_generator: Optional[AnchorGenerator] = None,
        logger=None,
    ) -> None:
        super().__init__()
        num_classes = ifnone(num_classes, NUM_CLASSES)
        backbone_kind = ifnone(backbone_kind, BACKBONE)
        prior = ifnone(prior, PRIOR)
        pretrained = ifnone(pretrained, PRETRAINED_BACKBONE)
        nms_thres = ifnone(nms_thres, NMS_THRES)
        score_thres = ifnone(score_thres, SCORE_THRES)
        max_detections_per_images = ifnone(max_detections_per_images, MAX_DETECTIONS_PER_IMAGE)
        freeze_bn = ifnone(freeze_bn, FREEZE_BN)
        min_size = ifnone(min_size, MIN_IMAGE_SIZE)
        max_size = ifnone(max_size, MAX_IMAGE_SIZE)
        image_mean = ifnone(image_mean, MEAN)
        image_std = ifnone(image_std, STD)
        anchor_generator = ifnone(anchor_generator, AnchorGenerator())
        logger = ifnone(logger, logging.getLogger(__name__))
        logger.name = __name__

        if backbone_kind not in __small__ + __big__:
            raise ValueError(f"Expected `backbone_kind` to be one of {__small__ + __big__} got {backbone_kind}")

        self.backbone_kind = backbone_kind
        self.transform = GeneralizedRCNNTransform(min_size, max_size, image_mean, image_std)
        self.backbone = get_backbone(backbone_kind, pretrained, freeze_bn=freeze_bn)
        c3, c4, c5 = self._get_backbone_ouputs()
        self.fpn = FeaturePyramid(c3, c4, c5, 256)
        self.anchor_generator = anchor_generator
        num_anchors = self.anchor_generator.num_cell_anchors[0]
        self.retinanet_head = RetinaNetHead(256, 256, num_anchors, num_classes, prior)

        self.score_thres = score_thres
        self.nms_thres = nms_thres
        self.detections_per_img = max_detections_per_images
        self.num_classes = num_classes

        logger.info(f"BACKBONE     : {backbone_kind}")
        logger.info(f"INPUT_PARAMS : MAX_SIZE={max_size}, MIN_SIZE={min_size}")
        logger.info(f"NUM_CLASSES  : {self.num_classes}")

    def _get_backbone_ouputs(self) -> List[int]:
        "Channel counts of C3, C4, C5 (models.py:135-150)."
        net = self.backbone.backbone
        last = "conv2" if self.backbone_kind in __small__ else "conv3"
        return [getattr(stage[-1], last).out_channels for stage in (net.layer2, net.layer3, net.layer4)]

    # -- conv stack ------------------------------------------------------------------------
    def _features(self, batch: Tensor) -> Tuple[List[Tensor], Dict[str, Tensor]]:
        if self.backbone.backbone.conv1.weight.is_contiguous(memory_format=torch.channels_last):
            batch = batch.contiguous(memory_format=torch.channels_last)   # model was moved to channels_last
        feature_maps = self.fpn(self.backbone(batch))
        return feature_maps, self.retinanet_head(feature_maps)

    def _batch_layout(self) -> Dict[str, object]:
        """Layout / dtype the conv stack will consume, for the fused transform kernel: channels-last when
        the model is, and autocast's dtype when autocast is on (conv1 would cast its input to it anyway,
        with the same round-to-nearest-even)."""
        w = self.backbone.backbone.conv1.weight
        cl = w.is_contiguous(memory_format=torch.channels_last)
        dt = torch.get_autocast_dtype("cuda") if (w.is_cuda and torch.is_autocast_enabled("cuda")) else None
        return {"out_dtype": dt, "channels_last": cl}

    # -- training ----------------------------------------------------------------------------
    def compute_loss(self, targets: List[Dict[str, Tensor]], outputs: Dict[str, Tensor],
                     anchors: List[Tensor]) -> Dict[str, Tensor]:
        return self.retinanet_head.compute_loss(targets, outputs, anchors)

    def forward(self, images: List[Tensor], targets: Optional[List[Dict[str, Tensor]]] = None):
        """Losses of the batch (models.py:274-288).  The reference requires `targets`
        (its Lightning wrapper's ``forward`` therefore raises); here
        ``targets=None`` means inference and returns ``predict(images)``."""
        if targets is None:
            return self.predict(images)
        images, targets = self.transform(images, targets, **self._batch_layout())
        batch = images.tensors
        if self.backbone.backbone.conv1.weight.is_contiguous(memory_format=torch.channels_last):
            batch = batch.contiguous(memory_format=torch.channels_last)        # no-op after the fused transform
        if batch.is_cuda and self.training and torch.is_grad_enabled():
            from . import biasact
            biasact.refresh_dgrad_weights(batch.device)            # the backward pass's flipped 3x3 weights: one launch for the whole model
        feature_maps = self.fpn(self.backbone(batch))
        anchors = self.anchor_generator(images, feature_maps)
        # K2 (IoU + matcher) needs only the anchors and the GT boxes: it goes out on a side stream here and runs beside the
        # head convolutions; K3 waits for it (losses.RetinaNetLosses.match_ahead).  NOT while the step is being captured into a
        # hipGraph: a graph with a forked branch costs hipGraphLaunch 20 ms of host time per replay on ROCm 7.0 (0.4 ms for
        # the linear graph of the same step, bench.py host_enqueue_ms_per_step) -- 25 us of GPU time are not worth that.
        ahead = None
        if batch.is_cuda and not torch.cuda.is_current_stream_capturing() and not self.retinanet_head.losses.fuses_match(targets):
            ahead = self.retinanet_head.losses.match_ahead(targets, anchors)      # (<= 64 GT per image: K2 runs inside K3, no launch at all)
        # same losses as compute_loss(targets, retinanet_head(feature_maps), anchors), but the loss kernel
        # reads the five per-level conv outputs in place instead of their torch.cat (layers.py:195, :259)
        outputs = self.retinanet_head.forward_levels(feature_maps)
        return self.retinanet_head.compute_loss_levels(targets, outputs, anchors, ahead=ahead)

    # -- inference ---------------------------------------------------------------------------
    def process_detections(self, outputs: Dict[str, Tensor], anchors: List[Tensor],
                           im_szs: List[Tuple[int, int]]) -> List[Dict[str, Tensor]]:
        """Raw head outputs -> per-image detections (models.py:160-243): one HIP call
        (``rn_detect``) for the whole batch instead of B*K sequential NMS launches."""
        class_logits = outputs.pop("cls_preds")
        bboxes = outputs.pop("bbox_preds")
        if any(w != 1.0 for w in BBOX_REG_WEIGHTS):
            bboxes.div_(bboxes.new_tensor([BBOX_REG_WEIGHTS]))          # box_utils.py:43 (in place, Q5)
        return ops.detect(class_logits, bboxes, anchors, im_szs, self.score_thres, MIN_BOX_SIZE, self.nms_thres,
                          self.detections_per_img)

    def process_detections_levels(self, outputs: Dict[str, List[Tensor]], anchors: List[Tensor],
                                  im_szs: List[Tuple[int, int]]) -> List[Dict[str, Tensor]]:
        """``process_detections`` on the per-level head outputs of ``retinanet_head.forward_levels``: the scan
        and decode kernels read the five conv outputs in place (no ``torch.cat``, layers.py:195, :259)."""
        return ops.detect_levels(outputs["cls_levels"], outputs["bbox_levels"], anchors, im_szs, self.score_thres,
                                 MIN_BOX_SIZE, self.nms_thres, self.detections_per_img, reg_w=BBOX_REG_WEIGHTS)

    def predict(self, images: List[Tensor]) -> List[Dict[str, Tensor]]:
        """Detections in original image coordinates (models.py:245-272).  Like the
        reference this only flips the TOP module's ``training`` flag (Q17): call
        ``.eval()`` first for eval-mode BN and for the rescale to original sizes.
        The conv stack runs under ``no_grad`` (nothing back-propagates through detections)."""
        if self.training:
            self.training = False
        original_image_sizes = []
        for img in images:
            val = img.shape[-2:]
            assert len(val) == 2
            original_image_sizes.append((int(val[0]), int(val[1])))
        with torch.no_grad():
            images, _ = self.transform(images, None, **self._batch_layout())
            batch = images.tensors
            if self.backbone.backbone.conv1.weight.is_contiguous(memory_format=torch.channels_last):
                batch = batch.contiguous(memory_format=torch.channels_last)
            feature_maps = self.fpn(self.backbone(batch))
            anchors = self.anchor_generator(images, feature_maps)
            detections = self.process_detections_levels(self.retinanet_head.forward_levels(feature_maps), anchors,
                                                        images.image_sizes)
            return self.transform.postprocess(detections, images.image_sizes, original_image_sizes)
