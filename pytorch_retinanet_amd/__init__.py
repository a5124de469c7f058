"""MI355X-native RetinaNet dense-head path behind the Python surface of
benihime91/pytorch_retinanet.

    from pytorch_retinanet_amd import Retinanet, AnchorGenerator      # reference: retinanet/__init__.py:1-2
    from pytorch_retinanet_amd import RetinaNetModel, load_hparams    # reference: model.py:18, hparams.yaml

Importing this package loads ``libretinanet_hip.so`` (hand-written HIP for gfx950,
C ABI in ``include/retinanet_hip.h``) and fails loudly if it has not been built.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is missing)
from . import ops  # noqa: F401
from .anchors import AnchorGenerator
from .box_utils import activ_2_bbox, bbox_2_activ, matcher
from .coco_eval import CocoEvaluator
from .datasets import CSVDetectionDataset
from .losses import RetinaNetLosses
from .model import RetinaNetModel, SimpleTrainer, SyntheticDetectionDataset
from .models import Retinanet
from .parallel import BucketedGradAllReduce, ExchangeGradScaler
from .utils import collate_fn, load_hparams, load_obj

__all__ = ["Retinanet", "AnchorGenerator", "RetinaNetLosses", "RetinaNetModel", "SimpleTrainer",
           "SyntheticDetectionDataset", "BucketedGradAllReduce", "ExchangeGradScaler", "matcher", "bbox_2_activ", "activ_2_bbox",
           "collate_fn", "load_obj", "load_hparams", "ops", "CocoEvaluator", "CSVDetectionDataset"]
