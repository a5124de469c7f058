"""Defaults of the RetinaNet surface (same names and values as the reference's
``retinanet/config.py:12-87`` -- these module constants ARE the configuration API:
every constructor argument left as ``None`` falls back to them)."""
from typing import List

# ---- input normalisation / resize (config.py:12-18) --------------------------
MEAN: List[float] = [0.485, 0.456, 0.406]
STD: List[float] = [0.229, 0.224, 0.225]
MIN_IMAGE_SIZE: int = 800
MAX_IMAGE_SIZE: int = 1333

# ---- anchors (config.py:27-42): 3 octave scales x 3 ratios on P3..P7 -----------
ANCHOR_SIZES: List[List[float]] = [[s * 2 ** (i / 3) for i in range(3)] for s in (32, 64, 128, 256, 512)]
ANCHOR_STRIDES: List[int] = [8, 16, 32, 64, 128]
ANCHOR_ASPECT_RATIOS: List[float] = [0.5, 1.0, 2.0]
ANCHOR_OFFSET: float = 0.0

# ---- head / backbone (config.py:48-67) ------------------------------------------
NUM_CLASSES: int = 90
BACKBONE: str = "resnet50"
PRETRAINED_BACKBONE: bool = True
PRIOR: float = 0.01
FREEZE_BN: bool = True
BBOX_REG_WEIGHTS: List[float] = [1.0, 1.0, 1.0, 1.0]

# ---- inference (config.py:71-75) ------------------------------------------------
SCORE_THRES: float = 0.05
NMS_THRES: float = 0.5
MAX_DETECTIONS_PER_IMAGE: int = 100

# ---- anchor labelling (config.py:81-82) -----------------------------------------
IOU_THRESHOLDS_FOREGROUND: float = 0.5
IOU_THRESHOLDS_BACKGROUND: float = 0.4

# ---- losses (config.py:85-87) -----------------------------------------------------
FOCAL_LOSS_GAMMA: float = 2.0
FOCAL_LOSS_ALPHA: float = 0.25
SMOOTH_L1_LOSS_BETA: float = 0.1

# ---- not in the reference: constants its code hard-wires --------------------------
LOGIT_SHIFT: float = 1.0        # losses.py:84  clas_pred = clas_pred + 1
ENCODE_LOG_EPS: float = 1e-8    # box_utils.py:32
MIN_BOX_SIZE: float = 1e-2      # models.py:203
