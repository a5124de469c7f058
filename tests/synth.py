"""Seeded synthetic inputs shared by gen_golden.py, the parity tests and bench.py.

numpy's PCG64 ``default_rng`` is used (not torch's RNG) so the same seed gives
the same bytes in the build container and on the GPU box; fixtures carry an
input checksum so a drift would be detected rather than silently compared.

Distributions follow SURVEY.md section 8d / BASELINE.md section 2.
"""
import hashlib
import math
from typing import List, Sequence, Tuple

import numpy as np

ANCHOR_SIZES = [[x, x * 2 ** (1 / 3), x * 2 ** (2 / 3)] for x in [32, 64, 128, 256, 512]]
ANCHOR_RATIOS = [0.5, 1.0, 2.0]
ANCHOR_STRIDES = [8, 16, 32, 64, 128]


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def fpn_grid_sizes(h: int, w: int) -> List[Tuple[int, int]]:
    """Feature-map sizes P3..P7 for a padded input of (h, w) (both multiples of 32).

    C3/C4/C5 are /8,/16,/32; P6,P7 follow a 3x3 stride-2 pad-1 conv: floor((n-1)/2)+1.
    """
    p3 = (h // 8, w // 8)
    p4 = (h // 16, w // 16)
    p5 = (h // 32, w // 32)
    nxt = lambda n: (n - 1) // 2 + 1
    p6 = (nxt(p5[0]), nxt(p5[1]))
    p7 = (nxt(p6[0]), nxt(p6[1]))
    return [p3, p4, p5, p6, p7]


def levels_for(h: int, w: int) -> List[Tuple[int, int, int]]:
    return [(gh, gw, s) for (gh, gw), s in zip(fpn_grid_sizes(h, w), ANCHOR_STRIDES)]


def gt_boxes(rng: np.random.Generator, T: int, img_h: int, img_w: int, num_classes: int = 90,
             wh_lo: float = 16.0, wh_hi: float = 316.0):
    """T boxes: centre uniform over the image, w,h ~ U(wh_lo, wh_hi), xyxy clamped, non-degenerate."""
    boxes = np.zeros((T, 4), dtype=np.float32)
    for i in range(T):
        while True:
            cx = rng.uniform(0, img_w)
            cy = rng.uniform(0, img_h)
            bw = rng.uniform(wh_lo, wh_hi)
            bh = rng.uniform(wh_lo, wh_hi)
            x1, y1 = max(cx - bw / 2, 0.0), max(cy - bh / 2, 0.0)
            x2, y2 = min(cx + bw / 2, float(img_w)), min(cy + bh / 2, float(img_h))
            b = np.array([x1, y1, x2, y2], dtype=np.float32)
            if b[2] > b[0] and b[3] > b[1]:
                boxes[i] = b
                break
    labels = rng.integers(1, num_classes + 1, size=(T,)).astype(np.int64)
    return boxes, labels


def head_outputs(rng: np.random.Generator, B: int, A: int, K: int, cls_mean: float = -4.6, cls_std: float = 1.0,
                 box_std: float = 0.1):
    cls = (rng.standard_normal((B, A, K), dtype=np.float32) * np.float32(cls_std) + np.float32(cls_mean)).astype(np.float32)
    box = (rng.standard_normal((B, A, 4), dtype=np.float32) * np.float32(box_std)).astype(np.float32)
    return cls, box


def round_bf16(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (values stay fp32)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(a.shape)


def round_f16(a: np.ndarray) -> np.ndarray:
    return a.astype(np.float16).astype(np.float32)


def sample_idx(n: int, k: int, seed: int = 123) -> np.ndarray:
    return np.sort(np.random.default_rng(seed).choice(n, size=min(k, n), replace=False)).astype(np.int64)


def state_dict_values(spec: Sequence[Tuple[str, Tuple[int, ...], str]], seed: int, prior: float = 0.01, cls_std: float = 0.0016):
    """Seed-reproducible model weights for the end-to-end fixture (tests/golden/e2e.npz): one array per
    ``(key, shape, dtype name)`` of a RetinaNet ``state_dict()``, drawn in key order from PCG64(seed).

    Scales keep activations O(1) through the stack in train-mode AND eval-mode BatchNorm and give the detector
    something to do: conv weights N(0, g / fan_in) with g = 1 in the backbone (residual sums would otherwise double the
    variance per block when BN runs on its running statistics) and g = 2 (He) in FPN and head towers, BatchNorm
    gamma / running_var ~ U(0.5, 1.5), beta / running_mean ~ N(0, 0.1), cls-output weights N(0, 0.0016) around the
    prior bias (so a few hundred anchors per image pass the 0.05 score threshold, none saturates), box-output
    weights N(0, 0.0008).  ``cls_std``: the class-output weights' standard deviation (the headline-shape fixture uses 0.0006: on 800 x 1333
    noise images the FPN outputs have an rms of 8 - 17, and 0.0016 would put 1.8 M candidates per image into the reference's O(n^2) NMS).
    ``anchor_generator.cell_anchors.*`` keys are skipped (buffers computed by the model itself).
    """
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape, dtype in spec:
        if key.startswith("anchor_generator."):
            continue
        leaf = key.rsplit(".", 1)[-1]
        is_bn = ".bn" in key or ".downsample.1." in key
        if leaf == "num_batches_tracked":
            v = np.zeros(shape, dtype=np.int64)
        elif leaf == "running_var" or (is_bn and leaf == "weight"):
            v = rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif leaf == "running_mean" or (is_bn and leaf == "bias"):
            v = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        elif leaf == "weight" and len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            std = math.sqrt((1.0 if key.startswith("backbone.") else 2.0) / fan_in)
            if "class_subnet_output" in key:
                std = cls_std
            elif "box_subnet_output" in key:
                std = 0.0008
            v = (rng.standard_normal(shape) * std).astype(np.float32)
        elif leaf == "bias":
            v = (rng.standard_normal(shape) * 0.05).astype(np.float32)
            if "class_subnet_output" in key:
                v = (v * 6 - math.log((1 - prior) / prior)).astype(np.float32)
        else:
            raise KeyError(f"no rule for state-dict entry {key} {shape}")
        out[key] = v
    return out


def e2e_inputs(seed: int = 31):
    """Images and targets of the end-to-end fixture: two images of different sizes (resized by the transform to
    min_size 128 / max_size 160), 3 and 2 GT boxes, labels in 1..5."""
    rng = np.random.default_rng(seed)
    sizes = [(120, 150), (140, 128)]
    images = [rng.random((3, h, w), dtype=np.float32) for h, w in sizes]
    targets = []
    for (h, w), T in zip(sizes, (3, 2)):
        b, l = gt_boxes(rng, T, h, w, num_classes=5, wh_lo=20.0, wh_hi=90.0)
        targets.append((b, l))
    return images, targets


# The headline configuration end to end (tests/golden/e2e_full.npz, gen_golden.py e2e_full): BASELINE configs[1]'s model and image size
E2E_FULL = dict(num_classes=90, backbone_kind="resnet50", pretrained=False, min_size=800, max_size=1333)
E2E_FULL_CLS_STD = 0.0006       # class-output weight std of that fixture's state dict (state_dict_values): ~1e4 candidates per image in eval mode


def e2e_full_inputs(seed: int = 47):
    """Two images of exactly 3 x 800 x 1333 (the transform's identity scale; padded to 800 x 1344) with 8 GT boxes each, labels in
    1..90 -- BASELINE configs[1]'s per-image shape.  Regenerated from the seed (25.6 MB of pixels: the fixture records their sha256)."""
    rng = np.random.default_rng(seed)
    images = [rng.random((3, 800, 1333), dtype=np.float32) for _ in range(2)]
    targets = [gt_boxes(rng, 8, 800, 1333) for _ in range(2)]
    return images, targets


TRAJ = dict(num_classes=5, backbone_kind="resnet18", pretrained=False, min_size=128, max_size=160)
# The reference's optimizer (hparams.yaml:63-68: SGD, weight decay 1e-3, momentum 0.9) at a learning rate of 2e-5 instead of its 1e-3:
# on this synthetic state dict 1e-3 diverges (total loss 3.2 -> 18.9 / 212.9 at step 1) and the 5-step map amplifies a 1e-6 relative
# perturbation of the initial weights to 9e-4 in the losses -- no tolerance could tell a wrong step from rounding.  At 2e-5 the
# losses fall (3.23 -> 3.01, 3.89 -> 3.55) and the same perturbation stays at 2e-6 (scratch-measured with the reference itself,
# tests/golden/gen_golden.py gen_traj prints the trajectory).
TRAJ_OPT = dict(lr=2e-5, weight_decay=1e-3, momentum=0.9)
TRAJ_STEPS = 5


def traj_inputs(kind: str, step: int):
    """Batch ``step`` of the training-trajectory fixture (tests/golden/traj.npz).
    ``live``: two images of different sizes (the transform resizes and pads them), 3 and 2 GT boxes -- the same shapes at every
    step, so a captured step serves all of them.  ``frozen``: four images of exactly 3 x 128 x 160 (no resize, no padding: a rank
    that holds images [2r, 2r + 1] sees the same feature maps as the global batch), 3 GT boxes each."""
    rng = np.random.default_rng(7000 + 31 * step + (0 if kind == "live" else 500))
    sizes = [(120, 150), (140, 128)] if kind == "live" else [(128, 160)] * 4
    counts = (3, 2) if kind == "live" else (3, 3, 3, 3)
    images = [rng.random((3, h, w), dtype=np.float32) for h, w in sizes]
    targets = [gt_boxes(rng, T, h, w, num_classes=5, wh_lo=20.0, wh_hi=90.0) for (h, w), T in zip(sizes, counts)]
    return images, targets


def fingerprint(key: str, delta: np.ndarray, nsample: int = 16):
    "(norm, projection on a direction seeded by the key, positions, samples) of a tensor -- the trajectory fixture's record of one parameter's movement"
    import zlib
    flat = np.asarray(delta, dtype=np.float64).reshape(-1)
    seed = zlib.crc32(key.encode())
    r = np.random.default_rng(seed).standard_normal(flat.size)
    pos = np.random.default_rng(seed ^ 0x5A5A5A5A).integers(0, flat.size, nsample)
    return float(np.linalg.norm(flat)), float(flat @ r), pos.astype(np.int64), flat[pos]
