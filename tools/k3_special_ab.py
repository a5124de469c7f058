#!/usr/bin/env python3
"""Same-box A/B of K3 with / without the special-row words (rn_iou_match_special): per-level tensors at the train shape,
events right around the streaming kernel; 'cold' = 1 GiB fill between launches, 'warm' = back to back.
    python tools/k3_special_ab.py [T] [dtype bf16|f16]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
from pytorch_retinanet_amd import ops  # noqa: E402
from pytorch_retinanet_amd.anchors import AnchorGenerator  # noqa: E402

DEV = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dt = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
B, K = 8, 90
shapes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
anc = ops.anchors_emit(synth.levels_for(800, 1344), list(AnchorGenerator().to(DEV).cell_anchors), 0.0)
g = torch.Generator(device=DEV).manual_seed(2)
cls = [(torch.randn((B, h * w * 9, K), device=DEV, generator=g) - 4.6).to(dt) for h, w in shapes]
box = [(torch.randn((B, h * w * 9, 4), device=DEV, generator=g) * 0.1).to(dt) for h, w in shapes]
rng = np.random.default_rng(0)
gtb, gtl = zip(*[synth.gt_boxes(rng, T, 800, 1333) for _ in range(B)])
gt_boxes, gt_labels = torch.from_numpy(np.concatenate(gtb)).to(DEV), torch.from_numpy(np.concatenate(gtl)).to(DEV)
off = ops.gt_offsets([T] * B, DEV)
m, nfg, sp = ops.iou_match(anc, gt_boxes, off, B, 0.5, 0.4, want_special=True)
p = ops.make_loss_params(0.25, 2.0, 0.1)
evict = torch.empty((1 << 30,), dtype=torch.uint8, device=DEV)
A = sum(h * w * 9 for h, w in shapes)
nbytes = B * (2 * A * K * 2 + 2 * A * 4 * 2 + A * 8 + T * 24)
res = {}
for rnd in range(3):
    for name, special in (("matches", None), ("special", sp)):
        for mode in ("cold", "warm"):
            ops.enable_timing(True)
            for _ in range(12):
                if mode == "cold":
                    evict.fill_(1)
                ops.loss_fwd_bwd_levels(cls, box, anc, gt_boxes, gt_labels, off, m, nfg, p, special=special)
            torch.cuda.synchronize()
            ev = ops.timing_events()["loss_stream_kernel"][2:]
            ops.enable_timing(False)
            res.setdefault((name, mode), []).append(float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3)
for k, v in res.items():
    us = float(np.median(v))
    print(json.dumps({"variant": k[0], "mode": k[1], "T": T, "dtype": str(dt), "us": round(us, 1), "rounds": [round(x, 1) for x in v],
                      "frac_of_8TBps": round(nbytes / (us * 1e-6) / 8e12, 4)}))
