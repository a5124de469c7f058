#!/usr/bin/env python3
"""K2 timing scan over T (GT boxes per image) and GT size: separates the load/store skeleton (T = 0) from the per-pair cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import synth
from pytorch_retinanet_amd import ops
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_kernels import anchors_for, timeit, DEV

B, A = 8, 201600
anc = anchors_for(800, 1344)
for wh in ((16.0, 316.0), (4.0, 12.0)):
    for T in (0, 1, 2, 4, 8, 16, 32, 33, 64, 128, 256, 500, 1000):
        rng = np.random.default_rng(T)
        b = [synth.gt_boxes(rng, T, 800, 1333, wh_lo=wh[0], wh_hi=wh[1])[0] for _ in range(B)]
        gt = torch.from_numpy(np.concatenate(b)).to(DEV) if T else torch.zeros((0, 4), device=DEV)
        off = ops.gt_offsets([T] * B, DEV)
        med, mn, mean = timeit(lambda: ops.iou_match(anc, gt, off, B, 0.5, 0.4), 40)
        print(f"wh={wh} T={T:5d}: median {med*1e3:8.1f} us  min {mn*1e3:8.1f} us   {B*A*T/(med*1e-3)/1e12 if T else 0:6.2f} Tpairs/s", flush=True)
