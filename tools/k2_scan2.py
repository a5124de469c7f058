#!/usr/bin/env python3
"""K2 load-balance probe: time per pyramid level (are the big-anchor levels the tail of the launch?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import synth
from pytorch_retinanet_amd import ops
from bench_kernels import anchors_for, timeit, DEV

B = 8
anc = anchors_for(800, 1344)
T = 500
for wh in ((16.0, 316.0), (4.0, 12.0)):
    rng = np.random.default_rng(T)
    b = [synth.gt_boxes(rng, T, 800, 1333, wh_lo=wh[0], wh_hi=wh[1])[0] for _ in range(B)]
    gt = torch.from_numpy(np.concatenate(b)).to(DEV)
    off = ops.gt_offsets([T] * B, DEV)
    for name, sl in [("all", slice(0, 201600)), ("P3", slice(0, 151200)), ("P4", slice(151200, 189000)), ("P5", slice(189000, 198450)),
                     ("P6+P7", slice(198450, 201600)), ("P3 x1.33 (same A as all)", None)]:
        a = anc[sl].contiguous() if sl is not None else torch.cat([anc[:151200], anc[:50400]]).contiguous()
        med, mn, mean = timeit(lambda: ops.iou_match(a, gt, off, B, 0.5, 0.4), 20)
        A = a.shape[0]
        print(f"wh={wh} {name:28s} A={A:7d}: median {med*1e3:8.1f} us   {B*A*T/(med*1e-3)/1e12:6.2f} Tpairs/s", flush=True)
