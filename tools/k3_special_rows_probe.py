#!/usr/bin/env python3
"""K3 at the BASELINE configs[4] shape (fp16, B = 8, A = 201 600, K = 90, 500 GT boxes per image) with the thresholds moved so that
the special rows change kind: what do the matched rows cost, what the ignored ones?  (round 4, one box: no special rows 124.5 us;
8 037 matched / image +14 us; + 10 634 ignored / image +32 us; 18 665 ignored / image +71 us.)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import bench_kernels as bk
from pytorch_retinanet_amd import ops
DEV = bk.DEV
B, A, K, T = 8, 201600, 90, 500
rng = np.random.default_rng(0)
anc = bk.anchors_for(800, 1344)
gt, gl, off = bk.gts(rng, B, T, 800, 1333)
g = torch.Generator(device=DEV).manual_seed(0)
cls = (torch.randn((B, A, K), device=DEV, generator=g) - 4.6).to(torch.float16)
box = (torch.randn((B, A, 4), device=DEV, generator=g) * 0.1).to(torch.float16)
p = ops.make_loss_params(0.25, 2.0, 0.1)
for fg, bgt in ((0.5, 0.4), (0.5, 0.49999), (0.9, 0.4), (0.99, 0.98)):
    m, nfg, sp = ops.iou_match(anc, gt, off, B, fg, bgt, want_special=True)
    ms = bk.timeit(lambda: ops.loss_fwd_bwd_levels([cls], [box], anc, gt, gl, off, m, nfg, p, True, special=sp), 30)
    print(f"fg {fg} bg {bgt}: matched/img {int((m >= 0).sum()) // B} ignored/img {int((m == -2).sum()) // B}  K3 {ms[0] * 1e3:.1f} us (min {ms[1] * 1e3:.1f})", flush=True)
