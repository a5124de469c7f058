#!/usr/bin/env python3
"""K3 timing with and without special rows: stream kernel, repair kernel and the whole call (events from the library)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import synth
from pytorch_retinanet_amd import ops
from bench_kernels import anchors_for, timeit, DEV

B, A, K = 8, 201600, 90
anc = anchors_for(800, 1344)
g = torch.Generator(device=DEV).manual_seed(0)
cls = (torch.randn((B, A, K), device=DEV, generator=g) - 4.6).to(torch.bfloat16)
box = (torch.randn((B, A, 4), device=DEV, generator=g) * 0.1).to(torch.bfloat16)
p = ops.make_loss_params(0.25, 2.0, 0.1)
for name, T, wh in (("no special rows (1 px GT boxes: every anchor background)", 8, (1.0, 1.5)), ("T=8", 8, (16.0, 316.0)), ("T=64", 64, (16.0, 316.0)),
                    ("T=500", 500, (16.0, 316.0))):
    rng = np.random.default_rng(T)
    bs, ls = zip(*[synth.gt_boxes(rng, T, 800, 1333, wh_lo=wh[0], wh_hi=wh[1]) for _ in range(B)])
    gt = torch.from_numpy(np.concatenate(bs)).to(DEV); gl = torch.from_numpy(np.concatenate(ls)).to(DEV)
    off = ops.gt_offsets([T] * B, DEV)
    m, nfg = ops.iou_match(anc, gt, off, B, 0.5, 0.4)
    ops.enable_timing(True)
    for _ in range(25):
        ops.loss_fwd_bwd_levels([cls], [box], anc, gt, gl, off, m, nfg, p, True)
    torch.cuda.synchronize()
    te = ops.timing_events()
    ops.enable_timing(False)
    ts = np.array([a.elapsed_time(b) for a, b in te["loss_stream_kernel"][5:]]) * 1e3
    tc = np.array([a.elapsed_time(b) for a, b in te["loss_fwd_bwd"][5:]]) * 1e3
    print(f"{name:60s} fg/img {int(nfg.float().mean()):6d} ignored/img {int((m == -2).sum()) // B:6d}: kernel {np.median(ts):6.1f} us (min {ts.min():6.1f})  "
          f"call incl. finalize {np.median(tc):6.1f} us", flush=True)
