import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, synth
from pytorch_retinanet_amd import ops
from tools.bench_kernels import anchors_for, gts, DEV, timeit
rng = np.random.default_rng(0)
B, A, K, T = 8, 201600, 90, 8
anc = anchors_for(800, 1344)
gt, gl, off = gts(rng, B, T, 800, 1333)
m, nfg = ops.iou_match(anc, gt, off, B, 0.5, 0.4)
print("fg", int((m >= 0).sum()), "ignored", int((m == -2).sum()), "bg", int((m == -1).sum()), "nfg", nfg.tolist())
