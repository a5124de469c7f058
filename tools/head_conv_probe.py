#!/usr/bin/env python3
"""How well does MIOpen run the head-tower convolutions (3x3, 256->256, bf16, channels-last) at each
pyramid level of the R50 config, and on one packed canvas holding all five levels?
usage: python tools/head_conv_probe.py [B]"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from pytorch_retinanet_amd import tuning  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
tuning.enable_conv_autotune()
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
CANVAS = (151, 168)


def bench(h, w, cout=256, reps=20):
    x = torch.randn(B, 256, h, w, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = torch.randn(cout, 256, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    for _ in range(3):
        y = F.conv2d(x, wt, padding=1)
        y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    g = torch.ones_like(y)
    for _ in range(reps):
        e[0].record()
        y = F.conv2d(x, wt, padding=1)
        e[1].record()
        y.backward(g)
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    flops = 2.0 * B * h * w * 256 * cout * 9
    return tf / reps, tb / reps, flops


if "--cout" in sys.argv:          # output-channel sweep of the final convs at the two big levels
    for cout in (36, 64, 810, 816, 832, 864, 896, 1024, 1152):
        for (h, w) in ((100, 168), (50, 84)):
            f, b, fl = bench(h, w, cout)
            print(f"cout {cout:5d} {h:4d}x{w:<4d} fwd {f * 1e3:8.1f} us ({fl / f / 1e9:7.1f} TFLOP/s)   bwd {b * 1e3:8.1f} us ({2 * fl / b / 1e9:7.1f} TFLOP/s)")
    sys.exit(0)

tot_f = tot_b = 0.0
for cout, name in ((256, "tower 256->256"), (810, "cls out 256->810")):
    tot_f = tot_b = 0.0
    print(f"== {name}, B={B}")
    for (h, w) in LEVELS:
        f, b, fl = bench(h, w, cout)
        tot_f += f; tot_b += b
        print(f"  {h:4d}x{w:<4d} fwd {f * 1e3:8.1f} us ({fl / f / 1e9:7.1f} TFLOP/s)   bwd {b * 1e3:8.1f} us ({2 * fl / b / 1e9:7.1f} TFLOP/s)")
    print(f"  sum of 5 levels: fwd {tot_f * 1e3:8.1f} us  bwd {tot_b * 1e3:8.1f} us")
    f, b, fl = bench(*CANVAS, cout)
    print(f"  canvas {CANVAS[0]}x{CANVAS[1]}: fwd {f * 1e3:8.1f} us ({fl / f / 1e9:7.1f} TFLOP/s)   bwd {b * 1e3:8.1f} us ({2 * fl / b / 1e9:7.1f} TFLOP/s)")
