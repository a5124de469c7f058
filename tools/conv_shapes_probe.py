#!/usr/bin/env python3
"""Per-shape MIOpen efficiency of every convolution in the headline model (R50-FPN, B=8, 800x1344, bf16,
channels-last): forward and backward time of each distinct (Cin, Cout, k, stride, pad, H, W), with how many
times the step runs it.  Finds shapes where MIOpen is far off the MFMA roofline (like the 810-channel cls conv).
usage: python tools/conv_shapes_probe.py [B]"""
import sys
from collections import OrderedDict

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
import pytorch_retinanet_amd as P  # noqa: E402
from pytorch_retinanet_amd import tuning  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
tuning.use_shipped_miopen_db()
tuning.enable_conv_autotune()
net = P.Retinanet(num_classes=90, backbone_kind="resnet50", pretrained=False, min_size=800, max_size=1333)
net = net.to(dev).to(memory_format=torch.channels_last).train()
shapes = OrderedDict()
orig = F.conv2d


def spy(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    key = (tuple(x.shape), tuple(w.shape), tuple(stride) if isinstance(stride, (tuple, list)) else (stride, stride),
           tuple(padding) if isinstance(padding, (tuple, list)) else (padding, padding))
    shapes[key] = shapes.get(key, 0) + 1
    return orig(x, w, b, stride, padding, dilation, groups)


F.conv2d = spy
torch.nn.functional.conv2d = spy
g = torch.Generator(device=dev).manual_seed(0)
images = [torch.rand((3, 800, 1333), device=dev, generator=g) for _ in range(B)]
targets = [{"boxes": torch.tensor([[100.0, 100.0, 400.0, 300.0]], device=dev), "labels": torch.tensor([1], device=dev)} for _ in range(B)]
with torch.autocast("cuda", dtype=torch.bfloat16):
    net(images, targets)
F.conv2d = orig
torch.nn.functional.conv2d = orig

rows = []
for (xs, ws, st, pd), count in shapes.items():
    x = torch.randn(xs, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(ws, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    for _ in range(2):
        y = orig(x, w, None, st, pd)
        y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    gy = torch.ones_like(y)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    reps = 5
    for _ in range(reps):
        e[0].record(); y = orig(x, w, None, st, pd); e[1].record(); y.backward(gy); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    tf /= reps; tb /= reps
    flops = 2.0 * y.numel() * ws[1] * ws[2] * ws[3]
    rows.append((count * (tf + tb), count, xs, ws, st, tf, tb, flops / tf / 1e9, 2 * flops / tb / 1e9))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print(f"{len(rows)} distinct conv shapes, {sum(r[1] for r in rows)} calls, isolated fwd+bwd total {tot:.2f} ms")
for total, count, xs, ws, st, tf, tb, ef, eb in rows:
    print(f"{total:7.3f} ms x{count:<2d} x{list(xs)} w{list(ws)} s{st[0]}  fwd {tf * 1e3:7.1f} us {ef:6.0f} TF/s  bwd {tb * 1e3:7.1f} us {eb:6.0f} TF/s")
