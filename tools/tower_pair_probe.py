"""tower_conv x2 vs tower_conv_pair (batched launch) on the R50 canvas, forward and backward, bf16."""
import sys
import torch
sys.path.insert(0, ".")
from pytorch_retinanet_amd import biasact, tuning
tuning.enable_conv_autotune()
dev = torch.device("cuda")
shapes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
cv = biasact.Canvas(shapes, dev, pad=1)
N, C = 8, 256
mask2d = cv.mask.view(cv.H, cv.W)
def mk():
    return (torch.randn(N, C, cv.H, cv.W, device=dev) * mask2d[None, None]).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
x0, x1 = mk(), mk()
w0 = (torch.randn(C, C, 3, 3, device=dev) * 0.03).contiguous(memory_format=torch.channels_last).requires_grad_(True)
w1 = (torch.randn(C, C, 3, 3, device=dev) * 0.03).contiguous(memory_format=torch.channels_last).requires_grad_(True)
b0 = torch.zeros(C, device=dev, requires_grad=True); b1 = torch.zeros(C, device=dev, requires_grad=True)
g0 = torch.randn_like(x0); g1 = torch.randn_like(x1)
def single():
    y0 = biasact.tower_conv(x0, w0, b0, cv.mask); y1 = biasact.tower_conv(x1, w1, b1, cv.mask)
    return y0, y1
def pair():
    return biasact.tower_conv_pair(x0, x1, w0, w1, b0, b1, cv.mask)
for name, fn in (("2 x tower_conv", single), ("tower_conv_pair", pair), ("2 x tower_conv", single), ("tower_conv_pair", pair)):
    for _ in range(3):
        y0, y1 = fn(); torch.autograd.backward([y0, y1], [g0, g1])
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(10):
        e[0].record(); y0, y1 = fn(); e[1].record(); torch.autograd.backward([y0, y1], [g0, g1]); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print(f"{name:18s} fwd {tf * 100:8.1f} us   bwd {tb * 100:8.1f} us")
