"""Effective HBM bandwidth of the fused BN kernels per ResNet-50 layer shape (bf16, channels-last, train mode).
fwd = 3 tensor passes (+1 with residual), bwd = 5 passes (no residual: mask recomputed from x) / 8 (residual)."""
import sys
import torch
sys.path.insert(0, ".")
from pytorch_retinanet_amd.norm import FusedBatchNorm2d
dev = torch.device("cuda:0")
SHAPES = [(8, 64, 400, 672), (8, 64, 200, 336), (8, 256, 200, 336), (8, 128, 100, 168), (8, 512, 100, 168), (8, 256, 50, 84),
          (8, 1024, 50, 84), (8, 512, 25, 42), (8, 2048, 25, 42)]
for res in (False, True):
    for (N, C, H, W) in SHAPES:
        bn = FusedBatchNorm2d(C).to(dev).train()
        x = torch.randn(N, C, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        r = torch.randn_like(x).requires_grad_(True) if res else None
        g = torch.randn_like(x)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        reps = 20
        for i in range(reps + 3):
            e[0].record(); y = bn(x, relu=True, residual=r); e[1].record(); y.backward(g); e[2].record()
            torch.cuda.synchronize()
            if i >= 3:
                tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
            x.grad = None; bn.weight.grad = None; bn.bias.grad = None
            if r is not None: r.grad = None
        mb = x.numel() * 2 / 1e6
        pf, pb = (4, 8) if res else (3, 5)
        tf, tb = tf / reps, tb / reps
        print(f"res={int(res)} [{N},{C},{H},{W}] {mb:6.1f} MB  fwd {tf * 1e3:7.1f} us {pf * mb / tf / 1e3:5.2f} TB/s   bwd {tb * 1e3:7.1f} us {pb * mb / tb / 1e3:5.2f} TB/s")
