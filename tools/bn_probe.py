"""Time BatchNorm2d fwd+bwd (train mode, bf16, channels_last) through MIOpen vs torch's native kernels."""
import sys, time
import torch
dev = torch.device("cuda:0")
def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for (N, C, H, W) in [(8, 64, 400, 672), (8, 256, 200, 336), (8, 512, 100, 168), (8, 1024, 50, 84), (8, 2048, 25, 42)]:
    bn = torch.nn.BatchNorm2d(C).to(dev).to(memory_format=torch.channels_last).train()
    x = torch.randn(N, C, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randn_like(x)
    def step(native):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if native:
                with torch.backends.cudnn.flags(enabled=False):
                    y = torch.relu(bn(x))
            else:
                y = torch.relu(bn(x))
        y.backward(g)
        x.grad = None; bn.weight.grad = None; bn.bias.grad = None
    mb = x.numel() * 2 / 1e6
    print(f"[{N},{C},{H},{W}] {mb:7.1f} MB  miopen {bench(lambda: step(False)):7.3f} ms   native {bench(lambda: step(True)):7.3f} ms")
