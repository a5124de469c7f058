import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from pytorch_retinanet_amd.pool import FusedMaxPool2d
dev = "cuda"
x = torch.relu(torch.randn(8, 64, 400, 672, device=dev)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
pool = FusedMaxPool2d(3, 2, 1)
g = None
for name, fn in (("torch", lambda: F.max_pool2d(x, 3, 2, 1)), ("fused", lambda: pool(x)), ("torch", lambda: F.max_pool2d(x, 3, 2, 1)), ("fused", lambda: pool(x))):
    for _ in range(3):
        y = fn(); g = torch.randn_like(y) if g is None else g; y.backward(g); x.grad = None
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(10):
        e[0].record(); y = fn(); e[1].record(); y.backward(g); e[2].record(); torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2]); x.grad = None
    print(f"{name}: fwd {tf * 100:7.1f} us  bwd {tb * 100:7.1f} us")
