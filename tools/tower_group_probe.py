"""Do the two head towers run faster as ONE grouped convolution per layer (groups=2, 512 channels) than as two
256-channel convolutions?  Canvas shape of the R50 config, bf16, channels-last, MIOpen find."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from pytorch_retinanet_amd import tuning
tuning.enable_conv_autotune()
dev = torch.device("cuda")
B, H, W = 8, 151, 168


def bench(cin, cout, groups, reps=20):
    x = torch.randn(B, cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(cout, cin // groups, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    for _ in range(3):
        y = F.conv2d(x, w, padding=1, groups=groups); y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    g = torch.randn_like(y)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(reps):
        e[0].record(); y = F.conv2d(x, w, padding=1, groups=groups); e[1].record(); y.backward(g); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    return tf / reps * 1e3, tb / reps * 1e3


for name, (cin, cout, groups, mult) in {"256->256 (x2 launches)": (256, 256, 1, 2), "512->512 groups=2": (512, 512, 2, 1),
                                        "256->512 (first layer, shared input)": (256, 512, 1, 1)}.items():
    f, b = bench(cin, cout, groups)
    print(f"{name:40s} fwd {f * mult:8.1f} us  bwd {b * mult:8.1f} us  total {mult * (f + b):8.1f} us")
