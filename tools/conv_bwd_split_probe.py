"""MIOpen's backward of the tower conv on the canvas, split into data-gradient and weight-gradient calls."""
import sys
import torch
sys.path.insert(0, ".")
from pytorch_retinanet_amd import tuning
tuning.enable_conv_autotune()
dev = torch.device("cuda")
for (H, W) in ((151, 168), (153, 170)):
    x = torch.randn(8, 256, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(256, 256, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = torch.randn(8, 256, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    def run(mask):
        return torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, mask)
    for name, mask in (("dgrad", [True, False, False]), ("wgrad", [False, True, False]), ("both", [True, True, False])):
        for _ in range(3): run(mask)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(mask)
        e1.record(); torch.cuda.synchronize()
        print(f"[8,256,{H},{W}] {name}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
    y = torch.nn.functional.conv2d(x, w, padding=1)
    for _ in range(3): torch.nn.functional.conv2d(x, w, padding=1)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.nn.functional.conv2d(x, w, padding=1)
    e1.record(); torch.cuda.synchronize()
    print(f"[8,256,{H},{W}] fwd  : {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
