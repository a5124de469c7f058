#!/usr/bin/env python3
"""pw_gemm (csrc/pw.hip) against hipBLASLt / MIOpen at the 1x1 shapes of the R50 trunk, isolated, bf16: plain forward, forward with the
BatchNorm-apply prologue + statistics epilogue.  (What decides whether `pwconv._BottleneckFn` can take layer3 / layer4.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                     # noqa: E402
import torch.nn.functional as F                                  # noqa: E402

from pytorch_retinanet_amd import pwconv                         # noqa: E402

DEV = torch.device("cuda:0")


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in ev)
    return ts[len(ts) // 2] * 1e3


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    shapes = [("l1.conv1", 8, 200, 336, 256, 64), ("l1.conv3", 8, 200, 336, 64, 256), ("l2.conv1", 8, 100, 168, 512, 128), ("l2.conv3", 8, 100, 168, 128, 512),
              ("l3.conv1", 8, 50, 84, 1024, 256), ("l3.conv3", 8, 50, 84, 256, 1024), ("l4.conv1", 8, 25, 42, 2048, 512), ("l4.conv3", 8, 25, 42, 512, 2048)]
    for name, N, H, W, cin, cout in shapes:
        x = (torch.randn((N, cin, H, W), device=DEV, generator=g)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((cout, cin, 1, 1), device=DEV, generator=g) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        M = N * H * W
        x2, w2 = x.permute(0, 2, 3, 1).reshape(M, cin), w.view(cout, cin)
        coef = torch.cat([torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.3])
        epi, partial, nb = pwconv.stats_epilogue(M, cout, DEV)
        t_mm = timeit(lambda: torch.mm(x2, w2.t()))
        t_conv = timeit(lambda: F.conv2d(x, w))
        t_pw = timeit(lambda: pwconv.pw_forward(x, w))
        t_pw_stats = timeit(lambda: pwconv.pw_forward(x, w, epi=epi))
        t_pw_full = timeit(lambda: pwconv.pw_forward(x, w, pro=pwconv.affine_relu(coef), epi=epi))
        gf = 2.0 * M * cin * cout / 1e9
        mb = (M * (cin + cout) * 2) / 1e6
        print(f"{name:9s} M={M:7d} {cin:4d}->{cout:4d}  {gf:6.1f} GF {mb:6.1f} MB | mm {t_mm:6.1f}  miopen {t_conv:6.1f}  pw {t_pw:6.1f}  pw+stats {t_pw_stats:6.1f}  "
              f"pw+bn+stats {t_pw_full:6.1f} us | pw {gf / t_pw * 1e3:6.0f} TF/s  {mb / t_pw:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
