#!/usr/bin/env python3
"""What the vendor GEMM (hipBLASLt through torch.mm, bf16) reaches on this box: square GEMMs, and the head's 3x3 convolutions written as the
explicit GEMMs they are (positions x 2304 -> 256 / 810 / 36, the im2col matrix given for free) -- the yardstick for `conv_mfma` in the bench line.
Graph-replayed, TFLOP/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                     # noqa: E402

import bench                                                     # noqa: E402

DEV = torch.device("cuda:0")


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    shapes = [("square 4096", 4096, 4096, 4096), ("square 8192", 8192, 8192, 8192), ("square 16384 x 8192 x 8192", 16384, 8192, 8192),
              ("tower conv as a GEMM (8 img)", 179200, 2304, 256), ("tower pair as ONE GEMM (2 x 179200 rows)", 358400, 2304, 256),
              ("class-output conv as a GEMM", 179200, 2304, 810), ("box-output conv as a GEMM", 179200, 2304, 36),
              ("tower conv as a GEMM (16 img @1344)", 601696, 2304, 256)]
    for name, M, K, N in shapes:
        a = torch.randn((M, K), device=DEV, generator=g).to(torch.bfloat16)
        b = (torch.randn((N, K), device=DEV, generator=g) * 0.05).to(torch.bfloat16)
        t = bench.graph_replay_ms(lambda: torch.mm(a, b.t()), reps=10, rounds=3)
        print(f"{name:45s} M={M:7d} K={K:5d} N={N:5d}  {t * 1e3:8.1f} us  {2.0 * M * K * N / t / 1e9:7.0f} TFLOP/s", flush=True)
        del a, b


if __name__ == "__main__":
    main()
