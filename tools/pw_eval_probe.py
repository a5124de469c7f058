#!/usr/bin/env python3
"""The inference 1x1 convolutions of the R101 trunk at predict's shape (16 x 1344 x 1344), isolated: pwconv.eval_conv1x1's GEMM with the folded
bias / identity / ReLU in its epilogue (csrc/pw.hip) against hipBLASLt's addmm (+ an add + ReLU pass for conv3).  Graph-replayed, us per call,
and the rate on the bytes every operand moves once (X + W + Y (+ R))."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                     # noqa: E402

import bench                                                     # noqa: E402
from pytorch_retinanet_amd import pwconv                         # noqa: E402

DEV = torch.device("cuda:0")


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    B = 16
    shapes = [("l1.conv1", 336, 256, 64, False), ("l1.conv3", 336, 64, 256, True), ("l2.conv1", 168, 512, 128, False), ("l2.conv3", 168, 128, 512, True),
              ("l3.conv1", 84, 1024, 256, False), ("l3.conv3", 84, 256, 1024, True), ("l4.conv1", 42, 2048, 512, False), ("l4.conv3", 42, 512, 2048, True)]
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for name, hw, cin, cout, res in shapes:
        if only and name not in only:
            continue
        x = torch.randn((B, cin, hw, hw), device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((cout, cin, 1, 1), device=DEV, generator=g) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        bias = torch.randn((cout,), device=DEV, generator=g)
        r = torch.randn((B, cout, hw, hw), device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if res else None
        M = B * hw * hw
        x2, w2, b16 = x.permute(0, 2, 3, 1).reshape(M, cin), w.view(cout, cin), bias.to(torch.bfloat16)
        epi = pwconv.bias_act_epilogue(bias, True, r)
        t_pw = bench.graph_replay_ms(lambda: pwconv.pw_forward(x, w, epi=epi)) * 1e3
        if res:
            r2 = r.permute(0, 2, 3, 1).reshape(M, cout)
            t_mm = bench.graph_replay_ms(lambda: torch.relu_(torch.addmm(b16, x2, w2.t()).add_(r2))) * 1e3
        else:
            t_mm = bench.graph_replay_ms(lambda: torch._addmm_activation(b16, x2, w2.t())) * 1e3
        mb = (M * (cin + cout * (2 if res else 1)) * 2 + cin * cout * 2) / 1e6
        gf = 2.0 * M * cin * cout / 1e9
        print(f"{name:9s} M={M:8d} {cin:4d}->{cout:4d} {'+res' if res else '    '} {gf:6.1f} GF {mb:7.1f} MB | hipblaslt{'+add+relu' if res else '         '} {t_mm:7.1f} us | "
              f"pw {t_pw:7.1f} us  {mb / t_pw:5.2f} TB/s  {gf / t_pw * 1e3:6.0f} TF/s", flush=True)
        del x, w, r


if __name__ == "__main__":
    main()
