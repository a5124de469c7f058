"""Phase times of pw_gemm_kernel's row tiles from the s_memrealtime stamps of the scratch build (tools/probes/make_stamped_pw.sh; install it with
tools/ab_libs_run.sh "python tools/probes/pw_stamps_probe.py l3.conv3" ab_libs/pwstamps.so).  s_memrealtime ticks at 100 MHz."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np                                               # noqa: E402
import torch                                                     # noqa: E402

from pytorch_retinanet_amd import pwconv                         # noqa: E402
import pytorch_retinanet_amd._lib as L                           # noqa: E402

DEV = torch.device("cuda:0")
so = ctypes.CDLL(os.path.join(os.path.dirname(L.__file__), "libretinanet_hip.so"))
so.rn_debug_pw_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
SHAPES = {"l1.conv3": (336, 64, 256, True), "l2.conv3": (168, 128, 512, True), "l3.conv1": (84, 1024, 256, False), "l3.conv3": (84, 256, 1024, True),
          "l4.conv3": (42, 512, 2048, True)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "l3.conv3"
    hw, cin, cout, res = SHAPES[name]
    B = 16
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn((B, cin, hw, hw), device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((cout, cin, 1, 1), device=DEV, generator=g) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bias = torch.randn((cout,), device=DEV, generator=g)
    r = torch.randn((B, cout, hw, hw), device=DEV, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if res else None
    epi = pwconv.bias_act_epilogue(bias, True, r)
    cap = 8192
    stamps = torch.zeros((cap, 4, 12), dtype=torch.int64, device=DEV)
    for _ in range(3):
        pwconv.pw_forward(x, w, epi=epi)
    torch.cuda.synchronize()
    so.rn_debug_pw_stamps(ctypes.c_void_p(stamps.data_ptr()), cap)
    pwconv.pw_forward(x, w, epi=epi)
    torch.cuda.synchronize()
    so.rn_debug_pw_stamps(ctypes.c_void_p(0), 0)
    s = stamps.cpu().numpy().reshape(-1, 12)
    s = s[s[:, 5] != 0]
    us = lambda v: v / 100.0
    t0 = s[:, 0].min()
    print(f"{name}: {len(s)} tiles stamped; kernel span {us(s[:, 5].max() - t0):.1f} us")
    names = ["first commit + barrier", "K loop", "acc -> LDS + barrier", "epilogue rows (LDS -> +bias/resid -> store issue)", "end barrier"]
    for i, nm in enumerate(names):
        d = us(s[:, i + 1] - s[:, i])
        print(f"  {nm:52s} mean {d.mean():6.2f}  p10 {np.percentile(d, 10):6.2f}  p50 {np.percentile(d, 50):6.2f}  p90 {np.percentile(d, 90):6.2f} us")
    tot = us(s[:, 5] - s[:, 0])
    print(f"  {'whole tile':52s} mean {tot.mean():6.2f}  p50 {np.percentile(tot, 50):6.2f} us")
    kt = cin // 64
    print(f"  the K loop per K-tile ({kt} K-tiles):")
    for i, nm in enumerate(["request (addresses + global loads)", "fragment reads + MFMAs", "commit (wait loads, ds_write, wait LDS)", "barrier"]):
        d = us(s[:, 6 + i]) / kt
        print(f"     {nm:49s} mean {d.mean():6.3f}  p50 {np.percentile(d, 50):6.3f}  p90 {np.percentile(d, 90):6.3f} us")
    mid = (s[:, 0].min() + s[:, 5].max()) // 2
    print(f"  tiles in flight at mid-kernel: {int(((s[:, 0] <= mid) & (s[:, 5] >= mid)).sum())}")


if __name__ == "__main__":
    main()
