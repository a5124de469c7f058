#!/usr/bin/env python3
"""Isolated timings of the dense-head HIP kernels at BASELINE.json's shapes (events on the launch stream).

    python tools/bench_kernels.py [k2] [k3] [k3f32] [detect] [detect_stress] [--reps N]

Prints one JSON line per kernel: average launch time, algorithmic bytes, achieved GB/s and
the fraction of the 8 TB/s HBM peak; pairs/s for the matcher.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth  # noqa: E402
from pytorch_retinanet_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
PEAK = 8000.0


def timeit(fn, reps, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = np.array([a.elapsed_time(b) for a, b in evs])
    return float(np.median(ts)), float(ts.min()), float(ts.mean())


def anchors_for(h, w):
    from pytorch_retinanet_amd.anchors import AnchorGenerator
    ag = AnchorGenerator().to(DEV)
    cells = list(ag.cell_anchors)
    return ops.anchors_emit(synth.levels_for(h, w), cells, 0.0)


def gts(rng, B, T, h, w):
    b, l = zip(*[synth.gt_boxes(rng, T, h, w) for _ in range(B)])
    gt = torch.from_numpy(np.concatenate(b)).to(DEV)
    gl = torch.from_numpy(np.concatenate(l)).to(DEV)
    off = ops.gt_offsets([T] * B, DEV)
    return gt, gl, off


def report(name, ms, nbytes, extra=None):
    med, mn, mean = ms
    d = {"kernel": name, "ms_median": round(med, 4), "ms_min": round(mn, 4), "ms_mean": round(mean, 4),
         "algorithmic_MB": round(nbytes / 1e6, 2), "GBps": round(nbytes / (med * 1e-3) / 1e9, 1),
         "frac_of_8TBps": round(nbytes / (med * 1e-3) / 1e9 / PEAK, 4)}
    if extra:
        d.update(extra)
    print(json.dumps(d), flush=True)


def bench_k2(reps, T):
    B, A = 8, 201600
    rng = np.random.default_rng(0)
    anc = anchors_for(800, 1344)
    gt, gl, off = gts(rng, B, T, 800, 1333)
    ms = timeit(lambda: ops.iou_match(anc, gt, off, B, 0.5, 0.4), reps)
    nbytes = B * (A * 16 + T * 16 + A * 8)                  # per image, anchors counted for every image
    unique = A * 16 + B * (T * 16 + A * 8)                  # one anchor set shared by the batch: read once
    report(f"K2 iou_match B={B} A={A} T={T}", ms, nbytes, {"Gpairs_per_s": round(B * A * T / (ms[0] * 1e-3) / 1e9, 2),
                                                           "unique_MB": round(unique / 1e6, 2),
                                                           "unique_GBps": round(unique / (ms[0] * 1e-3) / 1e9, 1)})


def bench_k3(reps, dtype, B=8, want_grad=True, T=8):
    A, K = 201600, 90
    rng = np.random.default_rng(0)
    anc = anchors_for(800, 1344)
    gt, gl, off = gts(rng, B, T, 800, 1333)
    g = torch.Generator(device=DEV).manual_seed(0)
    cls = (torch.randn((B, A, K), device=DEV, generator=g) - 4.6).to(dtype)
    box = (torch.randn((B, A, 4), device=DEV, generator=g) * 0.1).to(dtype)
    m, nfg, sp = ops.iou_match(anc, gt, off, B, 0.5, 0.4, want_special=True)
    p = ops.make_loss_params(0.25, 2.0, 0.1)
    # (the model's path: special-row words from K2, `matches` read only at flagged rows)
    # (... in the form the model picks for this GT count: losses.k3_form -- chunk by chunk at the train shape, the compact list from 32 GT
    # boxes per image on -- and with the in-kernel finalize, as RetinaNetLosses calls it)
    from pytorch_retinanet_amd import losses as L
    ms = timeit(lambda: ops.loss_fwd_bwd_levels([cls], [box], anc, gt, gl, off, m, nfg, p, want_grad, special=sp,
                                                in_kernel_finalize=L.IN_KERNEL_FINALIZE, form=L.k3_form(B * T, B)), reps)
    s = cls.element_size()
    nbytes = B * ((2 if want_grad else 1) * (A * K * s + A * 4 * s) + A * 8 + T * 24)
    report(f"K3 loss_{'fwd_bwd' if want_grad else 'fwd'} {str(dtype).split('.')[-1]} B={B} A={A} K={K} T={T}", ms, nbytes,
           {"num_fg_per_image": int(nfg.float().mean()), "ignored_rows_per_image": int((m == -2).sum() // B)})


def bench_k3_fused(reps, dtype, B=8, T=8):
    "K2 inside K3 (rn_loss_match_fwd_bwd_levels, one launch) next to K2 + K3 (two launches + the num_fg memset) on the same data."
    A, K = 201600, 90
    rng = np.random.default_rng(0)
    anc = anchors_for(800, 1344)
    gt, gl, off = gts(rng, B, T, 800, 1333)
    g = torch.Generator(device=DEV).manual_seed(0)
    cls = (torch.randn((B, A, K), device=DEV, generator=g) - 4.6).to(dtype)
    box = (torch.randn((B, A, 4), device=DEV, generator=g) * 0.1).to(dtype)
    p = ops.make_loss_params(0.25, 2.0, 0.1)
    s = cls.element_size()
    nbytes = B * (2 * (A * K * s + A * 4 * s) + A * 8 + T * 24)

    def two():
        m, nfg, sp = ops.iou_match(anc, gt, off, B, 0.5, 0.4, want_special=True)
        return ops.loss_fwd_bwd_levels([cls], [box], anc, gt, gl, off, m, nfg, p, True, special=sp)
    ms2 = timeit(two, reps)
    report(f"K2 + K3 two launches {str(dtype).split('.')[-1]} B={B} A={A} K={K} T={T}", ms2, nbytes)
    ms1 = timeit(lambda: ops.loss_match_fwd_bwd_levels([cls], [box], anc, gt, gl, off, T, 0.5, 0.4, p, True), reps)
    report(f"K2 inside K3 one launch {str(dtype).split('.')[-1]} B={B} A={A} K={K} T={T}", ms1, nbytes)
    msm = timeit(lambda: ops.loss_match_fwd_bwd_levels([cls], [box], anc, gt, gl, off, T, 0.5, 0.4, p, True, want_matches=True), reps)
    report(f"K2 inside K3 one launch + matches written {str(dtype).split('.')[-1]} T={T}", msm, nbytes)


def bench_detect(reps, mean, std, tag, B=16):
    A, K = 338454, 90
    anc = anchors_for(1344, 1344)
    g = torch.Generator(device=DEV).manual_seed(1)
    cls = (torch.randn((B, A, K), device=DEV, generator=g) * std + mean).to(torch.float16)
    box = (torch.randn((B, A, 4), device=DEV, generator=g) * 0.1).to(torch.float16)
    hw = [(1333, 1333)] * B
    ncand = int((torch.sigmoid(cls.float()) > 0.05).sum())
    t0 = time.perf_counter()
    ops.enable_timing(True)
    for _ in range(reps):
        ops.detect(cls, box, anc, hw, 0.05, 1e-2, 0.5, 100, max_candidates=max(1 << 18, 2 * ncand // B))
    torch.cuda.synchronize()
    ev = ops.timing_events()["detect"]
    ops.enable_timing(False)
    ts = np.array([a.elapsed_time(b) for a, b in ev])[1:]
    nbytes = B * (A * K * 2 + A * 4 * 2 + A * 16)
    report(f"K4-K7 rn_detect {tag} fp16 B={B} A={A} K={K}", (float(np.median(ts)), float(ts.min()), float(ts.mean())), nbytes,
           {"candidates_per_image": ncand // B, "wall_ms_per_call_incl_sync": round((time.perf_counter() - t0) / reps * 1e3, 3)})
    # the same data as five per-level tensors (what the conv stack produces) -> rn_detect_levels, and the cat it avoids
    counts = [254016, 63504, 15876, 3969, 1089]
    cl, bl, o = [], [], 0
    for n in counts:
        cl.append(cls[:, o:o + n].contiguous()); bl.append(box[:, o:o + n].contiguous()); o += n
    ops.enable_timing(True)
    for _ in range(reps):
        ops.detect_levels(cl, bl, anc, hw, 0.05, 1e-2, 0.5, 100, max_candidates=max(1 << 18, 2 * ncand // B))
    torch.cuda.synchronize()
    ev = ops.timing_events()["detect"]
    ops.enable_timing(False)
    ts = np.array([a.elapsed_time(b) for a, b in ev])[1:]
    report(f"K4-K7 rn_detect_levels {tag} fp16 B={B} A={A} K={K} L=5", (float(np.median(ts)), float(ts.min()), float(ts.mean())), nbytes)
    cat_ms = timeit(lambda: (torch.cat(cl, dim=1), torch.cat(bl, dim=1)), reps)
    report(f"torch.cat of the 5 levels (avoided) B={B}", cat_ms, 2 * B * (A * K * 2 + A * 4 * 2))


def bench_narrow3x3(reps, N=8, H=200, W=336):
    "conv2 of a layer1 bottleneck (64 -> 64, 3x3, stride 1): csrc/narrow3x3.hip next to the kernel MIOpen picks, same tensors."
    import torch.nn.functional as F
    from pytorch_retinanet_amd import biasact
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((N, 64, H, W), generator=g).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((64, 64, 3, 3), generator=g) * 0.05).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = biasact.conv3x3_narrow_forward(x, w)
    ref = F.conv2d(x.float(), w.float(), None, 1, 1)
    err = float((y.float() - ref).abs().max())
    err_lib = float((F.conv2d(x, w, None, 1, 1).float() - ref).abs().max())
    flop = 2.0 * N * H * W * 64 * 64 * 9
    nbytes = 2 * N * H * W * 64 * 2 + 64 * 64 * 9 * 2
    for name, fn in (("rn_conv3x3_narrow_forward", lambda: biasact.conv3x3_narrow_forward(x, w)), ("MIOpen F.conv2d", lambda: F.conv2d(x, w, None, 1, 1))):
        med, mn, mean = timeit(fn, reps)
        print(json.dumps({"kernel": f"{name} {N}x{H}x{W}x64 bf16", "ms_median": round(med, 4), "ms_min": round(mn, 4), "TFLOP/s": round(flop / med / 1e9, 1),
                          "GB/s": round(nbytes / med / 1e6, 1), "max_abs_err_vs_fp32": err if name.startswith("rn_") else err_lib}), flush=True)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reps = 30
    if "--reps" in sys.argv:
        reps = int(sys.argv[sys.argv.index("--reps") + 1])
        args = [a for a in args if a != str(reps)]
    which = args or ["k2", "k2_500", "k3", "k3_500", "k3f32", "k3fwd", "detect"]
    for w in which:
        if w == "k2":
            bench_k2(reps, 8)
        elif w == "k2_500":
            bench_k2(reps, 500)
        elif w == "k3":
            bench_k3(reps, torch.bfloat16)
        elif w == "k3f16":
            bench_k3(reps, torch.float16)
        elif w == "k3_500":          # BASELINE configs[4]: fp16, 500 GT boxes per image -> 25x the repair (phase B) work
            bench_k3(reps, torch.float16, T=500)
        elif w == "k3f32":
            bench_k3(reps, torch.float32)
        elif w == "k3fused8":
            bench_k3_fused(reps, torch.bfloat16, T=8)
        elif w == "k3fused":
            for T in (0, 8, 32, 64):
                bench_k3_fused(reps, torch.bfloat16, T=T)
        elif w == "k3fwd":
            bench_k3(reps, torch.bfloat16, want_grad=False)
        elif w == "detect":
            bench_detect(max(reps // 5, 4), -7.0, 1.2, "sparse")
        elif w == "detect_empty":
            bench_detect(6, -20.0, 0.5, "empty")
        elif w == "detect_stress":
            bench_detect(3, -6.0, 1.5, "stress")
        elif w == "narrow3x3":
            bench_narrow3x3(reps)
            bench_narrow3x3(reps, 2, 37, 131)
        elif w == "narrow3x3_tall":
            bench_narrow3x3(reps, 8, 400, 336)


if __name__ == "__main__":
    main()
