"""Same-box A/B of K3's two forms at the train shape (A = 201 600, K = 90, B = 8, per-level tensors): the three forms of rn_loss_fwd_bwd_levels_rp:
one launch with the in-wave repair chunk by chunk (form 0), one launch with the compact list (form 2), background stream + repair kernel (form 1).

    python tools/k3_rp_ab.py            # bf16 T = 8 and fp16 T = 500; graph-replayed back to back, after a logits writer, and cold

Prints one JSON object per (dtype, T): microseconds per call for both forms in the three conditions, and K2 beside them."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import synth  # noqa: E402
from pytorch_retinanet_amd import ops  # noqa: E402
from pytorch_retinanet_amd.anchors import AnchorGenerator  # noqa: E402


def case(device, B, T, dtype):
    K = 90
    shapes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    A = sum(h * w * 9 for h, w in shapes)
    ag = AnchorGenerator().to(device)
    anc = ops.anchors_emit(synth.levels_for(800, 1344), list(ag.cell_anchors), 0.0)
    g = torch.Generator(device=device).manual_seed(2)
    src = [(torch.randn((B, h * w * 9, K), device=device, generator=g) - 4.6).to(dtype) for h, w in shapes]
    cls = [t.clone() for t in src]
    box = [(torch.randn((B, h * w * 9, 4), device=device, generator=g) * 0.1).to(dtype) for h, w in shapes]
    rng = np.random.default_rng(0)
    gtb, gtl = zip(*[synth.gt_boxes(rng, T, 800, 1333) for _ in range(B)])
    gt_boxes = torch.from_numpy(np.concatenate(gtb)).to(device)
    gt_labels = torch.from_numpy(np.concatenate(gtl)).to(device)
    off = ops.gt_offsets([T] * B, device)
    m, nfg, sp = ops.iou_match(anc, gt_boxes, off, B, 0.5, 0.4, want_special=True)
    params = ops.make_loss_params(0.25, 2.0, 0.1)
    evict = torch.empty((1 << 30,), dtype=torch.uint8, device=device)

    def k3(form):
        return lambda: ops.loss_fwd_bwd_levels(cls, box, anc, gt_boxes, gt_labels, off, m, nfg, params, True, special=sp,
                                               in_kernel_finalize=True, form=form)

    def writer():
        for d, t in zip(cls, src):
            d.copy_(t)

    def cold(fn):
        def f():
            evict.fill_(1)
            return fn()
        return f

    out = {"dtype": str(dtype).replace("torch.", ""), "T": T, "matched_per_image": int(nfg.sum()) // B,
           "ignored_per_image": int((m == -2).sum()) // B}
    t_w = bench.graph_replay_ms(writer)
    t_e = bench.graph_replay_ms(lambda: evict.fill_(1))
    forms = (("chunks", 0), ("list", 2)) if "--in-kernel-only" in sys.argv else (("chunks", 0), ("list", 2), ("repair_pass", 1))
    for name, rp in forms:
        f = k3(rp)
        out[name] = {"back_to_back_us": round(1e3 * bench.graph_replay_ms(f), 2),
                     "after_writer_us": round(1e3 * (bench.graph_replay_ms(lambda: (writer(), f())[1]) - t_w), 2),
                     "cold_us": round(1e3 * (bench.graph_replay_ms(cold(f)) - t_e), 2)}
    nb = bench.k3_bytes(B, A, K, T, 2)
    for name, _ in forms:
        out[name]["frac_back_to_back"] = round(nb / (out[name]["back_to_back_us"] * 1e-6) / 8e12, 4)
        out[name]["frac_after_writer"] = round(nb / (out[name]["after_writer_us"] * 1e-6) / 8e12, 4)
        out[name]["frac_cold"] = round(nb / (out[name]["cold_us"] * 1e-6) / 8e12, 4)
    return out


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for T, dt in ((8, torch.bfloat16), (500, torch.float16), (64, torch.bfloat16)):
        print(json.dumps(case(dev, 8, T, dt)), flush=True)
