#!/usr/bin/env python3
"""Out-of-bounds probe for the hand-written kernels: every INPUT tensor is placed at the very END of its own device mapping
(no caching allocator: one hipMalloc per tensor, 2 MiB granules), so a kernel that reads or writes past an operand runs off the
mapping and the GPU raises a memory access fault (the process aborts) instead of silently touching a neighbour.  Found this way:
``stem_fwd_kernel`` -- a wave of the last workgroup that owns no tile loaded at a tile index past the last image.

usage: guard_probe.py stem B H W | w3 N H W C | n3 N H W | dense | pw | pool | bottleneck | step KIND MIN MAX | detect DT B H W K [MEAN] |
       loss DT B H W K T                                                                    (driven by tests/test_guard_gpu.py)
Prints one "ok ..." line per case; a fault kills the process (non-zero exit status, no "ok" line for the case)."""
import os
import sys

os.environ["PYTORCH_NO_CUDA_MEMORY_CACHING"] = "1"
import torch                                                       # noqa: E402
import torch.nn as nn                                              # noqa: E402

from pytorch_retinanet_amd._lib import RN_BF16, check, lib         # noqa: E402

DEV = torch.device("cuda:0")
GRAN = 2 << 20
_KEEP = []


def raw_at_end(nbytes: int, fill=None) -> torch.Tensor:
    tot = (nbytes + GRAN - 1) // GRAN * GRAN
    base = torch.empty((tot,), dtype=torch.uint8, device=DEV)
    base.fill_(0x7f if fill is None else fill)                     # bf16 0x7f7f = 3.4e38: a stray read of the padding shows in the results too
    _KEEP.append(base)
    off = (tot - nbytes) // 16 * 16
    return base[off: off + nbytes]


def cl_at_end(N: int, C: int, H: int, W: int, scale: float = 1.0, seed: int = 0) -> torch.Tensor:
    "bf16 channels-last [N, C, H, W] random tensor whose last byte is the last byte (mod 16) of its mapping"
    g = torch.Generator(device=DEV).manual_seed(seed)
    src = (torch.randn((N, H, W, C), device=DEV, generator=g) * scale).to(torch.bfloat16)
    raw = raw_at_end(src.numel() * 2)
    t = raw.view(torch.bfloat16).view(N, H, W, C)
    t.copy_(src)
    return t.permute(0, 3, 1, 2)


def main() -> None:
    which = sys.argv[1]
    st = torch.cuda.current_stream().cuda_stream
    if which == "stem":
        B, H, W = (int(v) for v in sys.argv[2:5])
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        x, w, g = raw_at_end(B * H * W * 3 * 2, 0x3c), raw_at_end(64 * 147 * 2, 0x3c), raw_at_end(B * Ho * Wo * 64 * 2, 0x3c)
        xp, wk, z = raw_at_end(lib.rn_stem_padded_bytes(B, H, W)), raw_at_end(64 * 7 * 32 * 2), raw_at_end(B * Ho * Wo * 64 * 2)
        part = raw_at_end(lib.rn_stem_partial_rows(B, H, W) * 2 * 64 * 4)
        check(lib.rn_stem_conv_forward(x.data_ptr(), w.data_ptr(), xp.data_ptr(), wk.data_ptr(), z.data_ptr(), part.data_ptr(), RN_BF16, B, H, W,
                                       st), "rn_stem_conv_forward")
        torch.cuda.synchronize()
        need = lib.rn_stem_wgrad_workspace_bytes(B, H, W)
        ws, dw = raw_at_end(need), raw_at_end(64 * 147 * 2)
        check(lib.rn_stem_conv_wgrad(g.data_ptr(), xp.data_ptr(), dw.data_ptr(), RN_BF16, B, H, W, ws.data_ptr(), need, st), "rn_stem_conv_wgrad")
        torch.cuda.synchronize()
        print("ok stem", B, H, W, flush=True)
    elif which == "w3":
        N, H, W, Cc = (int(v) for v in sys.argv[2:6])
        g, x = raw_at_end(N * H * W * Cc * 2, 0x3c), raw_at_end(N * H * W * Cc * 2, 0x3c)
        need = lib.rn_conv3x3_wgrad_narrow_workspace_bytes(Cc, Cc)
        dw, ws, zp = raw_at_end(Cc * Cc * 9 * 2), raw_at_end(need), raw_at_end(256, 0)
        check(lib.rn_conv3x3_wgrad_narrow(g.data_ptr(), x.data_ptr(), dw.data_ptr(), RN_BF16, N, H, W, Cc, Cc, zp.data_ptr(), ws.data_ptr(), need,
                                          st), "rn_conv3x3_wgrad_narrow")
        torch.cuda.synchronize()
        print("ok w3", N, H, W, Cc, flush=True)
    elif which == "n3":
        # forward product of the 64-channel 3x3 convolution (csrc/narrow3x3.hip): LDS-DMA row staging with a zero page for the halo
        N, H, W = (int(v) for v in sys.argv[2:5])
        x, w = raw_at_end(N * H * W * 64 * 2, 0x3c), raw_at_end(64 * 64 * 9 * 2, 0x3c)
        y, zp = raw_at_end(N * H * W * 64 * 2), raw_at_end(256, 0)
        check(lib.rn_conv3x3_narrow_forward(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), RN_BF16, N, H, W, 64, 0, zp.data_ptr(), st), "rn_conv3x3_narrow_forward")
        torch.cuda.synchronize()
        assert bool(torch.isfinite(y.view(torch.bfloat16).float()).all()), "a pixel outside the image was read"       # (padding is 3.4e38: 9 * 64 of them overflow)
        print("ok n3", N, H, W, flush=True)
    elif which == "dense":
        from pytorch_retinanet_amd import biasact
        for N, shapes in [(2, [(25, 42), (13, 21), (7, 11)]), (1, [(1, 300), (3, 1), (2, 2)]), (3, [(9, 30)])]:
            convs = [nn.Conv2d(256, 256, 3, 1, 1).to(DEV).to(torch.bfloat16).to(memory_format=torch.channels_last) for _ in shapes]
            for c in convs:
                c.bias.data = c.bias.data.float()
            xs = [cl_at_end(N, 256, h, w, seed=i).requires_grad_() for i, (h, w) in enumerate(shapes)]
            ys = biasact.dense_conv_group(xs, convs)
            torch.autograd.backward(ys, [cl_at_end(N, 256, h, w, seed=9 + i) for i, (h, w) in enumerate(shapes)])
            torch.cuda.synchronize()
            # layer3's conv2: forward and both gradients, P = 1
            c2 = nn.Conv2d(256, 256, 3, 1, 1, bias=False).to(DEV).to(torch.bfloat16).to(memory_format=torch.channels_last)
            x2 = cl_at_end(N, 256, *shapes[0], seed=5).requires_grad_()
            biasact.conv3x3_mfma_bwd(c2, x2).backward(cl_at_end(N, 256, *shapes[0], seed=6))
            torch.cuda.synchronize()
            print("ok dense", N, shapes, flush=True)
    elif which == "pw":
        from pytorch_retinanet_amd import pwconv
        for (N, cin, cout, H, W, k, s) in [(2, 64, 256, 9, 13, 1, 1), (1, 256, 64, 7, 5, 1, 1), (2, 128, 128, 11, 14, 3, 1), (2, 128, 128, 11, 14, 3, 2),
                                           (1, 512, 1024, 6, 10, 1, 2), (3, 1024, 256, 5, 3, 1, 1)]:
            x = cl_at_end(N, cin, H, W, seed=1)
            w = cl_at_end(cout, cin, k, k, 0.05, seed=2)
            y = pwconv.pw_forward(x, w, stride=s)
            g = cl_at_end(*y.shape, seed=3)
            pwconv.pw_wgrad(g, x, w, stride=s)
            torch.cuda.synchronize()
            print("ok pw", N, cin, cout, H, W, k, s, flush=True)
    elif which == "pool":
        from pytorch_retinanet_amd.pool import FusedMaxPool2d
        for (N, Cc, H, W) in [(2, 64, 37, 53), (1, 64, 1, 1), (1, 8, 5, 2), (2, 64, 30, 301)]:
            x = cl_at_end(N, Cc, H, W, seed=1).requires_grad_()
            y = FusedMaxPool2d(3, 2, 1)(x)
            y.backward(cl_at_end(*y.shape, seed=2))
            torch.cuda.synchronize()
            print("ok pool", N, Cc, H, W, flush=True)
    elif which == "bottleneck":
        from pytorch_retinanet_amd import backbone, optim
        for (inpl, planes, stride, H, W) in [(64, 64, 1, 19, 27), (256, 64, 1, 19, 27), (256, 128, 2, 18, 26), (1024, 512, 2, 7, 9), (2048, 512, 1, 5, 7)]:
            ds = None
            if stride != 1 or inpl != planes * 4:
                ds = nn.Sequential(nn.Conv2d(inpl, planes * 4, 1, stride, bias=False), backbone.FusedBatchNorm2d(planes * 4))
            blk = backbone.Bottleneck(inpl, planes, stride, ds).to(DEV).to(memory_format=torch.channels_last).train()
            for m in blk.modules():
                if isinstance(m, nn.Conv2d):
                    m.weight.data = m.weight.data.to(torch.bfloat16)
            x = cl_at_end(2, inpl, H, W, seed=1).requires_grad_()
            y = blk(x)
            y.backward(cl_at_end(*y.shape, seed=2))
            torch.cuda.synchronize()
            print("ok bottleneck", inpl, planes, stride, H, W, flush=True)
    elif which == "step":
        # whole train steps + one inference pass with one device allocation per tensor (page-granular mappings: an overrun of more
        # than a page past ANY tensor of the step faults), R50 trunk, odd image sizes, eager
        import numpy as np
        import pytorch_retinanet_amd as P
        import synth
        from pytorch_retinanet_amd.graph import CapturedTrainStep
        from pytorch_retinanet_amd.optim import MasterSGD, use_bf16_conv_weights
        torch.manual_seed(3)
        kind, lo, hi = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
        net = P.Retinanet(num_classes=7, backbone_kind=kind, pretrained=False, min_size=lo, max_size=hi).to(DEV)
        net = net.to(memory_format=torch.channels_last).train()
        use_bf16_conv_weights(net)
        opt = MasterSGD(net.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
        step = CapturedTrainStep(net, opt, amp_dtype=torch.bfloat16, enabled=False)
        rng = np.random.default_rng(1)
        for hw in [(lo - 3, hi - 7), (lo + 5, hi - 21)]:
            images = [torch.from_numpy(rng.random((3, *hw), dtype=np.float32)).to(DEV) for _ in range(2)]
            targets = []
            for _ in range(2):
                b, l = synth.gt_boxes(rng, 4, hw[0], hw[1], num_classes=7, wh_lo=20.0, wh_hi=90.0)
                targets.append({"boxes": torch.from_numpy(b).to(DEV), "labels": torch.from_numpy(l).to(DEV)})
            loss = float(step(images, targets)["loss"])
            assert loss == loss, "loss is NaN"
        net.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            net([torch.from_numpy(rng.random((3, lo - 1, hi - 2), dtype=np.float32)).to(DEV)])
        torch.cuda.synchronize()
        print("ok step", kind, lo, hi, flush=True)
    elif which == "detect":
        # K4-K7 (rn_detect_levels): every per-level logit / delta tensor, the anchors and the image sizes at the end of their own mappings
        import numpy as np
        import synth
        from pytorch_retinanet_amd import ops
        from pytorch_retinanet_amd.anchors import AnchorGenerator
        dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[2]]
        B, H, W, K = (int(v) for v in sys.argv[3:7])
        mean = float(sys.argv[7]) if len(sys.argv) > 7 else -4.6
        std, dstd = (float(sys.argv[8]), float(sys.argv[9])) if len(sys.argv) > 9 else (1.2, 0.1)
        levels = synth.levels_for(H, W)
        ag = AnchorGenerator().to(DEV)
        anc_src = ops.anchors_emit(levels, list(ag.cell_anchors), 0.0)
        anc = raw_at_end(anc_src.numel() * 4).view(torch.float32).view(-1, 4)
        anc.copy_(anc_src)
        g = torch.Generator(device=DEV).manual_seed(1)
        cl, bl = [], []
        for (h, w, _) in levels:
            n = h * w * 9
            c = raw_at_end(B * n * K * dt.itemsize).view(dt).view(B, n, K)
            c.copy_((torch.randn((B, n, K), device=DEV, generator=g) * std + mean).to(dt))
            d = raw_at_end(B * n * 4 * dt.itemsize).view(dt).view(B, n, 4)
            d.copy_((torch.randn((B, n, 4), device=DEV, generator=g) * dstd).to(dt))
            cl.append(c); bl.append(d)
        dets = ops.detect_levels(cl, bl, anc, [(H - 5, W - 3)] * B, 0.05, 1e-2, 0.5, 100)
        torch.cuda.synchronize()
        print("ok detect", sys.argv[2], B, H, W, K, mean, [int(d["scores"].numel()) for d in dets], flush=True)
    elif which == "loss":
        # K2 + K3 (rn_iou_match_special + rn_loss_fwd_bwd_levels_ex, and the one-launch form): per-level logits / deltas, anchors, GT
        # boxes / labels / offsets at the end of their own mappings; ragged GT counts incl. an image without GT
        import numpy as np
        import synth
        from pytorch_retinanet_amd import ops
        from pytorch_retinanet_amd.anchors import AnchorGenerator
        dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[2]]
        B, H, W, K, T = (int(v) for v in sys.argv[3:8])
        levels = synth.levels_for(H, W)
        ag = AnchorGenerator().to(DEV)
        anc_src = ops.anchors_emit(levels, list(ag.cell_anchors), 0.0)
        anc = raw_at_end(anc_src.numel() * 4).view(torch.float32).view(-1, 4)
        anc.copy_(anc_src)
        rng = np.random.default_rng(2)
        Ts = [T if i != 1 else 0 for i in range(B)]
        gtb, gtl = zip(*[synth.gt_boxes(rng, t, H, W, num_classes=K, wh_lo=20.0, wh_hi=min(H, W) * 0.6) for t in Ts])
        gb = np.concatenate(gtb).astype(np.float32); gl = np.concatenate(gtl).astype(np.int64)
        gt_t = raw_at_end(max(gb.size, 4) * 4).view(torch.float32)[:gb.size].view(-1, 4); gt_t.copy_(torch.from_numpy(gb))
        gl_t = raw_at_end(max(gl.size, 1) * 8).view(torch.int64)[:gl.size]; gl_t.copy_(torch.from_numpy(gl))
        off_src = ops.gt_offsets(Ts, DEV)
        off = raw_at_end(off_src.numel() * 4).view(torch.int32); off.copy_(off_src)
        g = torch.Generator(device=DEV).manual_seed(1)
        cl, bl = [], []
        for (h, w, _) in levels:
            n = h * w * 9
            c = raw_at_end(B * n * K * dt.itemsize).view(dt).view(B, n, K)
            c.copy_((torch.randn((B, n, K), device=DEV, generator=g) - 4.6).to(dt))
            d = raw_at_end(B * n * 4 * dt.itemsize).view(dt).view(B, n, 4)
            d.copy_((torch.randn((B, n, 4), device=DEV, generator=g) * 0.1).to(dt))
            cl.append(c); bl.append(d)
        p = ops.make_loss_params(0.25, 2.0, 0.1)
        m, nfg, sp = ops.iou_match(anc, gt_t, off, B, 0.5, 0.4, want_special=True)
        loss, gc, gbx = ops.loss_fwd_bwd_levels(cl, bl, anc, gt_t, gl_t, off, m, nfg, p, True, special=sp)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(loss).all())
        if max(Ts) <= 64:
            out = ops.loss_match_fwd_bwd_levels(cl, bl, anc, gt_t, gl_t, off, max(Ts), 0.5, 0.4, p, True, want_matches=True)
            torch.cuda.synchronize()
            assert out is None or (torch.equal(out[0], loss) and torch.equal(out[4], m))
        print("ok loss", sys.argv[2], B, H, W, K, T, [float(x) for x in loss], flush=True)
    else:
        raise SystemExit(f"unknown probe {which!r}")


if __name__ == "__main__":
    main()
