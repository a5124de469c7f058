#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference; torchvision is
replaced by the test-only stand-in ``_tv_standin.py``).  For every case it

  1. runs the reference's own function (file:line cited per case),
  2. asserts the CPU oracle (``oracle/rn_oracle.c``) reproduces it -- this is
     what pins the oracle -- and
  3. writes inputs (or their seed + sha256) and expected outputs as ``.npz``.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
"""
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore", message="torch.meshgrid")

import torch  # noqa: E402

import _tv_standin  # noqa: E402
import oracle  # noqa: E402
import synth  # noqa: E402

R = _tv_standin.import_reference("/root/reference")
from retinanet import box_utils as ref_box_utils  # noqa: E402
from retinanet import losses as ref_losses  # noqa: E402
from retinanet import models as ref_models  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def ref_anchor_set(h, w, **kw):
    """Reference anchors for a padded (h, w) input: anchors.py:172-197 + cat (:228)."""
    ag = R.AnchorGenerator(**kw)
    grids = synth.fpn_grid_sizes(h, w)[: ag.num_features]
    per_level = ag.grid_anchors([list(g) for g in grids], device=torch.device("cpu"))
    cells = [b.numpy().copy() for b in ag.cell_anchors]
    return torch.cat(per_level).numpy().copy(), cells, ag


# --------------------------------------------------------------------------- #
def gen_anchors():
    print("[anchors]  retinanet/anchors.py:110-228")
    out = {}
    for tag, (h, w) in {"r18_512": (512, 512), "r50_800x1344": (800, 1344), "r101_1344": (1344, 1344)}.items():
        ref, cells, _ = ref_anchor_set(h, w)
        levels = synth.levels_for(h, w)
        oc = [oracle.cell_anchors(s, synth.ANCHOR_RATIOS) for s in synth.ANCHOR_SIZES]
        for a, b in zip(oc, cells):
            assert np.array_equal(a, b), "oracle cell anchors != reference"
        got = oracle.anchors_emit(levels, oc, 0.0)
        assert got.shape == ref.shape and np.array_equal(got, ref), f"oracle anchors != reference ({tag})"
        out[f"{tag}_sha"] = np.array(synth.sha(ref))
        out[f"{tag}_count"] = np.array(ref.shape[0])
        out[f"{tag}_head"] = ref[:32]
        out[f"{tag}_tail"] = ref[-32:]
        out[f"{tag}_levels"] = np.array(levels, dtype=np.int32)
        print(f"  {tag}: A={ref.shape[0]} bit-exact")
    out["cells"] = np.stack(cells)
    # non-default generator: 2 levels, offset 0.5 / 0.3, custom sizes & ratios (anchors.py:69-99)
    for tag, off in (("toy_off05", 0.5), ("toy_off03", 0.3)):
        sizes = [[20.0, 33.5], [70.0, 91.25]]
        ratios = [0.4, 1.0, 3.0]
        strides = [8, 16]
        ag = R.AnchorGenerator(sizes=sizes, aspect_ratios=ratios, strides=strides, offset=off)
        grids = [[5, 7], [3, 4]]
        ref = torch.cat(ag.grid_anchors(grids, device=torch.device("cpu"))).numpy()
        oc = [oracle.cell_anchors(s, ratios) for s in sizes]
        got = oracle.anchors_emit([(5, 7, 8), (3, 4, 16)], oc, off)
        assert np.array_equal(got, ref), f"oracle anchors != reference ({tag})"
        out[f"{tag}_full"] = ref
    save("anchors.npz", **out)


# --------------------------------------------------------------------------- #
def ref_match(anchors, gt):
    return ref_box_utils.matcher(torch.from_numpy(anchors), torch.from_numpy(gt).reshape(-1, 4)).numpy()


def gen_match():
    print("[match]  retinanet/box_utils.py:51-80 + torchvision box_iou")
    out = {}
    anc18, _, _ = ref_anchor_set(512, 512)
    anc50, _, _ = ref_anchor_set(800, 1344)
    for T in (0, 1, 8, 64, 500):
        rng = np.random.default_rng(1000 + T)
        gt, _ = synth.gt_boxes(rng, T, 512, 512)
        ref = ref_match(anc18, gt)
        got, nfg = oracle.iou_match(anc18, [gt])
        assert np.array_equal(got[0], ref), f"oracle matcher != reference (T={T})"
        assert nfg[0] == (ref >= 0).sum()
        out[f"r18_T{T}_gt"] = gt
        out[f"r18_T{T}_matches"] = ref.astype(np.int16)
        print(f"  r18 T={T}: fg={int((ref >= 0).sum())} bg={int((ref == -1).sum())} ign={int((ref == -2).sum())} bit-exact")
    rng = np.random.default_rng(2008)
    gt, _ = synth.gt_boxes(rng, 8, 800, 1333)
    ref = ref_match(anc50, gt)
    got, _ = oracle.iou_match(anc50, [gt])
    assert np.array_equal(got[0], ref)
    out["r50_T8_gt"] = gt
    out["r50_T8_matches"] = ref.astype(np.int16)
    print(f"  r50 T=8: fg={int((ref >= 0).sum())} bit-exact")
    # handcrafted: ties, exact thresholds, degenerate / NaN
    hand_anchors = np.array([
        [0, 0, 10, 10],      # 0: vs gt0 IoU exactly 0.5 -> ignore; vs gt1 0.4 -> ignore
        [100, 100, 110, 110],  # 1: identical to gt2 and gt3 (tie) -> lowest index 2
        [200, 200, 200, 200],  # 2: zero-area anchor on zero-area gt4 -> 0/0 NaN -> ignore
        [300, 300, 310, 310],  # 3: no overlap -> bg
        [0, 0, 10, 6],       # 4: vs gt0: inter 50, union 60 -> 0.8333 -> gt0
        [100, 100, 110, 104],  # 5: vs gt2/gt3 0.4 exactly -> ignore (not < 0.4)
        [100, 100, 110, 103.9],  # 6: just under 0.4 -> bg
        [100, 100, 110, 105.1],  # 7: just over 0.5 -> 2
    ], dtype=np.float32)
    hand_gt = np.array([
        [0, 0, 10, 5],
        [0, 0, 10, 4],
        [100, 100, 110, 110],
        [100, 100, 110, 110],
        [200, 200, 200, 200],
    ], dtype=np.float32)
    ref = ref_match(hand_anchors, hand_gt)
    got, _ = oracle.iou_match(hand_anchors, [hand_gt])
    assert np.array_equal(got[0], ref), (got[0], ref)
    print("  handcrafted:", ref.tolist())
    out["hand_anchors"], out["hand_gt"], out["hand_matches"] = hand_anchors, hand_gt, ref.astype(np.int16)
    # NaN-first ordering: gt with NaN IoU not in first position
    hand_gt2 = hand_gt[[4, 0, 2]]
    ref = ref_match(hand_anchors, hand_gt2)
    got, _ = oracle.iou_match(hand_anchors, [hand_gt2])
    assert np.array_equal(got[0], ref), (got[0], ref)
    out["hand2_gt"], out["hand2_matches"] = hand_gt2, ref.astype(np.int16)
    save("match.npz", **out)


# --------------------------------------------------------------------------- #
def ref_losses_with_grads(cls, box, anchors, gtb, gtl, K):
    """RetinaNetLosses.forward  retinanet/losses.py:113-145, with autograd grads of each loss."""
    B = cls.shape[0]
    crit = ref_losses.RetinaNetLosses(K)
    c = torch.from_numpy(cls).clone().requires_grad_(True)
    b = torch.from_numpy(box).clone().requires_grad_(True)
    targets = [{"boxes": torch.from_numpy(gtb[i]).reshape(-1, 4), "labels": torch.from_numpy(gtl[i]).reshape(-1)}
               for i in range(B)]
    ancs = [torch.from_numpy(anchors) for _ in range(B)]
    out = crit(targets, {"cls_preds": c, "bbox_preds": b}, ancs)
    cl, rl = out["classification_loss"], out["regression_loss"]
    gc = torch.autograd.grad(cl, c, retain_graph=True)[0].numpy()
    gb = torch.autograd.grad(rl, b, allow_unused=True)[0] if rl.requires_grad else None
    gb = gb.numpy() if gb is not None else np.zeros_like(box)
    per = []
    for i in range(B):
        bb, cc = crit.calc_loss(ancs[i], torch.from_numpy(cls[i]), torch.from_numpy(box[i]),
                                targets[i]["labels"], targets[i]["boxes"])
        per.append([float(bb), float(cc)])
    return np.array([float(cl), float(rl)], np.float32), np.array(per, np.float32), gc, gb


def check_loss_oracle(tag, cls, box, anchors, gtb, gtl, K, ref):
    loss, per, gc, gb = ref
    matches, nfg = oracle.iou_match(anchors, gtb)
    o = oracle.loss_fwd_bwd(cls, box, anchors, gtb, gtl, matches)
    np.testing.assert_allclose(o["loss"], loss, rtol=1e-5, atol=1e-7, err_msg=tag)
    np.testing.assert_allclose(o["per_image"], per, rtol=1e-5, atol=1e-7, err_msg=tag)
    np.testing.assert_allclose(o["gcls"], gc, rtol=1e-5, atol=1e-9, err_msg=tag)
    np.testing.assert_allclose(o["gbox"], gb, rtol=1e-5, atol=1e-9, err_msg=tag)
    print(f"  {tag}: loss={loss.tolist()} nfg={nfg.tolist()} oracle within 1e-5")
    return matches, nfg


def gen_loss():
    print("[loss]  retinanet/losses.py:19-145 (+ box_utils.py:25-34)")
    out = {}
    # toy: B=3, 64x64 input, K=3; image 1 has EMPTY GT (Q7), image 2 has GT that matches nothing
    anc, _, _ = ref_anchor_set(64, 64)
    A, K = anc.shape[0], 3
    rng = np.random.default_rng(7)
    cls, box = synth.head_outputs(rng, 3, A, K, cls_mean=-2.0, cls_std=1.5, box_std=0.3)
    gtb = [np.array([[4, 6, 40, 44], [20, 10, 60, 58]], np.float32), np.zeros((0, 4), np.float32),
           np.array([[1, 1, 3, 3]], np.float32)]
    gtl = [np.array([1, 3], np.int64), np.zeros((0,), np.int64), np.array([2], np.int64)]
    ref = ref_losses_with_grads(cls, box, anc, gtb, gtl, K)
    matches, nfg = check_loss_oracle("toy", cls, box, anc, gtb, gtl, K, ref)
    assert nfg[0] > 0 and nfg[1] == 0
    assert ref[1][1, 1] == 0.0, "Q7: empty GT must give zero classification loss"
    out.update(toy_anchors=anc, toy_cls=cls, toy_box=box, toy_gtb0=gtb[0], toy_gtb2=gtb[2], toy_gtl0=gtl[0],
               toy_gtl2=gtl[2], toy_loss=ref[0], toy_per_image=ref[1], toy_gcls=ref[2], toy_gbox=ref[3],
               toy_matches=matches.astype(np.int16), toy_nfg=nfg)
    # R18 config (BASELINE.json configs[0]): B=2, A=49104, K=90; inputs regenerated from the seed in-test
    anc, _, _ = ref_anchor_set(512, 512)
    A, K, B = anc.shape[0], 90, 2
    for variant in ("f32", "bf16", "f16"):
        rng = np.random.default_rng(18)
        cls, box = synth.head_outputs(rng, B, A, K)
        gtb, gtl = [], []
        for T in (8, 3):
            g, l = synth.gt_boxes(rng, T, 512, 512)
            gtb.append(g)
            gtl.append(l)
        if variant == "bf16":
            cls, box = synth.round_bf16(cls), synth.round_bf16(box)
        elif variant == "f16":
            cls, box = synth.round_f16(cls), synth.round_f16(box)
        ref = ref_losses_with_grads(cls, box, anc, gtb, gtl, K)
        matches, nfg = check_loss_oracle(f"r18/{variant}", cls, box, anc, gtb, gtl, K, ref)
        idx_c = synth.sample_idx(cls.size, 2048, seed=5)
        # make sure positives' grads are in the sample: add every element of the fg rows' target class
        fg_rows = np.argwhere(matches >= 0)
        pos = np.array([(b * A + a) * K + (gtl[b][matches[b, a]] - 1) for b, a in fg_rows], dtype=np.int64)
        idx_c = np.unique(np.concatenate([idx_c, pos]))
        idx_b = np.unique(np.concatenate([synth.sample_idx(box.size, 512, seed=6),
                                          np.array([(b * A + a) * 4 + j for b, a in fg_rows for j in range(4)],
                                                   dtype=np.int64)]))
        pre = f"r18_{variant}_"
        out.update({
            pre + "in_sha": np.array(synth.sha(cls) + synth.sha(box) + synth.sha(np.concatenate(gtb))),
            pre + "loss": ref[0], pre + "per_image": ref[1], pre + "nfg": nfg,
            pre + "gcls_idx": idx_c, pre + "gcls_val": ref[2].reshape(-1)[idx_c],
            pre + "gbox_idx": idx_b, pre + "gbox_val": ref[3].reshape(-1)[idx_b],
            pre + "gcls_sum": np.array([ref[2].astype(np.float64).sum(), np.abs(ref[2]).astype(np.float64).sum()]),
            pre + "gbox_sum": np.array([ref[3].astype(np.float64).sum(), np.abs(ref[3]).astype(np.float64).sum()]),
        })
    save("loss.npz", **out)


# --------------------------------------------------------------------------- #
def gen_decode():
    print("[decode]  retinanet/box_utils.py:37-48 (Q4) + clip_boxes_to_image (models.py:189)")
    anc, _, _ = ref_anchor_set(512, 512)
    rows = synth.sample_idx(anc.shape[0], 1024, seed=9)
    a = anc[rows]
    rng = np.random.default_rng(9)
    d = (rng.standard_normal((1024, 4), dtype=np.float32) * np.float32(0.5)).astype(np.float32)
    dec = ref_box_utils.activ_2_bbox(torch.from_numpy(d).clone(), torch.from_numpy(a)).numpy()
    from torchvision.ops import boxes as tvops
    clip = tvops.clip_boxes_to_image(torch.from_numpy(dec), (480, 500)).numpy()
    got = oracle.decode_clip(d, a, None)
    np.testing.assert_allclose(got, dec, rtol=1e-5, atol=1e-3)
    got = oracle.decode_clip(d, a, [(480, 500)])
    np.testing.assert_allclose(got, clip, rtol=1e-5, atol=1e-3)
    # Q4: w,h come from exp(dx), exp(dy); dw,dh are ignored
    d2 = d.copy()
    d2[:, 2:] = 7.0
    dec2 = ref_box_utils.activ_2_bbox(torch.from_numpy(d2).clone(), torch.from_numpy(a)).numpy()
    assert np.array_equal(dec2, dec), "Q4 expected dw,dh to be ignored"
    # encode (bbox_2_activ, box_utils.py:25-34) on the same rows against shifted GT
    g = a + rng.uniform(-6, 6, size=a.shape).astype(np.float32)
    g[:, 2:] = np.maximum(g[:, 2:], g[:, :2] + 1.0)
    enc = ref_box_utils.bbox_2_activ(torch.from_numpy(g), torch.from_numpy(a)).numpy()
    np.testing.assert_allclose(oracle.encode(g, a), enc, rtol=1e-5, atol=1e-6)
    print("  oracle decode/clip/encode within tolerance")
    save("decode.npz", anchors=a, deltas=d, decoded=dec, clipped=clip, clip_hw=np.array([480, 500]), enc_gt=g, encoded=enc)


# --------------------------------------------------------------------------- #
def ref_detect(cls, box, anchors, hw, score_thr=0.05, nms_thr=0.5, max_det=100):
    """Retinanet.process_detections  retinanet/models.py:160-243 (unbound, on a namespace)."""
    ns = types.SimpleNamespace(score_thres=score_thr, nms_thres=nms_thr, detections_per_img=max_det)
    B = cls.shape[0]
    outputs = {"cls_preds": torch.from_numpy(cls).clone(), "bbox_preds": torch.from_numpy(box).clone()}
    dets = ref_models.Retinanet.process_detections(ns, outputs, [torch.from_numpy(anchors)] * B, hw)
    return [{k: v.numpy() for k, v in d.items()} for d in dets]


def compare_dets(tag, got, ref):
    for b, (g, r) in enumerate(zip(got, ref)):
        assert g["labels"].shape == r["labels"].shape, (tag, b, g["labels"].shape, r["labels"].shape)
        assert np.array_equal(g["labels"], r["labels"]), (tag, b)
        np.testing.assert_allclose(g["scores"], r["scores"], rtol=1e-6, atol=1e-7, err_msg=tag)
        np.testing.assert_allclose(g["boxes"], r["boxes"], rtol=1e-5, atol=1e-3, err_msg=tag)


def gen_detect():
    print("[detect]  retinanet/models.py:160-243 + torchvision nms/remove_small/clip")
    out = {}
    anc, _, _ = ref_anchor_set(64, 64)
    A, K = anc.shape[0], 3
    rng = np.random.default_rng(11)
    cls, box = synth.head_outputs(rng, 2, A, K, cls_mean=-2.0, cls_std=1.5, box_std=0.2)
    hw = [(60, 64), (64, 50)]
    ref = ref_detect(cls, box, anc, hw)
    compare_dets("toy", oracle.detect(cls, box, anc, hw), ref)
    print(f"  toy: dets/img={[len(r['scores']) for r in ref]} oracle ok")
    out.update(toy_anchors=anc, toy_cls=cls, toy_box=box, toy_hw=np.array(hw))
    for b, r in enumerate(ref):
        out.update({f"toy_boxes{b}": r["boxes"], f"toy_scores{b}": r["scores"], f"toy_labels{b}": r["labels"]})
    # max_det smaller than survivors + different thresholds
    ref = ref_detect(cls, box, anc, hw, score_thr=0.2, nms_thr=0.3, max_det=7)
    compare_dets("toy2", oracle.detect(cls, box, anc, hw, oracle.default_detect_params(0.2, 1e-2, 0.3, 7)), ref)
    for b, r in enumerate(ref):
        out.update({f"toy2_boxes{b}": r["boxes"], f"toy2_scores{b}": r["scores"], f"toy2_labels{b}": r["labels"]})
    # R18 config, sparse and denser regimes, inputs regenerated from seed in-test
    anc, _, _ = ref_anchor_set(512, 512)
    A, K = anc.shape[0], 90
    hw = [(512, 512), (480, 500)]
    for tag, mean, std, seed in (("sparse", -7.0, 1.2, 21), ("dense", -6.0, 1.5, 22)):
        rng = np.random.default_rng(seed)
        cls, box = synth.head_outputs(rng, 2, A, K, cls_mean=mean, cls_std=std, box_std=0.1)
        ref = ref_detect(cls, box, anc, hw)
        compare_dets(tag, oracle.detect(cls, box, anc, hw), ref)
        ncand = int((1 / (1 + np.exp(-cls.astype(np.float64))) > 0.05).sum())
        print(f"  r18 {tag}: candidates={ncand} dets/img={[len(r['scores']) for r in ref]} oracle ok")
        out[f"r18_{tag}_in_sha"] = np.array(synth.sha(cls) + synth.sha(box))
        out[f"r18_{tag}_hw"] = np.array(hw)
        for b, r in enumerate(ref):
            out.update({f"r18_{tag}_boxes{b}": r["boxes"], f"r18_{tag}_scores{b}": r["scores"],
                        f"r18_{tag}_labels{b}": r["labels"]})
    save("detect.npz", **out)


def gen_nms():
    print("[nms]  torchvision.ops.nms as called at retinanet/models.py:210")
    from torchvision.ops import boxes as tvops
    out = {}
    for n in (0, 1, 5, 200, 1500):
        rng = np.random.default_rng(300 + n)
        ctr = rng.uniform(0, 200, size=(max(n // 6, 1), 2))
        c = ctr[rng.integers(0, len(ctr), size=n)] + rng.normal(0, 4, size=(n, 2))
        wh = rng.uniform(10, 60, size=(n, 2))
        boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        scores = rng.uniform(0.05, 1, size=n).astype(np.float32)
        if n >= 200:
            scores[::7] = scores[3]          # score ties: stable order decides
            boxes[5] = boxes[4]              # identical boxes
        for thr in (0.5, 0.3):
            ref = tvops.nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr).numpy()
            got = oracle.nms(boxes, scores, thr)
            assert np.array_equal(got, ref), (n, thr)
            out[f"n{n}_keep_{int(thr * 10)}"] = ref
        out[f"n{n}_boxes"], out[f"n{n}_scores"] = boxes, scores
        print(f"  n={n}: keep={len(ref)} bit-exact")
    save("nms.npz", **out)


def gen_transform():
    print("[transform]  GeneralizedRCNNTransform as constructed at retinanet/models.py:116 and called at :262, :279")
    # The class the reference module holds is torchvision's; here it resolves to the stand-in, whose
    # arithmetic is torch's own ops (sub/div, F.interpolate bilinear on CPU, zero-pad copy).
    rng = np.random.default_rng(21)
    imgs = [rng.random((3, 19, 27), dtype=np.float32), rng.random((3, 32, 24), dtype=np.float32),
            rng.random((3, 20, 30), dtype=np.float32)]
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    out = {"mean": np.float32(mean), "std": np.float32(std)}
    for i, im in enumerate(imgs):
        out[f"img{i}"] = im
    # (min, max): up-scaling limited by max_size; identity for image 0 (19 -> 19); down-scaling
    for tag, (mn, mx) in {"up": (28, 40), "ident": (19, 100), "down": (12, 16)}.items():
        t = ref_models.GeneralizedRCNNTransform(mn, mx, mean, std)
        t.eval()
        gtb = [torch.tensor([[1.0, 2.0, 10.0, 12.0]]) for _ in imgs]
        il, tg = t([torch.from_numpy(i) for i in imgs], [{"boxes": b.clone()} for b in gtb])
        ref = il.tensors.numpy()
        got, sizes = oracle.transform_batch(imgs, mn, mx, mean, std)
        assert [tuple(s) for s in il.image_sizes] == sizes, (tag, il.image_sizes, sizes)
        assert got.shape == ref.shape, (tag, got.shape, ref.shape)
        err = float(np.abs(got - ref).max())
        assert err <= 2e-5, (tag, err)
        print(f"  {tag}: batch {ref.shape}, sizes {sizes}, oracle max abs err {err:.2e}")
        out[f"{tag}_cfg"] = np.int32([mn, mx])
        out[f"{tag}_batch"] = ref
        out[f"{tag}_sizes"] = np.int32(sizes)
        out[f"{tag}_boxes"] = np.stack([x["boxes"].numpy() for x in tg])
    save("transform.npz", **out)


def hash_key(k: str) -> int:
    "stable 32-bit hash of a parameter name (seeds the projection direction; Python's hash() is salted per process)"
    import zlib
    return zlib.crc32(k.encode())


def gen_e2e(kind: str = "resnet18", fname: str = "e2e.npz", cfg=None, inputs=None, cls_std: float = 0.0016, fp64: bool = False):
    """Retinanet.forward (models.py:274-288) and Retinanet.predict (models.py:245-272) of the REFERENCE model itself, for a
    seed-reproducible state dict: the fixture the assembled GPU model is held to.  ``resnet18`` -> e2e.npz (BasicBlock
    trunk); ``resnet50`` -> e2e_r50.npz (the Bottleneck trunk of the headline configuration, backbone.py:105-136)."""
    print(f"[e2e {kind}]  retinanet/models.py:245-288 (forward in train-mode BN, predict in eval mode)")
    E2E = dict(cfg) if cfg is not None else dict(num_classes=5, backbone_kind=kind, pretrained=False, min_size=128, max_size=160)
    torch.manual_seed(0)
    ref = R.Retinanet(**E2E)
    spec = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in ref.state_dict().items()]
    vals = synth.state_dict_values(spec, seed=4242, cls_std=cls_std)
    sd = ref.state_dict()
    for k, v in vals.items():
        sd[k] = torch.from_numpy(v)
    ref.load_state_dict(sd)
    images, targets = (inputs or synth.e2e_inputs)()
    timgs = [torch.from_numpy(i) for i in images]
    ttgts = [{"boxes": torch.from_numpy(b), "labels": torch.from_numpy(l)} for b, l in targets]
    out = {"spec_keys": np.array([k for k, _, _ in spec]), "spec_shapes": np.array([",".join(map(str, s)) for _, s, _ in spec]),
           "spec_dtypes": np.array([d for _, _, d in spec]),
           "weights_sha": np.array(synth.sha(np.concatenate([vals[k].astype(np.float64).reshape(-1) for k in sorted(vals)]))),
           "inputs_sha": np.array(synth.sha(np.concatenate([i.reshape(-1) for i in images]))), "cls_std": np.array(cls_std)}
    # (1) training forward: the whole module in train() -- BatchNorm uses batch statistics (Q18: train() un-freezes BN)
    ref.train()
    losses = ref(timgs, [{k: v.clone() for k, v in t.items()} for t in ttgts])
    total = losses["classification_loss"] + losses["regression_loss"]
    total.backward()
    out["train_losses"] = np.array([float(losses["classification_loss"].detach()), float(losses["regression_loss"].detach())], np.float64)
    # gradient fingerprints: L2 norms of a few parameter gradients spread over the stack (bwd parity of the assembled model)
    probes = ["backbone.backbone.conv1.weight", "backbone.backbone.layer2.0.conv1.weight", "backbone.backbone.layer4.1.bn2.weight",
              "fpn.conv_c3_3x3.weight", "fpn.conv_c7_3x3.bias", "retinanet_head.classification_head.class_subnet.0.weight",
              "retinanet_head.classification_head.class_subnet_output.weight", "retinanet_head.classification_head.class_subnet_output.bias",
              "retinanet_head.regression_head.box_subnet.6.weight", "retinanet_head.regression_head.box_subnet_output.weight"]
    named = dict(ref.named_parameters())
    out["grad_probe_keys"] = np.array(probes)
    out["grad_probe_norms"] = np.array([float(named[k].grad.double().norm()) for k in probes])
    out["grad_probe_head"] = np.stack([named[k].grad.reshape(-1)[:8].numpy().astype(np.float64) for k in probes])
    # ... and of EVERY parameter: norm, the projection on a seeded N(0, 1) direction (sign / direction, not only size) and 32
    # elements at seeded positions
    keys = [k for k, p in named.items() if p.grad is not None]
    rng = np.random.default_rng(20260101)
    norms, projs, samples, pos = [], [], [], []
    for k in keys:
        gflat = named[k].grad.reshape(-1).double().numpy()
        r = np.random.default_rng(abs(hash_key(k)) % (1 << 32)).standard_normal(gflat.size)
        idx = rng.integers(0, gflat.size, 32)
        norms.append(float(np.linalg.norm(gflat))); projs.append(float(gflat @ r)); samples.append(gflat[idx]); pos.append(idx)
    out["grad_all_keys"] = np.array(keys)
    out["grad_all_norms"] = np.array(norms)
    out["grad_all_proj"] = np.array(projs)
    out["grad_all_samples"] = np.stack(samples)
    out["grad_all_pos"] = np.stack(pos).astype(np.int64)
    if fp64:
        # The same training forward + backward of the reference in DOUBLE precision: what the fp32 numbers above approximate.  Through
        # ~50 train-mode BatchNorm backward steps (each subtracts the mean and the x-hat component of its incoming gradient) the fp32
        # run's own gradients sit 1 - 6 % (projection on a random direction, in units of the gradient's norm) off the fp64 ones at this
        # shape -- the yardstick a GPU fp32 run is measured with (tests/test_e2e_gpu.py), instead of a bar tuned until it passes.
        import copy
        ref64 = copy.deepcopy(ref).to(torch.float64)
        ref64.load_state_dict({k: v.to(torch.float64) if v.is_floating_point() else v for k, v in sd.items()})
        ref64.zero_grad()
        ref64.train()
        l64 = ref64([t.to(torch.float64) for t in timgs], [{"boxes": t["boxes"].to(torch.float64), "labels": t["labels"].clone()} for t in ttgts])
        (l64["classification_loss"] + l64["regression_loss"]).backward()
        named64 = dict(ref64.named_parameters())
        out["train_losses64"] = np.array([float(l64["classification_loss"].detach()), float(l64["regression_loss"].detach())], np.float64)
        n64, p64 = [], []
        for k in keys:
            gflat = named64[k].grad.reshape(-1).double().numpy()
            r = np.random.default_rng(abs(hash_key(k)) % (1 << 32)).standard_normal(gflat.size)
            n64.append(float(np.linalg.norm(gflat))); p64.append(float(gflat @ r))
        out["grad64_all_norms"], out["grad64_all_proj"] = np.array(n64), np.array(p64)
        e_n = np.abs(np.array(norms) - out["grad64_all_norms"]) / out["grad64_all_norms"]
        e_p = np.abs(np.array(projs) - out["grad64_all_proj"]) / out["grad64_all_norms"]
        print(f"  fp64 reference: cls={out['train_losses64'][0]:.8f} reg={out['train_losses64'][1]:.8f}; the fp32 reference's gradients against it: "
              f"norm median {np.median(e_n):.2e} max {e_n.max():.2e}, projection / norm median {np.median(e_p):.2e} p90 {np.percentile(e_p, 90):.2e} max {e_p.max():.2e}")
        del ref64, named64
    # running statistics after that ONE training forward (momentum update of every BN layer)
    out["bn1_running_mean_after"] = ref.backbone.backbone.bn1.running_mean.numpy().copy()
    print(f"  train: cls={out['train_losses'][0]:.6f} reg={out['train_losses'][1]:.6f}")
    # (2) same weights, frozen BN as constructed (freeze_bn=True leaves BN in eval inside a train()-less module): eval forward
    ref.load_state_dict(sd)
    ref.zero_grad()
    ref.eval()
    with torch.no_grad():
        losses_eval = ref(timgs, [{k: v.clone() for k, v in t.items()} for t in ttgts])
    out["eval_losses"] = np.array([float(losses_eval["classification_loss"]), float(losses_eval["regression_loss"])], np.float64)
    print(f"  eval-BN forward: cls={out['eval_losses'][0]:.6f} reg={out['eval_losses'][1]:.6f}")
    # (3) predict in eval mode: detections rescaled to the original image sizes (postprocess)
    with torch.no_grad():
        dets = ref.predict(timgs)
        # head outputs of the same pass, for the oracle cross-check and a tighter intermediate comparison
        il, _ = ref.transform(timgs, None)
        fm = ref.fpn(ref.backbone(il.tensors))
        ho = ref.retinanet_head(fm)
        anchors = ref.anchor_generator(il, fm)
    out["batch_shape"] = np.array(il.tensors.shape)
    out["image_sizes"] = np.array(il.image_sizes)
    out["cls_preds_sample_idx"] = synth.sample_idx(ho["cls_preds"].numel(), 4096, seed=77)
    out["cls_preds_sample"] = ho["cls_preds"].reshape(-1)[out["cls_preds_sample_idx"]].numpy()
    out["box_preds_sample_idx"] = synth.sample_idx(ho["bbox_preds"].numel(), 4096, seed=78)
    out["box_preds_sample"] = ho["bbox_preds"].reshape(-1)[out["box_preds_sample_idx"]].numpy()
    ncand = int((torch.sigmoid(ho["cls_preds"]) > 0.05).sum())
    print(f"  logits: mean {float(ho['cls_preds'].mean()):.3f} std {float(ho['cls_preds'].std()):.3f} max {float(ho['cls_preds'].max()):.3f};"
          f" box deltas std {float(ho['bbox_preds'].std()):.3f}")
    for b, d in enumerate(dets):
        out[f"det_boxes{b}"], out[f"det_scores{b}"], out[f"det_labels{b}"] = d["boxes"].numpy(), d["scores"].numpy(), d["labels"].numpy()
    print(f"  predict: A={anchors[0].shape[0]} candidates={ncand} dets/img={[len(d['scores']) for d in dets]}"
          f" top score={[float(d['scores'][0]) if len(d['scores']) else None for d in dets]}")
    # the oracle on the reference's own head outputs must reproduce process_detections (pins the chain on live logits)
    hw = [tuple(int(x) for x in s) for s in il.image_sizes]
    got = oracle.detect(ho["cls_preds"].numpy(), ho["bbox_preds"].numpy(), anchors[0].numpy(), hw)
    pre = ref.process_detections({"cls_preds": ho["cls_preds"].clone(), "bbox_preds": ho["bbox_preds"].clone()}, anchors, il.image_sizes)
    compare_dets("e2e", got, [{k: v.numpy() for k, v in d.items()} for d in pre])
    print("  oracle.detect == reference process_detections on the live head outputs")
    save(fname, **out)


def gen_e2e_r50():
    gen_e2e("resnet50", "e2e_r50.npz")


def gen_e2e_full():
    """The HEADLINE configuration end to end (BASELINE configs[1]'s model and per-image shape): the reference's
    ``Retinanet(num_classes=90, "resnet50", min_size=800, max_size=1333)`` on the seed-reproducible state dict and two 3 x 800 x 1333
    images with 8 GT boxes each -- train-mode loss dict + every parameter's gradient fingerprint, eval-mode losses, ``predict``.  At this
    size the assembled GPU model runs the kernels at their real tile counts (two-image canvas sheets, band / split-K / chain kernels)."""
    gen_e2e("resnet50", "e2e_full.npz", cfg=synth.E2E_FULL, inputs=synth.e2e_full_inputs, cls_std=synth.E2E_FULL_CLS_STD, fp64=True)


def gen_traj():
    """FIVE optimisation steps of the REFERENCE model: ``training_step`` (model.py:112-119: loss = sum of the loss dict) under the
    optimizer hparams.yaml:63-68 configures (``torch.optim.SGD`` on ``net.parameters()``, lr 1e-3, weight decay 1e-3, momentum 0.9,
    model.py:76-78).  Two trajectories from the seed-reproducible state dict of the e2e fixture:
      live    the module in train() -- BatchNorm on batch statistics with running-stat updates (Q18) -- on two images per step;
      frozen  the module as constructed (backbone.py:347-351: BatchNorm layers in eval(), everything else in training mode) on four
              images per step of one size: the run a 2 x 2 data-parallel split reproduces exactly (per-image normaliser, Q8).
    Recorded: the loss dict of every step, and for every parameter (and, live, every BatchNorm buffer) a fingerprint of
    final - initial: norm, projection on a seeded direction, 16 seeded elements."""
    print("[traj]  model.py:76-78,112-119 + hparams.yaml:63-68 (5 SGD steps of the reference model)")
    out = {"opt": np.array([synth.TRAJ_OPT["lr"], synth.TRAJ_OPT["weight_decay"], synth.TRAJ_OPT["momentum"]], np.float64),
           "steps": np.array(synth.TRAJ_STEPS)}
    for kind in ("live", "frozen"):
        torch.manual_seed(0)
        ref = R.Retinanet(**synth.TRAJ)
        spec = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in ref.state_dict().items()]
        vals = synth.state_dict_values(spec, seed=4242)
        sd = ref.state_dict()
        for k, v in vals.items():
            sd[k] = torch.from_numpy(v)
        ref.load_state_dict(sd)
        if kind == "live":
            ref.train()
        else:
            assert ref.training and not ref.backbone.backbone.bn1.training          # as constructed: freeze_bn=True
        opt = torch.optim.SGD(ref.parameters(), **synth.TRAJ_OPT)
        initial = {k: v.detach().clone() for k, v in ref.state_dict().items()}
        losses = []
        for step in range(synth.TRAJ_STEPS):
            images, targets = synth.traj_inputs(kind, step)
            timgs = [torch.from_numpy(i) for i in images]
            ttgts = [{"boxes": torch.from_numpy(b), "labels": torch.from_numpy(l)} for b, l in targets]
            loss_dict = ref(timgs, ttgts)
            total = sum(l for l in loss_dict.values())
            opt.zero_grad()
            total.backward()
            opt.step()
            losses.append([float(loss_dict["classification_loss"].detach()), float(loss_dict["regression_loss"].detach())])
            print(f"  {kind} step {step}: cls={losses[-1][0]:.6f} reg={losses[-1][1]:.6f}")
        out[f"{kind}_losses"] = np.array(losses, np.float64)
        final = ref.state_dict()
        pkeys = [k for k, _ in ref.named_parameters()]
        bkeys = [k for k, _ in ref.named_buffers() if "running_" in k] if kind == "live" else []
        for tag, keys in (("param", pkeys), ("buf", bkeys)):
            if not keys:
                continue
            fps = [synth.fingerprint(k, (final[k].double() - initial[k].double()).numpy()) for k in keys]
            out[f"{kind}_{tag}_keys"] = np.array(keys)
            out[f"{kind}_{tag}_norm"] = np.array([f[0] for f in fps])
            out[f"{kind}_{tag}_proj"] = np.array([f[1] for f in fps])
            out[f"{kind}_{tag}_pos"] = np.stack([f[2] for f in fps])
            out[f"{kind}_{tag}_samples"] = np.stack([f[3] for f in fps])
            out[f"{kind}_{tag}_scale"] = np.array([float(initial[k].double().norm()) for k in keys])     # |initial|: what "moved" is relative to
        if kind == "live":
            assert int(final["backbone.backbone.bn1.num_batches_tracked"]) == synth.TRAJ_STEPS
        else:
            assert torch.equal(final["backbone.backbone.bn1.running_mean"], initial["backbone.backbone.bn1.running_mean"])
        moved = np.array([f for f in out[f"{kind}_param_norm"]])
        print(f"  {kind}: {len(pkeys)} parameters, |delta| median {np.median(moved):.3e} max {moved.max():.3e}")
    save("traj.npz", **out)


if __name__ == "__main__":
    oracle.build()
    which = sys.argv[1:] or ["anchors", "match", "loss", "decode", "detect", "nms", "transform", "e2e", "e2e_r50", "traj"]
    for w in which:
        globals()["gen_" + w]()
    print("done")
