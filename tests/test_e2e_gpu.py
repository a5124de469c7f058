"""End-to-end parity of the ASSEMBLED model with the reference's own ``Retinanet.forward`` / ``Retinanet.predict``
(``/root/reference/retinanet/models.py:245-288``).

``tests/golden/e2e.npz`` was written by ``tests/golden/gen_golden.py e2e``, which imports the reference, loads the
seed-reproducible state dict of ``synth.state_dict_values`` into ``Retinanet(num_classes=5, backbone_kind="resnet18",
min_size=128, max_size=160)`` and records: the loss dict of a training forward (train-mode BN, Q18) with gradient
fingerprints, the loss dict in eval-mode BN, and the ``predict`` detection lists in eval mode.  The tests below
regenerate the same weights from the seed, load them into ``pytorch_retinanet_amd.Retinanet`` and hold the GPU model
(fused BN -> packed canvas -> tower convs -> dead-class padding -> per-level K3 / detect -> postprocess) to it.

Tolerances: fp32 losses rel <= 1e-4; detections >= 99 % box-set agreement at IoU >= 0.999 with equal
labels, scores abs <= 1e-4; bf16-autocast losses rel <= 2e-2.

Round 5: a second fixture, ``e2e_r50.npz`` (``gen_golden.py e2e_r50``: the same seed-reproducible recipe on the reference's
``resnet50`` -- the Bottleneck trunk of the headline configuration), runs through the same tests, and
``test_r50_trunk_on_the_fused_bottleneck_kernels_holds_the_reference_bf16`` holds the fused layer1 / layer2 kernels, the stem and the
head to the reference's own losses and gradient norms.
"""
import numpy as np
import pytest
import torch

import synth

E2E = dict(num_classes=5, backbone_kind="resnet18", pretrained=False, min_size=128, max_size=160)
# the same for the Bottleneck trunk of the headline configuration (tests/golden/gen_golden.py e2e_r50; reference backbone.py:105-136)
FIXTURES = {"e2e.npz": E2E, "e2e_r50.npz": dict(E2E, backbone_kind="resnet50"), "e2e_full.npz": synth.E2E_FULL}
# round 6: the HEADLINE configuration itself (``gen_golden.py e2e_full``: the reference's Retinanet(num_classes=90, "resnet50", min_size=800,
# max_size=1333) on two 3 x 800 x 1333 images with 8 GT boxes each) -- the assembled model at BASELINE configs[1]'s per-image shape
INPUTS = {"e2e.npz": synth.e2e_inputs, "e2e_r50.npz": synth.e2e_inputs, "e2e_full.npz": synth.e2e_full_inputs}
# fp32 gradient bars per fixture: (norm rtol, projection / norm, sample atol in units of the gradient's rms, probe-norm rtol, probe-head atol / rms).
# The R50 trunk at 128 x 160 px ends in 4 x 5 feature maps of a batch of two: BatchNorm over 40 - 160 values per channel and 53 layers of
# ReLU boundaries amplify the fp32 summation-order differences between the CPU reference and the GPU libraries (losses agree to 4e-7;
# measured over all 161 parameters: norm 6.0e-3 max / 1.2e-3 p90, projection 3.7e-2 / 1.6e-2, samples 0.36 rms max / 0.04 p90).
GRAD_BARS = {"e2e.npz": (3e-3, 1e-2, 2e-2, 2e-3, 1e-2), "e2e_r50.npz": (1.2e-2, 6e-2, 6e-1, 1.2e-2, 3e-1),
             "e2e_full.npz": (1.2e-2, None, 6e-1, 1.2e-2, 3e-1)}      # (projection: held to the fixture's fp64 gradients, see the test)
BOTH = pytest.mark.parametrize("fixture", list(FIXTURES))
SMALL = pytest.mark.parametrize("fixture", ["e2e.npz", "e2e_r50.npz"])
DEV = "cuda:0"


def _spec(g):
    return [(str(k), tuple(int(x) for x in str(s).split(",") if x), str(d))
            for k, s, d in zip(g["spec_keys"], g["spec_shapes"], g["spec_dtypes"])]


def _weights(g):
    vals = synth.state_dict_values(_spec(g), seed=4242, cls_std=float(g["cls_std"]) if "cls_std" in g.files else 0.0016)
    sha = synth.sha(np.concatenate([vals[k].astype(np.float64).reshape(-1) for k in sorted(vals)]))
    assert sha == str(g["weights_sha"]), "regenerated state dict differs from the one the fixture was made with"
    return vals


def _inputs(g, device, fixture="e2e.npz"):
    images, targets = INPUTS[fixture]()
    assert synth.sha(np.concatenate([i.reshape(-1) for i in images])) == str(g["inputs_sha"])
    timgs = [torch.from_numpy(i).to(device) for i in images]
    ttgts = [{"boxes": torch.from_numpy(b).to(device), "labels": torch.from_numpy(l).to(device)} for b, l in targets]
    return timgs, ttgts


def _model(g, device, cfg=None):
    import pytorch_retinanet_amd as P
    net = P.Retinanet(**(cfg or E2E))
    sd = net.state_dict()
    assert [k for k, _, _ in _spec(g)] == list(sd), "state-dict keys differ from the reference's"
    for k, v in _weights(g).items():
        assert tuple(sd[k].shape) == v.shape, k
        sd[k] = torch.from_numpy(v)
    net.load_state_dict(sd)
    return net.to(device).to(memory_format=torch.channels_last)


@BOTH
def test_fixture_weights_and_inputs_regenerate_from_the_seed(golden, fixture):
    "CPU: the seed reproduces the exact weights / images the reference was run on (sha256 recorded in the fixture)."
    g = golden(fixture)
    vals = _weights(g)
    assert len(vals) == len(_spec(g)) - 5                      # all entries but the 5 cell-anchor buffers
    images, targets = INPUTS[fixture]()
    assert synth.sha(np.concatenate([i.reshape(-1) for i in images])) == str(g["inputs_sha"])
    assert [len(t[1]) for t in targets] == ([8, 8] if fixture == "e2e_full.npz" else [3, 2])


def _iou(a, b):
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + ab[None, :] - inter)


def box_set_agreement(got, ref, iou_thr=0.999):
    """Fraction of the reference's detections that have a detection of the same label with IoU >= `iou_thr` in `got`,
    and vice versa; the smaller of the two."""
    if len(ref["labels"]) == 0 or len(got["labels"]) == 0:
        return 1.0 if len(ref["labels"]) == len(got["labels"]) else 0.0
    ok = (_iou(ref["boxes"], got["boxes"]) >= iou_thr) & (ref["labels"][:, None] == got["labels"][None, :])
    return float(min(ok.any(1).mean(), ok.any(0).mean()))


@pytest.mark.gpu
@BOTH
def test_forward_losses_and_gradients_match_the_reference_fp32(golden, fixture):
    "Retinanet.forward in train-mode BN, fp32: loss dict rel <= 1e-4; EVERY parameter gradient: norm 3e-3, a seeded projection, 32 elements."
    g = golden(fixture)
    net = _model(g, DEV, FIXTURES[fixture]).train()
    images, targets = _inputs(g, DEV, fixture)
    out = net(images, targets)
    got = np.array([float(out["classification_loss"].detach()), float(out["regression_loss"].detach())])
    np.testing.assert_allclose(got, g["train_losses"], rtol=1e-4)
    (out["classification_loss"] + out["regression_loss"]).backward()
    named = dict(net.named_parameters())
    bar_norm, bar_proj, bar_samp, bar_pnorm, bar_phead = GRAD_BARS[fixture]
    for k, norm, head in zip(g["grad_probe_keys"], g["grad_probe_norms"], g["grad_probe_head"]):
        gr = named[str(k)].grad
        assert gr is not None, k
        np.testing.assert_allclose(float(gr.double().norm()), norm, rtol=bar_pnorm, err_msg=str(k))
        # (single elements: fp32 convolutions whose algorithm MIOpen picks per run -- compare against the gradient's rms)
        np.testing.assert_allclose(gr.reshape(-1)[:8].double().cpu().numpy(), head, rtol=5e-2, atol=bar_phead * norm / np.sqrt(gr.numel()),
                                   err_msg=str(k))
    # every parameter: gradient norm, projection on a seeded random direction, 32 seeded elements (fixture: the reference's autograd)
    import zlib
    bad, gpu_norm, gpu_proj = [], [], []
    yardstick = "grad64_all_norms" in g.files          # (the headline-shape fixture also holds the reference's fp64 gradients: below)
    for k, norm, proj, samp, pos in zip(g["grad_all_keys"], g["grad_all_norms"], g["grad_all_proj"], g["grad_all_samples"], g["grad_all_pos"]):
        gr = named[str(k)].grad
        assert gr is not None, k
        flat = gr.reshape(-1).double().cpu().numpy()
        r = np.random.default_rng(zlib.crc32(str(k).encode())).standard_normal(flat.size)
        rms = norm / np.sqrt(flat.size)
        gpu_norm.append(float(np.linalg.norm(flat))); gpu_proj.append(float(flat @ r))
        ok = (abs(np.linalg.norm(flat) - norm) <= bar_norm * norm + 1e-9 and (yardstick or abs(flat @ r - proj) <= bar_proj * norm + 1e-9)
              and np.all(np.abs(flat[pos] - samp) <= 5e-2 * np.abs(samp) + bar_samp * rms + 1e-12))
        if not ok:
            bad.append((str(k), float(np.linalg.norm(flat)), float(norm), float(flat @ r), float(proj)))
    assert not bad, bad[:5]
    if yardstick:
        # The fp32 reference is itself an approximation: at this shape its gradients sit 1 - 6 % (projection on a random direction, in
        # units of the gradient's norm) off the SAME reference run in fp64 -- ~50 train-mode BatchNorm backward steps each subtract the
        # mean and the x-hat component of their incoming gradient (gen_golden.py e2e_full prints the figures).  So the GPU's fp32 run is
        # held to the fp64 gradients with the fp32 reference's own deviation as the yardstick: as a population (median / p90 / max of
        # both error kinds) no more than twice as far from fp64 as the reference's fp32 run is (measured: median 1.4e-2 / p90 3.7e-2 / max
        # 7.4e-2 against the reference's 9.0e-3 / 3.3e-2 / 6.2e-2 -- the GPU run is one more fp32 realisation of the same arithmetic);
        # the fp32 losses within 1e-5 of the fp64 ones.
        n64, p64 = g["grad64_all_norms"], g["grad64_all_proj"]
        np.testing.assert_allclose(got, g["train_losses64"], rtol=1e-5)
        ref_en, ref_ep = np.abs(g["grad_all_norms"] - n64) / n64, np.abs(g["grad_all_proj"] - p64) / n64
        gpu_en, gpu_ep = np.abs(np.array(gpu_norm) - n64) / n64, np.abs(np.array(gpu_proj) - p64) / n64
        for name, stat in (("median", np.median), ("p90", lambda a: np.percentile(a, 90)), ("max", np.max)):
            assert stat(gpu_en) <= 2.0 * stat(ref_en) + 1e-4, ("norm", name, float(stat(gpu_en)), float(stat(ref_en)))
            assert stat(gpu_ep) <= 2.0 * stat(ref_ep) + 2e-3, ("projection", name, float(stat(gpu_ep)), float(stat(ref_ep)))
        print(f"[{fixture} fp32 vs fp64] projection / norm: GPU median {np.median(gpu_ep):.2e} p90 {np.percentile(gpu_ep, 90):.2e} max {gpu_ep.max():.2e}; "
              f"reference fp32 median {np.median(ref_ep):.2e} p90 {np.percentile(ref_ep, 90):.2e} max {ref_ep.max():.2e}")
    assert len(g["grad_all_keys"]) == sum(1 for p in net.parameters() if p.requires_grad)
    # one training forward moved the BN running statistics exactly like the reference's
    np.testing.assert_allclose(net.backbone.backbone.bn1.running_mean.cpu().numpy(), g["bn1_running_mean_after"], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
@BOTH
def test_forward_losses_eval_bn_fp32(golden, fixture):
    g = golden(fixture)
    net = _model(g, DEV, FIXTURES[fixture]).eval()
    images, targets = _inputs(g, DEV, fixture)
    with torch.no_grad():
        out = net(images, targets)
    got = np.array([float(out["classification_loss"]), float(out["regression_loss"])])
    np.testing.assert_allclose(got, g["eval_losses"], rtol=1e-4)


@pytest.mark.gpu
@BOTH
def test_predict_matches_the_reference_fp32(golden, fixture):
    "Retinanet.predict in eval mode, fp32: >= 99 % box-set agreement at IoU 0.999 + label equality; scores 1e-4."
    g = golden(fixture)
    net = _model(g, DEV, FIXTURES[fixture]).eval()
    images, _ = _inputs(g, DEV, fixture)
    dets = net.predict(images)
    assert len(dets) == 2
    for b, d in enumerate(dets):
        ref = {"boxes": g[f"det_boxes{b}"], "scores": g[f"det_scores{b}"], "labels": g[f"det_labels{b}"]}
        got = {k: v.cpu().numpy() for k, v in d.items()}
        assert got["labels"].dtype == np.int64 and got["boxes"].dtype == np.float32
        assert abs(len(got["labels"]) - len(ref["labels"])) <= max(1, len(ref["labels"]) // 100)
        assert box_set_agreement(got, ref) >= 0.99, (b, box_set_agreement(got, ref))
        n = min(len(got["scores"]), len(ref["scores"]))
        np.testing.assert_allclose(got["scores"][:n], ref["scores"][:n], atol=1e-4)      # same descending score profile
    # head outputs of the same pass against the reference's sampled logits / deltas (tighter, intermediate)
    with torch.no_grad():
        il, _ = net.transform(images, None, **net._batch_layout())
        assert tuple(il.tensors.shape) == tuple(int(x) for x in g["batch_shape"])
        assert [tuple(s) for s in il.image_sizes] == [tuple(int(x) for x in s) for s in g["image_sizes"]]
        _, out = net._features(il.tensors)
    np.testing.assert_allclose(out["cls_preds"].reshape(-1)[torch.from_numpy(g["cls_preds_sample_idx"]).to(DEV)].cpu().numpy(),
                               g["cls_preds_sample"], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(out["bbox_preds"].reshape(-1)[torch.from_numpy(g["box_preds_sample_idx"]).to(DEV)].cpu().numpy(),
                               g["box_preds_sample"], rtol=1e-4, atol=2e-4)


@pytest.mark.gpu
def test_forward_and_predict_fp16_autocast(golden):
    """BASELINE configs[4] / the reference's own published run (native fp16 AMP, demo.ipynb precision=16): fp16 autocast takes the SAME
    hand-written MFMA kernels as bf16 (csrc/conv.hip, pw.hip, stem.hip, narrow3x3.hip, wgrad3x3.hip instantiated on
    v_mfma_f32_*_f16) -- asserted through the launch tags -- and holds the fp32 reference: losses within 2e-2 (measured far
    inside: fp16 carries 3 more mantissa bits than bf16), detections at the bf16 test's bar or better, gradients finite."""
    from pytorch_retinanet_amd import biasact, pwconv
    from pytorch_retinanet_amd.optim import use_16bit_conv_weights
    g = golden("e2e.npz")
    net = _model(g, DEV).train()
    assert use_16bit_conv_weights(net, torch.float16) > 0
    images, targets = _inputs(g, DEV)
    biasact.MFMA_FLOP.clear(); pwconv.PW_FLOP.clear()
    with torch.autocast("cuda", dtype=torch.float16):
        out = net(images, targets)
        total = out["classification_loss"] + out["regression_loss"]
    got = np.array([float(out["classification_loss"].detach()), float(out["regression_loss"].detach())])
    np.testing.assert_allclose(got, g["train_losses"], rtol=2e-2)
    (total * 1024.0).backward()                                            # (a static loss scale: what GradScaler does, without its bookkeeping)
    ran = set(biasact.MFMA_FLOP) | set(pwconv.PW_FLOP)
    # (the fixture's class-output conv has 9 * 5 = 45 channels: odd, so that one conv takes the library path at every dtype; its MFMA
    # form at fp16 is covered by tests/test_fp16_kernels_gpu.py)
    for tag in ("mfma_tower_fwd_x2", "mfma_box_output_fwd", "mfma_tower_dgrad_x2", "mfma_tower_wgrad_x2", "mfma_box_output_wgrad", "stem_fwd", "stem_wgrad"):
        assert tag in ran, (tag, sorted(ran))
    named = dict(net.named_parameters())
    bad = [k for k, p in named.items() if p.grad is None or not bool(torch.isfinite(p.grad.float()).all())]
    assert not bad, bad[:5]
    # gradient norms against the reference's (fp32) at 5e-2: fp16 activations + fp16 weight gradients
    for k, norm in zip(g["grad_all_keys"], g["grad_all_norms"]):
        gr = named[str(k)].grad.double() / 1024.0
        assert abs(float(gr.norm()) - norm) <= 5e-2 * norm + 1e-7, (str(k), float(gr.norm()), float(norm))
    net = _model(g, DEV).eval()
    with torch.autocast("cuda", dtype=torch.float16):
        dets = net.predict(images)
    n_top = 0
    for b, d in enumerate(dets):
        ref = {"boxes": g[f"det_boxes{b}"], "scores": g[f"det_scores{b}"], "labels": g[f"det_labels{b}"]}
        got = {k: v.float().cpu().numpy() if v.dtype != torch.int64 else v.cpu().numpy() for k, v in d.items()}
        top = ref["scores"] >= 0.06
        n_top += int(top.sum())
        if top.any():
            iou = _iou(ref["boxes"][top], got["boxes"])
            same = ref["labels"][top][:, None] == got["labels"][None, :]
            near = np.abs(ref["scores"][top][:, None] - got["scores"][None, :]) <= 5e-3
            assert ((iou >= 0.9) & same & near).any(1).mean() >= 0.85, b
            assert np.median(np.where(same, iou, 0.0).max(1)) >= 0.97, b
    assert n_top > 0


@pytest.mark.gpu
@BOTH
def test_forward_and_predict_bf16_autocast(golden, fixture):
    """The headline numeric configuration (bf16 autocast, MFMA towers, fp32 masters): losses within 2e-2 of the fp32
    reference; detections: >= 85 % of the reference's confident boxes have a same-label partner at IoU >= 0.9 with a score
    within 5e-3, median IoU >= 0.97 -- bf16 logits move scores near the 0.05 threshold, the top-100 cut-off and near-tie NMS
    decisions, so the fp32 criterion (99 % at IoU 0.999) does not apply; the exact check is the oracle test below."""
    g = golden(fixture)
    net = _model(g, DEV, FIXTURES[fixture]).train()
    images, targets = _inputs(g, DEV, fixture)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(images, targets)
    got = np.array([float(out["classification_loss"].detach()), float(out["regression_loss"].detach())])
    # (R50 fixture: the regression loss of its few matched anchors sits 1.6 - 2.1 % off from run to run -- the library picks its bf16 solvers per process)
    np.testing.assert_allclose(got, g["train_losses"], rtol=2e-2 if fixture == "e2e.npz" else 4e-2)
    net = _model(g, DEV, FIXTURES[fixture]).eval()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        dets = net.predict(images)
    n_top = 0
    for b, d in enumerate(dets):
        ref = {"boxes": g[f"det_boxes{b}"], "scores": g[f"det_scores{b}"], "labels": g[f"det_labels{b}"]}
        got = {k: v.float().cpu().numpy() if v.dtype != torch.int64 else v.cpu().numpy() for k, v in d.items()}
        top = ref["scores"] >= 0.06                                   # clear of the 0.05 threshold
        n_top += int(top.sum())
        if top.any():
            # the fp32 reference's confident detections are found by the bf16 pipeline.  What bf16 head outputs allow (measured on
            # this fixture): box deltas carry 8 bits of mantissa, so the median IoU with the fp32 box is 0.98 - 0.99, and 93 - 96 %
            # of the reference's boxes have a same-label partner at IoU >= 0.9 whose score is within 5e-3 (the rest sit at the
            # top-100 cut-off of an image with more than 100 candidates, where 1e-3 of score decides membership).  The EXACT
            # check of this configuration's detection chain is the oracle test below.
            iou = _iou(ref["boxes"][top], got["boxes"])
            same = ref["labels"][top][:, None] == got["labels"][None, :]
            near = np.abs(ref["scores"][top][:, None] - got["scores"][None, :]) <= 5e-3
            ok = (iou >= 0.9) & same & near
            assert ok.any(1).mean() >= 0.85, (b, ok.any(1).mean())      # (measured 0.93 - 0.96; the summation order of the conv kernels moves it by a box or two)
            best = np.where(same, iou, 0.0).max(1)
            # (headline-shape fixture: its box deltas have a standard deviation of 0.48 -- four times the small fixtures' -- and a bf16
            # delta moves a decoded box by 2^-9 |delta| of its size: measured median 0.968)
            assert np.median(best) >= (0.95 if fixture == "e2e_full.npz" else 0.97), (b, np.median(best))
    assert n_top > 0                                                   # the fixture does have confident detections to find


@pytest.mark.gpu
def test_r50_trunk_on_the_fused_bottleneck_kernels_holds_the_reference_bf16(golden):
    """The headline configuration's TRUNK against the reference's own R50 (``e2e_r50.npz``): bf16 conv weights + bf16 autocast put layer1 / layer2
    on the fused bottleneck kernels of csrc/pw.hip (conv1 + statistics, conv3 walker, block output + next conv1, conv3's two gradients in one pass,
    conv1's data gradient + the previous block's sums), the stem on csrc/stem.hip and conv2 on the narrow / band kernels -- asserted through the
    launch tags -- and the result is held to the reference's fp32 losses (4e-2; measured 1.7e-3 / 1.6e-2, the regression loss up to 2.1e-2 in other processes) and to every parameter's gradient norm.
    Bars: convolution weights 0.25 (measured: median 0.015, p90 0.045, max 0.115 at P7's 1 x 2 map -- and once above 0.15 on another box: the
    library picks its bf16 solvers per process), BatchNorm weights / biases and conv biases 0.4 (median 0.035, p90 0.088, max 0.18): a per-channel gradient is a SUM over 160 - 2 560 positions of terms that nearly cancel, so bf16's
    2^-9 per term becomes sqrt(N) 2^-9 of the sum."""
    from pytorch_retinanet_amd import biasact, pwconv
    from pytorch_retinanet_amd.optim import use_16bit_conv_weights
    g = golden("e2e_r50.npz")
    net = _model(g, DEV, FIXTURES["e2e_r50.npz"]).train()
    assert use_16bit_conv_weights(net, torch.bfloat16) > 0
    images, targets = _inputs(g, DEV)
    biasact.MFMA_FLOP.clear(); pwconv.PW_FLOP.clear()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(images, targets)
        total = out["classification_loss"] + out["regression_loss"]
    got = np.array([float(out["classification_loss"].detach()), float(out["regression_loss"].detach())])
    np.testing.assert_allclose(got, g["train_losses"], rtol=4e-2)
    total.backward()
    ran = set(biasact.MFMA_FLOP) | set(pwconv.PW_FLOP)
    for tag in ("stem_fwd", "stem_wgrad", "pw_conv1_fwd", "pw_conv3_fwd", "pw_conv3_bwd", "pw_block_out_conv1", "pw_conv1_dgrad_sums", "pw_conv1_wgrad",
                "mfma_tower_fwd_x2", "mfma_tower_dgrad_x2", "mfma_tower_wgrad_x2"):
        assert tag in ran, (tag, sorted(ran))
    named = dict(net.named_parameters())
    bad = []
    for k, norm in zip(g["grad_all_keys"], g["grad_all_norms"]):
        gr = named[str(k)].grad
        assert gr is not None and bool(torch.isfinite(gr.float()).all()), k
        rel = abs(float(gr.double().norm()) - norm) / (norm + 1e-12)
        if rel > (0.25 if gr.dim() == 4 else 0.4):
            bad.append((str(k), rel))
    assert not bad, bad[:8]


@pytest.mark.gpu
def test_headline_shape_bf16_on_the_own_kernels_holds_the_reference(golden):
    """the HEADLINE configuration end to end against the reference itself (``e2e_full.npz``: R50-FPN, K = 90, two
    3 x 800 x 1333 images, 8 GT boxes each) in the headline numeric mode -- bf16 autocast, bf16 working copies.  At this size the
    assembled model runs its kernels at their real tile counts: the two-image canvas sheet of the towers, the dense 810-channel
    class-output conv on MFMA in all three directions, the band kernel (conv2 of layer2), the split-K dense kernel (conv2 of layer4), the
    fused bottleneck chain of layer1 / layer2, the stem -- asserted through the launch tags -- and the result is held to the reference's
    fp32 losses (2e-2 / 4e-2) and to every parameter's gradient norm at the bars of the small R50 fixture."""
    from pytorch_retinanet_amd import biasact, pwconv
    from pytorch_retinanet_amd.optim import use_16bit_conv_weights
    g = golden("e2e_full.npz")
    net = _model(g, DEV, FIXTURES["e2e_full.npz"]).train()
    assert use_16bit_conv_weights(net, torch.bfloat16) > 0
    images, targets = _inputs(g, DEV, "e2e_full.npz")
    biasact.MFMA_FLOP.clear(); pwconv.PW_FLOP.clear()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(images, targets)
        total = out["classification_loss"] + out["regression_loss"]
    got = np.array([float(out["classification_loss"].detach()), float(out["regression_loss"].detach())])
    np.testing.assert_allclose(got, g["train_losses"], rtol=4e-2)
    np.testing.assert_allclose(got[0], g["train_losses"][0], rtol=2e-2)
    total.backward()
    ran = set(biasact.MFMA_FLOP) | set(pwconv.PW_FLOP)
    for tag in ("stem_fwd", "stem_wgrad", "pw_conv1_fwd", "pw_conv3_fwd", "pw_conv3_bwd", "pw_block_out_conv1", "pw_conv1_dgrad_sums", "pw_conv1_wgrad",
                "mfma_tower_fwd_x2", "mfma_tower_dgrad_x2", "mfma_tower_wgrad_x2", "mfma_cls_output_fwd", "mfma_cls_output_dgrad", "mfma_cls_output_wgrad",
                "mfma_box_output_fwd", "mfma_box_output_dgrad", "mfma_box_output_wgrad", "mfma_conv2_band", "mfma_conv2_splitk", "mfma_conv2_narrow_fwd",
                "mfma_fpn_output_fwd_x3", "mfma_fpn_output_dgrad_x3", "mfma_fpn_output_wgrad_x3"):
        assert tag in ran, (tag, sorted(ran))
    assert net.retinanet_head.mfma_cls_output
    named = dict(net.named_parameters())
    bad, rels = [], []
    for k, norm in zip(g["grad_all_keys"], g["grad_all_norms"]):
        gr = named[str(k)].grad
        assert gr is not None and bool(torch.isfinite(gr.float()).all()), k
        rel = abs(float(gr.double().norm()) - norm) / (norm + 1e-12)
        rels.append((rel, str(k)))
        # (bars: the small R50 fixture's + a third -- measured here: median 0.025, p90 0.096, max 0.25; a per-channel BatchNorm gradient
        # is a sum over 8 400 - 537 600 positions of bf16 terms that nearly cancel)
        if rel > (0.33 if gr.dim() == 4 else 0.5):
            bad.append((str(k), rel))
    rv = np.array([r for r, _ in rels])
    print(f"[e2e_full bf16] losses {got} vs {g['train_losses']}; gradient-norm rel err median {np.median(rv):.4f} p90 {np.percentile(rv, 90):.4f} "
          f"max {rv.max():.4f}; worst {sorted(rels, reverse=True)[:3]}")
    assert not bad, bad[:8]
    assert np.median(rv) <= 0.06 and np.percentile(rv, 90) <= 0.2


@pytest.mark.gpu
@BOTH
def test_bf16_head_outputs_through_the_oracle_give_the_models_detections(golden, oracle_lib, fixture):
    """The detection chain (K4-K7) of the headline numeric configuration, exactly: the model's OWN bf16 head outputs (packed
    canvas, MFMA towers, dense class-output conv) fed to the CPU oracle's process_detections (reference models.py:160-243)
    must give the labels, boxes and scores ``process_detections_levels`` returns for them -- no tolerance on labels and order."""
    g = golden(fixture)
    net = _model(g, DEV, FIXTURES[fixture]).eval()
    images, _ = _inputs(g, DEV, fixture)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        il, _ = net.transform(images, None, **net._batch_layout())
        fmaps = net.fpn(net.backbone(il.tensors))
        anchors = net.anchor_generator(il, fmaps)
        lv = net.retinanet_head.forward_levels(fmaps)
        assert lv["cls_levels"][0].dtype == torch.bfloat16
        cls = torch.cat([c.float() for c in lv["cls_levels"]], 1).cpu().numpy()
        box = torch.cat([b_.float() for b_ in lv["bbox_levels"]], 1).cpu().numpy()
        dets = net.process_detections_levels(lv, anchors, il.image_sizes)
    ref = oracle_lib.detect(cls[..., :net.num_classes], box, anchors[0].cpu().numpy(), il.image_sizes)
    total = 0
    for got, r in zip(dets, ref):
        assert np.array_equal(got["labels"].cpu().numpy(), r["labels"])
        np.testing.assert_allclose(got["boxes"].cpu().numpy(), r["boxes"], rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(got["scores"].cpu().numpy(), r["scores"], rtol=0, atol=1e-6)
        total += len(r["labels"])
    assert total > 0


@pytest.mark.reference
@BOTH
def test_fixture_is_what_the_reference_computes_now(golden, fixture):
    """Build container only: re-run the reference on the regenerated weights and compare with the committed fixture
    (guards against a stale e2e.npz)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import _tv_standin
    R = _tv_standin.import_reference()
    g = golden(fixture)
    ref = R.Retinanet(**FIXTURES[fixture])
    sd = ref.state_dict()
    for k, v in _weights(g).items():
        sd[k] = torch.from_numpy(v)
    ref.load_state_dict(sd)
    images, targets = _inputs(g, "cpu", fixture)
    ref.eval()
    with torch.no_grad():
        out = ref(images, [{k: v.clone() for k, v in t.items()} for t in targets])
        dets = ref.predict(images) if fixture != "e2e_full.npz" else []      # (headline shape: the eval forward alone re-checks the fixture within the CPU suite's minutes)
    np.testing.assert_allclose([float(out["classification_loss"]), float(out["regression_loss"])], g["eval_losses"], rtol=1e-6)
    for b, d in enumerate(dets):
        assert np.array_equal(d["labels"].numpy(), g[f"det_labels{b}"])
        np.testing.assert_allclose(d["boxes"].numpy(), g[f"det_boxes{b}"], rtol=1e-5, atol=1e-4)


@pytest.mark.gpu
def test_headline_shape_fp16_loss_scale_reaches_the_class_head_gradients(golden):
    """ADVICE r5 (medium) at the model level, on the headline-shape fixture (K = 90, A = 201 600, logits around the prior): under fp16
    autocast the loss kernel stores d loss / d logits in fp16 in its forward pass.  With the GradScaler's scale handed to it
    (``losses.grad_prescale``, what ``CapturedTrainStep`` / ``SimpleTrainer`` do) the class head's parameter gradients match the reference's
    fp32 ones at the fp16 bar of the small fixture (5e-2); with the scale multiplied in only afterwards -- the round-5 path -- the
    background elements' gradients (~1e-8) are flushed to zero or rounded to a subnormal before the scale arrives, and the class-output
    BIAS gradient, which is the sum over 18 M mostly-background elements, is 400 x further from the reference's."""
    from pytorch_retinanet_amd import losses as L
    from pytorch_retinanet_amd.optim import use_16bit_conv_weights
    g = golden("e2e_full.npz")
    S = 65536.0
    ref = {str(k): float(n) for k, n in zip(g["grad_all_keys"], g["grad_all_norms"])}
    head = [k for k in ref if "classification_head" in k]
    assert len(head) == 10
    errs = {}
    for mode in ("prescaled", "scaled_afterwards"):
        net = _model(g, DEV, FIXTURES["e2e_full.npz"]).train()
        use_16bit_conv_weights(net, torch.float16)
        images, targets = _inputs(g, DEV, "e2e_full.npz")
        scale = torch.full((1,), S, device=DEV)
        with torch.autocast("cuda", dtype=torch.float16), L.grad_prescale(scale if mode == "prescaled" else None):
            out = net(images, targets)
            total = out["classification_loss"] + out["regression_loss"]
        (total * scale[0]).backward()
        named = dict(net.named_parameters())
        errs[mode] = {k: abs(float(named[k].grad.double().norm()) / S - ref[k]) / ref[k] for k in head}
        assert all(bool(torch.isfinite(named[k].grad.float()).all()) for k in head)
    bias = "retinanet_head.classification_head.class_subnet_output.bias"
    print(f"[fp16 class head] prescaled: max rel err {max(errs['prescaled'].values()):.4f} (bias {errs['prescaled'][bias]:.4f}); "
          f"scaled afterwards: max {max(errs['scaled_afterwards'].values()):.4f} (bias {errs['scaled_afterwards'][bias]:.4f})")
    assert max(errs["prescaled"].values()) <= 5e-2, errs["prescaled"]
    # (measured at this fixture's B = 2, where a background gradient is 4 x the bench shape's: bias 4.5e-5 prescaled against 1.7e-2)
    assert errs["prescaled"][bias] <= 1e-3 and errs["scaled_afterwards"][bias] > 5e-3, (errs["scaled_afterwards"][bias], errs["prescaled"][bias])
