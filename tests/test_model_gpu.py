"""GPU: the reference's Python surface end to end on the HIP path (autograd, Retinanet, RetinaNetModel)."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _toy(golden):
    g = golden("loss.npz")
    gtb = [g["toy_gtb0"], np.zeros((0, 4), np.float32), g["toy_gtb2"]]
    gtl = [g["toy_gtl0"], np.zeros((0,), np.int64), g["toy_gtl2"]]
    targets = [{"boxes": torch.from_numpy(b).to(DEV), "labels": torch.from_numpy(l).to(DEV)} for b, l in zip(gtb, gtl)]
    return g, targets


def test_retinanet_losses_module_autograd_matches_reference(golden):
    """RetinaNetLosses.forward + loss.backward() == the reference's losses and autograd gradients (toy golden)."""
    import pytorch_retinanet_amd as P
    g, targets = _toy(golden)
    cls = torch.from_numpy(g["toy_cls"]).to(DEV).requires_grad_(True)
    box = torch.from_numpy(g["toy_box"]).to(DEV).requires_grad_(True)
    anc = torch.from_numpy(g["toy_anchors"]).to(DEV)
    crit = P.RetinaNetLosses(3)
    out = crit(targets, {"cls_preds": cls, "bbox_preds": box}, [anc] * 3)
    assert set(out) == {"classification_loss", "regression_loss"}
    np.testing.assert_allclose([float(out["classification_loss"]), float(out["regression_loss"])], g["toy_loss"], rtol=1e-5)
    (out["classification_loss"] + out["regression_loss"]).backward()
    np.testing.assert_allclose(cls.grad.cpu().numpy(), g["toy_gcls"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(box.grad.cpu().numpy(), g["toy_gbox"], rtol=1e-5, atol=1e-7)
    # upstream gradients other than 1 (and different per loss) go through the device-side scale
    cls2 = cls.detach().clone().requires_grad_(True)
    box2 = box.detach().clone().requires_grad_(True)
    out = crit(targets, {"cls_preds": cls2, "bbox_preds": box2}, [anc] * 3)
    (3.0 * out["classification_loss"] - 0.5 * out["regression_loss"]).backward()
    np.testing.assert_allclose(cls2.grad.cpu().numpy(), 3.0 * g["toy_gcls"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(box2.grad.cpu().numpy(), -0.5 * g["toy_gbox"], rtol=1e-5, atol=1e-7)
    # calc_loss: one image, returns (bb_loss, clas_loss) like the reference
    bb, cl = crit.calc_loss(anc, cls.detach()[0], box.detach()[0], targets[0]["labels"], targets[0]["boxes"])
    np.testing.assert_allclose([float(bb), float(cl)], g["toy_per_image"][0], rtol=1e-5)
    bb, cl = crit.calc_loss(anc, cls.detach()[1], box.detach()[1], targets[1]["labels"], targets[1]["boxes"])
    assert float(bb) == 0.0 and float(cl) == 0.0                       # Q7


def test_box_utils_surface(golden):
    import pytorch_retinanet_amd as P
    g = golden("match.npz")
    m = P.matcher(torch.from_numpy(g["hand_anchors"]).to(DEV), torch.from_numpy(g["hand_gt"]).to(DEV))
    assert m.dtype == torch.int64 and m.tolist() == [-2, 2, -2, -1, 0, -2, -1, 2]
    assert P.matcher(torch.from_numpy(g["hand_anchors"]).to(DEV), torch.zeros((0, 4), device=DEV)).tolist() == [-2] * 8
    d = golden("decode.npz")
    dec = P.activ_2_bbox(torch.from_numpy(d["deltas"]).to(DEV), torch.from_numpy(d["anchors"]).to(DEV))
    np.testing.assert_allclose(dec.cpu().numpy(), d["decoded"], rtol=1e-5, atol=1e-3)
    enc = P.bbox_2_activ(torch.from_numpy(d["enc_gt"]).to(DEV), torch.from_numpy(d["anchors"]).to(DEV))
    np.testing.assert_allclose(enc.cpu().numpy(), d["encoded"], rtol=1e-5, atol=1e-6)


def test_anchor_generator_module_on_device(golden):
    import pytorch_retinanet_amd as P
    from pytorch_retinanet_amd.transform import ImageList
    g = golden("anchors.npz")
    ag = P.AnchorGenerator().to(DEV)
    fmaps = [torch.zeros(2, 256, h, w, device=DEV) for h, w in synth.fpn_grid_sizes(512, 512)]
    out = ag(ImageList(torch.zeros(2, 3, 512, 512, device=DEV), [(512, 512), (480, 500)]), fmaps)
    assert len(out) == 2 and out[0] is out[1]                          # one cached tensor per shape set (Q12)
    assert synth.sha(out[0].cpu().numpy()) == str(g["r18_512_sha"])
    per_level = ag.grid_anchors([f.shape[-2:] for f in fmaps], torch.device(DEV))
    assert [p.shape[0] for p in per_level] == [36864, 9216, 2304, 576, 144]


def _head_outputs_as_the_kernels_see_them(net, images, targets=None):
    """(cls [B,A,K] f32, box [B,A,4] f32, anchors, transformed targets, image_sizes) from the per-level path the loss / detect
    kernels read (packed canvas, MFMA towers under bf16, dead classes stripped)."""
    K = net.num_classes
    il, tg = net.transform(images, targets, **net._batch_layout())
    fmaps = net.fpn(net.backbone(il.tensors))
    lv = net.retinanet_head.forward_levels(fmaps)
    cls = torch.cat([c[..., :K] for c in lv["cls_levels"]], 1).float()
    box = torch.cat(list(lv["bbox_levels"]), 1).float()
    return cls, box, net.anchor_generator(il, fmaps)[0], tg, il.image_sizes


def test_retinanet_train_step_and_predict_r18(oracle_lib):
    """BASELINE configs[0] (R18-FPN, 2 x 3x512x512): forward + loss + backward, then predict, on the device -- the loss
    dict and the detections are held to the CPU oracle evaluated on the model's own head outputs."""
    import pytorch_retinanet_amd as P
    from test_e2e_gpu import box_set_agreement
    torch.manual_seed(0)
    net = P.Retinanet(backbone_kind="resnet18", pretrained=False, min_size=512, max_size=512).to(DEV)
    net = net.to(memory_format=torch.channels_last).train()
    rng = np.random.default_rng(0)
    images = [torch.rand(3, 512, 512, device=DEV) for _ in range(2)]
    targets = []
    for T in (2, 1):
        b, l = synth.gt_boxes(rng, T, 512, 512)
        targets.append({"boxes": torch.from_numpy(b).to(DEV), "labels": torch.from_numpy(l).to(DEV)})
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(images, targets)
    loss = sum(out.values())
    loss.backward()
    assert torch.isfinite(loss) and out["classification_loss"].dtype == torch.float32
    g = net.retinanet_head.classification_head.class_subnet_output.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0
    assert net.backbone.backbone.conv1.weight.grad.abs().sum() > 0
    # the same forward again (train-mode BN uses batch statistics, so the head outputs repeat up to conv rounding): oracle losses on them
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        cls, box, anchors, tg, _ = _head_outputs_as_the_kernels_see_them(net, images, targets)
    a_np = anchors.cpu().numpy()
    gtb = [t["boxes"].cpu().numpy() for t in tg]
    gtl = [t["labels"].cpu().numpy() for t in tg]
    m, nfg = oracle_lib.iou_match(a_np, gtb)
    ref = oracle_lib.loss_fwd_bwd(cls.cpu().numpy(), box.cpu().numpy(), a_np, gtb, gtl, m)
    assert nfg.sum() > 0
    # (1) the loss kernels on exactly these head outputs == the oracle; (2) Retinanet.forward's own loss dict agrees with it up
    # to what a repeated forward can differ (MIOpen's convolutions are not bit-reproducible from call to call)
    got = net.retinanet_head.losses(tg, {"cls_preds": cls, "bbox_preds": box}, [anchors, anchors])
    np.testing.assert_allclose([float(got["classification_loss"]), float(got["regression_loss"])], ref["loss"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose([float(out["classification_loss"].detach()), float(out["regression_loss"].detach())], ref["loss"], rtol=2e-3)
    net.eval()
    with torch.no_grad():
        # an untrained head scores every anchor at the 0.01 prior: spread the logits, then put the score threshold where
        # about 3000 anchors per image pass it, so the scan, NMS and top-k all have work
        net.retinanet_head.classification_head.class_subnet_output.weight.normal_(0.0, 1.0)
        cls, box, anchors, _, hw = _head_outputs_as_the_kernels_see_them(net, images)
        top = torch.topk(torch.sigmoid(cls).reshape(-1), 6001).values
        thr = float((top[5999].double() + top[6000].double()) / 2) if top[5999] > top[6000] else float(top[6000])
    net.score_thres = thr
    dets = net.predict(images)
    assert len(dets) == 2
    ref_d = oracle_lib.detect(cls.cpu().numpy(), box.cpu().numpy(), anchors.cpu().numpy(), [tuple(int(x) for x in s) for s in hw],
                              oracle_lib.default_detect_params(thr, 1e-2, net.nms_thres, net.detections_per_img))
    for d, r in zip(dets, ref_d):
        assert d["boxes"].shape[1] == 4 and d["labels"].dtype == torch.int64 and len(d["scores"]) <= 100
        assert (d["scores"][:-1] >= d["scores"][1:]).all()
        if len(d["labels"]):
            assert d["labels"].min() >= 1 and d["labels"].max() <= 90
        got = {k: v.cpu().numpy() for k, v in d.items()}
        assert len(r["labels"]) > 0 and box_set_agreement(got, r, 0.99) >= 0.95       # original size == resized size: no rescale
    assert net(images) is not None                                     # targets=None -> predict (Q19)


def test_lightning_module_with_simple_trainer():
    import pytorch_retinanet_amd as P
    conf = P.load_hparams()
    conf.model.update(backbone_kind="resnet18", pretrained=False, num_classes=5, min_size=128, max_size=160)
    conf.dataset.kind = "synthetic"
    conf.dataset.update(length=4, height=128, width=160, boxes_per_image=3)
    conf.dataloader.train_bs = 2
    conf.dataloader.valid_bs = 2
    conf.dataloader.test_bs = 2
    conf.dataloader.args.pin_memory = False
    model = P.RetinaNetModel(conf)
    trainer = P.SimpleTrainer(max_epochs=1, device=DEV)
    before = model.net.retinanet_head.regression_head.box_subnet_output.weight.detach().clone()
    steps = trainer.fit(model)
    assert steps == 2
    after = model.net.retinanet_head.regression_head.box_subnet_output.weight.detach().cpu()
    assert not torch.equal(before, after)
    res, outs = trainer.test(model)
    assert len(outs) == 2 and all("detections" in o for o in outs)


def test_simple_trainer_replays_the_captured_step():
    """the training path itself (``SimpleTrainer.fit``) runs the step bench.py measures -- ``graph.CapturedTrainStep``:
    two eager steps per input signature, then hipGraph replays; the weights move as in the eager trainer (same data, same seed)."""
    import pytorch_retinanet_amd as P

    def run(capture):
        torch.manual_seed(7)
        conf = P.load_hparams()
        conf.model.update(backbone_kind="resnet18", pretrained=False, num_classes=5, min_size=128, max_size=160)
        conf.dataset.kind = "synthetic"
        conf.dataset.update(length=12, height=128, width=160, boxes_per_image=3)
        conf.dataloader.train_bs = 2
        conf.dataloader.valid_bs = 2
        conf.dataloader.args.pin_memory = False
        model = P.RetinaNetModel(conf)                     # (same seed: same weights and the same shuffled order in both runs)
        trainer = P.SimpleTrainer(max_epochs=1, device=DEV, capture=capture)
        steps = trainer.fit(model)
        w = model.net.retinanet_head.classification_head.class_subnet_output.bias.detach().float().cpu()
        return steps, trainer.captured_steps, w

    steps, replays, w_graph = run(True)
    assert steps == 6 and replays >= 3, (steps, replays)
    steps_e, replays_e, w_eager = run(False)
    assert steps_e == 6 and replays_e == 0
    assert bool(torch.isfinite(w_graph).all())
    # same trajectory up to the run-to-run noise of the bf16 conv stack (a captured step IS the eager step: tests/test_graph_gpu.py)
    assert float((w_graph - w_eager).abs().max()) <= 5e-3 * max(1.0, float(w_eager.abs().max())), float((w_graph - w_eager).abs().max())


def test_test_step_through_the_evaluator_equals_bbox_eval_on_the_oracles_detections(oracle_lib):
    """The evaluation path end to end (reference model.py:132-146: test_step -> CocoEvaluator.update -> test_epoch_end ->
    accumulate / summarize -> stats[0]) on GPU detections, against ``BBoxEval`` fed the CPU oracle's detections
    (reference models.py:160-243 restated) for the same head outputs: all 12 COCO statistics agree.  The ground truth is a
    few of the model's own detections (so the statistics are not all zero) plus boxes it cannot find.  (``BBoxEval`` itself is
    pinned to the published COCO protocol by hand-derived cases, not to a pycocotools run: tests/test_coco_eval.py.)"""
    import pytorch_retinanet_amd as P
    from pytorch_retinanet_amd.coco_eval import BBoxEval, gt_from_dataset, prepare_for_coco_detection
    conf = P.load_hparams()
    conf.model.update(backbone_kind="resnet18", pretrained=False, num_classes=5, min_size=128, max_size=160)
    conf.dataset.kind = "synthetic"
    conf.dataset.update(length=4, height=128, width=160, boxes_per_image=3)
    conf.dataloader.test_bs = 2
    conf.dataloader.args.pin_memory = False
    torch.manual_seed(21)
    model = P.RetinaNetModel(conf).to(DEV).eval()
    with torch.no_grad():                                        # scores around 0.1 .. 0.5 instead of the 0.01 prior: detections exist
        model.net.retinanet_head.classification_head.class_subnet_output.bias.fill_(-1.0)
        model.net.retinanet_head.classification_head.class_subnet_output.weight.mul_(30.0)
    g = torch.Generator().manual_seed(3)
    images = [torch.rand(3, 128, 160, generator=g) for _ in range(4)]
    with torch.no_grad():
        first = model.net.predict([im.to(DEV) for im in images])
    assert all(len(d["labels"]) >= 6 for d in first)

    class DS(torch.utils.data.Dataset):
        def __len__(self): return 4
        def __getitem__(self, i):
            d = first[i]
            pick = [0, 3, 5]                                     # three of its own detections + one box nothing will match
            boxes = torch.cat([d["boxes"][pick].cpu(), torch.tensor([[2.0, 2.0, 9.0, 9.0]])])
            labels = torch.cat([d["labels"][pick].cpu(), torch.tensor([1])])
            return images[i], {"boxes": boxes, "labels": labels, "image_id": torch.tensor([10 + i])}, 10 + i
    model.test_ds = DS()
    res, outs = P.SimpleTrainer(device=DEV, precision="32").test(model)
    stats = np.asarray(model.test_evaluator.coco_eval["bbox"].stats, dtype=np.float64)
    assert stats.shape == (12,) and float(res["AP"]) == pytest.approx(stats[0])
    # the oracle's detections for the same head outputs
    rows = []
    with torch.no_grad():
        for lo in (0, 2):
            ims = [im.to(DEV) for im in images[lo:lo + 2]]
            il, _ = model.net.transform(ims, None)
            fm, out = model.net._features(il.tensors)
            anchors = model.net.anchor_generator(il, fm)[0].cpu().numpy()
            ref = oracle_lib.detect(out["cls_preds"].float().cpu().numpy(), out["bbox_preds"].float().cpu().numpy(), anchors, il.image_sizes)
            rows += prepare_for_coco_detection({10 + lo + j: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in r.items()}
                                                for j, r in enumerate(ref)})
    want = BBoxEval(gt_from_dataset(model.test_ds), rows).evaluate().summarize(verbose=False)
    np.testing.assert_allclose(stats, np.asarray(want, dtype=np.float64), rtol=0, atol=2e-3)
    assert 0.0 < stats[0] < 1.0 and stats[8] > 0.0              # a real precision / recall curve, not a degenerate one


def test_per_level_loss_equals_concatenated(golden):
    """rn_loss_fwd_bwd_levels (head outputs left per pyramid level) == rn_loss_fwd_bwd on their concatenation:
    values and gradients bit-identical per element, incl. ragged level sizes and an image without GT."""
    import pytorch_retinanet_amd as P
    rng = np.random.default_rng(12)
    B, K = 3, 7
    counts = [333, 90, 27, 9, 2]                     # odd sizes: every level has a ragged tail somewhere
    A = sum(counts)
    anc = np.sort(rng.uniform(0, 300, (A, 2, 2)).astype(np.float32), axis=1).reshape(A, 4)
    targets = []
    for T in (5, 0, 3):
        b, l = synth.gt_boxes(rng, T, 300, 300, num_classes=K, wh_lo=30, wh_hi=200)
        targets.append({"boxes": torch.from_numpy(b).to(DEV), "labels": torch.from_numpy(l).to(DEV)})
    crit = P.RetinaNetLosses(K)
    for dtype in (torch.float32, torch.bfloat16):
        cls_l = [torch.randn(B, n, K, device=DEV).mul(1.5).sub(1.0).to(dtype).requires_grad_(True) for n in counts]
        box_l = [torch.randn(B, n, 4, device=DEV).mul(0.4).to(dtype).requires_grad_(True) for n in counts]
        anchors = torch.from_numpy(anc).to(DEV)
        out_l = crit.forward_levels(targets, cls_l, box_l, [anchors] * B)
        (out_l["classification_loss"] + 2.0 * out_l["regression_loss"]).backward()
        cls_c = torch.cat([c.detach() for c in cls_l], 1).requires_grad_(True)
        box_c = torch.cat([b.detach() for b in box_l], 1).requires_grad_(True)
        out_c = crit(targets, {"cls_preds": cls_c, "bbox_preds": box_c}, [anchors] * B)
        (out_c["classification_loss"] + 2.0 * out_c["regression_loss"]).backward()
        np.testing.assert_allclose(float(out_l["classification_loss"]), float(out_c["classification_loss"]), rtol=1e-6)
        np.testing.assert_allclose(float(out_l["regression_loss"]), float(out_c["regression_loss"]), rtol=1e-6)
        assert torch.equal(torch.cat([c.grad for c in cls_l], 1), cls_c.grad)
        assert torch.equal(torch.cat([b.grad for b in box_l], 1), box_c.grad)


def test_train_path_uses_level_outputs_and_matches_cat_path():
    """Retinanet.forward (per-level loss path) == compute_loss on the concatenated head outputs."""
    import pytorch_retinanet_amd as P
    torch.manual_seed(1)
    net = P.Retinanet(num_classes=6, backbone_kind="resnet18", pretrained=False, min_size=128, max_size=160).to(DEV)
    net = net.to(memory_format=torch.channels_last).train()
    images = [torch.rand(3, 128, 160, device=DEV) for _ in range(2)]
    targets = [{"boxes": torch.tensor([[10., 12., 90., 100.]], device=DEV), "labels": torch.tensor([2], device=DEV)},
               {"boxes": torch.tensor([[20., 20., 70., 110.], [60., 30., 150., 120.]], device=DEV), "labels": torch.tensor([3, 6], device=DEV)}]
    net.eval()   # same BN statistics for both passes
    a = net(images, targets)
    il, tg = net.transform(images, targets)
    fmaps, outputs = net._features(il.tensors)
    b = net.compute_loss(tg, outputs, net.anchor_generator(il, fmaps))
    for k in a:
        np.testing.assert_allclose(float(a[k]), float(b[k]), rtol=1e-5)
    lv = net.retinanet_head.forward_levels(fmaps)["cls_levels"][0]
    assert lv.is_contiguous()          # a view of the channels_last conv output: nothing was copied


def test_transform_module_fused_equals_torch_path():
    """GeneralizedRCNNTransform: CUDA images take the one-launch HIP path; it must agree with the module's
    own PyTorch-op path (what CPU tensors take) on sizes, rescaled GT boxes and pixels."""
    from pytorch_retinanet_amd.transform import GeneralizedRCNNTransform
    torch.manual_seed(3)
    imgs = [torch.rand(3, 120, 180), torch.rand(3, 200, 150), torch.rand(3, 128, 128)]
    tgts = [{"boxes": torch.tensor([[10.0, 20.0, 90.0, 100.0]]), "labels": torch.tensor([1])} for _ in imgs]
    t = GeneralizedRCNNTransform(128, 192, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]).eval()
    ref, rt = t(imgs, tgts)
    got, gt = t([i.cuda() for i in imgs], [{k: v.cuda() for k, v in d.items()} for d in tgts])
    assert got.image_sizes == ref.image_sizes and got.tensors.shape == ref.tensors.shape
    torch.testing.assert_close(got.tensors.cpu(), ref.tensors, rtol=0, atol=2e-5)
    for a, b in zip(gt, rt):
        torch.testing.assert_close(a["boxes"].cpu(), b["boxes"])
    cl, _ = t([i.cuda() for i in imgs], None, out_dtype=torch.bfloat16, channels_last=True)
    assert cl.tensors.dtype == torch.bfloat16 and cl.tensors.is_contiguous(memory_format=torch.channels_last)
    torch.testing.assert_close(cl.tensors.float().cpu(), ref.tensors, rtol=1e-2, atol=1e-2)


def test_padded_class_head_matches_reference_shaped_head():
    """forward_levels(pad_classes=True): real columns == the unpadded conv (fp32: 1e-5), dead columns == -80,
    gradients reach the [A*K, C, 3, 3] parameters with zero contribution from the dead classes."""
    import pytorch_retinanet_amd as P
    torch.manual_seed(4)
    net = P.Retinanet(num_classes=6, backbone_kind="resnet18", pretrained=False, min_size=128, max_size=160).to(DEV)
    net = net.to(memory_format=torch.channels_last).eval()
    head = net.retinanet_head.classification_head
    assert head.padded_classes == 8
    fm = [torch.randn(2, 256, 16, 20, device=DEV).contiguous(memory_format=torch.channels_last),
          torch.randn(2, 256, 8, 10, device=DEV).contiguous(memory_format=torch.channels_last)]
    plain = head.forward_levels(fm)
    padded = head.forward_levels(fm, pad_classes=True)
    for p_, q_ in zip(plain, padded):
        assert q_.shape == (*p_.shape[:2], 8) and q_.is_contiguous()
        torch.testing.assert_close(q_[..., :6], p_, rtol=1e-5, atol=1e-5)
        assert torch.equal(q_[..., 6:], torch.full_like(q_[..., 6:], -80.0))
    g0 = torch.autograd.grad(sum((p_ ** 2).sum() for p_ in plain), head.class_subnet_output.weight)[0]
    g1 = torch.autograd.grad(sum((q_[..., :6] ** 2).sum() for q_ in padded), head.class_subnet_output.weight)[0]
    torch.testing.assert_close(g1, g0, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("relu,use_mask", [(True, True), (True, False), (False, False), (False, True)])
def test_bias_act_kernel_vs_torch(dtype, relu, use_mask):
    from pytorch_retinanet_amd import biasact
    torch.manual_seed(0)
    N, Cc, H, W = 3, 64, 13, 21
    x = torch.randn(N, Cc, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bias = torch.randn(Cc, device=DEV, requires_grad=True)
    mask = (torch.rand(H * W, device=DEV) > 0.3).to(torch.uint8) if use_mask else None
    assert biasact.fusable(x, bias)
    y = biasact.bias_act(x, bias, mask, relu)
    g = torch.randn_like(y)
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    br = bias.detach().clone().requires_grad_(True)
    yr = xr + br[None, :, None, None]
    if relu:
        yr = torch.relu(yr)
    if mask is not None:
        yr = yr * mask.view(1, 1, H, W).float()
    yr.backward(g.float())
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(y.float(), yr, **tol)
    # the ReLU decision is taken on the rounded y; compare dx only where the reference's y is not within rounding of 0
    sure = (yr.abs() > 1e-2) | (yr == 0) if dtype != torch.float32 else torch.ones_like(yr, dtype=torch.bool)
    torch.testing.assert_close(x.grad.float()[sure], xr.grad[sure], **tol)
    torch.testing.assert_close(bias.grad, br.grad, rtol=2e-2 if dtype != torch.float32 else 1e-5, atol=5e-2 if dtype != torch.float32 else 1e-5)


def test_head_on_canvas_equals_per_level_head():
    """forward_levels(canvas=True) == forward_levels(canvas=False): same logits / deltas (fp32: 1e-4) and the same
    parameter gradients -- the packed canvas must not leak between levels."""
    import pytorch_retinanet_amd as P
    from pytorch_retinanet_amd import biasact
    torch.manual_seed(5)
    net = P.Retinanet(num_classes=6, backbone_kind="resnet18", pretrained=False, min_size=128, max_size=160).to(DEV)
    net = net.to(memory_format=torch.channels_last).eval()
    head = net.retinanet_head
    for p in head.parameters():                     # the reference's init (std 0.01) makes everything tiny: use O(1) weights
        if p.dim() == 4:
            torch.nn.init.normal_(p, std=0.03)
    shapes = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    fm = [torch.randn(2, 256, h, w, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True) for h, w in shapes]
    cv = biasact.Canvas.of(fm)
    assert cv.W == 20 and cv.slots == 1 and int(cv.mask.sum()) == sum(h * w for h, w in shapes)      # (no border: the MIOpen canvas)
    outs = {}
    for canvas in (False, True):
        head.zero_grad()
        for f in fm:
            f.grad = None
        o = head.forward_levels(fm, pad_classes=True, canvas=canvas)
        loss = sum((c.float() ** 2).sum() for c in o["cls_levels"]) * 1e-4 + sum((b.float() ** 2).sum() for b in o["bbox_levels"])
        loss.backward()
        outs[canvas] = (o, [f.grad.clone() for f in fm], {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None})
    for key in ("cls_levels", "bbox_levels"):
        for a, b in zip(outs[False][0][key], outs[True][0][key]):
            assert a.shape == b.shape
            torch.testing.assert_close(b, a, rtol=1e-4, atol=1e-4)
    for a, b in zip(outs[False][1], outs[True][1]):
        torch.testing.assert_close(b, a, rtol=1e-3, atol=1e-4)
    for n, a in outs[False][2].items():
        torch.testing.assert_close(outs[True][2][n], a, rtol=1e-3, atol=1e-3)


def _canvas_conv_reference(x, w, b, mask2d):
    y = torch.relu(torch.nn.functional.conv2d(x.float(), w.float(), b, padding=1))
    return y * mask2d[None, None].float()


@pytest.mark.parametrize("shape", [(2, 64, 256, 9, 11), (1, 256, 256, 14, 37), (3, 512, 256, 6, 8), (2, 256, 512, 5, 7),
                                   (1, 256, 256, 250, 300)])      # 293 row tiles: more than one round of workgroups
def test_mfma_canvas_conv_forward_backward_vs_torch(shape):
    """rn_conv3x3_canvas (forward + data gradient) and the MIOpen weight gradient, against torch's conv2d in fp32 on
    the same bf16 values; zero-bordered canvas with a random interior mask (gaps)."""
    from pytorch_retinanet_amd import biasact
    torch.manual_seed(0)
    N, Cin, Cout, Hp, Wp = shape
    mask2d = torch.zeros(Hp, Wp, dtype=torch.uint8, device=DEV)
    mask2d[1:-1, 1:-1] = (torch.rand(Hp - 2, Wp - 2, device=DEV) > 0.2).to(torch.uint8)
    x = (torch.randn(N, Cin, Hp, Wp, device=DEV) * mask2d[None, None]).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, device=DEV) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = (torch.randn(Cout, device=DEV) * 0.1).requires_grad_(True)
    y = biasact.tower_conv(x, w, b, mask2d.reshape(-1))
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    g = (torch.randn_like(y) * mask2d[None, None]).to(torch.bfloat16)
    y.backward(g)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)       # the kernel sees the bf16-rounded weights
    br = b.detach().clone().requires_grad_(True)
    yr = _canvas_conv_reference(xr, wr, br, mask2d)
    # ONE bf16 rounding of the fp32 result (the old bars -- 2e-2 relative + 1e-2 of the maximum, 5e-2 + 3e-2 on the
    # gradients -- would have let a wrong ragged-tile column through): 2^-9 per element => ~1.1e-3 in L2; bars 2.5e-3 in L2 and 2^-8 of
    # the tensor's largest magnitude per element, as tests/test_fp16_kernels_gpu.py does at fp16
    def one_rounding(got, ref, what):
        got, ref = got.float(), ref.float()
        rel = float((got - ref).norm() / (ref.norm() + 1e-20))
        err, top = float((got - ref).abs().max()), float(ref.abs().max())
        assert rel < 2.5e-3 and err <= 2.0 ** -8 * top + 1e-6, (what, shape, rel, err, top)
    one_rounding(y, yr.detach(), "forward")
    assert not y.float()[:, :, mask2d == 0].any()                         # border and gaps are exact zeros
    # gradients: through the ReLU decisions the KERNEL took (y > 0 on its bf16 output) -- a pre-activation within rounding of 0
    # may fall on the other side in fp32, and at 75 000 positions x 256 channels a few always do
    pre = torch.nn.functional.conv2d(xr, wr, br, padding=1)
    (pre * (y.detach().float() > 0) * mask2d[None, None].float()).backward(g.float())
    interior = mask2d[None, None].bool().expand_as(xr.grad)
    one_rounding(x.grad.float()[interior], xr.grad[interior], "data gradient")
    one_rounding(w.grad, wr.grad, "weight gradient")
    one_rounding(b.grad, br.grad, "bias gradient")


def test_head_mfma_towers_equal_miopen_towers_bf16():
    """Under bf16 autocast the canvas towers run on the MFMA conv; same logits / deltas / parameter gradients as the
    MIOpen + fused-epilogue towers (bf16 tolerance)."""
    import pytorch_retinanet_amd as P
    torch.manual_seed(7)
    net = P.Retinanet(num_classes=6, backbone_kind="resnet18", pretrained=False, min_size=128, max_size=160).to(DEV)
    net = net.to(memory_format=torch.channels_last).eval()
    head = net.retinanet_head
    for p in head.parameters():
        if p.dim() == 4:
            torch.nn.init.normal_(p, std=0.03)
    shapes = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    fm = [torch.randn(2, 256, h, w, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
          for h, w in shapes]
    outs = {}
    for mfma in (False, True):
        head.mfma_towers = mfma
        head.zero_grad()
        for f in fm:
            f.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            o = head.forward_levels(fm, pad_classes=True, canvas=True)
        loss = sum((c.float() ** 2).sum() for c in o["cls_levels"]) * 1e-4 + sum((b.float() ** 2).sum() for b in o["bbox_levels"])
        loss.backward()
        outs[mfma] = (o, [f.grad.clone() for f in fm], {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None})
    head.mfma_towers = True
    K = head.classification_head.num_classes
    assert outs[True][0]["cls_levels"][0].shape[-1] == K            # MFMA class-output conv: dense logits, no dead classes
    assert outs[False][0]["cls_levels"][0].shape[-1] == 8           # MIOpen path: class dimension padded to a multiple of 8
    for key in ("cls_levels", "bbox_levels"):
        for a, b in zip(outs[False][0][key], outs[True][0][key]):
            a = a[..., :K] if key == "cls_levels" else a
            assert a.shape == b.shape
            torch.testing.assert_close(b.float(), a.float(), rtol=5e-2, atol=5e-2 * float(a.float().abs().max()))
    for a, b in zip(outs[False][1], outs[True][1]):
        torch.testing.assert_close(b.float(), a.float(), rtol=1e-1, atol=5e-2 * float(a.float().abs().max()))
    for n, a in outs[False][2].items():
        torch.testing.assert_close(outs[True][2][n].float(), a.float(), rtol=1e-1, atol=5e-2 * float(a.float().abs().max()))


@pytest.mark.parametrize("N,shapes", [(2, [(9, 12), (5, 6), (3, 3)]), (3, [(9, 12), (5, 6), (3, 3)]),
                                      (2, [(120, 150), (60, 75), (30, 38)])])     # 2 x ~170 row tiles: more than one round of workgroups
def test_tower_relu_backward_fused_into_the_data_gradient_kernel(N, shapes):
    """A chain of tower_conv_pair layers with TowerLink hand-over (the ReLU backward and bias gradient of layer l - 1 ride in
    layer l's data-gradient kernel: rn_conv3x3_canvas_dgrad_relu_batched) == the same chain with the separate
    rn_bias_act_backward passes: input gradients bit-equal, weight gradients bit-equal, bias gradients to fp32 summation
    order; and a layer whose output has a second consumer falls back to the separate pass."""
    from pytorch_retinanet_amd import biasact
    torch.manual_seed(3)
    feats = [torch.randn(N, 256, h, w, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for h, w in shapes]
    cv = biasact.Canvas.of(feats, pad=1)
    ws = [[(torch.randn(256, 256, 3, 3, device=DEV) * 0.02).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
           for _ in range(2)] for _ in range(3)]
    bs = [[(torch.randn(256, device=DEV) * 0.1).requires_grad_(True) for _ in range(2)] for _ in range(3)]

    def run(linked, tap_middle=False):
        x = biasact.pack_levels(cv, feats).requires_grad_(True)
        for t in [w for l in ws for w in l] + [b for l in bs for b in l]:
            t.grad = None
        a = b = x
        prev, extra = None, 0.0
        for i in range(3):
            link = biasact.TowerLink() if (linked and i < 2) else None
            a, b = biasact.tower_conv_pair(a, b, ws[i][0], ws[i][1], bs[i][0], bs[i][1], cv.mask, prev if linked else None, link)
            prev = link
            if tap_middle and i == 1:
                extra = (a.float() * 0.5).sum()                  # a second consumer of layer 1's output
        ((a.float() ** 2).sum() + (b.float() ** 3).sum() + extra).backward()
        return x.grad.clone(), [[w.grad.clone() for w in l] for l in ws], [[t.grad.clone() for t in l] for l in bs]

    ref = run(False)
    got = run(True)
    assert torch.equal(got[0], ref[0])
    for l in range(3):
        for k in range(2):
            assert torch.equal(got[1][l][k], ref[1][l][k]), (l, k)
            torch.testing.assert_close(got[2][l][k], ref[2][l][k], rtol=1e-4, atol=1e-4 * float(ref[2][l][k].abs().max()))
    ref2, got2 = run(False, True), run(True, True)                # autograd sums two gradients for layer 1's output: no hand-over there
    assert torch.equal(got2[0], ref2[0])
    for l in range(3):
        for k in range(2):
            torch.testing.assert_close(got2[2][l][k], ref2[2][l][k], rtol=1e-4, atol=1e-4 * float(ref2[2][l][k].abs().max()))
            assert torch.equal(got2[1][l][k], ref2[1][l][k]), (l, k)


@pytest.mark.parametrize("N,shapes", [(2, [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]), (3, [(40, 31), (7, 9)])])
def test_first_tower_layer_sums_both_data_gradients_in_one_launch(N, shapes):
    """``rn_conv3x3_canvas_sum2``: both towers read the same canvas, so its gradient is dgrad(cls) + dgrad(box).  One launch with a two-source
    contraction (f32 sum, one rounding) against the two-output launch + autograd's bf16 add, and against fp32 convolutions of the same values;
    weight / bias gradients are untouched (bit-equal)."""
    from pytorch_retinanet_amd import biasact
    torch.manual_seed(4)
    feats = [torch.randn(N, 256, h, w, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for h, w in shapes]
    cv = biasact.Canvas.of(feats, pad=1)
    ws = [(torch.randn(256, 256, 3, 3, device=DEV) * 0.02).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(2)]
    bs = [(torch.randn(256, device=DEV) * 0.1).requires_grad_(True) for _ in range(2)]
    x0 = biasact.pack_levels(cv, feats)
    g0, g1 = torch.randn_like(x0), torch.randn_like(x0)

    def run(flag):
        old = biasact.TOWER_SUM2
        biasact.TOWER_SUM2 = flag
        try:
            x = x0.clone().requires_grad_(True)
            for t in ws + bs:
                t.grad = None
            a, b = biasact.tower_conv_pair(x, x, ws[0], ws[1], bs[0], bs[1], cv.mask)
            torch.autograd.backward([a, b], [g0, g1])
            return x.grad.clone(), [w.grad.clone() for w in ws], [t.grad.clone() for t in bs], a.detach(), b.detach()
        finally:
            biasact.TOWER_SUM2 = old

    one, two = run(True), run(False)
    for k in range(2):
        assert torch.equal(one[1][k], two[1][k]) and torch.equal(one[2][k], two[2][k])
    # fp32 reference of the input gradient from the same 16-bit operands: g' = g * [y > 0] * mask, dx = sum of two transposed convolutions
    m = cv.mask.view(1, 1, x0.shape[2], x0.shape[3]).float()
    ref = torch.zeros_like(x0, dtype=torch.float32)
    for g, y, w in ((g0, one[3], ws[0]), (g1, one[4], ws[1])):
        gp = (g.float() * (y.float() > 0) * m).to(torch.bfloat16).float()
        ref += torch.ops.aten.convolution_backward(gp, x0.float(), w.detach().float(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    ref = ref * m
    rel = lambda t: float((t.float() - ref).norm() / ref.norm())
    assert rel(one[0]) < 4e-3 and rel(one[0]) <= rel(two[0]) * 1.02, (rel(one[0]), rel(two[0]))
    assert float((one[0].float() - two[0].float()).norm() / two[0].float().norm()) < 1e-2


@pytest.mark.parametrize("N,K,shapes,slots", [(2, 6, [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)], 1),   # Cout = 54: one tile, 54 % 8 = 6
                                              (2, 6, [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)], 2),   # the same, both images on one sheet
                                              (1, 90, [(13, 17), (7, 9), (4, 5)], 1),                 # Cout = 810: 4 tiles, 810 % 8 = 2
                                              (3, 90, [(13, 17), (7, 9), (4, 5)], 2),                 # odd batch: the last sheet's second slot is empty
                                              (2, 6, [(200, 180), (8, 10)], 1),                     # ~300 row tiles: more than one round of workgroups
                                              (3, 32, [(9, 11), (5, 6), (3, 3), (2, 2), (1, 1)], 1),  # Cout = 288: 2 tiles, multiple of 8
                                              (4, 32, [(9, 11), (5, 6), (3, 3), (2, 2), (1, 1)], 2)])
def test_cls_output_conv_on_canvas_vs_torch(N, K, shapes, slots):
    """rn_conv3x3_canvas_to_levels / rn_conv3x3_levels_to_canvas / rn_conv3x3_levels_wgrad (the class-output conv of
    retinanet/layers.py:163-167 with dense 9*K-channel logits) vs torch's conv2d per level on the same bf16 values:
    outputs, input gradient, weight gradient, bias gradient."""
    from pytorch_retinanet_amd import biasact
    torch.manual_seed(K)
    A = 9
    conv = torch.nn.Conv2d(256, A * K, 3, padding=1).to(DEV).to(memory_format=torch.channels_last)
    torch.nn.init.normal_(conv.weight, std=0.05)
    torch.nn.init.normal_(conv.bias, std=0.5)
    feats = [torch.randn(N, 256, h, w, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
             for h, w in shapes]
    cv = biasact.Canvas.of(feats, pad=1, slots=slots)
    packed = biasact.pack_levels(cv, feats)
    assert packed.shape[0] == (N + slots - 1) // slots
    assert biasact.cls_output_conv_fusable(packed, conv, cv)
    ys = biasact.cls_output_conv(packed, conv, cv, K, N)
    gys = [torch.randn_like(y) for y in ys]
    torch.autograd.backward(ys, gys)
    got = ([y.detach().float() for y in ys], [f.grad.float() for f in feats], conv.weight.grad.float().clone(), conv.bias.grad.float().clone())
    conv.zero_grad()
    refs, ref_dx = [], []
    w32, b32 = conv.weight.detach().to(torch.bfloat16).float().requires_grad_(True), conv.bias.detach().clone().requires_grad_(True)
    for f, g, (h, w) in zip(feats, gys, shapes):
        x32 = f.detach().float().requires_grad_(True)
        y = torch.nn.functional.conv2d(x32, w32, b32, padding=1)                       # [N, A*K, h, w]
        y3 = y.permute(0, 2, 3, 1).reshape(N, h * w * A, K)                             # layers.py:189-191
        y3.backward(g.float())
        refs.append(y3.detach())
        ref_dx.append(x32.grad)
    for a, b in zip(got[0], refs):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=2e-2, atol=2e-2 * float(b.abs().max()))              # one bf16 rounding of the output
    for a, b in zip(got[1], ref_dx):
        torch.testing.assert_close(a, b, rtol=2e-2, atol=2e-2 * float(b.abs().max()))
    torch.testing.assert_close(got[2], w32.grad, rtol=2e-2, atol=2e-2 * float(w32.grad.abs().max()))
    torch.testing.assert_close(got[3], b32.grad, rtol=2e-2, atol=2e-2 * float(b32.grad.abs().max()))


@pytest.mark.parametrize("nesterov,dampening", [(False, 0.0), (True, 0.0), (False, 0.1)])
def test_master_sgd_follows_torch_sgd_under_autocast(nesterov, dampening):
    """fp32 masters + bf16 conv weights + MasterSGD == fp32 parameters + autocast + torch.optim.SGD: after several steps the
    masters equal torch's fp32 weights (1e-6: torch's foreach kernels may contract mul+add), BN/bias parameters too."""
    from pytorch_retinanet_amd.optim import MasterSGD, master_state_dict, use_bf16_conv_weights
    from pytorch_retinanet_amd.norm import FusedBatchNorm2d

    def make():
        torch.manual_seed(11)
        m = torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3, padding=1), FusedBatchNorm2d(16), torch.nn.ReLU(),
                                torch.nn.Conv2d(16, 8, 1, bias=False)).to(DEV).to(memory_format=torch.channels_last)
        return m.train()
    a, b = make(), make()
    kw = dict(lr=0.05, momentum=0.9, weight_decay=1e-2, nesterov=nesterov, dampening=dampening)
    oa = torch.optim.SGD(a.parameters(), **kw)
    assert use_bf16_conv_weights(b) == 2
    ob = MasterSGD(b.parameters(), **kw)
    x = torch.randn(4, 8, 12, 10, device=DEV).contiguous(memory_format=torch.channels_last)
    for _ in range(4):
        for m, o in ((a, oa), (b, ob)):
            o.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = (m(x).float() ** 2).mean()
            loss.backward()
            o.step()
    sa, sb = a.state_dict(), master_state_dict(b)
    assert b[0].weight.dtype == torch.bfloat16 and sb["0.weight"].dtype == torch.float32
    for k in sa:
        torch.testing.assert_close(sb[k].float(), sa[k].float(), rtol=2e-5, atol=1e-6, msg=k)
    assert torch.equal(b[0].weight.float(), sb["0.weight"].to(torch.bfloat16).float())      # working copy == bf16(master)


@pytest.mark.parametrize("shape", [(2, 9, 11), (1, 14, 37), (3, 23, 31)])
def test_mfma_canvas_wgrad_vs_miopen(shape):
    """rn_conv3x3_canvas_wgrad_batched (position-contraction MFMA GEMM with transposed LDS reads + split reduction) vs the
    fp32 weight gradient of torch's conv2d, two problems in one launch, zero-bordered canvas with gaps."""
    from pytorch_retinanet_amd import biasact
    torch.manual_seed(5)
    N, Hp, Wp = shape
    mask2d = torch.zeros(Hp, Wp, dtype=torch.uint8, device=DEV)
    mask2d[1:-1, 1:-1] = (torch.rand(Hp - 2, Wp - 2, device=DEV) > 0.2).to(torch.uint8)
    mk = lambda: (torch.randn(N, 256, Hp, Wp, device=DEV) * mask2d[None, None]).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xs, gs = [mk(), mk()], [mk(), mk()]
    w = torch.zeros(256, 256, 3, 3, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    got = biasact._canvas_wgrad(gs, xs, [w, w], Wp, torch.cuda.current_stream().cuda_stream)
    assert got is not None
    for g, x, dw in zip(gs, xs, got):
        ref = torch.ops.aten.convolution_backward(g.float(), x.float(), w.float(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                  [False, True, False])[1]
        assert dw.shape == ref.shape and dw.is_contiguous(memory_format=torch.channels_last)
        torch.testing.assert_close(dw.float(), ref, rtol=2e-2, atol=1e-2 * float(ref.abs().max()))


@pytest.mark.parametrize("K", [5, 90])
def test_bucketed_ddp_world1_with_master_sgd_equals_plain_step(K):
    """9*K-element head biases used to leave the following fp32 bucket views 8-byte aligned and
    rn_sgd_master_step refused them (RN_EALIGN).  R18 with K = 5 / 90 under BucketedGradAllReduce at world size 1 +
    MasterSGD.step(grads=grad_views()) must run and give the same parameters as the plain MasterSGD step."""
    import pytorch_retinanet_amd as P
    from pytorch_retinanet_amd.optim import MasterSGD, use_bf16_conv_weights

    def run(use_ddp):
        torch.manual_seed(5)
        net = P.Retinanet(num_classes=K, backbone_kind="resnet18", pretrained=False, min_size=128, max_size=160).to(DEV)
        net = net.to(memory_format=torch.channels_last).train()
        use_bf16_conv_weights(net)
        opt = MasterSGD(net.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-3)
        ddp = P.BucketedGradAllReduce(net, bucket_mb=8.0) if use_ddp else None
        rng = np.random.default_rng(3)
        for _ in range(2):
            images = [torch.from_numpy(rng.random((3, 128, 160), dtype=np.float32)).to(DEV) for _ in range(2)]
            targets = []
            for _ in range(2):
                b, l = synth.gt_boxes(rng, 3, 128, 160, num_classes=K, wh_lo=20.0, wh_hi=90.0)
                targets.append({"boxes": torch.from_numpy(b).to(DEV), "labels": torch.from_numpy(l).to(DEV)})
            ddp.zero_grad() if ddp else opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = net(images, targets)
            (out["classification_loss"] + out["regression_loss"]).backward()
            if ddp:
                ddp.finish()
                for v in ddp.grad_views().values():
                    assert v.data_ptr() % 16 == 0
                opt.step(grads=ddp.grad_views())
            else:
                opt.step()
        return {n: (p.master if hasattr(p, "master") else p.data).detach().float().cpu() for n, p in net.named_parameters()}

    a, b = run(True), run(False)
    # not bit-equal: MIOpen's weight-gradient kernels accumulate with atomics, so even two identical runs differ in the
    # last bf16 bits of the gradients (lr 1e-2, 2 steps)
    for k in a:
        torch.testing.assert_close(a[k], b[k], rtol=0, atol=5e-4, msg=k)


@pytest.mark.parametrize("kind", ["resnet18", "resnet50"])
def test_frozen_bn_folding_equals_the_unfused_eval_path(kind):
    """Inference with BN frozen (eval mode, no_grad): conv weights carry gamma / sqrt(var + eps) and the epilogue adds the
    folded bias (backbone.conv_bn; reference backbone.py:348-351) -- same C3/C4/C5 as bn(conv(x)) on the
    running statistics, fp32 1e-4 and bf16 autocast within bf16 rounding; the fold is rebuilt when the statistics change."""
    import pytorch_retinanet_amd as P
    from pytorch_retinanet_amd import backbone
    torch.manual_seed(3)
    net = P.Retinanet(num_classes=4, backbone_kind=kind, pretrained=False, min_size=128, max_size=160).to(DEV)
    net = net.to(memory_format=torch.channels_last).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):                  # non-trivial statistics and affine parameters
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    x = torch.randn(2, 3, 128, 160, device=DEV).contiguous(memory_format=torch.channels_last)

    def run(fold, autocast):
        backbone.FOLD_FROZEN_BN = fold
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                return [o.float() for o in net.backbone(x)]
        finally:
            backbone.FOLD_FROZEN_BN = True
    for autocast, tol in ((False, 1e-4), (True, 4e-2)):
        ref, got = run(False, autocast), run(True, autocast)
        for a, b in zip(ref, got):
            torch.testing.assert_close(b, a, rtol=tol, atol=tol * float(a.abs().max()))
    assert len(net.backbone.backbone.bn1.__dict__.get("_rn_fold", {})) > 0        # the fold lives on its BN module
    before = run(True, False)
    with torch.no_grad():
        net.backbone.backbone.bn1.running_mean.add_(0.5)             # statistics change -> version counter -> the fold is rebuilt
    after, ref = run(True, False), run(False, False)
    assert not torch.allclose(before[0], after[0])
    for a, b in zip(ref, after):
        torch.testing.assert_close(b, a, rtol=1e-4, atol=1e-4 * float(a.abs().max()))
    # with gradients enabled (or BN in train mode) nothing is folded: the training path is untouched
    net.train()
    y = net.backbone(x)
    assert y[0].requires_grad


def test_frozen_bn_fold_is_rebuilt_after_the_librarys_own_training_writes():
    """MasterSGD.step and the fused training BN forward write weights / running statistics through raw
    pointers, which never bump torch's ``_version`` counters.  fold -> train step -> fold again must see the NEW tensors:
    the folded eval output equals the unfolded one after the step, and differs from the one cached before it."""
    import pytorch_retinanet_amd as P
    from pytorch_retinanet_amd import backbone
    from pytorch_retinanet_amd.optim import MasterSGD, use_bf16_conv_weights
    torch.manual_seed(5)
    net = P.Retinanet(num_classes=4, backbone_kind="resnet18", pretrained=False, min_size=128, max_size=160).to(DEV)
    net = net.to(memory_format=torch.channels_last)
    use_bf16_conv_weights(net)
    opt = MasterSGD(net.parameters(), lr=0.05, momentum=0.9)
    x = torch.randn(2, 3, 128, 160, device=DEV).contiguous(memory_format=torch.channels_last)

    def eval_features(fold):
        backbone.FOLD_FROZEN_BN = fold
        net.eval()
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                return [o.float() for o in net.backbone(x)]
        finally:
            backbone.FOLD_FROZEN_BN = True
    before = eval_features(True)                                     # populates the fold cache
    net.train()
    images = [torch.rand(3, 128, 160, device=DEV) for _ in range(2)]
    targets = [{"boxes": torch.tensor([[10., 12., 90., 100.]], device=DEV), "labels": torch.tensor([1], device=DEV)} for _ in range(2)]
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(images, targets)
        (out["classification_loss"] + out["regression_loss"]).backward()
        opt.step()
    after, ref = eval_features(True), eval_features(False)
    assert not torch.allclose(before[0], after[0])                   # the stale fold would have reproduced `before`
    for a, b in zip(ref, after):
        torch.testing.assert_close(b, a, rtol=4e-2, atol=4e-2 * float(a.abs().max()))


def test_bottleneck_identity_gradient_in_conv1_gemm_equals_autograd_add():
    """backbone._Conv1x1Skip / _SkipLink (the identity branch's gradient goes into conv1's data-gradient GEMM as its accumulator
    input) == the plain block, where autograd adds the two gradients of the block input: outputs bit-equal, input and
    parameter gradients equal up to one bf16 rounding of the sum."""
    from pytorch_retinanet_amd import backbone as bb
    torch.manual_seed(2)
    blocks = torch.nn.Sequential(bb.Bottleneck(256, 64), bb.Bottleneck(256, 64)).to(DEV).to(memory_format=torch.channels_last).train()
    for p in blocks.parameters():
        if p.dim() == 4:
            p.data = p.data.to(torch.bfloat16)
    x0 = torch.randn(2, 256, 20, 24, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = torch.randn_like(x0)
    res = {}
    for fuse in (False, True):
        bb.FUSE_SKIP_ADD = fuse
        x = x0.clone().requires_grad_(True)
        blocks.zero_grad()
        y = blocks(x)
        y.backward(g)
        res[fuse] = (y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in blocks.named_parameters()})
    bb.FUSE_SKIP_ADD = True
    assert torch.equal(res[True][0], res[False][0])
    ref = res[False][1].float()
    torch.testing.assert_close(res[True][1].float(), ref, rtol=2e-2, atol=2e-2 * float(ref.abs().max()))
    for n, a in res[False][2].items():
        b = res[True][2][n]
        torch.testing.assert_close(b.float(), a.float(), rtol=3e-2, atol=3e-2 * float(a.float().abs().max()), msg=n)


def test_joined_gradients_of_c3_c4_equal_autograds_adds():
    """C3 / C4 feed the next layer's conv1, its stride-2 downsample conv and an FPN lateral (reference: backbone.py:246-263 +
    layers.py:44-64; autograd adds the three data gradients): with ``pwconv.share_gradients`` the donors' gradients join the
    receiver's GEMM.  Every parameter gradient of trunk + FPN with joins == without, to bf16 summation-order noise; and a donor whose
    receiver does not take part in the backward pass returns its gradient the ordinary way."""
    from pytorch_retinanet_amd import backbone, pwconv
    from pytorch_retinanet_amd.layers import FeaturePyramid
    from pytorch_retinanet_amd.optim import use_bf16_conv_weights
    torch.manual_seed(0)
    m = torch.nn.ModuleDict({"trunk": backbone.resnet50(pretrained=False), "fpn": FeaturePyramid(512, 1024, 2048)})
    # (BatchNorm on its running statistics: with batch statistics over the 40 - 160 positions of these small maps the comparison of two
    # bf16 pipelines is ill-conditioned -- a rounding difference anywhere moves every gradient upstream by tens of percent)
    m = m.to(DEV).to(memory_format=torch.channels_last).eval()
    use_bf16_conv_weights(m)
    x = torch.randn(2, 3, 192, 256, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def run(flag):
        pwconv.JOIN_GRADS = flag
        for p in m.parameters():
            p.grad = None
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                f = m["trunk"](x)
                outs = m["fpn"]([f["layer_2"], f["layer_3"], f["layer_4"]])
                loss = sum((o.float() ** 2).mean() for o in outs)
            loss.backward()
        finally:
            pwconv.JOIN_GRADS = True
        return {n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None}

    before = dict(pwconv.JOIN_STATS)
    joined = run(True)
    assert pwconv.JOIN_STATS["full"] - before["full"] == 2 and pwconv.JOIN_STATS["compact"] - before["compact"] == 2      # C3 and C4
    plain, plain2 = run(False), run(False)
    assert joined.keys() == plain.keys()

    def rel(u, v):
        return {n: float((u[n] - v[n]).norm() / v[n].norm().clamp_min(1e-12)) for n in v}
    # two runs of the SAME bf16 pipeline already differ (MIOpen's weight gradients accumulate with atomics, and every difference is
    # carried upstream through 50 layers): the joined run must sit inside that spread
    floor, err = rel(plain2, plain), rel(joined, plain)
    bad = [(n, err[n], floor[n]) for n in err if err[n] > 3.0 * max(floor.values()) + 1e-2]
    assert not bad, (bad[:6], max(floor.values()))
    # a receiver that never runs: the loss uses C3's lateral alone (layer3 and everything behind it get no gradient) -- the donor must
    # notice and return its gradient the ordinary way, or layer1 / layer2 / the stem would be left without one
    def lateral_only(flag):
        pwconv.JOIN_GRADS = flag
        for p in m.parameters():
            p.grad = None
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                f = m["trunk"](x)
                (pwconv.conv1x1(m["fpn"].conv_c3_1x1, f["layer_2"]).float() ** 2).mean().backward()
        finally:
            pwconv.JOIN_GRADS = True
        return {n: p.grad.float().clone() for n, p in m.named_parameters() if p.grad is not None}
    a, b, b2 = lateral_only(True), lateral_only(False), lateral_only(False)
    assert a.keys() == b.keys() and "trunk.conv1.weight" in a and "trunk.layer2.3.conv3.weight" in a
    floor, err = rel(b2, b), rel(a, b)
    bad = [(n, err[n], floor[n]) for n in err if err[n] > 3.0 * max(floor.values()) + 1e-2]
    assert not bad, (bad[:6], max(floor.values()))


def test_top_tower_layer_relu_backward_rides_in_the_output_convs_data_gradient():
    """The last tower layer's ReLU backward + bias gradient in the epilogue of the class- / box-output convs' data-gradient kernel
    (``rn_conv3x3_levels_to_canvas_relu``) == the separate ``rn_bias_act_backward`` pass: every head parameter gradient and the input
    gradients bit-equal (bias gradients: fp32 summation order)."""
    from pytorch_retinanet_amd import biasact
    from pytorch_retinanet_amd.layers import RetinaNetHead
    torch.manual_seed(1)
    head = RetinaNetHead(256, 256, 9, 6, 0.01).to(DEV).to(memory_format=torch.channels_last)
    for m in head.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.normal_(m.weight, std=0.03)
            torch.nn.init.normal_(m.bias, std=0.1)
    feats = [torch.randn(2, 256, h, w, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
             for h, w in ((16, 20), (8, 10), (4, 5), (2, 3), (1, 2))]
    calls = {"n": 0}
    real = biasact.lib.rn_conv3x3_levels_to_canvas_relu

    def run(flag):
        biasact.FUSE_TOWER_RELU_BWD = flag
        head.zero_grad()
        for f in feats:
            f.grad = None
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = head.forward_levels(feats)
                loss = sum((t.float() ** 2).sum() for t in out["cls_levels"]) + sum((t.float() ** 3).sum() for t in out["bbox_levels"])
            loss.backward()
        finally:
            biasact.FUSE_TOWER_RELU_BWD = True
        return [f.grad.clone() for f in feats], {n: p.grad.clone() for n, p in head.named_parameters()}

    half_seen = []
    orig = biasact._levels_dgrad

    def spy(relu_link, *a, **k):
        rc = orig(relu_link, *a, **k)
        half_seen.append(relu_link is not None and relu_link[0] is not None and relu_link[0].half[relu_link[1]] is not None)
        return rc
    biasact._levels_dgrad = spy
    try:
        fused = run(True)
        assert half_seen == [True, True] or half_seen == [True, True][::-1], half_seen      # both output convs deposited their half
        half_seen.clear()
        plain = run(False)
        assert half_seen == [False, False]
    finally:
        biasact._levels_dgrad = orig
    for a, b in zip(fused[0], plain[0]):
        assert torch.equal(a, b)
    for n in plain[1]:
        if n.endswith("bias"):
            torch.testing.assert_close(fused[1][n], plain[1][n], rtol=1e-4, atol=1e-4 * float(plain[1][n].abs().max()))
        else:
            assert torch.equal(fused[1][n], plain[1][n]), n


def test_bucket_gather_is_one_widening_launch_and_equals_copy():
    "parallel._gather == per-tensor copy_ (bf16 / f16 -> fp32 views, same-dtype views, odd sizes, channels-last and strided leftovers)."
    from pytorch_retinanet_amd import parallel
    g = torch.Generator(device=DEV).manual_seed(3)
    shapes = [(64, 3, 7, 7), (256, 64, 1, 1), (90 * 9,), (5,), (64, 64, 3, 3), (1,), (1000, 7)]
    grads, views, want = [], [], []
    for i, shp in enumerate(shapes):
        dt = [torch.bfloat16, torch.float16, torch.float32][i % 3]
        t = torch.randn(shp, device=DEV, generator=g).to(dt)
        if len(shp) == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        grads.append(t)
        flat = torch.full((t.numel() + 64,), float("nan"), device=DEV, dtype=torch.float32)
        views.append(torch.as_strided(flat, t.size(), t.stride(), 64 - (i % 2) * 3 if dt == torch.float32 else 64))
        want.append(t.float())
    # a pair whose strides differ (the gradient is a transposed view): must take the fallback
    t = torch.randn((6, 10), device=DEV, generator=g).to(torch.bfloat16).t()
    grads.append(t); views.append(torch.empty((10, 6), device=DEV, dtype=torch.float32)); want.append(t.float())
    parallel._gather(views, grads)
    torch.cuda.synchronize()
    for v, w in zip(views, want):
        assert torch.equal(v, w)
