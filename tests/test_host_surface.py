"""CPU suite: the host-side mirror of the reference surface, the C-ABI library's symbols, and the
no-CPU-fallback rule.  (Numerical parity of the HIP path is in test_hip_parity.py, -m gpu.)"""
import os
import re

import numpy as np
import pytest
import torch

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_loads_and_exports_every_declared_symbol():
    """Every function declared in include/retinanet_hip.h is exported by libretinanet_hip.so and bound."""
    from pytorch_retinanet_amd import _lib
    header = open(os.path.join(ROOT, "include", "retinanet_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rn_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(_lib.lib, name), name
    assert _lib.lib.rn_version() == 10
    assert b"alignment" in _lib.lib.rn_status_string(-2)


def test_no_cpu_fallback_and_no_oracle_import_in_product():
    import pytorch_retinanet_amd as P
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.matcher(torch.zeros(4, 4), torch.zeros(1, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.AnchorGenerator().grid_anchors([[4, 4]] * 5, torch.device("cpu"))
    # the product package never references the oracle
    pkg = os.path.join(ROOT, "pytorch_retinanet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "rn_oracle" not in src, f


def test_config_defaults_and_ifnone():
    from pytorch_retinanet_amd import config as c
    from pytorch_retinanet_amd.utilities import ifnone
    assert (c.NUM_CLASSES, c.BACKBONE, c.PRIOR) == (90, "resnet50", 0.01)
    assert (c.SCORE_THRES, c.NMS_THRES, c.MAX_DETECTIONS_PER_IMAGE) == (0.05, 0.5, 100)
    assert (c.IOU_THRESHOLDS_FOREGROUND, c.IOU_THRESHOLDS_BACKGROUND) == (0.5, 0.4)
    assert (c.FOCAL_LOSS_GAMMA, c.FOCAL_LOSS_ALPHA, c.SMOOTH_L1_LOSS_BETA) == (2.0, 0.25, 0.1)
    assert c.ANCHOR_STRIDES == [8, 16, 32, 64, 128] and c.ANCHOR_OFFSET == 0.0
    assert np.allclose(c.ANCHOR_SIZES, synth.ANCHOR_SIZES)
    assert ifnone(None, 3) == 3 and ifnone(0, 3) == 0


def test_cell_anchors_match_golden(golden):
    """A1 (anchors.py:110-135): size-major, double arithmetic -> fp32; state-dict buffer names."""
    from pytorch_retinanet_amd import AnchorGenerator
    ag = AnchorGenerator()
    cells = np.stack([b.numpy() for b in ag.cell_anchors])
    assert np.array_equal(cells, golden("anchors.npz")["cells"])
    assert ag.num_cell_anchors == [9] * 5 == ag.num_anchors
    assert list(ag.state_dict()) == [f"cell_anchors.{i}" for i in range(5)]
    ag2 = AnchorGenerator(sizes=[[20.0, 33.5]], aspect_ratios=[0.4, 1.0, 3.0], strides=[8, 16], offset=0.5)
    assert ag2.num_anchors == [6, 6] and len(ag2.sizes) == 2
    with pytest.raises(AssertionError):
        AnchorGenerator(sizes=[[1.0], [2.0], [3.0]], strides=[8, 16])


STATE_KEYS_R18_HEAD = [
    "fpn.conv_c3_1x1.weight", "fpn.conv_c3_3x3.bias", "fpn.conv_c6_3x3.weight", "fpn.conv_c7_3x3.bias",
    "anchor_generator.cell_anchors.0", "anchor_generator.cell_anchors.4",
    "retinanet_head.classification_head.class_subnet.0.weight", "retinanet_head.classification_head.class_subnet.6.bias",
    "retinanet_head.classification_head.class_subnet_output.bias",
    "retinanet_head.regression_head.box_subnet.4.weight", "retinanet_head.regression_head.box_subnet_output.weight",
    "backbone.backbone.conv1.weight", "backbone.backbone.layer4.1.bn2.running_var",
]


def test_retinanet_construction_and_state_dict_surface():
    import pytorch_retinanet_amd as P
    with pytest.raises(ValueError, match="backbone_kind"):
        P.Retinanet(backbone_kind="vgg16", pretrained=False)
    net = P.Retinanet(num_classes=7, backbone_kind="resnet18", pretrained=False)
    sd = net.state_dict()
    for k in STATE_KEYS_R18_HEAD:
        assert k in sd, k
    assert sd["retinanet_head.classification_head.class_subnet_output.weight"].shape == (9 * 7, 256, 3, 3)
    assert sd["retinanet_head.regression_head.box_subnet_output.weight"].shape == (36, 256, 3, 3)
    prior_bias = sd["retinanet_head.classification_head.class_subnet_output.bias"]
    assert torch.allclose(prior_bias, torch.full_like(prior_bias, -np.log(99.0)))          # layers.py:175-178
    # freeze_bn only flips BN to eval at construction (Q18); .train() un-freezes
    bn = net.backbone.backbone.bn1
    assert not bn.training
    net.train()
    assert bn.training
    assert net._get_backbone_ouputs() == [128, 256, 512]
    assert P.Retinanet(backbone_kind="resnet50", pretrained=False)._get_backbone_ouputs() == [512, 1024, 2048]


def test_head_layout_matches_reference_permutation():
    """Q20: [N, A*K, H, W] -> [N, H*W*A, K] with anchor index (h*W+w)*A+a; zero-copy for channels_last."""
    from pytorch_retinanet_amd.layers import _to_anchor_major
    n, a, k, h, w = 2, 9, 5, 3, 4
    x = torch.randn(n, a * k, h, w)
    ref = x.view(n, a, k, h, w).permute(0, 3, 4, 1, 2).contiguous().view(n, -1, k)      # layers.py:189-191
    assert torch.equal(_to_anchor_major(x, k), ref)
    xc = x.contiguous(memory_format=torch.channels_last)
    out = _to_anchor_major(xc, k)
    assert torch.equal(out, ref) and out.data_ptr() == xc.data_ptr()


def test_transform_resize_pad_and_postprocess():
    from pytorch_retinanet_amd.transform import GeneralizedRCNNTransform
    t = GeneralizedRCNNTransform(800, 1333, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]).eval()
    imgs = [torch.rand(3, 800, 1333), torch.rand(3, 400, 500)]
    tg = [{"boxes": torch.tensor([[10., 20., 100., 200.]]), "labels": torch.tensor([1])},
          {"boxes": torch.tensor([[10., 20., 100., 200.]]), "labels": torch.tensor([2])}]
    il, out = t(imgs, tg)
    assert il.tensors.shape == (2, 3, 800, 1344) and il.image_sizes == [(800, 1333), (800, 1000)]
    assert torch.equal(out[0]["boxes"], tg[0]["boxes"])                      # scale 1: untouched
    assert torch.allclose(out[1]["boxes"], tg[1]["boxes"] * 2.0)
    assert tg[1]["boxes"][0, 0] == 10.0                                      # caller's targets not mutated
    ref0 = (imgs[0] - torch.tensor(t.image_mean)[:, None, None]) / torch.tensor(t.image_std)[:, None, None]
    assert torch.equal(il.tensors[0, :, :800, :1333], ref0) and not il.tensors[0, :, :, 1333:].any()
    dets = t.postprocess([{"boxes": torch.tensor([[0., 0., 1000., 800.]])}] * 2, il.image_sizes, [(800, 1333), (400, 500)])
    assert torch.allclose(dets[1]["boxes"], torch.tensor([[0., 0., 500., 400.]]))
    with pytest.raises(ValueError):
        t([torch.rand(800, 1333)])


def test_hparams_utils_and_lightning_surface():
    import pytorch_retinanet_amd as P
    conf = P.load_hparams()
    assert conf.model.backbone_kind == "resnet50" and conf.optimizer.params.lr == 0.001
    assert conf.scheduler.monitor == "val_loss" and conf.dataloader.args.num_workers == 0
    assert P.load_obj("torch.optim.SGD") is torch.optim.SGD
    with pytest.raises(AttributeError):
        P.load_obj("torch.optim.NoSuchOptimizer")
    assert P.collate_fn([(1, "a", 0), (2, "b", 1)]) == ((1, 2), ("a", "b"), (0, 1))
    conf.model.pretrained = False
    conf.model.backbone_kind = "resnet18"
    m = P.RetinaNetModel(conf)
    for hook in ("forward", "prepare_data", "configure_optimizers", "train_dataloader", "val_dataloader", "test_dataloader",
                 "training_step", "validation_step", "test_step", "test_epoch_end"):
        assert callable(getattr(m, hook)), hook
    opts, scheds = m.configure_optimizers()
    assert isinstance(opts[0], torch.optim.SGD) and scheds[0]["monitor"] == "val_loss"
    conf.dataset.kind = "coco"
    with pytest.raises(NotImplementedError):
        m.prepare_data()
    conf.dataset.kind = "synthetic"
    conf.dataset.length, conf.dataset.height, conf.dataset.width = 4, 64, 96
    m.prepare_data()
    img, tgt, idx = m.trn_ds[1]
    assert img.shape == (3, 64, 96) and tgt["boxes"].shape == (8, 4) and tgt["labels"].min() >= 1
    batch = next(iter(m.train_dataloader()))
    assert len(batch) == 3 and len(batch[0]) == 2


def test_helper_losses_match_their_definitions():
    """The stand-alone focal / smooth-L1 helpers keep the reference's definitions (Q2, Q3, Q10)."""
    import pytorch_retinanet_amd as P
    crit = P.RetinaNetLosses(3)
    x = torch.tensor([[0.3, -1.2, 2.0]], requires_grad=True)
    t = torch.tensor([[1.0, 0.0, 0.0]])
    loss = crit.focal_loss(x, t)
    p = torch.sigmoid(x.detach())
    w = torch.where(t > 0, 0.75 * (1 - p) ** 2, 0.25 * p ** 2)                       # positives get 1-alpha (Q2)
    ref = (w * torch.nn.functional.binary_cross_entropy_with_logits(x.detach(), t, reduction="none")).sum()
    assert torch.allclose(loss, ref)
    loss.backward()
    assert torch.allclose(x.grad, w * (p - t))                                        # detached weight (Q3)
    d = torch.tensor([0.05, -0.3])
    assert torch.allclose(crit.smooth_l1_loss(d, torch.zeros(2)), torch.tensor(0.5 * 0.05 ** 2 / 0.1 + 0.3 - 0.05))


@pytest.mark.reference
def test_conv_stack_equals_reference_with_same_weights():
    """Live check in the build container: load the reference's state dict into this framework's model and
    compare head outputs (covers key names, FPN wiring incl. P6-from-C5, head layout)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import _tv_standin
    R = _tv_standin.import_reference()
    import pytorch_retinanet_amd as P
    torch.manual_seed(0)
    ref = R.Retinanet(num_classes=4, backbone_kind="resnet18", pretrained=False, min_size=96, max_size=128).eval()
    mine = P.Retinanet(num_classes=4, backbone_kind="resnet18", pretrained=False, min_size=96, max_size=128).eval()
    assert list(ref.state_dict()) == list(mine.state_dict())
    mine.load_state_dict(ref.state_dict())
    imgs = [torch.rand(3, 96, 120), torch.rand(3, 80, 128)]
    with torch.no_grad():
        il_r, _ = ref.transform(imgs, None)
        il_m, _ = mine.transform(imgs, None)
        assert torch.allclose(il_r.tensors, il_m.tensors, atol=1e-6) and list(il_r.image_sizes) == list(il_m.image_sizes)
        f_r = ref.fpn(ref.backbone(il_r.tensors))
        out_r = ref.retinanet_head(f_r)
        f_m, out_m = mine._features(il_m.tensors)
    for a, b in zip(f_r, f_m):
        assert torch.allclose(a, b, atol=1e-5)
    assert torch.allclose(out_r["cls_preds"], out_m["cls_preds"], atol=1e-5)
    assert torch.allclose(out_r["bbox_preds"], out_m["bbox_preds"], atol=1e-5)
    # R50 key parity too (bottleneck blocks)
    assert list(R.Retinanet(backbone_kind="resnet50", pretrained=False).state_dict()) == \
        list(P.Retinanet(backbone_kind="resnet50", pretrained=False).state_dict())


def test_canvas_layout_and_pack_unpack_roundtrip_cpu():
    """biasact.Canvas (pure host logic): level placement with and without the zero border, one and two images per sheet,
    the mask and the position map of the level-mode kernels, and that unpack(pack(levels)) is the identity with zeros
    everywhere else, also for an odd batch on two-slot sheets (the torch fallbacks run on CPU)."""
    import numpy as np
    import torch
    from pytorch_retinanet_amd import biasact
    shapes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    for pad, slots, (H, W) in ((0, 1, (151, 168)), (1, 1, (153, 170)), (1, 2, (280, 171)), (0, 2, (278, 169))):
        cv = biasact.Canvas(shapes, torch.device("cpu"), pad=pad, slots=slots)
        assert (cv.H, cv.W) == (H, W) and int(cv.mask.sum()) == slots * sum(h * w for h, w in shapes)
        assert len(cv.regions) == slots * len(shapes) and len(cv.origin) == len(shapes)
        m = cv.mask.view(cv.H, cv.W)
        mp = cv.map.view(cv.H, cv.W).numpy()
        assert ((mp >= 0) == (m.numpy() == 1)).all()
        for l, s_, r, c, h, w in cv.regions:
            assert (h, w) == cv.shapes[l] and m[r:r + h, c:c + w].all()
            ring = m[max(r - 1, 0):r + h + 1, max(c - 1, 0):c + w + 1].sum()
            assert int(ring) == h * w                      # a zero ring (or the canvas edge) around every rectangle
            want = (s_ << 28) | (l << 24) | (np.arange(h)[:, None] * w + np.arange(w)[None, :])
            assert (mp[r:r + h, c:c + w] == want).all()
        if pad:
            assert not m[0].any() and not m[-1].any() and not m[:, 0].any() and not m[:, -1].any()
    # two images per sheet: 8 % fewer positions per image on the standard pyramid
    one, two = biasact.Canvas(shapes, torch.device("cpu"), 1, 1), biasact.Canvas(shapes, torch.device("cpu"), 1, 2)
    assert two.H * two.W / 2 < 0.925 * one.H * one.W and two.sheets(8) == 4 and two.sheets(3) == 2
    small = [(6, 8), (3, 4), (2, 2)]
    for slots, N in ((1, 2), (2, 2), (2, 3)):
        cv = biasact.Canvas(small, torch.device("cpu"), pad=1, slots=slots)
        feats = [torch.randn(N, 8, h, w).contiguous(memory_format=torch.channels_last).requires_grad_(True) for h, w in small]
        packed = biasact.pack_levels(cv, feats)
        assert packed.shape == (cv.sheets(N), 8, cv.H, cv.W) and not packed[:, :, cv.mask.view(cv.H, cv.W) == 0].any()
        if slots == 2 and N == 3:                          # image 2 = sheet 1, slot 0; slot 1 of sheet 1 stays empty
            l, s_, r, c, h, w = [q for q in cv.regions if q[0] == 0 and q[1] == 1][0]
            assert not packed[1, :, r:r + h, c:c + w].any()
            l, s_, r, c, h, w = [q for q in cv.regions if q[0] == 1 and q[1] == 0][0]
            assert torch.equal(packed[1, :, r:r + h, c:c + w], feats[1][2])
        back = biasact.unpack_levels(cv, packed, N)
        for a, b in zip(back, feats):
            assert torch.equal(a, b)
        sum((b * (i + 1)).sum() for i, b in enumerate(back)).backward()
        for i, f in enumerate(feats):
            assert torch.equal(f.grad, torch.full_like(f, float(i + 1)))
    cv = biasact.Canvas(small, torch.device("cpu"), pad=1)
    # bias_act's PyTorch path (what CPU tensors take)
    x = torch.randn(2, 8, cv.H, cv.W)
    y = biasact.bias_act(x, torch.arange(8.0), cv.mask, relu=True)
    ref = torch.relu(x + torch.arange(8.0)[None, :, None, None]) * cv.mask.view(1, 1, cv.H, cv.W)
    assert torch.equal(y, ref)


def test_weight_gradient_partials_stay_at_two_workgroups_per_cu():
    """Host logic of csrc/pw.hip and csrc/wgrad3x3.hip (no GPU call): the f32 split partials of a 1x1 weight gradient are two
    workgroups per CU x one 64 KiB tile = 32 MiB at every trunk width -- rounds 2-3 never went below 32 splits and layer4's
    GEMMs wrote 113-256 MB --, and the narrow 3x3 weight gradient keeps one partial per CU whatever the sub-problem count."""
    import ctypes as C
    from pytorch_retinanet_amd._lib import RnPwConv, lib
    for (M, N, Cin, H, W) in [(33600, 256, 1024, 50, 84), (33600, 1024, 256, 50, 84), (8400, 512, 2048, 25, 42), (8400, 2048, 512, 25, 42),
                              (8400, 2048, 1024, 25, 42), (134400, 256, 512, 100, 168), (534400, 64, 256, 200, 334), (534400, 256, 64, 200, 334)]:
        d = RnPwConv(M, Cin, N, 1, 1, 0, H, W, H, W)
        need = lib.rn_pw_wgrad_workspace_bytes(C.byref(d))
        assert 0 < need <= (32 << 20), (M, N, Cin, need)
        assert need % (N * Cin * 4) == 0                       # whole splits
    for cout, cin in [(64, 64), (128, 128), (256, 256), (512, 512), (128, 64)]:
        need = lib.rn_conv3x3_wgrad_narrow_workspace_bytes(cout, cin)
        subs = (cout // 64) * (cin // 64)
        assert need % (subs * 64 * 9 * 64 * 4) == 0 and 0 < need <= 256 * 64 * 9 * 64 * 4, (cout, cin, need)
    assert lib.rn_conv3x3_wgrad_narrow_workspace_bytes(96, 64) == 0 and lib.rn_conv3x3_wgrad_narrow_workspace_bytes(0, 64) == 0


def test_block_links_are_not_submodules_and_survive_copies():
    """``pwconv.link_blocks`` (a fused bottleneck forms its successor's conv1 with its own output): the links live in ``__dict__`` -- no
    extra children, parameters or state-dict keys -- and a deepcopy / pickle round trip points them into the copy."""
    import copy
    import io
    import torch
    from pytorch_retinanet_amd import backbone
    m = backbone.resnet50(pretrained=False)
    keys = list(m.state_dict().keys())
    assert not any("_rn_next" in k for k in keys)
    assert [n for n, _ in m.layer1[0].named_children()] == ["conv1", "bn1", "conv2", "bn2", "conv3", "bn3", "relu", "downsample"]
    assert m.layer1[0].__dict__["_rn_next"][0] is m.layer1[1] and m.layer1[2].__dict__["_rn_next"][0] is m.layer2[0]
    assert "_rn_next" not in m.layer4[2].__dict__
    m2 = copy.deepcopy(m)
    assert list(m2.state_dict().keys()) == keys and m2.layer1[0].__dict__["_rn_next"][0] is m2.layer1[1]
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert m3.layer2[3].__dict__["_rn_next"][0] is m3.layer3[0]


def test_round5_entry_points_reject_bad_arguments_before_any_gpu_call():
    """Argument validation of the entry points round 5 added (include/retinanet_hip.h): null pointers, unsupported shapes and misaligned pointers are
    refused with their error codes before a HIP call is made -- checked here, where there is no GPU."""
    from pytorch_retinanet_amd._lib import lib
    EINVAL, EALIGN, EUNSUP = -1, -2, -4
    p = 4096                                                          # a 16-byte-aligned non-null "pointer": never dereferenced on these paths
    # band-staged dense 3x3 (+ statistics partials)
    assert lib.rn_conv3x3_dense_band(0, p, p, 1, 1, 8, 8, 128, 128, p, 0) == EINVAL
    assert lib.rn_conv3x3_dense_band(p, p, p, 1, 1, 8, 8, 96, 128, p, 0) == EUNSUP          # Cin % 64
    assert lib.rn_conv3x3_dense_band(p, p, p, 1, 1, 8, 8, 128, 192, p, 0) == EUNSUP         # Cout % 128
    assert lib.rn_conv3x3_dense_band(p + 2, p, p, 1, 1, 8, 8, 128, 128, p, 0) == EALIGN
    assert lib.rn_conv3x3_dense_band_stats(p, p, p, 0, 1, 1, 8, 8, 128, 128, p, 0) == EINVAL  # no partial buffer
    assert lib.rn_conv3x3_dense_band_tiles(8, 100, 168) == 525 and lib.rn_conv3x3_dense_band_tiles(0, 100, 168) == 0
    # K-split dense 3x3: the workspace query says when a split does not help
    assert lib.rn_conv3x3_dense_splitk_workspace_bytes(8, 25, 42, 512) > 0
    assert lib.rn_conv3x3_dense_splitk_workspace_bytes(8, 50, 84, 256) == 0
    assert lib.rn_conv3x3_dense_splitk(p, p, p, 1, 8, 25, 42, 512, 512, p, 0, 0, 0) == EINVAL
    # the two-source data gradient of the first tower layer
    assert lib.rn_conv3x3_canvas_sum2(p, 0, p, 0, p, 1, 1000, 500, 20, 256, 256, 0) == EINVAL
    assert lib.rn_conv3x3_canvas_sum2(p, p, p, 0, p, 1, 1000, 500, 20, 96, 256, 0) == EUNSUP
    assert lib.rn_conv3x3_canvas_sum2(p, p + 8, p, 0, p, 1, 1000, 500, 20, 256, 256, 0) == EALIGN
    # the bottleneck seam kernels: walkers() == 0 outside their shapes
    assert lib.rn_pw_conv3_backward_walkers(134400, 128, 512) > 0 and lib.rn_pw_conv3_backward_walkers(33600, 256, 1024) == 0
    assert lib.rn_pw_block_out_conv1_walkers(537600, 256, 64) > 0 and lib.rn_pw_block_out_conv1_walkers(33600, 1024, 256) == 0
    assert lib.rn_pw_dgrad_resid_sums_walkers(537600, 64, 256) > 0 and lib.rn_pw_dgrad_resid_sums_walkers(33600, 256, 1024) == 0
    assert lib.rn_pw_conv3_forward_walkers(134400, 128, 512) > 0 and lib.rn_pw_conv3_forward_walkers(134400, 96, 512) == 0
