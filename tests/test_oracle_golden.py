"""CPU suite: the oracle (oracle/rn_oracle.c) against the committed golden vectors.

The vectors were produced by the reference itself (tests/golden/gen_golden.py);
this keeps the oracle pinned on every run, including on the GPU box where
/root/reference does not exist.
"""
import numpy as np
import pytest

import synth


def _cells(oracle):
    return [oracle.cell_anchors(s, synth.ANCHOR_RATIOS) for s in synth.ANCHOR_SIZES]


@pytest.mark.parametrize("tag,hw", [("r18_512", (512, 512)), ("r50_800x1344", (800, 1344)), ("r101_1344", (1344, 1344))])
def test_anchors_bit_exact(golden, oracle_lib, tag, hw):
    g = golden("anchors.npz")
    cells = _cells(oracle_lib)
    assert np.array_equal(np.stack(cells), g["cells"])
    a = oracle_lib.anchors_emit(synth.levels_for(*hw), cells, 0.0)
    assert a.shape[0] == int(g[f"{tag}_count"])
    assert np.array_equal(a[:32], g[f"{tag}_head"]) and np.array_equal(a[-32:], g[f"{tag}_tail"])
    assert synth.sha(a) == str(g[f"{tag}_sha"])


@pytest.mark.parametrize("tag,off", [("toy_off05", 0.5), ("toy_off03", 0.3)])
def test_anchors_custom_generator(golden, oracle_lib, tag, off):
    g = golden("anchors.npz")
    sizes, ratios = [[20.0, 33.5], [70.0, 91.25]], [0.4, 1.0, 3.0]
    cells = [oracle_lib.cell_anchors(s, ratios) for s in sizes]
    a = oracle_lib.anchors_emit([(5, 7, 8), (3, 4, 16)], cells, off)
    assert np.array_equal(a, g[f"{tag}_full"])


@pytest.mark.parametrize("T", [0, 1, 8, 64, 500])
def test_matcher_r18(golden, oracle_lib, T):
    g = golden("match.npz")
    anc = oracle_lib.anchors_emit(synth.levels_for(512, 512), _cells(oracle_lib), 0.0)
    m, nfg = oracle_lib.iou_match(anc, [g[f"r18_T{T}_gt"]])
    assert np.array_equal(m[0], g[f"r18_T{T}_matches"].astype(np.int64))
    assert nfg[0] == (m[0] >= 0).sum()


def test_matcher_r50_and_handcrafted(golden, oracle_lib):
    g = golden("match.npz")
    anc = oracle_lib.anchors_emit(synth.levels_for(800, 1344), _cells(oracle_lib), 0.0)
    m, _ = oracle_lib.iou_match(anc, [g["r50_T8_gt"]])
    assert np.array_equal(m[0], g["r50_T8_matches"].astype(np.int64))
    m, _ = oracle_lib.iou_match(g["hand_anchors"], [g["hand_gt"]])
    assert m[0].tolist() == g["hand_matches"].tolist() == [-2, 2, -2, -1, 0, -2, -1, 2]
    m, _ = oracle_lib.iou_match(g["hand_anchors"], [g["hand2_gt"]])
    assert m[0].tolist() == g["hand2_matches"].tolist()


def test_matcher_batched_equals_per_image(golden, oracle_lib):
    g = golden("match.npz")
    anc = oracle_lib.anchors_emit(synth.levels_for(512, 512), _cells(oracle_lib), 0.0)
    gts = [g[f"r18_T{T}_gt"] for T in (8, 0, 64)]
    m, nfg = oracle_lib.iou_match(anc, gts)
    for i, T in enumerate((8, 0, 64)):
        assert np.array_equal(m[i], g[f"r18_T{T}_matches"].astype(np.int64))


def test_loss_toy(golden, oracle_lib):
    g = golden("loss.npz")
    gtb = [g["toy_gtb0"], np.zeros((0, 4), np.float32), g["toy_gtb2"]]
    gtl = [g["toy_gtl0"], np.zeros((0,), np.int64), g["toy_gtl2"]]
    m, nfg = oracle_lib.iou_match(g["toy_anchors"], gtb)
    assert np.array_equal(m, g["toy_matches"].astype(np.int64)) and np.array_equal(nfg, g["toy_nfg"])
    o = oracle_lib.loss_fwd_bwd(g["toy_cls"], g["toy_box"], g["toy_anchors"], gtb, gtl, m)
    np.testing.assert_allclose(o["loss"], g["toy_loss"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(o["per_image"], g["toy_per_image"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(o["gcls"], g["toy_gcls"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(o["gbox"], g["toy_gbox"], rtol=1e-5, atol=1e-9)
    assert o["per_image"][1].tolist() == [0.0, 0.0]          # Q7: empty GT -> zero loss
    assert not o["gcls"][1].any() and not o["gbox"][1].any()


def r18_loss_inputs(variant):
    rng = np.random.default_rng(18)
    cls, box = synth.head_outputs(rng, 2, 49104, 90)
    gtb, gtl = [], []
    for T in (8, 3):
        b, l = synth.gt_boxes(rng, T, 512, 512)
        gtb.append(b)
        gtl.append(l)
    if variant == "bf16":
        cls, box = synth.round_bf16(cls), synth.round_bf16(box)
    elif variant == "f16":
        cls, box = synth.round_f16(cls), synth.round_f16(box)
    return cls, box, gtb, gtl


@pytest.mark.parametrize("variant", ["f32", "bf16", "f16"])
def test_loss_r18(golden, oracle_lib, variant):
    g = golden("loss.npz")
    cls, box, gtb, gtl = r18_loss_inputs(variant)
    pre = f"r18_{variant}_"
    assert synth.sha(cls) + synth.sha(box) + synth.sha(np.concatenate(gtb)) == str(g[pre + "in_sha"]), "input RNG drift"
    anc = oracle_lib.anchors_emit(synth.levels_for(512, 512), _cells(oracle_lib), 0.0)
    m, nfg = oracle_lib.iou_match(anc, gtb)
    assert np.array_equal(nfg, g[pre + "nfg"])
    o = oracle_lib.loss_fwd_bwd(cls, box, anc, gtb, gtl, m)
    np.testing.assert_allclose(o["loss"], g[pre + "loss"], rtol=1e-5)
    np.testing.assert_allclose(o["per_image"], g[pre + "per_image"], rtol=1e-5)
    np.testing.assert_allclose(o["gcls"].reshape(-1)[g[pre + "gcls_idx"]], g[pre + "gcls_val"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(o["gbox"].reshape(-1)[g[pre + "gbox_idx"]], g[pre + "gbox_val"], rtol=1e-5, atol=1e-9)
    s = np.array([o["gcls"].astype(np.float64).sum(), np.abs(o["gcls"]).astype(np.float64).sum()])
    np.testing.assert_allclose(s, g[pre + "gcls_sum"], rtol=1e-5)
    s = np.array([o["gbox"].astype(np.float64).sum(), np.abs(o["gbox"]).astype(np.float64).sum()])
    np.testing.assert_allclose(s, g[pre + "gbox_sum"], rtol=1e-5, atol=1e-7)


def test_decode_encode(golden, oracle_lib):
    g = golden("decode.npz")
    np.testing.assert_allclose(oracle_lib.decode_clip(g["deltas"], g["anchors"], None), g["decoded"], rtol=1e-5, atol=1e-3)
    hw = tuple(int(v) for v in g["clip_hw"])
    np.testing.assert_allclose(oracle_lib.decode_clip(g["deltas"], g["anchors"], [hw]), g["clipped"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(oracle_lib.encode(g["enc_gt"], g["anchors"]), g["encoded"], rtol=1e-5, atol=1e-6)
    d2 = g["deltas"].copy()
    d2[:, 2:] = -3.0   # Q4: dw, dh never reach the decoded box
    assert np.array_equal(oracle_lib.decode_clip(d2, g["anchors"], None), oracle_lib.decode_clip(g["deltas"], g["anchors"], None))


def _cmp_dets(got, g, pre, B=2):
    for b in range(B):
        assert np.array_equal(got[b]["labels"], g[f"{pre}_labels{b}"])
        np.testing.assert_allclose(got[b]["scores"], g[f"{pre}_scores{b}"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(got[b]["boxes"], g[f"{pre}_boxes{b}"], rtol=1e-5, atol=1e-3)


def test_detect_toy(golden, oracle_lib):
    g = golden("detect.npz")
    hw = [tuple(int(x) for x in r) for r in g["toy_hw"]]
    _cmp_dets(oracle_lib.detect(g["toy_cls"], g["toy_box"], g["toy_anchors"], hw), g, "toy")
    p = oracle_lib.default_detect_params(0.2, 1e-2, 0.3, 7)
    _cmp_dets(oracle_lib.detect(g["toy_cls"], g["toy_box"], g["toy_anchors"], hw, p), g, "toy2")


def r18_detect_inputs(tag):
    mean, std, seed = {"sparse": (-7.0, 1.2, 21), "dense": (-6.0, 1.5, 22)}[tag]
    rng = np.random.default_rng(seed)
    return synth.head_outputs(rng, 2, 49104, 90, cls_mean=mean, cls_std=std, box_std=0.1)


@pytest.mark.parametrize("tag", ["sparse", "dense"])
def test_detect_r18(golden, oracle_lib, tag):
    g = golden("detect.npz")
    cls, box = r18_detect_inputs(tag)
    assert synth.sha(cls) + synth.sha(box) == str(g[f"r18_{tag}_in_sha"]), "input RNG drift"
    anc = oracle_lib.anchors_emit(synth.levels_for(512, 512), _cells(oracle_lib), 0.0)
    hw = [tuple(int(x) for x in r) for r in g[f"r18_{tag}_hw"]]
    _cmp_dets(oracle_lib.detect(cls, box, anc, hw), g, f"r18_{tag}")


@pytest.mark.parametrize("n", [0, 1, 5, 200, 1500])
def test_nms_keep_indices(golden, oracle_lib, n):
    g = golden("nms.npz")
    for thr in (0.5, 0.3):
        keep = oracle_lib.nms(g[f"n{n}_boxes"], g[f"n{n}_scores"], thr)
        assert np.array_equal(keep, g[f"n{n}_keep_{int(thr * 10)}"])


# ------------------------------------------------------------------ T1 transform
@pytest.mark.parametrize("tag", ["up", "ident", "down"])
def test_transform_oracle_vs_golden(oracle_lib, golden, tag):
    """Oracle T1 == the reference's transform (torchvision semantics on torch CPU ops): sizes exact,
    pixels within 2e-5 abs (values are O(1); torch's CPU bilinear blends in a different order)."""
    g = golden("transform.npz")
    imgs = [g[f"img{i}"] for i in range(3)]
    mn, mx = (int(v) for v in g[f"{tag}_cfg"])
    got, sizes = oracle_lib.transform_batch(imgs, mn, mx, g["mean"], g["std"])
    assert sizes == [tuple(int(v) for v in r) for r in g[f"{tag}_sizes"]]
    assert got.shape == g[f"{tag}_batch"].shape
    np.testing.assert_allclose(got, g[f"{tag}_batch"], rtol=0, atol=2e-5)
    if tag == "ident":      # image 0 is not resized: bit-exact normalisation, exact zero padding
        h, w = sizes[0]
        assert np.array_equal(got[0, :, :h, :w], (imgs[0] - g["mean"][:, None, None]) / g["std"][:, None, None])
        assert not got[0, :, h:, :].any() and not got[0, :, :, w:].any()
