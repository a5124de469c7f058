"""COCO wire format + pycocotools-free bbox mAP.  pycocotools is not in the image, so the
metric is checked against hand-computed known answers of the published COCOeval protocol."""
import numpy as np
import pytest
import torch

from pytorch_retinanet_amd.coco_eval import BBoxEval, CocoEvaluator, convert_to_xywh, gt_from_dataset, prepare_for_coco_detection


def _gt(img, cat, box, **kw):
    return dict(image_id=img, category_id=cat, bbox=list(box), **kw)


def _dt(img, cat, box, score):
    return dict(image_id=img, category_id=cat, bbox=list(box), score=score)


def test_wire_format():
    "coco_eval.py:71-93 / :159-161: xyxy -> xywh rows, python scalars, empty predictions skipped."
    preds = {7: {"boxes": torch.tensor([[10.0, 20.0, 50.0, 80.0], [0.0, 0.0, 5.0, 5.0]]), "scores": torch.tensor([0.9, 0.25]),
                 "labels": torch.tensor([3, 1])}, 9: {}}
    rows = prepare_for_coco_detection(preds)
    assert rows == [{"image_id": 7, "category_id": 3, "bbox": [10.0, 20.0, 40.0, 60.0], "score": pytest.approx(0.9)},
                    {"image_id": 7, "category_id": 1, "bbox": [0.0, 0.0, 5.0, 5.0], "score": pytest.approx(0.25)}]
    assert torch.equal(convert_to_xywh(torch.tensor([[1.0, 2.0, 4.0, 8.0]])), torch.tensor([[1.0, 2.0, 3.0, 6.0]]))


def test_perfect_detections():
    gt = [_gt(1, 1, (0, 0, 50, 50)), _gt(1, 2, (100, 100, 20, 20)), _gt(2, 1, (10, 10, 200, 200))]
    dt = [_dt(g["image_id"], g["category_id"], g["bbox"], 0.9) for g in gt]
    s = BBoxEval(gt, dt).evaluate().summarize(verbose=False)
    assert s[[0, 1, 2, 8]] == pytest.approx(1.0) and s[[3, 4, 5]] == pytest.approx(1.0)            # 400 px^2 small, 2500 medium, 40000 large


def test_interleaved_false_positive_known_ap():
    """2 GT, detections TP(.9) FP(.8) TP(.7): precision envelope [1, 2/3, 2/3] at recalls [.5, .5, 1]
    -> 51 recall points at 1 and 50 at 2/3, for every IoU threshold."""
    gt = [_gt(1, 1, (0, 0, 40, 40)), _gt(1, 1, (100, 100, 40, 40))]
    dt = [_dt(1, 1, (0, 0, 40, 40), 0.9), _dt(1, 1, (300, 300, 40, 40), 0.8), _dt(1, 1, (100, 100, 40, 40), 0.7)]
    s = BBoxEval(gt, dt).evaluate().summarize(verbose=False)
    want = (51 * 1.0 + 50 * 2.0 / 3.0) / 101
    assert s[0] == pytest.approx(want, abs=1e-9) and s[1] == pytest.approx(want, abs=1e-9)
    assert s[6] == pytest.approx(0.5) and s[7] == pytest.approx(1.0)      # AR@1 sees one detection, AR@10 all


def test_iou_thresholds():
    "IoU 0.78 counts at thresholds .50 .. .75 (6 of 10) and is a false positive above."
    gt = [_gt(1, 1, (0, 0, 100, 100))]
    dt = [_dt(1, 1, (0, 0, 100, 78), 0.9)]
    s = BBoxEval(gt, dt).evaluate().summarize(verbose=False)
    assert s[0] == pytest.approx(0.6) and s[1] == pytest.approx(1.0) and s[2] == pytest.approx(1.0) and s[8] == pytest.approx(0.6)


def test_area_ranges_and_missing_bins():
    gt = [_gt(1, 1, (0, 0, 10, 10))]                                  # 100 px^2: small only
    s = BBoxEval(gt, [_dt(1, 1, (0, 0, 10, 10), 0.5)]).evaluate().summarize(verbose=False)
    assert s[3] == pytest.approx(1.0) and s[4] == -1.0 and s[5] == -1.0 and s[9] == pytest.approx(1.0) and s[10] == -1.0


def test_crowd_ground_truth_is_ignored():
    """A detection inside a crowd region is neither TP nor FP; the crowd box is not a recall target."""
    gt = [_gt(1, 1, (0, 0, 40, 40)), _gt(1, 1, (200, 200, 100, 100), iscrowd=1)]
    dt = [_dt(1, 1, (0, 0, 40, 40), 0.9), _dt(1, 1, (210, 210, 30, 30), 0.8)]
    s = BBoxEval(gt, dt).evaluate().summarize(verbose=False)
    assert s[0] == pytest.approx(1.0) and s[8] == pytest.approx(1.0)
    dt.append(_dt(1, 1, (500, 500, 30, 30), 0.95))                    # a real false positive ranked first
    s = BBoxEval(gt, dt).evaluate().summarize(verbose=False)
    assert s[0] == pytest.approx(0.5)                                 # precision 1/2 at every recall point


def test_categories_are_evaluated_separately_and_missing_detections_cost_recall():
    gt = [_gt(1, 1, (0, 0, 40, 40)), _gt(1, 2, (0, 0, 40, 40))]
    dt = [_dt(1, 1, (0, 0, 40, 40), 0.9)]                             # right box, but only for category 1
    s = BBoxEval(gt, dt).evaluate().summarize(verbose=False)
    assert s[0] == pytest.approx(0.5) and s[8] == pytest.approx(0.5)


def test_evaluator_flow_matches_reference_call_sequence():
    "model.py:141-157: update per batch -> accumulate -> summarize -> coco_eval['bbox'].stats[0]."
    class DS:
        def __len__(self): return 2
        def __getitem__(self, i):
            t = {"boxes": torch.tensor([[0.0, 0.0, 40.0, 40.0], [100.0, 100.0, 140.0, 140.0]]), "labels": torch.tensor([1, 1]),
                 "image_id": torch.tensor([i + 10])}
            return torch.zeros(3, 8, 8), t, i + 10
    ev = CocoEvaluator(gt_from_dataset(DS()), ["bbox"])
    for img in (10, 11):
        ev.update({img: {"boxes": torch.tensor([[0.0, 0.0, 40.0, 40.0], [300.0, 300.0, 340.0, 340.0], [100.0, 100.0, 140.0, 140.0]]),
                         "scores": torch.tensor([0.9, 0.8, 0.7]), "labels": torch.tensor([1, 1, 1])}})
    ev.synchronize_between_processes()
    ev.accumulate()
    ev.summarize(verbose=False)
    # merged ranking over both images: TP TP FP FP TP TP -> envelope precision 1 up to recall .5, then 2/3
    want = (51 * 1.0 + 50 * 2.0 / 3.0) / 101
    assert ev.coco_eval["bbox"].stats[0] == pytest.approx(want, abs=1e-9)
    with pytest.raises(ValueError):
        CocoEvaluator([], ["segm"])


def test_csv_dataset_reader(tmp_path):
    "README.md:103-112 format; items carry the reference's target keys (pascal_utils.py:98-142)."
    from PIL import Image
    from pytorch_retinanet_amd.datasets import CSVDetectionDataset
    rng = np.random.default_rng(0)
    for name, (h, w) in {"a.png": (20, 30), "b.png": (16, 16)}.items():
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(tmp_path / name)
    (tmp_path / "d.csv").write_text(
        "filename,width,height,class,xmin,ymin,xmax,ymax,labels\n"
        "a.png,30,20,chair,2,3,12,13,9\n"
        "b.png,16,16,table,1,1,9,11,11\n"
        "a.png,30,20,table,5,5,25,15,11\n")
    ds = CSVDetectionDataset(str(tmp_path / "d.csv"))
    assert len(ds) == 2
    im, tg, idx = ds[0]
    assert im.shape == (3, 20, 30) and im.dtype == torch.float32 and 0.0 <= float(im.min()) and float(im.max()) <= 1.0
    assert tg["boxes"].tolist() == [[2.0, 3.0, 12.0, 13.0], [5.0, 5.0, 25.0, 15.0]] and tg["labels"].tolist() == [9, 11]
    assert tg["area"].tolist() == [100.0, 200.0] and tg["iscrowd"].tolist() == [0, 0] and int(idx) == 0 and int(tg["image_id"]) == 0
    rows = gt_from_dataset(ds)
    assert len(rows) == 3 and rows[0]["bbox"] == [2.0, 3.0, 10.0, 10.0] and rows[2]["image_id"] == 1
    with pytest.raises(ValueError):
        (tmp_path / "bad.csv").write_text("filename,xmin\nx,1\n")
        CSVDetectionDataset(str(tmp_path / "bad.csv"))


# ------------------------------------------------------------------ properties (hypothesis)
from hypothesis import given, settings, strategies as st


@st.composite
def _scene(draw):
    n_img = draw(st.integers(1, 3))
    gts, dts = [], []
    for img in range(n_img):
        for _ in range(draw(st.integers(0, 4))):
            x, y = draw(st.integers(0, 200)), draw(st.integers(0, 200))
            w, h = draw(st.integers(5, 120)), draw(st.integers(5, 120))
            gts.append(_gt(img, draw(st.integers(1, 2)), (x, y, w, h)))
        for _ in range(draw(st.integers(0, 6))):
            x, y = draw(st.integers(0, 200)), draw(st.integers(0, 200))
            w, h = draw(st.integers(5, 120)), draw(st.integers(5, 120))
            dts.append(_dt(img, draw(st.integers(1, 2)), (x, y, w, h), draw(st.integers(1, 1000)) / 1000.0))
    return gts, dts


@settings(max_examples=40, deadline=None)
@given(_scene())
def test_property_stats_are_bounded_and_order_independent(scene):
    gts, dts = scene
    s1 = BBoxEval([dict(g) for g in gts], [dict(d) for d in dts]).evaluate().summarize(verbose=False)
    assert all(v == -1.0 or -1e-9 <= v <= 1.0 + 1e-9 for v in s1)
    # the order of detections only matters for exact score ties, the order of ground truth only for boxes of one
    # (image, category) that tie on IoU with a detection
    scores = [d["score"] for d in dts]
    if len(set(scores)) == len(scores):
        s3 = BBoxEval([dict(g) for g in gts], [dict(d) for d in reversed(dts)]).evaluate().summarize(verbose=False)
        assert np.allclose(s1, s3)
    if len({(g["image_id"], g["category_id"]) for g in gts}) == len(gts):
        s2 = BBoxEval([dict(g) for g in reversed(gts)], [dict(d) for d in dts]).evaluate().summarize(verbose=False)
        assert np.allclose(s1, s2)


@settings(max_examples=25, deadline=None)
@given(_scene())
def test_property_detecting_every_ground_truth_exactly_gives_ap_one(scene):
    gts, _ = scene
    if not gts:
        return
    dts = [_dt(g["image_id"], g["category_id"], g["bbox"], 0.5 + 0.001 * i) for i, g in enumerate(gts)]
    # identical boxes of one (image, category) would steal each other's matches only at equal IoU: still all TP
    s = BBoxEval([dict(g) for g in gts], dts).evaluate().summarize(verbose=False)
    assert s[0] == pytest.approx(1.0) and s[8] == pytest.approx(1.0)


def test_hand_computed_protocol_fixtures():
    """tests/golden/coco_protocol_cases.json: multi-class / area-range / maxDets cases whose 12 statistics were derived by
    hand from the published COCO protocol (each case carries its derivation); reference evaluator call sites:
    utils/coco/coco_eval.py:15-156, model.py:136-146."""
    import json
    import os
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "coco_protocol_cases.json")))
    for case in cases["cases"]:
        s = BBoxEval(case["gt"], case["dt"]).evaluate().summarize(verbose=False)
        assert len(s) == 12
        np.testing.assert_allclose(s, case["stats"], atol=1e-9, err_msg=case["name"])
