"""Property tests (hypothesis) on the CPU oracle: the invariants the domain offers, independent of any
golden vector.  The same invariants are asserted on the HIP path at full size in test_hip_parity.py."""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle


def _boxes(rng, n, span=200.0):
    c = rng.uniform(0, span, size=(n, 2))
    wh = rng.uniform(2, span / 3, size=(n, 2))
    return np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)


def _iou(a, b):
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + ab[None, :] - inter)


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 120), st.sampled_from([0.3, 0.5, 0.7]))
def test_nms_greedy_invariants(seed, n, thr):
    rng = np.random.default_rng(seed)
    boxes, scores = _boxes(rng, n), rng.uniform(0, 1, n).astype(np.float32)
    keep = oracle.nms(boxes, scores, thr)
    assert len(set(keep.tolist())) == len(keep) and np.all(np.diff(scores[keep]) <= 0)       # unique, score order
    iou = _iou(boxes.astype(np.float64), boxes.astype(np.float64))
    kk = iou[np.ix_(keep, keep)]
    np.fill_diagonal(kk, 0)
    assert np.all(kk <= thr + 1e-6)                                # no kept pair overlaps more than thr
    dropped = np.setdiff1d(np.arange(n), keep)
    for d in dropped:                                              # every dropped box lost to a higher-scored kept box
        better = keep[(scores[keep] > scores[d]) | ((scores[keep] == scores[d]) & (keep < d))]
        assert np.any(iou[d, better] > thr - 1e-6)
    assert np.array_equal(oracle.nms(boxes[keep], scores[keep], thr), np.arange(len(keep)))  # idempotent


@settings(max_examples=30, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(0, 12), st.integers(1, 300))
def test_matcher_invariants(seed, T, A):
    rng = np.random.default_rng(seed)
    anchors, gt = _boxes(rng, A), _boxes(rng, T)
    m, nfg = oracle.iou_match(anchors, [gt])
    m = m[0]
    assert m.min() >= -2 and m.max() < max(T, 1) and nfg[0] == (m >= 0).sum()
    if T == 0:
        assert np.all(m == -2)
        return
    iou = _iou(gt, anchors).astype(np.float32)
    best = iou.max(0)
    assert np.all(best[m >= 0] > 0.5) and np.all(best[m == -1] < 0.4)
    assert np.all((best[m == -2] >= 0.4) & (best[m == -2] <= 0.5))
    fg = m >= 0
    assert np.all(iou[m[fg], np.nonzero(fg)[0]] == best[fg])       # matched to an arg-max ...
    assert np.all(m[fg] == iou[:, fg].argmax(0))                   # ... and the FIRST one


@settings(max_examples=30, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 200))
def test_encode_decode_round_trip_under_q4(seed, n):
    """decode(encode(gt)) reproduces the GT centre exactly-ish, while w,h follow exp(dx),exp(dy) (Q4)."""
    rng = np.random.default_rng(seed)
    anchors = _boxes(rng, n)
    gt = anchors + rng.uniform(-3, 3, size=anchors.shape).astype(np.float32)
    gt[:, 2:] = np.maximum(gt[:, 2:], gt[:, :2] + 1)
    d = oracle.encode(gt, anchors)
    dec = oracle.decode_clip(d, anchors, None)
    cx, cy = (dec[:, 0] + dec[:, 2]) / 2, (dec[:, 1] + dec[:, 3]) / 2
    np.testing.assert_allclose(cx, (gt[:, 0] + gt[:, 2]) / 2, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(cy, (gt[:, 1] + gt[:, 3]) / 2, rtol=1e-4, atol=1e-3)
    aw, ah = anchors[:, 2] - anchors[:, 0], anchors[:, 3] - anchors[:, 1]
    np.testing.assert_allclose(dec[:, 2] - dec[:, 0], aw * np.exp(d[:, 0]), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(dec[:, 3] - dec[:, 1], ah * np.exp(d[:, 1]), rtol=1e-4, atol=1e-3)


@settings(max_examples=15, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 3), st.integers(1, 5))
def test_loss_linearity_and_ignore_independence(seed, B, K):
    """Gradients sum like the loss (finite differences), scale with 1/B, and do not depend on ignored rows' logits."""
    rng = np.random.default_rng(seed)
    A = 64
    anchors = _boxes(rng, A)
    gtb = [_boxes(rng, 3) for _ in range(B)]
    gtl = [rng.integers(1, K + 1, 3).astype(np.int64) for _ in range(B)]
    cls = rng.normal(-2, 1.5, (B, A, K)).astype(np.float32)
    box = rng.normal(0, 0.3, (B, A, 4)).astype(np.float32)
    m, _ = oracle.iou_match(anchors, gtb)
    o = oracle.loss_fwd_bwd(cls, box, anchors, gtb, gtl, m)
    assert np.isclose(o["per_image"].sum(0)[::-1] / B, o["loss"], rtol=1e-5, atol=1e-7).all()
    cls2 = cls.copy()
    cls2[m == -2] += 5.0                                           # ignored rows: no effect on loss or other grads
    o2 = oracle.loss_fwd_bwd(cls2, box, anchors, gtb, gtl, m)
    assert np.array_equal(o2["loss"], o["loss"]) and np.array_equal(o2["gcls"], o["gcls"])
    assert not o["gcls"][m == -2].any() and not o["gbox"][m < 0].any()
    # regression gradient = finite difference of the regression loss (it IS the true derivative)
    if (m >= 0).any():
        b, a = np.argwhere(m >= 0)[0]
        eps = 1e-3
        bp, bm = box.copy(), box.copy()
        bp[b, a, 0] += eps
        bm[b, a, 0] -= eps
        fd = (oracle.loss_fwd_bwd(cls, bp, anchors, gtb, gtl, m, want_grads=False)["loss"][1]
              - oracle.loss_fwd_bwd(cls, bm, anchors, gtb, gtl, m, want_grads=False)["loss"][1]) / (2 * eps)
        assert abs(fd - o["gbox"][b, a, 0]) < 2e-2 * max(1.0, abs(fd))
