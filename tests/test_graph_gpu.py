"""GPU: the whole train step captured in a hipGraph (``graph.CapturedTrainStep``) -- one host call per step instead of
~700 launches -- performs the same optimisation as the eager step.

The reference has no analogue (it enqueues ~40 torch ops per image from a Python loop, retinanet/losses.py:66-126); the bar
is equality with this package's own eager step: same loss trajectory and same parameters after N steps, up to the
non-determinism of MIOpen's atomically accumulated weight gradients (the tolerance ``test_model_gpu`` uses for two eager
runs), BN running statistics advancing once per replay, and derived caches (frozen-BN fold) seeing the replayed updates.
"""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(seed=11):
    import pytorch_retinanet_amd as P
    from pytorch_retinanet_amd.optim import MasterSGD, use_bf16_conv_weights
    torch.manual_seed(seed)
    net = P.Retinanet(num_classes=5, backbone_kind="resnet18", pretrained=False, min_size=128, max_size=160).to(DEV)
    net = net.to(memory_format=torch.channels_last).train()
    use_bf16_conv_weights(net)
    opt = MasterSGD(net.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-3)
    return net, opt


def _batches(n, T=3, seed=5):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        images = [torch.from_numpy(rng.random((3, 128, 160), dtype=np.float32)).to(DEV) for _ in range(2)]
        targets = []
        for _ in range(2):
            b, l = synth.gt_boxes(rng, T, 128, 160, num_classes=5, wh_lo=20.0, wh_hi=90.0)
            targets.append({"boxes": torch.from_numpy(b).to(DEV), "labels": torch.from_numpy(l).to(DEV)})
        out.append((images, targets))
    return out


def _params(net):
    return {n: (p.master if hasattr(p, "master") else p.data).detach().float().cpu() for n, p in net.named_parameters()}


def test_captured_step_equals_the_eager_step():
    from pytorch_retinanet_amd.graph import CapturedTrainStep
    data = _batches(8)
    res = {}
    for captured in (False, True):
        net, opt = _setup()
        initial = _params(net)
        step = CapturedTrainStep(net, opt, amp_dtype=torch.bfloat16, eager_steps=2, enabled=captured)
        losses = [float(step(im, tg)["loss"]) for im, tg in data]
        torch.cuda.synchronize()
        res[captured] = (losses, _params(net), {n: b.detach().float().cpu() for n, b in net.named_buffers()}, step.replays, step.captures)
    assert res[False][3] == 0 and res[True][3] == len(data) - 2 and res[True][4] == 1       # 2 eager steps, 1 capture, 6 replays
    la, lb = np.array(res[False][0]), np.array(res[True][0])
    assert np.all(np.isfinite(lb))
    np.testing.assert_allclose(lb, la, rtol=2e-2)             # bf16 step, atomics in MIOpen's weight gradients
    for k, a in res[False][1].items():
        torch.testing.assert_close(res[True][1][k], a, rtol=0, atol=2e-3, msg=k)
    moved = sum(float((res[True][1][k] - initial[k]).abs().max()) > 0 for k in initial)
    assert moved > len(initial) // 2                            # the steps (6 of 8 of them replays) really moved the parameters
    for k, a in res[False][2].items():
        if "num_batches_tracked" in k:
            assert int(res[True][2][k]) == int(a) == len(data)  # BN statistics advanced once per replay
        elif "running_" in k:
            # (deep layers see 4 x 5 positions per image here: their batch statistics amplify the bf16 / atomics noise)
            torch.testing.assert_close(res[True][2][k], a, rtol=5e-2, atol=3e-2 * float(a.abs().max()) + 1e-3, msg=k)


def test_a_new_input_signature_gets_its_own_graph_and_folds_see_replayed_updates():
    from pytorch_retinanet_amd import backbone
    from pytorch_retinanet_amd.graph import CapturedTrainStep
    net, opt = _setup(3)
    step = CapturedTrainStep(net, opt, amp_dtype=torch.bfloat16, eager_steps=1)
    a, b = _batches(4, T=3, seed=1), _batches(4, T=5, seed=2)   # different GT counts -> different signatures
    x = torch.randn(2, 3, 128, 160, device=DEV).contiguous(memory_format=torch.channels_last)

    def eval_c3():
        net.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            y = net.backbone(x)[0].float().clone()
        net.train()
        return y
    for (ia, ta), (ib, tb) in zip(a, b):
        la, lb = step(ia, ta), step(ib, tb)
        assert np.isfinite(float(la["loss"])) and np.isfinite(float(lb["loss"]))
    assert step.captures == 2 and step.replays == 6
    before = eval_c3()                                          # folds the frozen BN into the conv weights (cached)
    step(*a[0])                                                 # a replay: weights and running statistics change by raw pointers
    after = eval_c3()
    assert not torch.allclose(before, after)
    backbone.FOLD_FROZEN_BN = False
    try:
        ref = eval_c3()
    finally:
        backbone.FOLD_FROZEN_BN = True
    torch.testing.assert_close(after, ref, rtol=4e-2, atol=4e-2 * float(ref.abs().max()))
    # a changed learning rate is part of the signature: the next call is a fresh (eager) step, not a stale replay
    for g in opt.param_groups:
        g["lr"] = 5e-3
    n = step.replays
    step(*a[1])
    assert step.replays == n
