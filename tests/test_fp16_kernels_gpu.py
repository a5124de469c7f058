"""GPU: the hand-written MFMA conv kernels instantiated on ``v_mfma_f32_*_f16`` (BASELINE configs[4] "R50-FPN fp16"; the reference's
own published run is native fp16 AMP, demo.ipynb precision=16) -- every kernel family against an fp32 PyTorch reference of the same
op on the same fp16 inputs.  The bf16 instantiations have their exhaustive shape coverage in ``test_model_gpu`` / ``test_dense_conv_gpu``
/ ``test_pwconv_gpu``; the kernels are the same templates, so this file checks one or two shapes per family at fp16's tolerance: the
result is ONE fp16 rounding (2^-11 relative) of an fp32-accumulated sum.  Also here: ``MasterSGD`` with fp16 working copies under
``torch.amp.GradScaler`` (device-side unscale and skip) against ``torch.optim.SGD`` with the same scaler.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H = torch.float16


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _rand(shape, scale=1.0, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return _cl((torch.randn(shape, device=DEV, generator=g) * scale).to(H))


def _rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-20))


def _close(got, ref, rel=2e-3, msg=""):
    "max error within `rel` of the reference's largest magnitude (one fp16 rounding is 4.9e-4 of the element)"
    ref = ref.float()
    err, tol = float((got.float() - ref).abs().max()), rel * float(ref.abs().max()) + 1e-6
    assert err <= tol, f"{msg}: max err {err} > {tol}"


def test_tower_conv_forward_and_gradients():
    "csrc/conv.hip, MODE_CANVAS + the canvas weight gradient (head towers, layers.py:143-171)."
    from pytorch_retinanet_amd import biasact
    torch.manual_seed(0)
    N, C, Hp, Wp = 2, 256, 14, 37
    mask2d = torch.zeros(Hp, Wp, dtype=torch.uint8, device=DEV)
    mask2d[1:-1, 1:-1] = (torch.rand(Hp - 2, Wp - 2, device=DEV) > 0.2).to(torch.uint8)
    x = _cl((torch.randn(N, C, Hp, Wp, device=DEV) * mask2d[None, None]).to(H)).requires_grad_(True)
    w = _cl(torch.randn(C, C, 3, 3, device=DEV) * 0.03).requires_grad_(True)
    b = (torch.randn(C, device=DEV) * 0.1).requires_grad_(True)
    assert biasact.tower_conv_fusable(x, torch.nn.Conv2d(C, C, 3, padding=1))
    y = biasact.tower_conv(x, w, b, mask2d.reshape(-1))
    assert y.dtype == H
    g = (torch.randn_like(y) * mask2d[None, None]).to(H)
    y.backward(g)
    xr, wr, br = x.detach().float().requires_grad_(True), w.detach().to(H).float().requires_grad_(True), b.detach().clone().requires_grad_(True)
    pre = F.conv2d(xr, wr, br, padding=1)
    _close(y, F.relu(pre.detach()) * mask2d[None, None].float(), 2e-3, "forward")
    assert not y.float()[:, :, mask2d == 0].any()
    (pre * (y.detach().float() > 0) * mask2d[None, None].float()).backward(g.float())
    interior = mask2d[None, None].bool().expand_as(xr.grad)
    _close(x.grad[interior], xr.grad[interior], 3e-3, "data gradient")
    _close(w.grad, wr.grad, 3e-3, "weight gradient")
    _close(b.grad, br.grad, 3e-3, "bias gradient")


@pytest.mark.parametrize("N,K,slots", [(2, 90, 2), (3, 6, 1)])
def test_cls_and_box_output_convs_on_the_canvas(N, K, slots):
    "csrc/conv.hip level modes (class- / box-output conv, layers.py:163-167, 235-251): dense per-level outputs, all three gradients."
    from pytorch_retinanet_amd import biasact
    torch.manual_seed(K)
    A, shapes = 9, [(13, 17), (7, 9), (4, 5)]
    for cout, per in ((A * K, K), (A * 4, 4)):
        conv = torch.nn.Conv2d(256, cout, 3, padding=1).to(DEV).to(memory_format=torch.channels_last)
        torch.nn.init.normal_(conv.weight, std=0.05)
        torch.nn.init.normal_(conv.bias, std=0.5)
        feats = [_cl(torch.randn(N, 256, h, w, device=DEV).to(H)).requires_grad_(True) for h, w in shapes]
        cv = biasact.Canvas.of(feats, pad=1, slots=slots)
        packed = biasact.pack_levels(cv, feats)
        if per == 4:
            assert biasact.box_output_conv_fusable(packed, conv, cv)
            ys = biasact.box_output_conv(packed, conv, cv, N)
        else:
            assert biasact.cls_output_conv_fusable(packed, conv, cv)
            ys = biasact.cls_output_conv(packed, conv, cv, K, N)
        assert ys[0].dtype == H
        gys = [torch.randn_like(y) for y in ys]
        torch.autograd.backward(ys, gys)
        w32, b32 = conv.weight.detach().to(H).float().requires_grad_(True), conv.bias.detach().clone().requires_grad_(True)
        for f, y, g, (h, w) in zip(feats, ys, gys, shapes):
            x32 = f.detach().float().requires_grad_(True)
            y3 = F.conv2d(x32, w32, b32, padding=1).permute(0, 2, 3, 1).reshape(N, h * w * A, per)
            y3.backward(g.float())
            _close(y, y3.detach(), 2e-3, f"forward {cout}")
            _close(f.grad, x32.grad, 3e-3, f"data gradient {cout}")
        _close(conv.weight.grad, w32.grad, 3e-3, f"weight gradient {cout}")
        _close(conv.bias.grad, b32.grad, 3e-3, f"bias gradient {cout}")


def test_dense_conv_group_and_conv2_of_layer3():
    "csrc/conv.hip MODE_DENSE: the FPN output convs in one launch (layers.py:34-38) and a 256-channel conv2 in all three directions."
    from pytorch_retinanet_amd import biasact
    g = torch.Generator(device=DEV).manual_seed(5)
    N, shapes = 2, [(25, 42), (13, 21), (7, 11)]
    convs = [torch.nn.Conv2d(256, 256, 3, 1, 1).to(DEV) for _ in shapes]
    xs = [_cl(torch.randn((N, 256, h, w), device=DEV, generator=g).to(H)).requires_grad_() for h, w in shapes]
    dys = [_cl(torch.randn((N, 256, h, w), device=DEV, generator=g).to(H)) for h, w in shapes]
    assert biasact.dense_group_fusable(xs, convs)
    ys = biasact.dense_conv_group(xs, convs)
    torch.autograd.backward(ys, dys)
    for x, c, dy, y in zip(xs, convs, dys, ys):
        xf, wf, bf = x.detach().float().requires_grad_(), c.weight.detach().to(H).float().requires_grad_(), c.bias.detach().clone().requires_grad_()
        yf = F.conv2d(xf, wf, bf, 1, 1)
        yf.backward(dy.float())
        assert y.dtype == H and _rel(y, yf) < 6e-4 and _rel(x.grad, xf.grad) < 6e-4 and _rel(c.weight.grad, wf.grad) < 6e-4
        _close(c.bias.grad, bf.grad, 1e-3, "bias gradient")
    conv = torch.nn.Conv2d(256, 256, 3, 1, 1, bias=False).to(DEV).to(H).to(memory_format=torch.channels_last)
    x = _rand((2, 256, 25, 42), 1.0, 9).requires_grad_()
    assert biasact.conv3x3_bwd_fusable(conv, x)
    y = biasact.conv3x3_mfma_bwd(conv, x)
    dy = _rand(tuple(y.shape), 1.0, 10)
    y.backward(dy)
    xf, wf = x.detach().float().requires_grad_(), conv.weight.detach().float().requires_grad_()
    yf = F.conv2d(xf, wf, None, 1, 1)
    yf.backward(dy.float())
    assert _rel(y, yf) < 6e-4 and _rel(x.grad, xf.grad) < 6e-4 and _rel(conv.weight.grad, wf.grad) < 6e-4


def test_narrow_forward_and_narrow_weight_gradient():
    "csrc/narrow3x3.hip and csrc/wgrad3x3.hip (conv2 of layer1 / layer2 / layer4, backbone.py:112,128)."
    from pytorch_retinanet_amd import biasact
    x = _rand((2, 64, 40, 112), 1.0, 23)
    w = _rand((64, 64, 3, 3), 0.06, 24)
    assert biasact.narrow_fwd_ok(x, w)
    y = biasact.conv3x3_narrow_forward(x, w)
    ref = F.conv2d(x.float(), w.float(), None, 1, 1)
    assert y.dtype == H and float((y.float() - ref).abs().max()) <= 2.0 ** -10 * float(ref.abs().max()) + 1e-6
    bias = torch.randn(64, device=DEV)
    yb = biasact.conv3x3_narrow_forward(x, w, bias, True)
    _close(yb, F.relu(ref + bias[None, :, None, None]), 1e-3, "bias + relu epilogue")
    for cout, cin, hw in ((64, 64, (9, 97)), (128, 128, (25, 34)), (512, 512, (13, 21))):
        xx, dy = _rand((2, cin, *hw), 1.0, 1), _rand((2, cout, *hw), 1.0, 2)
        ww = torch.empty((cout, cin, 3, 3), device=DEV, dtype=H).contiguous(memory_format=torch.channels_last)
        assert biasact.wgrad_narrow_ok(ww, (1, 1), xx)
        dw = biasact.conv3x3_wgrad_narrow(dy, xx, ww)
        ref = torch.zeros((cout, cin * 9), device=DEV)
        for i in range(2):
            ref += dy[i].float().contiguous().reshape(cout, -1) @ F.unfold(xx[i:i + 1].float().contiguous(), 3, padding=1)[0].t()
        assert dw.dtype == H and _rel(dw, ref.reshape(cout, cin, 3, 3)) < 6e-4, (cout, cin)


@pytest.mark.parametrize("shape,cout,k,stride", [((2, 256, 37, 45), 64, 1, 1), ((2, 64, 21, 25), 64, 3, 1), ((2, 256, 38, 46), 512, 1, 2), ((1, 1024, 9, 11), 256, 1, 1)])
def test_pw_gemm_forward_prologue_statistics_and_weight_gradient(shape, cout, k, stride):
    "csrc/pw.hip: plain product, relu(bn(x)) in the operand load + column statistics in the epilogue, and the position-contraction weight gradient."
    from pytorch_retinanet_amd import pwconv
    x, w = _rand(shape, 1.0, 1), _rand((cout, shape[1], k, k), 0.05, 2)
    pad = k // 2
    y = pwconv.pw_forward(x, w, stride=stride)
    ref = F.conv2d(x.float(), w.float(), None, stride, pad)
    assert y.dtype == H and y.shape == ref.shape
    _close(y, ref, 1e-3, "plain conv")
    Cin = shape[1]
    g = torch.Generator(device=DEV).manual_seed(3)
    coef = torch.cat([torch.rand(Cin, device=DEV, generator=g) + 0.5, torch.randn(Cin, device=DEV, generator=g) * 0.3])
    act = F.relu(torch.addcmul(coef[Cin:][None, :, None, None], x.float(), coef[:Cin][None, :, None, None])).to(H)
    M = y.shape[0] * y.shape[2] * y.shape[3]
    epi, partial, nb = pwconv.stats_epilogue(M, cout, x.device)
    y2 = pwconv.pw_forward(x, w, stride=stride, pro=pwconv.affine_relu(coef), epi=epi)
    _close(y2, F.conv2d(act.float(), w.float(), None, stride, pad), 1e-3, "conv of relu(bn(x))")
    part = partial.view(nb, 2, cout).double().sum(0)
    yr = y2.float().permute(0, 2, 3, 1).reshape(-1, cout).double()
    np.testing.assert_allclose(part[0].cpu().numpy(), yr.sum(0).cpu().numpy(), rtol=1e-3, atol=1e-3 * float(yr.abs().sum(0).max()))
    np.testing.assert_allclose(part[1].cpu().numpy(), (yr * yr).sum(0).cpu().numpy(), rtol=1e-3)
    gy = _rand(tuple(y.shape), 1.0, 4)
    dw = pwconv.pw_wgrad(gy, x, w, stride=stride)
    refw = torch.ops.aten.convolution_backward(gy.float(), x.float(), w.float(), None, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1,
                                               [False, True, False])[1]
    assert dw.dtype == H
    _close(dw, refw, 1e-3, "weight gradient")


def test_stem_forward_statistics_and_weight_gradient():
    "csrc/stem.hip (conv 7x7 / stride 2 with bn1's statistics, backbone.py:152-156, 246-251)."
    from pytorch_retinanet_amd import pwconv
    from pytorch_retinanet_amd.norm import FusedBatchNorm2d
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(DEV).to(H).to(memory_format=torch.channels_last)
    bn = FusedBatchNorm2d(64).to(DEV).train()
    x = _cl(torch.rand(2, 3, 64, 96, device=DEV).to(H))
    assert pwconv.stem_fusable(conv, bn, x)
    y = pwconv.stem(conv, bn, x)
    assert y.dtype == H
    z = F.conv2d(x.float(), conv.weight.float(), None, 2, 3).to(H).float()           # the stored (rounded) conv output
    ref = F.relu(F.batch_norm(z, None, None, bn.weight, bn.bias, True, 0.1, bn.eps))
    _close(y, ref, 3e-3, "relu(bn(conv))")
    gy = torch.randn_like(y)
    y.backward(gy)
    wr = conv.weight.detach().float().requires_grad_()
    yr = F.relu(F.batch_norm(F.conv2d(x.float(), wr, None, 2, 3), None, None, bn.weight.detach(), bn.bias.detach(), True, 0.1, bn.eps))
    yr.backward(gy.float())
    assert _rel(conv.weight.grad, wr.grad) < 2e-2, _rel(conv.weight.grad, wr.grad)      # (ReLU decisions at the rounding boundary differ)


def test_fused_bottleneck_block_in_fp16():
    "pwconv._BottleneckFn (backbone.py:105-136) on the fp16 instantiations: as close to the fp32 block as one fp16 rounding per stored tensor allows."
    from pytorch_retinanet_amd import backbone as bb, pwconv
    torch.manual_seed(5)
    blk = bb.Bottleneck(256, 64).to(DEV).to(memory_format=torch.channels_last).train()
    ref = bb.Bottleneck(256, 64).to(DEV).train()
    ref.load_state_dict(blk.state_dict())
    for m in blk.modules():
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = m.weight.data.to(H)
    x = _rand((2, 256, 25, 34), 1.0, 8).requires_grad_()
    assert pwconv.bottleneck_fusable(blk, x)
    y = blk(x)
    gy = _rand(tuple(y.shape), 1.0, 9)
    y.backward(gy)
    old = pwconv.FUSED_BOTTLENECK
    pwconv.FUSED_BOTTLENECK = False
    try:
        xr = x.detach().float().requires_grad_()
        yr = ref(xr)
        yr.backward(gy.float())
    finally:
        pwconv.FUSED_BOTTLENECK = old
    assert y.dtype == H and _rel(y, yr) < 3e-3, _rel(y, yr)
    assert _rel(x.grad, xr.grad) < 4e-2, _rel(x.grad, xr.grad)             # (ReLU decisions at the fp16 rounding boundary flip whole gradient elements)
    for (n, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        assert _rel(p.grad, q.grad) < 5e-2, (n, _rel(p.grad, q.grad))
    for (n, a), (_, b) in zip(blk.named_buffers(), ref.named_buffers()):
        if "running" in n:
            torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-3, msg=n)


def test_master_sgd_with_fp16_copies_under_a_grad_scaler_follows_torch_sgd():
    """fp32 masters + fp16 conv weights + MasterSGD under torch.amp.GradScaler == fp32 parameters + fp16 autocast + torch.optim.SGD under
    the same scaler (hparams.yaml:63-68 optimizer, Lightning precision=16): unscale on the device, a step with a non-finite gradient moves
    NOTHING (parameters, masters, momentum) and halves the scale -- on both sides."""
    from pytorch_retinanet_amd.norm import FusedBatchNorm2d
    from pytorch_retinanet_amd.optim import MasterSGD, master_state_dict, use_16bit_conv_weights

    def make():
        torch.manual_seed(11)
        return torch.nn.Sequential(torch.nn.Conv2d(8, 16, 3, padding=1), FusedBatchNorm2d(16), torch.nn.ReLU(),
                                   torch.nn.Conv2d(16, 8, 1, bias=False)).to(DEV).to(memory_format=torch.channels_last).train()
    a, b = make(), make()
    kw = dict(lr=0.05, momentum=0.9, weight_decay=1e-2)
    oa, ob = torch.optim.SGD(a.parameters(), **kw), None
    assert use_16bit_conv_weights(b, H) == 2 and b[0].weight.dtype == H
    ob = MasterSGD(b.parameters(), **kw)
    sa, sb = torch.amp.GradScaler("cuda", init_scale=256.0, growth_interval=100), torch.amp.GradScaler("cuda", init_scale=256.0, growth_interval=100)
    x = _cl(torch.randn(4, 8, 12, 10, device=DEV))
    for step in range(5):
        poison = step == 2                                    # this step's loss is infinite: both optimizers must skip it
        for m, o, s in ((a, oa, sa), (b, ob, sb)):
            o.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=H):
                loss = (m(x).float() ** 2).mean()
            if poison:
                loss = loss * float("inf")
            before = {k: v.clone() for k, v in (master_state_dict(m) if m is b else m.state_dict()).items()}
            s.scale(loss).backward()
            s.step(o)
            s.update()
            if poison:
                after = master_state_dict(m) if m is b else m.state_dict()
                for k in before:
                    if "running" not in k and "num_batches" not in k:
                        assert torch.equal(before[k], after[k]), (k, "moved on a skipped step")
    assert float(sa.get_scale()) == float(sb.get_scale()) == 128.0
    da, db = a.state_dict(), master_state_dict(b)
    for k in da:
        torch.testing.assert_close(db[k].float(), da[k].float(), rtol=2e-4, atol=2e-5, msg=k)
    assert torch.equal(b[0].weight.float(), db["0.weight"].to(H).float())                # working copy == fp16(master)
