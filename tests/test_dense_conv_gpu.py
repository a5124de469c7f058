"""GPU: the dense multi-geometry 3x3 MFMA convolution (``rn_conv3x3_dense_batched`` / ``_dense_wgrad_batched``, csrc/conv.hip
MODE_DENSE) that runs the FPN's output convs of P3 / P4 / P5 (reference: retinanet/layers.py:34-38, 62-64, there one
``nn.Conv2d`` per level) against torch's fp32 convolution of the same bf16 inputs: outputs, data gradients, weight
gradients and bias gradients, on shapes with ragged last tiles, one-pixel-wide levels and tiles that straddle images."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))


@pytest.mark.parametrize("N,shapes", [
    (2, [(25, 42), (13, 21), (7, 11)]),          # three levels, nothing a multiple of the 256-position tile
    (3, [(9, 30)]),                              # one level: 270 positions per image, tiles straddle images
    (1, [(1, 300), (300, 1), (1, 1), (2, 2)]),   # degenerate geometry: every tap row / column leaves the image somewhere
    (8, [(50, 84), (25, 42)]),                   # FPN P4 / P5 of the 800 x 1333 bucket
])
def test_dense_conv_group_matches_torch(N, shapes):
    from pytorch_retinanet_amd import biasact
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    convs = [nn.Conv2d(256, 256, 3, 1, 1).to(dev) for _ in shapes]
    for c in convs:
        with torch.no_grad():
            c.bias.copy_(torch.randn(256, device=dev, generator=g))
    xs = [torch.randn((N, 256, h, w), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
          for h, w in shapes]
    dys = [torch.randn((N, 256, h, w), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
           for h, w in shapes]
    assert biasact.dense_group_fusable(xs, convs)
    ys = biasact.dense_conv_group(xs, convs)
    torch.autograd.backward(ys, dys)
    got = [(y.detach(), x.grad.clone(), c.weight.grad.clone(), c.bias.grad.clone()) for y, x, c in zip(ys, xs, convs)]
    for x, c in zip(xs, convs):
        x.grad = None; c.weight.grad = None; c.bias.grad = None
    for p, (x, c, dy) in enumerate(zip(xs, convs, dys)):
        xf = x.detach().float().requires_grad_()
        wf = c.weight.detach().to(torch.bfloat16).float().requires_grad_()       # the kernel multiplies bf16 weights
        bf = c.bias.detach().clone().requires_grad_()
        yf = F.conv2d(xf, wf, bf, 1, 1)
        yf.backward(dy.float())
        y, dx, dw, db = got[p]
        assert y.shape == yf.shape and y.dtype == torch.bfloat16
        assert _rel(y, yf) < 4e-3, (p, "fwd", _rel(y, yf))                        # bf16 rounding of the output: 2^-9 relative per element
        assert _rel(dx, xf.grad) < 4e-3, (p, "dgrad", _rel(dx, xf.grad))
        assert _rel(dw, wf.grad) < 4e-3, (p, "wgrad", _rel(dw, wf.grad))
        assert _rel(db, bf.grad) < 1e-3, (p, "dbias", _rel(db, bf.grad))
        # element-wise as well: no position may be off by more than bf16 rounding of a sum this size
        tol = 0.02 * float(yf.abs().max())
        assert float((y.float() - yf).abs().max()) < tol, (p, float((y.float() - yf).abs().max()), tol)
        tol = 0.02 * float(xf.grad.abs().max())
        assert float((dx.float() - xf.grad).abs().max()) < tol
        tol = 0.02 * float(wf.grad.abs().max())
        assert float((dw.float() - wf.grad).abs().max()) < tol


def test_feature_pyramid_uses_the_dense_group_and_matches_the_per_level_path():
    "FeaturePyramid.forward under bf16 autocast: the grouped launch against the module-by-module (MIOpen) path, outputs and gradients."
    from pytorch_retinanet_amd import biasact
    from pytorch_retinanet_amd.layers import FeaturePyramid
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    fpn = FeaturePyramid(512, 1024, 2048).to(dev).to(memory_format=torch.channels_last)
    cs = [torch.randn((2, c, h, w), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
          for c, h, w in ((512, 28, 40), (1024, 14, 20), (2048, 7, 10))]

    def run(flag):
        biasact.DENSE_GROUP = flag
        for p in fpn.parameters(): p.grad = None
        for c in cs: c.grad = None
        biasact.MFMA_FLOP.pop("mfma_fpn_output_fwd_x3", None)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            outs = fpn(cs)
            loss = sum((o.float() ** 2).mean() for o in outs)
        loss.backward()
        used = "mfma_fpn_output_fwd_x3" in biasact.MFMA_FLOP
        return used, [o.detach().float() for o in outs], [c.grad.float().clone() for c in cs], {n: p.grad.clone() for n, p in fpn.named_parameters()}
    try:
        used1, o1, g1, p1 = run(True)
        used0, o0, g0, p0 = run(False)
    finally:
        biasact.DENSE_GROUP = True
    assert used1 and not used0
    for a, b in zip(o1, o0):
        assert _rel(a, b) < 1e-2
    for a, b in zip(g1, g0):
        assert _rel(a, b) < 2e-2
    for n in p1:
        assert _rel(p1[n], p0[n]) < 2e-2, n


def test_conv3x3_gradients_on_the_dense_kernels_match_torch():
    "Bottleneck conv2 of layer3 (256 -> 256, no bias): forward and both gradients on the dense MFMA kernels."
    from pytorch_retinanet_amd import biasact
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    conv = nn.Conv2d(256, 256, 3, 1, 1, bias=False).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    x = torch.randn((4, 256, 50, 84), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    dy = torch.randn((4, 256, 50, 84), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert biasact.conv3x3_bwd_fusable(conv, x)
    y = biasact.conv3x3_mfma_bwd(conv, x)
    y.backward(dy)
    dx, dw = x.grad.clone(), conv.weight.grad.clone()
    xf, wf = x.detach().float().requires_grad_(), conv.weight.detach().float().requires_grad_()
    yf = F.conv2d(xf, wf, None, 1, 1)
    yf.backward(dy.float())
    assert _rel(y, yf) < 4e-3 and _rel(dx, xf.grad) < 4e-3 and _rel(dw, wf.grad) < 4e-3, (_rel(y, yf), _rel(dx, xf.grad), _rel(dw, wf.grad))


@pytest.mark.parametrize("cout,cin,hw", [(64, 64, (40, 52)), (128, 128, (25, 34)), (512, 512, (13, 21)), (64, 128, (17, 19))])
def test_data_gradient_issued_as_a_forward_convolution(cout, cin, hw):
    "conv2 of layer1 / layer2 / layer4: dx = conv(dy, flipped transposed weights) -- against autograd on fp32 (backbone.py:112,128)."
    from pytorch_retinanet_amd import biasact
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=False).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
    x = torch.randn((2, cin, *hw), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    dy = torch.randn((2, cout, *hw), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert biasact.conv3x3_dgrad_fwd_fusable(conv, x)
    y = biasact.conv3x3_dgrad_fwd(conv, x)
    y.backward(dy)
    dx, dw = x.grad.clone(), conv.weight.grad.clone()
    xf, wf = x.detach().float().requires_grad_(), conv.weight.detach().float().requires_grad_()
    yf = F.conv2d(xf, wf, None, 1, 1)
    yf.backward(dy.float())
    assert dx.shape == x.shape and dw.shape == conv.weight.shape
    assert _rel(y, yf) < 4e-3 and _rel(dx, xf.grad) < 4e-3 and _rel(dw, wf.grad) < 4e-3, (_rel(y, yf), _rel(dx, xf.grad), _rel(dw, wf.grad))
    # strided / dilated / grouped convolutions do not take this path
    assert not biasact.conv3x3_dgrad_fwd_fusable(nn.Conv2d(cin, cout, 3, 2, 1, bias=False).to(dev).to(torch.bfloat16), x)
