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
    # reference on the CPU (fp32): independent of MIOpen's solver choice for these odd shapes
    xf, wf = x.detach().float().cpu().requires_grad_(), conv.weight.detach().float().cpu().requires_grad_()
    yf = F.conv2d(xf, wf, None, 1, 1)
    yf.backward(dy.float().cpu())
    assert dx.shape == x.shape and dw.shape == conv.weight.shape
    rels = (_rel(y.cpu(), yf), _rel(dx.cpu(), xf.grad), _rel(dw.cpu(), wf.grad))
    assert max(rels) < 4e-3, rels
    # strided / dilated / grouped convolutions do not take this path
    assert not biasact.conv3x3_dgrad_fwd_fusable(nn.Conv2d(cin, cout, 3, 2, 1, bias=False).to(dev).to(torch.bfloat16), x)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,cout,cin,hw", [(2, 128, 128, (25, 34)), (1, 128, 64, (1, 1)), (3, 128, 128, (7, 300)), (2, 256, 128, (40, 9)), (1, 128, 128, (100, 168)),
                                           (2, 128, 192, (5, 2))])
def test_band_staged_dense_convolution_matches_torch(N, cout, cin, hw, dtype):
    """``rn_conv3x3_dense_band`` (conv2 of the layer2 bottlenecks): one band of 258 positions per (chunk, kernel row), image-edge taps zeroed at
    the fragment -- against fp32 convolution of the same 16-bit inputs, on shapes whose tiles straddle rows, images and the tensor's end."""
    from pytorch_retinanet_amd import biasact
    dev = torch.device("cuda:0")
    torch.manual_seed(10)
    x = torch.randn((N, cin, *hw), device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((cout, cin, 3, 3), device=dev) * 0.05).to(dtype).contiguous(memory_format=torch.channels_last)
    y = biasact.conv3x3_dense_band(x, w)
    ref = F.conv2d(x.float().cpu(), w.float().cpu(), None, 1, 1)
    assert y.shape == ref.shape and y.dtype == dtype
    assert _rel(y.cpu(), ref) < 4e-3
    assert float((y.float().cpu() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,hw", [(2, (50, 84)), (3, (9, 30)), (1, (1, 300)), (1, (2, 2))])
def test_band_staged_dense_convolution_with_statistics_in_the_epilogue(N, hw, dtype):
    """``rn_conv3x3_dense_band_stats``: the output is bit-identical to the plain launch and the per-tile partial sums finalize to what
    ``rn_bn_stats`` gives over the stored output (mean | invstd | a | b, running statistics)."""
    from pytorch_retinanet_amd import biasact, pwconv
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    Cc = 128
    M = N * hw[0] * hw[1]
    x = torch.randn((N, Cc, *hw), device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Cc, Cc, 3, 3), device=dev) * 0.05).to(dtype).contiguous(memory_format=torch.channels_last)
    y0 = biasact.conv3x3_dense_band(x, w)
    bn = nn.BatchNorm2d(Cc).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cc, device=dev) + 0.5)
        bn.bias.copy_(torch.randn(Cc, device=dev) * 0.1)
    y1, part, tiles = biasact.conv3x3_dense_band_stats(x, w)
    assert torch.equal(y1, y0) and tiles == (M + 255) // 256 and part.numel() == tiles * 2 * Cc
    bn_a = nn.BatchNorm2d(Cc).to(dev)
    bn_a.load_state_dict(bn.state_dict())
    bn_b = nn.BatchNorm2d(Cc).to(dev)
    bn_b.load_state_dict(bn.state_dict())
    got = pwconv.bn_finalize(part, tiles, M, bn_a)
    want = pwconv.bn_stats(y0, bn_b)
    torch.cuda.synchronize()
    assert torch.allclose(got, want, rtol=2e-5, atol=2e-6), float((got - want).abs().max())
    assert torch.allclose(bn_a.running_mean, bn_b.running_mean, rtol=2e-5, atol=1e-7)
    assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=2e-5, atol=1e-7)
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 1


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,cout,cin,hw", [(8, 512, 512, (25, 42)), (2, 256, 512, (7, 9)), (1, 512, 64, (3, 130)), (3, 256, 256, (31, 17))])
def test_k_split_dense_convolution_matches_torch(N, cout, cin, hw, dtype):
    """``rn_conv3x3_dense_splitk`` (conv2 of the layer4 bottlenecks, backbone.py:112,128; forward and, with flipped weights, data gradient):
    2 - 3 K ranges of the dense MFMA kernel + the f32 reduction, against fp32 convolution of the same 16-bit inputs; and the entry point says
    no (0 workspace bytes) when the plain launch already fills half the chip."""
    from pytorch_retinanet_amd import biasact
    from pytorch_retinanet_amd._lib import lib
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    x = torch.randn((N, cin, *hw), device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((cout, cin, 3, 3), device=dev) * 0.05).to(dtype).contiguous(memory_format=torch.channels_last)
    need = biasact.dense_splitk_bytes(x, w)
    assert need > 0
    y = biasact.conv3x3_same(x, w)
    ref = F.conv2d(x.float().cpu(), w.float().cpu(), None, 1, 1)
    assert y.shape == ref.shape and y.dtype == dtype
    assert _rel(y.cpu(), ref) < 4e-3
    # 33 600 positions x 256 channels = 132 workgroups: more than half the CUs, no split
    assert lib.rn_conv3x3_dense_splitk_workspace_bytes(8, 50, 84, 256) == 0
    assert lib.rn_conv3x3_dense_splitk_workspace_bytes(8, 25, 42, 500) == 0          # Cout not a multiple of 256


@pytest.mark.parametrize("N,cout,cin,hw", [
    (2, 64, 64, (40, 52)),        # one sub-problem; a row is one stage of two k-steps (52 > 32)
    (1, 64, 64, (3, 30)),         # rows shorter than one k-step, three image rows: every vertical tap leaves the image somewhere
    (2, 64, 64, (9, 97)),         # 97 = 64 + 33: second stage of a row runs both k-steps with 31 zero pixels
    (1, 64, 64, (5, 96)),         # 96 = 64 + 32: second stage runs ONE k-step
    (2, 128, 128, (25, 34)),      # four sub-problems
    (1, 128, 64, (7, 70)),        # rectangular: Cout != Cin
    (1, 512, 512, (13, 21)),      # layer4: 64 sub-problems, 4 workgroups each
    (8, 64, 64, (200, 334)),      # layer1 of the 800 x 1333 bucket (full size)
])
def test_narrow_weight_gradient_matches_torch(N, cout, cin, hw):
    "csrc/wgrad3x3.hip against autograd's weight gradient of F.conv2d on fp32 copies of the same bf16 tensors (backbone.py:112,128)."
    from pytorch_retinanet_amd import biasact
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(17)
    x = torch.randn((N, cin, *hw), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((N, cout, *hw), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.empty((cout, cin, 3, 3), device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert biasact.wgrad_narrow_ok(w, (1, 1), x)
    dw = biasact.conv3x3_wgrad_narrow(dy, x, w)
    assert dw.shape == w.shape and dw.is_contiguous(memory_format=torch.channels_last)
    # fp32 reference without MIOpen: dW[o][(c, kh, kw)] = sum_m dy[o][m] * unfold(x)[(c, kh, kw)][m], image by image (a GEMM each)
    ref = torch.zeros((cout, cin * 9), device=dev, dtype=torch.float32)
    for i in range(N):
        xu = F.unfold(x[i:i + 1].float().contiguous(), 3, padding=1)[0]                       # [cin * 9, H * W]
        ref += dy[i].float().contiguous().reshape(cout, -1) @ xu.t()
    ref = ref.reshape(cout, cin, 3, 3)
    # bf16 rounding of the result only: fp32 accumulation on both sides
    assert _rel(dw, ref) < 3e-3, _rel(dw, ref)
    # every tap separately (a swapped or shifted tap would pass a norm test on white noise only by luck -- it would not: checked anyway)
    for t in range(9):
        assert _rel(dw[:, :, t // 3, t % 3], ref[:, :, t // 3, t % 3]) < 4e-3, t


def test_narrow_weight_gradient_rejects_what_it_cannot_do():
    from pytorch_retinanet_amd import biasact
    from pytorch_retinanet_amd._lib import lib, RN_BF16, RN_F16, RN_F32
    dev = torch.device("cuda:0")
    x = torch.zeros((1, 64, 4, 4), device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert not biasact.wgrad_narrow_ok(torch.empty((96, 64, 3, 3), dtype=torch.bfloat16), (1, 1), x)
    assert not biasact.wgrad_narrow_ok(torch.empty((64, 64, 3, 3), dtype=torch.bfloat16), (2, 2), x)
    assert not biasact.wgrad_narrow_ok(torch.empty((64, 64, 1, 1), dtype=torch.bfloat16), (1, 1), x)
    ws = torch.empty((lib.rn_conv3x3_wgrad_narrow_workspace_bytes(64, 64),), dtype=torch.uint8, device=dev)
    dw = torch.empty((64 * 9 * 64,), dtype=torch.bfloat16, device=dev)
    z = biasact._zero_page(dev).data_ptr()
    assert lib.rn_conv3x3_wgrad_narrow(x.data_ptr(), x.data_ptr(), dw.data_ptr(), RN_F32, 1, 4, 4, 64, 64, z, ws.data_ptr(), ws.numel(), 0) != 0      # (bf16 and fp16 only)
    assert lib.rn_conv3x3_wgrad_narrow(x.data_ptr(), x.data_ptr(), dw.data_ptr(), RN_BF16, 1, 4, 4, 96, 64, z, ws.data_ptr(), ws.numel(), 0) != 0
    assert lib.rn_conv3x3_wgrad_narrow(x.data_ptr(), x.data_ptr(), dw.data_ptr(), RN_BF16, 1, 4, 4, 64, 64, z, ws.data_ptr(), 16, 0) != 0
    assert lib.rn_conv3x3_wgrad_narrow(x.data_ptr(), x.data_ptr(), dw.data_ptr(), RN_BF16, 1, 4, 4, 64, 64, 0, ws.data_ptr(), ws.numel(), 0) != 0
    assert lib.rn_conv3x3_wgrad_narrow_workspace_bytes(96, 64) == 0


@pytest.mark.parametrize("N,hw", [
    (2, (40, 52)),         # one strip, masked tail (52 < 128)
    (1, (1, 1)),           # a single pixel: every tap but the centre leaves the image
    (1, (3, 130)),         # two strips, the second holds two pixels; three rows: bands of one row on a 256-CU part
    (2, (9, 128)),         # exactly one full strip: the right halo pixel is outside the image
    (3, (5, 257)),         # three strips, one pixel in the last
    (1, (300, 129)),       # bands of several rows with a short last band
    (8, (200, 336)),       # layer1 of the 800 x 1333 bucket (full size)
    (2, (40, 112)),        # W % 128 == 112: wave 2's last half tile starts inside the image, the DMA piece behind it holds pixel W - 1
    (2, (40, 240)),        # the same in the second strip
    (2, (24, 144)),        # second strip with exactly 16 valid pixels: ONE store per row (the counted vmcnt follows the stores issued)
    (2, (24, 137)),        # ... 9 valid pixels
    (2, (24, 176)),        # 48 valid pixels: three stores
    (4, (64, 80)),         # wave 1 of the only strip: 16 valid pixels of its 64
])
def test_narrow_forward_matches_torch(N, hw):
    "csrc/narrow3x3.hip against F.conv2d on fp32 copies of the same bf16 tensors (conv2 of a layer1 bottleneck, backbone.py:112,128)."
    from pytorch_retinanet_amd import biasact
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(23)
    x = torch.randn((N, 64, *hw), device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((64, 64, 3, 3), device=dev, generator=g) * 0.06).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert biasact.narrow_fwd_ok(x, w)
    y = biasact.conv3x3_narrow_forward(x, w)
    assert y.shape == x.shape and y.is_contiguous(memory_format=torch.channels_last)
    ref = F.conv2d(x.float(), w.float(), None, 1, 1)
    # bf16 rounding of an fp32 sum of 576 products: half an ulp of the result
    assert _rel(y, ref) < 4e-3, _rel(y, ref)
    assert float((y.float() - ref).abs().max()) <= 2.0 ** -8 * float(ref.abs().max()) + 1e-6
    # a different weight per output channel and per tap: a transposed or mirrored tap order cannot pass
    w2 = torch.zeros_like(w)
    w2[5, 9, 0, 2] = 1.0                                   # y[:, 5, h, w] = x[:, 9, h - 1, w + 1]
    y2 = biasact.conv3x3_narrow_forward(x, w2)
    want = torch.zeros_like(x)
    if hw[0] > 1 and hw[1] > 1:
        want[:, 5, 1:, :-1] = x[:, 9, :-1, 1:]
    assert torch.equal(y2, want)
    # bias + ReLU in the epilogue (the folded-BatchNorm inference path): act(conv + bias), one rounding
    bias = torch.randn(64, device=dev, generator=g)
    yb = biasact.conv3x3_narrow_forward(x, w, bias, True)
    refb = F.relu(ref + bias[None, :, None, None])
    assert float((yb.float() - refb).abs().max()) <= 2.0 ** -8 * float(refb.abs().max()) + 1e-6 and float(yb.float().min()) >= 0.0
    yn = biasact.conv3x3_narrow_forward(x, w, bias, False)
    assert float((yn.float() - (ref + bias[None, :, None, None])).abs().max()) <= 2.0 ** -8 * float((ref + bias[None, :, None, None]).abs().max()) + 1e-6
    # other channel counts and dtypes are declined, not mis-computed
    assert not biasact.narrow_fwd_ok(x[:, :32], w[:, :32]) and not biasact.narrow_fwd_ok(x.float(), w.float())


def test_step_table_of_flipped_weights_is_used_only_while_current():
    "biasact.refresh_dgrad_weights: one launch flips every registered 3x3 weight; an entry serves a backward pass only while it is current."
    from pytorch_retinanet_amd import biasact
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    convs = [nn.Conv2d(c, c, 3, 1, 1, bias=False).to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last) for c in (64, 128, 64)]
    xs = [torch.randn((2, c.in_channels, 9, 13), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for c in convs]

    def dx_of(conv, x):
        x = x.detach().requires_grad_()
        biasact.conv3x3_dgrad_fwd(conv, x).sum().backward()
        return x.grad

    def want(conv, x):
        xf = x.detach().float().requires_grad_()
        F.conv2d(xf, conv.weight.detach().float(), None, 1, 1).sum().backward()
        return xf.grad

    st = torch.cuda.current_stream().cuda_stream
    for c, x in zip(convs, xs):                            # first use: flipped on the spot, registered
        assert _rel(dx_of(c, x), want(c, x)) < 4e-3
    assert all(biasact._DW_TABLE[id(c.weight)].ref() is c.weight for c in convs)
    assert biasact.refresh_dgrad_weights(dev) >= 3
    for c in convs:                                        # the table's entries are the flipped weights, bit for bit
        e = biasact._DW_TABLE[id(c.weight)]
        assert biasact.dgrad_weights([c.weight], st)[0] is e.flipped
        assert torch.equal(e.flipped, c.weight.detach().flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last))
    for c, x in zip(convs, xs):
        assert _rel(dx_of(c, x), want(c, x)) < 4e-3
    # an in-place update autograd sees (version counter) retires the entry without a refresh ...
    with torch.no_grad():
        convs[0].weight.mul_(-2.0)
    assert biasact.dgrad_weights([convs[0].weight], st)[0] is not biasact._DW_TABLE[id(convs[0].weight)].flipped
    assert _rel(dx_of(convs[0], xs[0]), want(convs[0], xs[0])) < 4e-3
    # ... a write behind its back (raw pointers: this package's optimizer) needs invalidate_dgrad_weights()
    biasact.refresh_dgrad_weights(dev)
    convs[1].weight.data.mul_(3.0)
    biasact.invalidate_dgrad_weights()
    assert _rel(dx_of(convs[1], xs[1]), want(convs[1], xs[1])) < 4e-3
    # a weight that no longer exists leaves the table at the next refresh
    key = id(convs[2].weight)
    del convs[2:], c, e
    import gc
    gc.collect()
    biasact.refresh_dgrad_weights(dev)
    assert key not in biasact._DW_TABLE
