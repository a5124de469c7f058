"""GPU: the backbone convolutions on the MFMA GEMMs of csrc/pw.hip (``pwconv``) against plain PyTorch fp32 references of
the same ops on the same bf16 inputs (reference semantics: the Bottleneck of retinanet/backbone.py:105-136 and its autograd
backward).  Kernel level: every prologue / epilogue combination, 1x1 and 3x3, stride 1 and 2, ragged row counts.
Block level: ``pwconv._BottleneckFn`` == the layer-by-layer path of ``backbone.Bottleneck`` (outputs, input gradient,
parameter gradients, running statistics).  Tolerances: bf16 outputs to one rounding of the fp32 reference (rel 1e-2 of the
tensor's max), fp32-accumulated statistics rel <= 1e-3, weight gradients rel 2e-2 of their max.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _rand(shape, scale=1.0, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return _cl((torch.randn(shape, device=DEV, generator=g) * scale).to(torch.bfloat16))


def _close(got, ref, rel, msg=""):
    ref = ref.float()
    tol = rel * float(ref.abs().max()) + 1e-6
    err = float((got.float() - ref).abs().max())
    assert err <= tol, f"{msg}: max err {err} > {tol}"


def _close_grad(got, ref, rel, msg=""):
    """Gradients of two forward passes that differ in the last bf16 bits: where an activation sits at the ReLU boundary the
    mask flips and the whole upstream element appears / disappears, so isolated outliers are legitimate.  Bars: relative L2
    error and the fraction of elements outside the elementwise tolerance."""
    got, ref = got.float(), ref.float()
    l2 = float((got - ref).norm() / (ref.norm() + 1e-12))
    out = float(((got - ref).abs() > rel * float(ref.abs().max()) + 1e-6).float().mean())
    assert l2 <= rel and out <= 2e-3, f"{msg}: relative L2 error {l2}, outlier fraction {out}"


def _alive(t):              # a bf16-rounded activation is positive
    return t.to(torch.bfloat16).float() > 0


@pytest.mark.parametrize("shape,cout,k,stride", [((2, 256, 37, 45), 64, 1, 1), ((2, 64, 37, 45), 256, 1, 1), ((3, 128, 20, 24), 512, 1, 1),
                                                 ((2, 256, 38, 46), 512, 1, 2), ((2, 64, 21, 25), 64, 3, 1), ((2, 128, 21, 25), 128, 3, 2),
                                                 ((1, 1024, 9, 11), 256, 1, 1), ((1, 512, 9, 11), 2048, 1, 1)])
def test_forward_plain_and_with_bn_relu_prologue_and_statistics(shape, cout, k, stride):
    from pytorch_retinanet_amd import pwconv
    from pytorch_retinanet_amd._lib import lib
    x = _rand(shape, 1.0, 1)
    w = _rand((cout, shape[1], k, k), 0.05, 2)
    pad = k // 2
    # plain
    y = pwconv.pw_forward(x, w, stride=stride)
    ref = F.conv2d(x.float(), w.float(), None, stride, pad)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    _close(y, ref, 1e-2, "plain conv")
    # relu(x * a + b) in the operand load (zero padding applies to the ACTIVATION) + column statistics of the output
    Cin = shape[1]
    g = torch.Generator(device=DEV).manual_seed(3)
    coef = torch.cat([torch.rand(Cin, device=DEV, generator=g) + 0.5, torch.randn(Cin, device=DEV, generator=g) * 0.3])
    act = F.relu(torch.addcmul(coef[Cin:][None, :, None, None], x.float(), coef[:Cin][None, :, None, None])).to(torch.bfloat16)
    M = y.shape[0] * y.shape[2] * y.shape[3]
    epi, partial, nb = pwconv.stats_epilogue(M, cout, x.device)
    y2 = pwconv.pw_forward(x, w, stride=stride, pro=pwconv.affine_relu(coef), epi=epi)
    ref2 = F.conv2d(act.float(), w.float(), None, stride, pad)
    _close(y2, ref2, 1e-2, "conv of relu(bn(x))")
    part = partial.view(nb, 2, cout).double().sum(0)
    yr = y2.float().permute(0, 2, 3, 1).reshape(-1, cout).double()
    np.testing.assert_allclose(part[0].cpu().numpy(), yr.sum(0).cpu().numpy(), rtol=1e-3, atol=1e-3 * float(yr.abs().sum(0).max()))
    np.testing.assert_allclose(part[1].cpu().numpy(), (yr * yr).sum(0).cpu().numpy(), rtol=1e-3)


def _bn_bwd_coefs(C, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    a = torch.rand(C, device=DEV, generator=g) + 0.5
    k0 = torch.randn(C, device=DEV, generator=g) * 0.05
    k1 = torch.randn(C, device=DEV, generator=g) * 0.05
    return torch.cat([a, k0, k1])


@pytest.mark.parametrize("relu_mode", [0, 2, 3])
@pytest.mark.parametrize("cin,cout", [(256, 64), (512, 128)])
def test_data_gradient_with_bn_backward_prologue_and_relu_backward_epilogue(relu_mode, cin, cout):
    "The conv3 data gradient of a bottleneck: dz3 = a g' + k1 z3 + k0 formed in the operand load, ReLU mask + BN-backward sums out."
    from pytorch_retinanet_amd import pwconv
    from pytorch_retinanet_amd._lib import RN_PW_EPI_RELU_BWD, RnPwEpilogue, lib
    shape = (2, cin, 19, 23)
    gup, z = _rand(shape, 1.0, 1), _rand(shape, 1.0, 2)
    coef3 = _bn_bwd_coefs(cin, 3)
    gen = torch.Generator(device=DEV).manual_seed(4)
    fwd = torch.cat([torch.rand(cin, device=DEV, generator=gen) + 0.5, torch.randn(cin, device=DEV, generator=gen) * 0.3])
    M = shape[0] * shape[2] * shape[3]
    zr = z.float().permute(0, 2, 3, 1).reshape(M, cin)
    gr = gup.float().permute(0, 2, 3, 1).reshape(M, cin)
    bits = None
    if relu_mode == 2:
        mask = _alive(torch.addcmul(fwd[cin:], zr, fwd[:cin]))
    elif relu_mode == 3:
        mask = torch.rand(M, cin, device=DEV, generator=gen) > 0.4
        bits = (mask.view(M, cin // 8, 8).to(torch.int32) * (2 ** torch.arange(8, device=DEV, dtype=torch.int32))).sum(-1).to(torch.uint8).contiguous().view(-1)
    else:
        mask = torch.ones_like(zr, dtype=torch.bool)
    dz = (coef3[:cin] * (gr * mask) + (coef3[2 * cin:] * zr + coef3[cin:2 * cin])).to(torch.bfloat16).float()
    w = _rand((cout, cin, 1, 1), 0.05, 5)                       # the transposed forward weight
    zp = _rand((2, cout, 19, 23), 1.0, 6)
    st = torch.cat([torch.randn(cout, device=DEV, generator=gen) * 0.2, torch.rand(cout, device=DEV, generator=gen) + 0.5,
                    torch.rand(cout, device=DEV, generator=gen) + 0.5, torch.randn(cout, device=DEV, generator=gen) * 0.3])   # mean | invstd | a | b
    nb = lib.rn_pw_walkers(M)
    part = torch.empty((nb * 2 * cout,), dtype=torch.float32, device=DEV)
    p = st.data_ptr()
    epi = RnPwEpilogue(RN_PW_EPI_RELU_BWD, part.data_ptr(), 0, 0, zp.data_ptr(), p + 8 * cout, p + 12 * cout, p, p + 4 * cout)
    pro = pwconv.bn_bwd(coef3, z, relu_mode, fwd_coef=fwd if relu_mode == 2 else None, bits=bits)
    dy = pwconv.pw_forward(gup, w, pro=pro, epi=epi)
    zpr = zp.float().permute(0, 2, 3, 1).reshape(M, cout)
    alive = _alive(torch.addcmul(st[3 * cout:], zpr, st[2 * cout:3 * cout]))
    ref = (dz @ w.float().view(cout, cin).t()).to(torch.bfloat16).float() * alive
    got = dy.float().permute(0, 2, 3, 1).reshape(M, cout)
    _close(got, ref, 1.5e-2, "masked data gradient")
    assert float((got[~alive]).abs().max()) == 0.0
    sums = part.view(nb, 2, cout).double().sum(0)
    xhat = ((zpr - st[:cout]) * st[cout:2 * cout]).double()
    np.testing.assert_allclose(sums[0].cpu().numpy(), got.double().sum(0).cpu().numpy(), rtol=1e-3, atol=1e-3 * float(got.abs().sum(0).max()))
    np.testing.assert_allclose(sums[1].cpu().numpy(), (got.double() * xhat).sum(0).cpu().numpy(), rtol=1e-3,
                               atol=1e-3 * float((got.double() * xhat).abs().sum(0).max()))


def test_data_gradient_with_residual_epilogue():
    "conv1's data gradient of an identity bottleneck: + g_out * bits in the epilogue, one rounding."
    from pytorch_retinanet_amd import pwconv
    from pytorch_retinanet_amd._lib import RN_PW_EPI_RESID, RnPwEpilogue
    cm, cin = 64, 256
    dz1 = _rand((2, cm, 19, 23), 1.0, 1)
    w1t = _rand((cin, cm, 1, 1), 0.05, 2)
    gout = _rand((2, cin, 19, 23), 1.0, 3)
    M = 2 * 19 * 23
    gen = torch.Generator(device=DEV).manual_seed(4)
    mask = torch.rand(M, cin, device=DEV, generator=gen) > 0.5
    bits = (mask.view(M, cin // 8, 8).to(torch.int32) * (2 ** torch.arange(8, device=DEV, dtype=torch.int32))).sum(-1).to(torch.uint8).contiguous().view(-1)
    epi = RnPwEpilogue(RN_PW_EPI_RESID, 0, gout.data_ptr(), bits.data_ptr(), 0, 0, 0, 0, 0)
    dx = pwconv.pw_forward(dz1, w1t, epi=epi)
    ref = dz1.float().permute(0, 2, 3, 1).reshape(M, cm) @ w1t.float().view(cin, cm).t() + gout.float().permute(0, 2, 3, 1).reshape(M, cin) * mask
    _close(dx.float().permute(0, 2, 3, 1).reshape(M, cin), ref, 1e-2, "data gradient + residual")


@pytest.mark.parametrize("H,W,cm,cin", [(19, 23, 64, 256), (20, 24, 128, 256), (1, 7, 64, 64)])
def test_data_gradient_joined_by_the_stride2_downsample_gradient(H, W, cm, cin):
    """conv1's data gradient of a bottleneck with a 1x1 / stride-2 downsample branch: the branch's data gradient, a GEMM on the
    stride-2 grid, is added at the rows with even (y, x) in the epilogue (no mask) == conv1 dgrad + the scattered gradient."""
    from pytorch_retinanet_amd import pwconv
    from pytorch_retinanet_amd._lib import RN_PW_EPI_RESID, RnPwEpilogue
    N = 2
    dz1 = _rand((N, cm, H, W), 1.0, 1)
    w1t = _rand((cin, cm, 1, 1), 0.05, 2)
    Hc, Wc = (H + 1) // 2, (W + 1) // 2
    dxd = _rand((N, cin, Hc, Wc), 1.0, 3)
    M = N * H * W
    for sd, res in ((2, dxd), (1, _rand((N, cin, H, W), 1.0, 5))):
        epi = RnPwEpilogue(RN_PW_EPI_RESID, 0, res.data_ptr(), 0, 0, 0, 0, 0, 0, sd, H, W)
        dx = pwconv.pw_forward(dz1, w1t, epi=epi)
        full = torch.zeros((N, cin, H, W), device=DEV)
        if sd == 2:
            full[:, :, ::2, ::2] = res.float()
        else:
            full = res.float()
        ref = (dz1.float().permute(0, 2, 3, 1).reshape(M, cm) @ w1t.float().view(cin, cm).t()).view(N, H, W, cin).permute(0, 3, 1, 2) + full
        _close(dx.float(), ref, 1e-2, f"data gradient + stride-{sd} branch gradient")


@pytest.mark.parametrize("shape,cout", [((2, 512, 25, 34), 256), ((1, 256, 9, 7), 256), ((3, 512, 13, 21), 128)])
def test_1x1_with_bias_in_the_gemm_epilogue(shape, cout):
    """``pwconv.conv1x1`` of a 1x1 convolution WITH bias (the FPN's C3 lateral, retinanet/layers.py:31): forward as one GEMM with the bias in its
    epilogue (one rounding of conv + bias) instead of a convolution + an add pass; gradients as for every ``_Conv1x1`` -- against fp32."""
    from pytorch_retinanet_amd import pwconv
    torch.manual_seed(6)
    conv = torch.nn.Conv2d(shape[1], cout, 1).to(DEV)
    conv.weight.data = _cl(conv.weight.data.to(torch.bfloat16))
    torch.nn.init.normal_(conv.bias, std=0.5)
    x = _rand(shape, 1.0, 1).requires_grad_(True)
    g = _rand((shape[0], cout, shape[2], shape[3]), 1.0, 2)
    outs = {}
    for flag in (True, False):
        pwconv.BIAS_1X1_MM = flag
        try:
            x.grad = None
            conv.zero_grad()
            y = pwconv.conv1x1(conv, x)
            y.backward(g)
            outs[flag] = (y.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
        finally:
            pwconv.BIAS_1X1_MM = True
    ref = F.conv2d(x.detach().float(), conv.weight.detach().float(), conv.bias.detach().float())
    _close(outs[True][0], ref, 6e-3, "1x1 + bias forward")
    err = lambda t: float((t.float() - ref).norm())
    assert err(outs[True][0]) <= err(outs[False][0]) * 1.05, (err(outs[True][0]), err(outs[False][0]))      # one rounding instead of two
    for k, what in ((1, "dx"), (2, "dw"), (3, "dbias")):            # the gradients do not depend on who ran the forward
        _close(outs[True][k], outs[False][k], 1e-3, what)


@pytest.mark.parametrize("n_out,cin,k,stride", [(64, 256, 1, 1), (256, 64, 1, 1), (128, 512, 1, 1), (512, 128, 1, 1), (64, 64, 3, 1), (128, 128, 3, 2),
                                               (512, 256, 1, 2), (256, 1024, 1, 1), (2048, 512, 1, 1), (256, 256, 3, 2), (512, 512, 3, 2), (1024, 512, 1, 2), (256, 2048, 3, 2)])
def test_weight_gradient_plain(n_out, cin, k, stride):
    from pytorch_retinanet_amd import pwconv
    H, W = (21, 25) if n_out * cin < 512 * 512 else (9, 11)
    x = _rand((2, cin, H, W), 1.0, 1)
    w = _rand((n_out, cin, k, k), 0.05, 2)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    g = _rand((2, n_out, Ho, Wo), 1.0, 3)
    dw = pwconv.pw_wgrad(g, x, w, stride=stride)
    ref = torch.ops.aten.convolution_backward(g.float(), x.float(), w.float(), None, [stride, stride], [k // 2, k // 2], [1, 1], False, [0, 0], 1,
                                              [False, True, False])[1]
    assert dw.shape == w.shape and dw.stride() == w.stride()
    _close(dw, ref, 1e-2, "weight gradient")


@pytest.mark.parametrize("n_out,cin,k", [(192, 64, 1), (192, 192, 1), (384, 64, 1), (64, 320, 1), (320, 192, 1), (192, 64, 3), (64, 192, 3)])
def test_weight_gradient_widths_that_are_not_powers_of_two(n_out, cin, k):
    """The C ABI accepts every multiple of 64: the tile must DIVIDE its axis or part of dW is never written (N = 192
    picked a 128-wide tile and one tile).  The workspace is poisoned with NaN patterns first, so an unwritten partial shows."""
    from pytorch_retinanet_amd import pwconv
    H, W = 13, 17
    x = _rand((2, cin, H, W), 1.0, 1)
    w = _rand((n_out, cin, k, k), 0.05, 2)
    g = _rand((2, n_out, H, W), 1.0, 3)
    pwconv.pw_wgrad(g, x, w)                                   # (allocates the cached workspace)
    for ws in pwconv._WG_WS.values():
        ws.fill_(0xFF)                                         # f32 0xFFFFFFFF = NaN
    dw = pwconv.pw_wgrad(g, x, w)
    ref = torch.ops.aten.convolution_backward(g.float(), x.float(), w.float(), None, [1, 1], [k // 2, k // 2], [1, 1], False, [0, 0], 1,
                                              [False, True, False])[1]
    assert bool(torch.isfinite(dw.float()).all()), "part of dW was summed from unwritten workspace"
    _close(dw, ref, 1e-2, "weight gradient")
    # deferred form (private partial buffer + one reduce launch for several problems): same answer
    lst = []
    dw2 = pwconv.pw_wgrad(g, x, w, defer=lst)
    pwconv.pw_wgrad_flush(lst)
    assert torch.equal(dw2, dw)


@pytest.mark.parametrize("c4,cm", [(256, 64), (512, 128), (1024, 256)])
def test_weight_gradient_with_both_operand_transforms(c4, cm):
    "conv3's weight gradient of a bottleneck: dz3 from (g, z3, bits) and a2 = relu(bn2(z2)), both formed in the operand loads."
    from pytorch_retinanet_amd import pwconv
    H, W = 19, 23
    M = 2 * H * W
    gup, z3, z2 = _rand((2, c4, H, W), 1.0, 1), _rand((2, c4, H, W), 1.0, 2), _rand((2, cm, H, W), 1.0, 3)
    w3 = _rand((c4, cm, 1, 1), 0.05, 4)
    coef3 = _bn_bwd_coefs(c4, 5)
    gen = torch.Generator(device=DEV).manual_seed(6)
    mask = torch.rand(M, c4, device=DEV, generator=gen) > 0.4
    bits = (mask.view(M, c4 // 8, 8).to(torch.int32) * (2 ** torch.arange(8, device=DEV, dtype=torch.int32))).sum(-1).to(torch.uint8).contiguous().view(-1)
    fwd2 = torch.cat([torch.rand(cm, device=DEV, generator=gen) + 0.5, torch.randn(cm, device=DEV, generator=gen) * 0.3])
    dw = pwconv.pw_wgrad(gup, z2, w3, gpro=pwconv.bn_bwd(coef3, z3, 3, bits=bits), xpro=pwconv.affine_relu(fwd2))
    zr, gr = z3.float().permute(0, 2, 3, 1).reshape(M, c4), gup.float().permute(0, 2, 3, 1).reshape(M, c4)
    dz = (coef3[:c4] * (gr * mask) + (coef3[2 * c4:] * zr + coef3[c4:2 * c4])).to(torch.bfloat16).float()
    a2 = F.relu(torch.addcmul(fwd2[cm:], z2.float().permute(0, 2, 3, 1).reshape(M, cm), fwd2[:cm])).to(torch.bfloat16).float()
    ref = dz.t() @ a2
    _close(dw.float().view(c4, cm), ref, 1e-2, "weight gradient with transforms")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("c4,cm,H,W", [(256, 64, 19, 23), (512, 128, 19, 23), (256, 64, 260, 260), (512, 128, 150, 130), (256, 64, 1, 5), (512, 128, 1, 5)])
def test_conv3_backward_in_one_pass(c4, cm, H, W, dtype):
    """``rn_pw_conv3_backward``: the data gradient (bn3-backward prologue, ReLU-backward epilogue + bn2-backward sums) and the weight
    gradient of a bottleneck's conv3 from ONE pass over (g_out, z3, bits) -- against fp32 PyTorch on the same inputs, and against the
    two separate kernels it replaces (same bars as their own tests).  260 x 260 / 150 x 130: more row tiles than walkers, ragged end."""
    import ctypes as C
    from pytorch_retinanet_amd import pwconv
    from pytorch_retinanet_amd._lib import RN_BF16, RN_F16, lib
    M = 2 * H * W
    gup, z3, z2 = (_rand((2, c4, H, W), 1.0, 1).to(dtype), _rand((2, c4, H, W), 1.0, 2).to(dtype), _rand((2, cm, H, W), 1.0, 3).to(dtype))
    w3 = _rand((c4, cm, 1, 1), 0.05, 4).to(dtype)
    w3t = _cl(w3.view(c4, cm).t().contiguous().view(cm, c4, 1, 1))
    coef3 = _bn_bwd_coefs(c4, 5)
    gen = torch.Generator(device=DEV).manual_seed(6)
    mask = torch.rand(M, c4, device=DEV, generator=gen) > 0.4
    bits = (mask.view(M, c4 // 8, 8).to(torch.int32) * (2 ** torch.arange(8, device=DEV, dtype=torch.int32))).sum(-1).to(torch.uint8).contiguous().view(-1)
    st2 = torch.cat([torch.randn(cm, device=DEV, generator=gen) * 0.2, torch.rand(cm, device=DEV, generator=gen) + 0.5,
                     torch.rand(cm, device=DEV, generator=gen) + 0.5, torch.randn(cm, device=DEV, generator=gen) * 0.3])    # mean | invstd | a | b
    nb = lib.rn_pw_conv3_backward_walkers(M, cm, c4)
    assert nb > 0 and nb <= 512
    part = torch.full((nb * 2 * cm,), float("nan"), dtype=torch.float32, device=DEV)
    dy2 = torch.full_like(z2, float("nan"))
    ws = torch.empty((lib.rn_pw_conv3_backward_workspace_bytes(M, cm, c4),), dtype=torch.uint8, device=DEV)
    S = C.c_int(0)
    p3, p2 = coef3.data_ptr(), st2.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.rn_pw_conv3_backward(M, cm, c4, RN_F16 if dtype == torch.float16 else RN_BF16, gup.data_ptr(), z3.data_ptr(), bits.data_ptr(), p3,
                                  p3 + 4 * c4, p3 + 8 * c4, w3t.data_ptr(), z2.data_ptr(), p2 + 8 * cm, p2 + 12 * cm, p2, p2 + 4 * cm,
                                  dy2.data_ptr(), part.data_ptr(), ws.data_ptr(), ws.numel(), C.byref(S), st)
    assert rc == 0 and S.value == nb
    dw = torch.empty_like(w3)
    pend = [(ws, int(S.value), dw.numel(), dw)]
    pwconv.pw_wgrad_flush(pend)
    torch.cuda.synchronize()
    # fp32 references
    zr, gr = z3.float().permute(0, 2, 3, 1).reshape(M, c4), gup.float().permute(0, 2, 3, 1).reshape(M, c4)
    dz = (coef3[:c4] * (gr * mask) + (coef3[2 * c4:] * zr + coef3[c4:2 * c4])).to(dtype).float()
    z2r = z2.float().permute(0, 2, 3, 1).reshape(M, cm)
    pre = torch.addcmul(st2[3 * cm:], z2r, st2[2 * cm:3 * cm])
    alive = pre.to(dtype).float() > 0
    a2 = F.relu(pre).to(dtype).float()
    ref_dy = (dz @ w3.float().view(c4, cm)).to(dtype).float() * alive
    got = dy2.float().permute(0, 2, 3, 1).reshape(M, cm)
    _close(got, ref_dy, 1.5e-2, "masked data gradient")
    assert float((got[~alive]).abs().max()) == 0.0
    sums = part.view(nb, 2, cm).double().sum(0)
    xhat = ((z2r - st2[:cm]) * st2[cm:2 * cm]).double()
    np.testing.assert_allclose(sums[0].cpu().numpy(), got.double().sum(0).cpu().numpy(), rtol=1e-3, atol=1e-3 * float(got.abs().sum(0).max()))
    np.testing.assert_allclose(sums[1].cpu().numpy(), (got.double() * xhat).sum(0).cpu().numpy(), rtol=1e-3,
                               atol=1e-3 * float((got.double() * xhat).abs().sum(0).max()))
    _close(dw.float().view(c4, cm), dz.t() @ a2, 1e-2, "weight gradient")
    # the two kernels it replaces: same data gradient bit for bit (same products in the same order), same weight gradient to f32 summation order
    from pytorch_retinanet_amd._lib import RN_PW_EPI_RELU_BWD, RnPwEpilogue
    nb1 = lib.rn_pw_walkers(M)
    part1 = torch.empty((nb1 * 2 * cm,), dtype=torch.float32, device=DEV)
    epi = RnPwEpilogue(RN_PW_EPI_RELU_BWD, part1.data_ptr(), 0, 0, z2.data_ptr(), p2 + 8 * cm, p2 + 12 * cm, p2, p2 + 4 * cm)
    pro = pwconv.bn_bwd(coef3, z3, 3, bits=bits)
    dy_two = pwconv.pw_forward(gup, w3t, pro=pro, epi=epi)
    dw_two = pwconv.pw_wgrad(gup, z2, w3, gpro=pro, xpro=pwconv.affine_relu(st2[2 * cm:]))
    if dtype == torch.bfloat16 and cm == 64:
        assert torch.equal(dy_two, dy2)
    elif dtype == torch.bfloat16:   # two column halves: the first sums its K-chunks in the separate kernel's order, the second half a turn ahead
        assert torch.equal(dy_two[:, :64], dy2[:, :64])
        _close(dy2.float(), dy_two.float(), 1e-2, "data gradient vs the separate kernel")
    else:       # fp16: the compiler may round fma + conversion once (v_fma_mixlo_f16) in one kernel and twice in the other: rare 1-ulp differences
        _close(dy2.float(), dy_two.float(), 1e-3, "data gradient vs the separate kernel")
    _close(dw.float(), dw_two.float(), 4e-3, "weight gradient vs the separate kernel")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("ra", [False, True])
@pytest.mark.parametrize("c4,cn,H,W", [(256, 64, 19, 23), (256, 128, 19, 23), (512, 128, 19, 23), (256, 64, 260, 260), (512, 128, 1, 5)])
def test_block_output_and_next_conv1_in_one_pass(c4, cn, H, W, ra, dtype):
    """``rn_pw_block_out_conv1`` against the launches it replaces: ``rn_bn_apply`` / ``rn_bn_apply_res_affine`` (block output + ReLU bits) and
    ``rn_pw_conv_forward(.., RN_PW_EPI_STATS)`` on that output (the next block's conv1 + bn1 statistics partials) -- y, bits and z1 bit for
    bit, the statistics to f32 summation order; plus z1 against fp32 PyTorch."""
    from pytorch_retinanet_amd import pwconv
    from pytorch_retinanet_amd._lib import RN_BF16, RN_F16, lib
    M = 2 * H * W
    z3, res = _rand((2, c4, H, W), 1.0, 1).to(dtype), _rand((2, c4, H, W), 1.0, 2).to(dtype)
    w1 = _rand((cn, c4, 1, 1), 0.05, 3).to(dtype)
    gen = torch.Generator(device=DEV).manual_seed(4)
    def stats():    # mean | invstd | a | b
        return torch.cat([torch.randn(c4, device=DEV, generator=gen) * 0.2, torch.rand(c4, device=DEV, generator=gen) + 0.5,
                          torch.rand(c4, device=DEV, generator=gen) + 0.5, torch.randn(c4, device=DEV, generator=gen) * 0.3])
    st3, std = stats(), stats()
    nb = lib.rn_pw_block_out_conv1_walkers(M, c4, cn)
    assert nb > 0
    y = torch.full_like(z3, float("nan"))
    bits = torch.zeros((M * c4 // 8,), dtype=torch.uint8, device=DEV)
    z1 = torch.full((2, cn, H, W), float("nan"), dtype=dtype, device=DEV).contiguous(memory_format=torch.channels_last)
    part = torch.full((nb * 2 * cn,), float("nan"), dtype=torch.float32, device=DEV)
    p3, pd = st3.data_ptr(), std.data_ptr()
    rc = lib.rn_pw_block_out_conv1(M, c4, cn, RN_F16 if dtype == torch.float16 else RN_BF16, z3.data_ptr(), res.data_ptr(),
                                   pd + 8 * c4 if ra else 0, pd + 12 * c4 if ra else 0, p3 + 8 * c4, p3 + 12 * c4, w1.data_ptr(), y.data_ptr(),
                                   bits.data_ptr(), z1.data_ptr(), part.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    y_ref, bits_ref = pwconv.bn_apply(z3, st3, relu=True, residual=res, want_bits=True, res_stats=std if ra else None)
    e1, p1, nb1 = pwconv.stats_epilogue(M, cn, torch.device(DEV))
    z1_ref = pwconv.pw_forward(y_ref, w1, epi=e1)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref) and torch.equal(bits, bits_ref)
    assert torch.equal(z1, z1_ref)
    a, b = part.view(nb, 2, cn).double().sum(0), p1.view(nb1, 2, cn).double().sum(0)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(b.abs().max()))
    ref = (y_ref.float().permute(0, 2, 3, 1).reshape(M, c4) @ w1.float().view(cn, c4).t())
    _close(z1.float().permute(0, 2, 3, 1).reshape(M, cn), ref, 1e-2, "next conv1 output")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cm,c4,H,W,rs", [(64, 256, 19, 23, 1), (128, 512, 19, 23, 1), (64, 256, 20, 24, 2), (128, 256, 19, 23, 2), (64, 256, 260, 260, 1),
                                          (128, 512, 1, 5, 1)])
def test_conv1_data_gradient_with_the_previous_blocks_bn3_sums(cm, c4, H, W, rs, dtype):
    """``rn_pw_dgrad_resid_sums``: dx == ``rn_pw_conv_forward(.., RN_PW_EPI_RESID)`` bit for bit (identity branch with ReLU bits, or the
    stride-2 downsample join), and the previous block's two bn3-backward sums over that dx against float64 PyTorch."""
    from pytorch_retinanet_amd import pwconv
    from pytorch_retinanet_amd._lib import RN_BF16, RN_F16, RN_PW_EPI_RESID, RnPwEpilogue, lib
    M = 2 * H * W
    dz1 = _rand((2, cm, H, W), 1.0, 1).to(dtype)
    w1t = _rand((c4, cm, 1, 1), 0.05, 2).to(dtype)
    gen = torch.Generator(device=DEV).manual_seed(3)
    def make_bits():
        mask = torch.rand(M, c4, device=DEV, generator=gen) > 0.4
        return mask, (mask.view(M, c4 // 8, 8).to(torch.int32) * (2 ** torch.arange(8, device=DEV, dtype=torch.int32))).sum(-1).to(torch.uint8).contiguous().view(-1)
    if rs == 1:
        resid = _rand((2, c4, H, W), 1.0, 4).to(dtype)
        _, rbits = make_bits()
        epi = RnPwEpilogue(RN_PW_EPI_RESID, 0, resid.data_ptr(), rbits.data_ptr(), 0, 0, 0, 0, 0)
    else:
        resid = _rand((2, c4, (H + 1) // 2, (W + 1) // 2), 1.0, 4).to(dtype)
        rbits = None
        epi = RnPwEpilogue(RN_PW_EPI_RESID, 0, resid.data_ptr(), 0, 0, 0, 0, 0, 0, 2, H, W)
    pz3 = _rand((2, c4, H, W), 1.0, 5).to(dtype)
    pmask, pbits = make_bits()
    pst = torch.cat([torch.randn(c4, device=DEV, generator=gen) * 0.2, torch.rand(c4, device=DEV, generator=gen) + 0.5])     # mean | invstd
    nb = lib.rn_pw_dgrad_resid_sums_walkers(M, cm, c4)
    assert nb > 0
    dx = torch.full((2, c4, H, W), float("nan"), dtype=dtype, device=DEV).contiguous(memory_format=torch.channels_last)
    part = torch.full((nb * 2 * c4,), float("nan"), dtype=torch.float32, device=DEV)
    rc = lib.rn_pw_dgrad_resid_sums(M, cm, c4, RN_F16 if dtype == torch.float16 else RN_BF16, dz1.data_ptr(), w1t.data_ptr(), resid.data_ptr(),
                                    rbits.data_ptr() if rbits is not None else 0, rs, H, W, pz3.data_ptr(), pbits.data_ptr(), pst.data_ptr(),
                                    pst.data_ptr() + 4 * c4, dx.data_ptr(), part.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    dx_ref = pwconv.pw_forward(dz1, w1t, epi=epi)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref)
    g = dx.double().permute(0, 2, 3, 1).reshape(M, c4) * pmask
    xhat = (pz3.double().permute(0, 2, 3, 1).reshape(M, c4) - pst[:c4].double()) * pst[c4:].double()
    sums = part.view(nb, 2, c4).double().sum(0)
    np.testing.assert_allclose(sums[0].cpu().numpy(), g.sum(0).cpu().numpy(), rtol=1e-4, atol=1e-4 * float(g.abs().sum(0).max()))
    np.testing.assert_allclose(sums[1].cpu().numpy(), (g * xhat).sum(0).cpu().numpy(), rtol=1e-4, atol=1e-4 * float((g * xhat).abs().sum(0).max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cm,c4,H,W", [(64, 256, 19, 23), (128, 512, 19, 23), (64, 256, 260, 260), (128, 512, 1, 5), (64, 128, 9, 7)])
def test_conv3_forward_on_the_row_tile_walker(cm, c4, H, W, dtype):
    "``rn_pw_conv3_forward`` == ``rn_pw_conv_forward(z2, w3, RN_PW_PRO_AFFINE_RELU, RN_PW_EPI_STATS)``: z3 bit for bit, statistics to summation order."
    from pytorch_retinanet_amd import pwconv
    from pytorch_retinanet_amd._lib import RN_BF16, RN_F16, lib
    M = 2 * H * W
    z2 = _rand((2, cm, H, W), 1.0, 1).to(dtype)
    w3 = _rand((c4, cm, 1, 1), 0.05, 2).to(dtype)
    gen = torch.Generator(device=DEV).manual_seed(3)
    coef = torch.cat([torch.rand(cm, device=DEV, generator=gen) + 0.5, torch.randn(cm, device=DEV, generator=gen) * 0.3])
    nb = lib.rn_pw_conv3_forward_walkers(M, cm, c4)
    assert nb > 0
    z3 = torch.full((2, c4, H, W), float("nan"), dtype=dtype, device=DEV).contiguous(memory_format=torch.channels_last)
    part = torch.full((nb * 2 * c4,), float("nan"), dtype=torch.float32, device=DEV)
    assert lib.rn_pw_conv3_forward(M, cm, c4, RN_F16 if dtype == torch.float16 else RN_BF16, z2.data_ptr(), coef.data_ptr(), w3.data_ptr(), z3.data_ptr(),
                                   part.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    e, p1, nb1 = pwconv.stats_epilogue(M, c4, torch.device(DEV))
    ref = pwconv.pw_forward(z2, w3, pro=pwconv.affine_relu(coef), epi=e)
    torch.cuda.synchronize()
    assert torch.equal(z3, ref)
    a, b = part.view(nb, 2, c4).double().sum(0), p1.view(nb1, 2, c4).double().sum(0)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(b.abs().max()))


def test_chained_blocks_equal_the_unchained_ones():
    """Two consecutive fused bottlenecks (downsample block -> identity block -> identity block) with and without ``pwconv.FUSE_CHAIN``:
    the same tensors reach the same kernels (only bn1's statistics are summed in another order), so outputs and gradients agree far
    inside the fused-vs-fp32 bars; the hand-over is consumed (the chained run launches no conv1 GEMM for the later blocks)."""
    from pytorch_retinanet_amd import backbone as bb
    from pytorch_retinanet_amd import ops, pwconv
    torch.manual_seed(5)
    ds = torch.nn.Sequential(bb._conv1x1(64, 256, 1), bb.FusedBatchNorm2d(256))
    blocks = [bb.Bottleneck(64, 64, 1, ds), bb.Bottleneck(256, 64), bb.Bottleneck(256, 64)]
    pwconv.link_blocks(blocks)
    seq = torch.nn.Sequential(*blocks).to(DEV).to(memory_format=torch.channels_last).train()
    with torch.no_grad():
        for m in seq.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    for p in seq.parameters():
        if p.dim() == 4:
            p.data = p.data.to(torch.bfloat16)
    x0, g = _rand((2, 64, 40, 44), 1.0, 1), _rand((2, 256, 40, 44), 1.0, 2)
    init = {n: b.clone() for n, b in seq.named_buffers()}
    res = {}
    try:
        for chain in (False, True, "forward only"):
            pwconv.FUSE_CHAIN = bool(chain)
            pwconv.FUSE_BWD_CHAIN = chain is True
            pwconv.PW_TIMES.clear() if hasattr(pwconv, "PW_TIMES") else None
            with torch.no_grad():
                for n, b in seq.named_buffers():
                    b.copy_(init[n])
            x = x0.clone().requires_grad_(True)
            seq.zero_grad()
            ops.TIMINGS.clear() if hasattr(ops, "TIMINGS") else None
            mid = []
            hooks = [b.register_forward_hook(lambda m, i, o: mid.append(hasattr(o, "_rn_chain"))) for b in blocks]
            y = seq(x)
            for h in hooks:
                h.remove()
            assert mid == [bool(chain), bool(chain), False]     # handed over between the blocks; the last one has no successor
            y.backward(g)
            torch.cuda.synchronize()
            assert not pwconv._BWD_CHAIN                 # every backward hand-over was consumed
            res[chain] = [y.detach().float(), x.grad.float()] + [p.grad.float().clone() for p in seq.parameters()] + [b.float().clone() for b in seq.buffers()]
    finally:
        pwconv.FUSE_CHAIN = pwconv.FUSE_BWD_CHAIN = True
    for i, (a, b) in enumerate(zip(res["forward only"], res[False])):
        assert _l2(a, b) <= 1e-2, f"forward chain, tensor {i}: relative L2 distance {_l2(a, b)}"
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        assert _l2(a, b) <= 1e-2, f"tensor {i}: relative L2 distance {_l2(a, b)}"     # (a1 rounds differently where bn1's statistics differ in their last bits)


def test_backward_hand_over_is_not_taken_when_the_block_output_has_a_second_consumer():
    """``pwconv.FUSE_BWD_CHAIN`` forms a block's bn3-backward sums from the gradient its successor produces.  When the block's output feeds
    something else as well, the gradient that reaches it is the SUM of two: the hand-over must not be used (the entry holds the
    successor's tensor, so the engine cannot add into it in place; the sum arrives as another tensor).  Checked against the unchained run."""
    from pytorch_retinanet_amd import backbone as bb
    from pytorch_retinanet_amd import pwconv
    torch.manual_seed(6)
    blocks = [bb.Bottleneck(256, 64), bb.Bottleneck(256, 64)]
    pwconv.link_blocks(blocks)
    for b in blocks:
        b.to(DEV).to(memory_format=torch.channels_last).train()
        for p in b.parameters():
            if p.dim() == 4:
                p.data = p.data.to(torch.bfloat16)
    x0, g = _rand((2, 256, 24, 28), 1.0, 1), _rand((2, 256, 24, 28), 1.0, 2)
    init = {i: {n: t.clone() for n, t in b.named_buffers()} for i, b in enumerate(blocks)}
    res = {}
    try:
        for chain in (False, True):
            pwconv.FUSE_CHAIN = pwconv.FUSE_BWD_CHAIN = chain
            for i, b in enumerate(blocks):
                with torch.no_grad():
                    for n, t in b.named_buffers():
                        t.copy_(init[i][n])
                b.zero_grad()
            x = x0.clone().requires_grad_(True)
            mid = blocks[0](x)
            out = blocks[1](mid)
            (out.float() * g.float()).sum().add((mid.float() ** 2).sum() * 0.5).backward()      # second consumer of `mid`: d/dmid = mid
            torch.cuda.synchronize()
            assert not pwconv._BWD_CHAIN or chain            # (an unconsumed entry may remain: it is dropped by the next trunk forward)
            pwconv._BWD_CHAIN.clear()
            res[chain] = [x.grad.float()] + [p.grad.float().clone() for b in blocks for p in b.parameters()]
    finally:
        pwconv.FUSE_CHAIN = pwconv.FUSE_BWD_CHAIN = True
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        assert _l2(a, b) <= 1e-2, f"tensor {i}: relative L2 distance {_l2(a, b)}"


def _truth_block(blk, x0, g):
    "The block in plain fp32 PyTorch ops with autograd (bf16 weights up-cast, batch statistics): outputs and every gradient."
    def bn(z, m):
        return F.batch_norm(z, None, None, m.weight, m.bias, True, 0.1, m.eps)
    params = {n: p.detach().float().requires_grad_(True) for n, p in blk.named_parameters()}
    x = x0.float().requires_grad_(True)

    class M:        # a BatchNorm's affine parameters from the fp32 copies
        def __init__(self, prefix, eps): self.weight, self.bias, self.eps = params[prefix + ".weight"], params[prefix + ".bias"], eps
    s = blk.conv2.stride
    a1 = F.relu(bn(F.conv2d(x, params["conv1.weight"]), M("bn1", blk.bn1.eps)))
    a2 = F.relu(bn(F.conv2d(a1, params["conv2.weight"], None, s, 1), M("bn2", blk.bn2.eps)))
    y3 = bn(F.conv2d(a2, params["conv3.weight"]), M("bn3", blk.bn3.eps))
    idt = x
    if blk.downsample is not None:
        idt = bn(F.conv2d(x, params["downsample.0.weight"], None, blk.downsample[0].stride), M("downsample.1", blk.downsample[1].eps))
    out = F.relu(y3 + idt)
    out.backward(g.float())
    return out.detach(), x.grad, {n: p.grad for n, p in params.items()}


def _l2(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


@pytest.mark.parametrize("inplanes,planes,stride,down", [(256, 64, 1, False), (64, 64, 1, True), (256, 128, 2, True), (512, 128, 1, False)])
def test_fused_bottleneck_is_as_close_to_fp32_as_the_layer_by_layer_block(inplanes, planes, stride, down):
    """``pwconv._BottleneckFn`` against the block in fp32 PyTorch ops (the reference's Bottleneck, retinanet/backbone.py:105-136,
    with autograd).  Two bf16 pipelines that sum in different orders cannot be compared element by element -- a pre-activation
    at the ReLU boundary flips its mask and a whole upstream gradient element with it -- so both the fused block and the
    layer-by-layer block (MIOpen convs + fused BN kernels, round 2's path) are measured by their relative L2 distance to the
    fp32 result: the fused block must be within 1.25 x the layer-by-layer block's distance (+ 1e-3) for the output, the input
    gradient and every parameter gradient, and the running statistics of the two must agree to 1e-3 (fp32-accumulated)."""
    from pytorch_retinanet_amd import backbone as bb
    from pytorch_retinanet_amd import pwconv
    torch.manual_seed(7)
    ds = None
    if down:
        ds = torch.nn.Sequential(bb._conv1x1(inplanes, planes * 4, stride), bb.FusedBatchNorm2d(planes * 4))
    blk = bb.Bottleneck(inplanes, planes, stride, ds).to(DEV).to(memory_format=torch.channels_last).train()
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    for p in blk.parameters():
        if p.dim() == 4:
            p.data = p.data.to(torch.bfloat16)
    x0 = _rand((2, inplanes, 24, 28), 1.0, 1)
    g = _rand((2, planes * 4, 24 // stride, 28 // stride), 1.0, 2)
    truth = _truth_block(blk, x0, g)
    init = {n: b.clone() for n, b in blk.named_buffers()}
    res = {}
    for fused in (False, True):
        pwconv.FUSED_BOTTLENECK = fused
        with torch.no_grad():
            for n, b in blk.named_buffers():
                b.copy_(init[n])
        x = x0.clone().requires_grad_(True)
        blk.zero_grad()
        assert pwconv.bottleneck_fusable(blk, x) == fused
        y = blk(x)
        y.backward(g)
        torch.cuda.synchronize()
        res[fused] = (y.detach().float(), x.grad.float(), {n: p.grad.float().clone() for n, p in blk.named_parameters()},
                      {n: b.float().clone() for n, b in blk.named_buffers()})
    pwconv.FUSED_BOTTLENECK = True
    checks = [("output", res[True][0], res[False][0], truth[0]), ("input gradient", res[True][1], res[False][1], truth[1])]
    checks += [(f"gradient of {n}", res[True][2][n], res[False][2][n], truth[2][n]) for n in truth[2]]
    for what, fused_v, plain_v, ref in checks:
        ef, ep = _l2(fused_v, ref), _l2(plain_v, ref)
        assert ef <= 1.25 * ep + 1e-3 and ef <= 0.15, f"{what}: fused {ef} vs layer-by-layer {ep} from the fp32 block"
    for n, a in res[False][3].items():
        if "num_batches" in n:
            assert int(res[True][3][n]) == int(a) == 1
        else:
            torch.testing.assert_close(res[True][3][n], a, rtol=1e-3, atol=1e-3 * float(a.abs().max()) + 1e-5, msg=n)


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 37, 53), (3, 32, 1000), (1, 7, 7)])
def test_stem_conv_forward_and_statistics(B, H, W):
    """rn_stem_conv_forward (7x7 / stride 2 / pad 3, 3 -> 64, the zero-bordered NHWC4 copy + MFMA kernel) against torch's fp32
    convolution of the same bf16 values; the statistics partials against the sums of the stored output."""
    import ctypes as C
    from pytorch_retinanet_amd._lib import RN_BF16, check, lib
    x = _rand((B, 3, H, W), 1.0, 1)
    w = _rand((64, 3, 7, 7), 0.1, 2)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    xp = torch.empty((lib.rn_stem_padded_bytes(B, H, W),), dtype=torch.uint8, device=DEV)
    wk = torch.empty((64 * 7 * 32,), dtype=torch.bfloat16, device=DEV)
    z = torch.empty((B, 64, Ho, Wo), dtype=torch.bfloat16, device=DEV, memory_format=torch.channels_last)
    nb = lib.rn_stem_partial_rows(B, H, W)
    part = torch.full((nb * 2 * 64,), float("nan"), device=DEV)
    check(lib.rn_stem_conv_forward(x.data_ptr(), w.data_ptr(), xp.data_ptr(), wk.data_ptr(), z.data_ptr(), part.data_ptr(), RN_BF16, B, H, W,
                                   torch.cuda.current_stream().cuda_stream), "rn_stem_conv_forward")
    ref = torch.nn.functional.conv2d(x.float(), w.float(), None, 2, 3)
    assert tuple(z.shape) == tuple(ref.shape)
    _close(z.float(), ref, 1e-2, "stem conv")
    assert float((z.float() - ref).abs().max()) < 0.02 * float(ref.abs().max())
    sums = part.view(nb, 2, 64).double().sum(0)
    zz = z.float().permute(0, 2, 3, 1).reshape(-1, 64).double()
    np.testing.assert_allclose(sums[0].cpu().numpy(), zz.sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3 * float(zz.abs().sum(0).max()))
    np.testing.assert_allclose(sums[1].cpu().numpy(), (zz * zz).sum(0).cpu().numpy(), rtol=1e-4)


def test_stem_function_matches_the_layer_by_layer_path():
    "backbone stem in training mode: _StemFn (conv + statistics epilogue + apply) vs conv_bn on MIOpen + the BatchNorm kernels."
    from pytorch_retinanet_amd import backbone, pwconv
    torch.manual_seed(2)
    outs = {}
    for flag in (True, False):
        pwconv.FUSED_STEM = flag
        try:
            torch.manual_seed(5)
            net = backbone.resnet18(pretrained=False).to(DEV).to(memory_format=torch.channels_last)
            net.conv1.weight.data = net.conv1.weight.data.to(torch.bfloat16)
            net.train()
            x = _rand((2, 3, 96, 128), 1.0, 7)
            assert pwconv.stem_fusable(net.conv1, net.bn1, x) == flag
            y = pwconv.stem(net.conv1, net.bn1, x, pool=net.maxpool) if flag else net.maxpool(backbone.conv_bn(net.conv1, net.bn1, x, relu=True))
            (y.float() ** 2).mean().backward()
            outs[flag] = (y.detach().float(), net.conv1.weight.grad.float().clone(), net.bn1.weight.grad.clone(), net.bn1.bias.grad.clone(),
                          net.bn1.running_mean.clone(), net.bn1.running_var.clone())
        finally:
            pwconv.FUSED_STEM = True
    names = ("output", "conv weight gradient", "bn weight gradient", "bn bias gradient", "running mean", "running var")
    for n, a, b in zip(names, outs[True], outs[False]):
        if "gradient" in n:
            # two bf16 pipelines: a conv output that differs by one ulp moves a ReLU decision or a pooling arg-max, and the gradient
            # of that window goes elsewhere -- compare in the l2 norm, as the block tests above do
            rel = float((a - b).norm() / b.norm().clamp_min(1e-12))
            assert rel < 5e-2, (n, rel)
        else:
            _close(a, b, 2e-2, n)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W,pool", [(2, 64, 96, True), (1, 37, 53, True), (2, 30, 600, False), (1, 7, 7, True)])
def test_stem_weight_gradient_with_bn_backward_in_its_operand_load(B, H, W, pool, dtype):
    "``pwconv.STEM_WGRAD_BN``: the stem's backward with the BatchNorm apply step inside ``rn_stem_conv_wgrad_bn`` == the three-launch form, bit for bit."
    from pytorch_retinanet_amd import backbone as bb
    from pytorch_retinanet_amd import pwconv
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(DEV).to(memory_format=torch.channels_last)
    conv.weight.data = conv.weight.data.to(dtype)
    bn = bb.FusedBatchNorm2d(64).to(DEV).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    mp = bb.FusedMaxPool2d(kernel_size=3, stride=2, padding=1) if pool else None
    x = _rand((B, 3, H, W), 1.0, 1).to(dtype)
    init = {n: b.clone() for n, b in bn.named_buffers()}
    res = {}
    try:
        for fused in (False, True, "pool sums"):
            pwconv.STEM_WGRAD_BN = bool(fused)
            pwconv.STEM_POOL_BN_SUMS = fused == "pool sums"
            with torch.no_grad():
                for n, b in bn.named_buffers():
                    b.copy_(init[n])
            conv.zero_grad(); bn.zero_grad()
            assert pwconv.stem_fusable(conv, bn, x)
            y = pwconv.stem(conv, bn, x, pool=mp)
            g = _rand(tuple(y.shape), 1.0, 2).to(dtype)
            y.backward(g)
            torch.cuda.synchronize()
            res[fused] = [conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()]
    finally:
        pwconv.STEM_WGRAD_BN = pwconv.STEM_POOL_BN_SUMS = True
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
    assert float(res[True][0].float().abs().max()) > 0
    # bn1's sums taken inside the pooling backward: another summation order -> coefficients equal to f32 rounding, gradients to the last bits
    for a, b in zip(res["pool sums"], res[False]):
        torch.testing.assert_close(a.float(), b.float(), rtol=2e-2, atol=2e-3 * float(b.float().abs().max()) + 1e-6)
    torch.testing.assert_close(res["pool sums"][1], res[False][1], rtol=1e-4, atol=1e-4 * float(res[False][1].abs().max()))
    torch.testing.assert_close(res["pool sums"][2], res[False][2], rtol=1e-4, atol=1e-4 * float(res[False][2].abs().max()))


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 37, 53), (2, 30, 600), (1, 7, 7)])
def test_stem_conv_weight_gradient(B, H, W):
    "rn_stem_conv_wgrad (transposing LDS reads over the raw padded strips) against torch's fp32 weight gradient."
    from pytorch_retinanet_amd._lib import RN_BF16, check, lib
    x = _rand((B, 3, H, W), 1.0, 1)
    w = _rand((64, 3, 7, 7), 0.1, 2)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    g = _rand((B, 64, Ho, Wo), 1.0, 3)
    st = torch.cuda.current_stream().cuda_stream
    xp = torch.empty((lib.rn_stem_padded_bytes(B, H, W),), dtype=torch.uint8, device=DEV)
    xp.view(torch.int16).fill_(0x7fc0)                                           # bf16 NaNs everywhere: the pad kernel must overwrite all of it
    wk = torch.empty((64 * 7 * 32,), dtype=torch.bfloat16, device=DEV)
    z = torch.empty((B, 64, Ho, Wo), dtype=torch.bfloat16, device=DEV, memory_format=torch.channels_last)
    check(lib.rn_stem_conv_forward(x.data_ptr(), w.data_ptr(), xp.data_ptr(), wk.data_ptr(), z.data_ptr(), 0, RN_BF16, B, H, W, st), "fwd")
    need = lib.rn_stem_wgrad_workspace_bytes(B, H, W)
    wsb = torch.empty((need,), dtype=torch.uint8, device=DEV)
    dw = torch.full_like(w, float("nan"))
    check(lib.rn_stem_conv_wgrad(g.data_ptr(), xp.data_ptr(), dw.data_ptr(), RN_BF16, B, H, W, wsb.data_ptr(), need, st), "rn_stem_conv_wgrad")
    # fp32 reference on the CPU (small problems; MIOpen's fp32 weight-gradient search for these odd shapes aborted the process once
    # in a while when it ran late in a long test session)
    ref = torch.ops.aten.convolution_backward(g.float().cpu(), x.float().cpu(), w.float().cpu(), None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1,
                                              [False, True, False])[1].to(DEV)
    assert dw.shape == ref.shape and dw.stride() == w.stride()
    _close(dw, ref, 1e-2, "stem weight gradient")


def test_bn_relu_inside_maxpool_equals_the_two_pass_form():
    "rn_bn_relu_maxpool3x3s2_forward == rn_bn_apply (ReLU) followed by rn_maxpool3x3s2_forward: values and arg-max codes, bit for bit."
    from pytorch_retinanet_amd._lib import RN_BF16, check, lib
    N, Cc, H, W = 2, 64, 37, 53
    z = _rand((N, Cc, H, W), 2.0, 1)
    gen = torch.Generator(device=DEV).manual_seed(3)
    coef = torch.cat([torch.rand(Cc, device=DEV, generator=gen) + 0.5, torch.randn(Cc, device=DEV, generator=gen)]).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    a = torch.empty_like(z)
    check(lib.rn_bn_apply(z.data_ptr(), 0, a.data_ptr(), RN_BF16, N * H * W, Cc, coef.data_ptr(), 1, 0, st), "rn_bn_apply")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    outs = []
    for fused in (False, True):
        y = torch.empty((N, Cc, Ho, Wo), dtype=torch.bfloat16, device=DEV, memory_format=torch.channels_last)
        arg = torch.empty(y.shape, dtype=torch.uint8, device=DEV, memory_format=torch.channels_last)
        if fused:
            check(lib.rn_bn_relu_maxpool3x3s2_forward(z.data_ptr(), coef.data_ptr(), y.data_ptr(), arg.data_ptr(), RN_BF16, N, H, W, Cc, st), "fused")
        else:
            check(lib.rn_maxpool3x3s2_forward(a.data_ptr(), y.data_ptr(), arg.data_ptr(), RN_BF16, N, H, W, Cc, st), "pool")
        outs.append((y, arg))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("shape,cout,stride,relu,with_res", [
    ((2, 64, 37, 45), 256, 1, True, True),        # conv3 of a layer1 block: + identity, ReLU
    ((2, 256, 37, 45), 64, 1, True, False),       # conv1: bias + ReLU
    ((2, 256, 38, 46), 512, 2, False, False),     # the downsample branch: stride 2, no ReLU
    ((1, 512, 9, 11), 2048, 1, True, True),       # layer4 conv3 (two column tiles of 128 per 256 ... sixteen)
    ((3, 128, 5, 7), 192, 1, False, True),        # 64-wide column tiles, residual without ReLU
    ((2, 128, 21, 25), 128, 1, True, False),      # 3x3 (conv2 of a layer2 block)
    ((2, 256, 22, 26), 256, 2, True, False),      # 3x3 / stride 2 (conv2 of the first block of layer3)
], ids=lambda v: str(v))
def test_forward_with_bias_residual_relu_epilogue(shape, cout, stride, relu, with_res):
    "Inference epilogue (folded BatchNorm + identity + ReLU, backbone.py:118-136 under eval()): act(conv + bias (+ residual)), one rounding."
    from pytorch_retinanet_amd import pwconv
    k = 3 if (shape[1] == cout and shape[2] in (21, 22)) else 1
    x = _rand(shape, 1.0, 1)
    w = _rand((cout, shape[1], k, k), 0.05 if k == 1 else 0.02, 2)
    gen = torch.Generator(device=DEV).manual_seed(3)
    bias = torch.randn(cout, device=DEV, generator=gen)
    ref = F.conv2d(x.float(), w.float(), bias, stride, k // 2)
    res = _rand(tuple(ref.shape), 1.0, 4) if with_res else None
    if res is not None:
        ref = ref + res.float()
    if relu:
        ref = F.relu(ref)
    y = pwconv.pw_forward(x, w, stride=stride, epi=pwconv.bias_act_epilogue(bias, relu, res))
    assert y.shape == ref.shape and y.dtype == torch.bfloat16
    _close(y, ref, 6e-3, "bias / residual / ReLU epilogue")
    if relu:
        assert float(y.float().min()) >= 0.0
    # the epilogue is exclusive with the training epilogues and prologues
    from pytorch_retinanet_amd._lib import RN_PW_EPI_BIAS, RN_PW_EPI_STATS, RnPwEpilogue, lib
    bad = RnPwEpilogue(RN_PW_EPI_BIAS | RN_PW_EPI_STATS, 0)
    bad.bias = bias.data_ptr()
    with pytest.raises(RuntimeError):
        pwconv.pw_forward(x, w, stride=stride, epi=bad)


def test_frozen_bn_bottleneck_runs_its_convs_with_fused_epilogues():
    "backbone.conv_bn under eval(): the GEMM-epilogue path == the convolution + epilogue-pass path == the fp32 block (backbone.py:118-136)."
    from pytorch_retinanet_amd import backbone, pwconv
    torch.manual_seed(5)
    for inpl, planes, stride, hw in ((256, 64, 1, (19, 27)), (256, 128, 2, (18, 26)), (1024, 256, 1, (9, 13)), (2048, 512, 1, (5, 7))):
        ds = None
        if stride != 1 or inpl != planes * 4:
            ds = torch.nn.Sequential(torch.nn.Conv2d(inpl, planes * 4, 1, stride, bias=False), backbone.FusedBatchNorm2d(planes * 4))
        blk = backbone.Bottleneck(inpl, planes, stride, ds).to(DEV).to(memory_format=torch.channels_last)
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0.0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0.0, 0.2)
        blk.eval()
        x = _rand((2, inpl, *hw), 1.0, 6)
        with torch.no_grad():
            from pytorch_retinanet_amd import biasact
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pwconv.EVAL_1X1_FUSED = biasact.DENSE_EVAL = True      # GEMM epilogues for the 1x1 convs, own kernels for conv2
                y1 = blk(x)
                pwconv.EVAL_1X1_FUSED = biasact.DENSE_EVAL = False     # library convolutions + one epilogue pass each
                try:
                    y0 = blk(x)
                finally:
                    pwconv.EVAL_1X1_FUSED = biasact.DENSE_EVAL = True
            backbone.FOLD_FROZEN_BN = False
            try:
                yf = blk(x.float())                                  # fp32, BatchNorm applied layer by layer
            finally:
                backbone.FOLD_FROZEN_BN = True
        assert y1.dtype == torch.bfloat16 and y1.shape == y0.shape == yf.shape
        e1 = float((y1.float() - yf).norm() / yf.norm())
        e0 = float((y0.float() - yf).norm() / yf.norm())
        assert e1 <= max(1.15 * e0, 6e-3), (inpl, planes, stride, e1, e0)   # one rounding less per convolution: not further from fp32 than the two-pass path


@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (1, 37, 53), (2, 131, 173)])
def test_stem_inference_on_the_mfma_kernel_equals_the_folded_library_path(B, H, W):
    "ResNet._stem_inference: conv7x7 with the folded BatchNorm on csrc/stem.hip + bias / ReLU inside the pooling == conv + epilogue pass + pooling."
    from pytorch_retinanet_amd import backbone, pwconv
    torch.manual_seed(3)
    net = backbone.resnet18(pretrained=False).to(DEV).to(memory_format=torch.channels_last).eval()
    with torch.no_grad():
        net.bn1.running_mean.normal_(0.0, 0.3); net.bn1.running_var.uniform_(0.5, 1.5); net.bn1.weight.uniform_(0.5, 1.5); net.bn1.bias.normal_(0.0, 0.3)
        x = _rand((B, 3, H, W), 1.0, 9)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pwconv.STEM_EVAL = True
            y1 = net._stem_inference(x)
            pwconv.STEM_EVAL = False
            try:
                y0 = net._stem_inference(x)
            finally:
                pwconv.STEM_EVAL = True
        ref = F.max_pool2d(F.relu(F.batch_norm(F.conv2d(x.float(), net.conv1.weight.float(), None, 2, 3), net.bn1.running_mean, net.bn1.running_var,
                                               net.bn1.weight, net.bn1.bias, False, 0.0, net.bn1.eps)), 3, 2, 1)
    assert y1.shape == y0.shape == ref.shape and y1.dtype == torch.bfloat16
    e1, e0 = float((y1.float() - ref).norm() / ref.norm()), float((y0.float() - ref).norm() / ref.norm())
    assert e1 <= max(1.2 * e0, 6e-3), (e1, e0)
