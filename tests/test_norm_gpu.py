"""GPU: fused BatchNorm2d (+residual)(+ReLU) kernels (csrc/norm.hip, through the C ABI) against PyTorch's own
batch_norm / add / relu on the same tensors."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(bn, x, relu, res):
    y = F.batch_norm(x.float(), bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)
    if res is not None:
        y = y + res.float()
    return F.relu(y) if relu else y


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("shape", [(2, 64, 17, 23), (3, 256, 9, 11), (2, 2048, 4, 5), (1, 8, 3, 3), (2, 4096, 3, 4), (2, 2056, 2, 3),
                                   (5, 24, 31, 29)])
@pytest.mark.parametrize("relu,use_res", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("training", [True, False])
def test_fused_bn_matches_torch(dtype, tol, shape, relu, use_res, training):
    from pytorch_retinanet_amd.norm import FusedBatchNorm2d
    torch.manual_seed(0)
    N, C, H, W = shape
    bn = FusedBatchNorm2d(C).to(DEV)
    ref = torch.nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.3, 0.3); bn.running_var.uniform_(0.5, 2.0)
    ref.load_state_dict(bn.state_dict())
    bn.train(training); ref.train(training)
    x = (torch.randn(N, C, H, W, device=DEV) * 1.7 + 0.4).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    res = torch.randn(N, C, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True) if use_res else None
    g = torch.randn(N, C, H, W, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)

    y = bn(x, relu=relu, residual=res)
    assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(g)

    x2 = x.detach().clone().requires_grad_(True)
    res2 = res.detach().clone().requires_grad_(True) if use_res else None
    y2 = _ref(ref, x2, relu, res2)
    y2.backward(g.float())

    torch.testing.assert_close(y.float(), y2, rtol=tol, atol=tol)
    torch.testing.assert_close(x.grad.float(), x2.grad.float(), rtol=tol * 5, atol=tol * 5)
    if use_res:
        torch.testing.assert_close(res.grad.float(), res2.grad.float(), rtol=tol, atol=tol)
    scale = max(1.0, float(ref.weight.grad.abs().max()))
    torch.testing.assert_close(bn.weight.grad, ref.weight.grad, rtol=tol * 5, atol=tol * 5 * scale)
    torch.testing.assert_close(bn.bias.grad, ref.bias.grad, rtol=tol * 5, atol=tol * 5 * scale)
    torch.testing.assert_close(bn.running_mean, ref.running_mean, rtol=1e-4, atol=1e-4 if dtype == torch.float32 else 1e-2)
    torch.testing.assert_close(bn.running_var, ref.running_var, rtol=1e-4 if dtype == torch.float32 else 2e-2, atol=1e-4 if dtype == torch.float32 else 1e-2)
    assert int(bn.num_batches_tracked) == (1 if training else 0)       # nn.BatchNorm2d.forward's counter, kept by the kernel


def test_unfusable_inputs_take_the_torch_path():
    from pytorch_retinanet_amd.norm import FusedBatchNorm2d
    bn = FusedBatchNorm2d(12)                       # CPU, C % 8 != 0
    x = torch.randn(2, 12, 5, 5)
    y = bn(x, relu=True, residual=x)
    ref = F.relu(F.batch_norm(x, None, None, bn.weight, bn.bias, True) + x)
    torch.testing.assert_close(y, ref)
    bn = bn.to(DEV)
    y = bn(x.to(DEV))                               # NCHW on the device: torch path too
    assert y.shape == x.shape


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 64, 17, 23), (1, 8, 1, 1), (3, 16, 8, 8), (2, 24, 5, 2)])
@pytest.mark.parametrize("ties", [False, True])
def test_fused_maxpool_matches_torch_bitwise(dtype, shape, ties):
    """rn_maxpool3x3s2_* == F.max_pool2d forward and backward, bit for bit, incl. massive ties (small-integer data:
    the first maximum of a window takes the gradient) and windows clipped by the border."""
    from pytorch_retinanet_amd.pool import FusedMaxPool2d
    torch.manual_seed(1)
    N, C, H, W = shape
    src = torch.randint(0, 3, (N, C, H, W), device=DEV).float() if ties else torch.randn(N, C, H, W, device=DEV)
    x = src.to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    pool = FusedMaxPool2d(3, 2, 1)
    y = pool(x)
    yr = F.max_pool2d(xr, 3, 2, 1)
    assert y.shape == yr.shape and torch.equal(y, yr)
    g = torch.randn_like(yr)
    y.backward(g)
    yr.backward(g)
    assert torch.equal(x.grad, xr.grad)


def test_fused_maxpool_nan_follows_torch():
    from pytorch_retinanet_amd.pool import FusedMaxPool2d
    x = torch.randn(1, 8, 6, 6, device=DEV)
    x[0, :, 2, 2] = float("nan"); x[0, 0, 3, 3] = float("nan")
    x = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    y = FusedMaxPool2d(3, 2, 1)(x); yr = F.max_pool2d(xr, 3, 2, 1)
    assert torch.equal(torch.isnan(y), torch.isnan(yr)) and torch.equal(torch.nan_to_num(y), torch.nan_to_num(yr))
    g = torch.ones_like(yr)
    y.backward(g); yr.backward(g)
    assert torch.equal(x.grad, xr.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_fpn_lateral_plus_upsample_matches_torch_bitwise(dtype):
    """rn_fpn_add_upsample2x / rn_fpn_upsample2x_backward == lat + nn.Upsample(scale_factor=2)(top) (layers.py:36,52-53) forward
    and backward, bit for bit: one rounding of an f32 sum either way (small-integer gradients keep the 2 x 2 sums exact)."""
    from pytorch_retinanet_amd.pool import add_upsample2x
    torch.manual_seed(3)
    dev = "cuda:0"
    lat = torch.randn(2, 16, 10, 12, device=dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    top = torch.randn(2, 16, 5, 6, device=dev).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randint(-4, 5, (2, 16, 10, 12), device=dev).to(dtype).contiguous(memory_format=torch.channels_last)
    out = add_upsample2x(lat, top)
    assert out is not None
    out.backward(g)
    got = (out.detach().clone(), lat.grad.clone(), top.grad.clone())
    lat.grad = None; top.grad = None
    ref = lat + torch.nn.Upsample(scale_factor=2, mode="nearest")(top)
    ref.backward(g)
    assert torch.equal(got[0], ref.detach()) and torch.equal(got[1], lat.grad) and torch.equal(got[2], top.grad)
    assert add_upsample2x(lat, torch.randn(2, 16, 5, 7, device=dev).to(dtype).contiguous(memory_format=torch.channels_last)) is None
