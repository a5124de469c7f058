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
