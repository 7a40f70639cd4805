"""InstanceNorm2d / IBN on the plane-statistics kernels vs torch.nn's own modules (what the reference's IBN
backbones instantiate, models/imagenet/resnet_ibn_cnsn.py:24-44) evaluated on the CPU in fp64."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from cnsn_amd.callers import IBN, InstanceNorm2d  # noqa: E402

DEV = torch.device("cuda:0")


@pytest.mark.parametrize("shape", [(4, 6, 8, 8), (3, 5, 9, 11), (2, 8, 56, 56), (2, 3, 136, 136)])
@pytest.mark.parametrize("affine", [True, False])
def test_instance_norm_matches_torch(shape, affine):
    torch.manual_seed(1)
    c = shape[1]
    x = torch.randn(shape, dtype=torch.float64) * 2 + 0.5
    gy = torch.randn(shape, dtype=torch.float64)
    x[0, 0] = 3.0                                        # a constant plane: var = 0, rstd = 1/sqrt(eps), finite grads
    gy[0, 0] = 0.0                                       # (kept out of the parameter gradients: rounding of its mean
                                                         #  is amplified by 1/sqrt(eps) = 316 on either side)
    ref = nn.InstanceNorm2d(c, affine=affine).double()
    mod = InstanceNorm2d(c, affine=affine).to(DEV)
    if affine:
        w, b = torch.rand(c, dtype=torch.float64) + 0.5, torch.randn(c, dtype=torch.float64)
        ref.weight.data.copy_(w), ref.bias.data.copy_(b)
        mod.weight.data.copy_(w), mod.bias.data.copy_(b)
        assert list(mod.state_dict().keys()) == list(ref.state_dict().keys())
    xr = x.clone().requires_grad_()
    ref(xr).backward(gy)
    xg = x.float().to(DEV).requires_grad_()
    y = mod(xg)
    y.backward(gy.float().to(DEV))
    live = torch.ones(shape[:2], dtype=torch.bool)
    live[0, 0] = False
    yr = ref(x)
    ey = (y.detach().cpu().double() - yr.detach()).abs()
    assert float(ey[live].max()) <= 1e-5 * max(1.0, float(yr.abs().max()))
    assert float(ey.max()) <= 1e-3     # constant plane: rounding of its mean is amplified by 1/sqrt(eps) = 316
    # (gradient of the constant plane: 0 * huge in exact arithmetic; compared loosely)
    err = (xg.grad.cpu().double() - xr.grad).abs()
    assert float(err[live].max()) <= 1e-5 * max(1.0, float(xr.grad.abs().max()))
    assert torch.isfinite(xg.grad).all()
    if affine:
        assert float((mod.weight.grad.cpu().double() - ref.weight.grad).abs().max()) <= 1e-4 * float(ref.weight.grad.abs().max())
        assert float((mod.bias.grad.cpu().double() - ref.bias.grad).abs().max()) <= 1e-4 * max(1.0, float(ref.bias.grad.abs().max()))


def test_ibn_matches_reference_layout():
    torch.manual_seed(2)
    x = torch.randn(4, 10, 12, 12, dtype=torch.float64)
    ref_in, ref_bn = nn.InstanceNorm2d(5, affine=True).double(), nn.BatchNorm2d(5).double()
    mod = IBN(10).to(DEV).train()
    assert sorted(k for k in mod.state_dict()) == sorted(
        [f"IN.{k}" for k in ref_in.state_dict()] + [f"BN.{k}" for k in ref_bn.state_dict()])
    want = torch.cat((ref_in(x[:, :5].contiguous()), ref_bn.train()(x[:, 5:].contiguous())), 1)
    got = mod(x.float().to(DEV))
    assert float((got.cpu().double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_instance_norm_bf16():
    x = torch.randn(4, 8, 28, 28)
    mod = InstanceNorm2d(8).to(DEV)
    y = mod(x.to(DEV).bfloat16())
    want = nn.InstanceNorm2d(8, affine=True)(x.bfloat16().float())
    assert y.dtype == torch.bfloat16
    assert float((y.float().cpu() - want).abs().max()) <= 2e-2 * float(want.abs().max())
