"""`saved` is one contract for every strategy: the forward of any strategy can be followed by the backward of any other
(AUTO does mix them — e.g. 16-bit 13-slot planes run the two-pass forward and the cluster-resident backward).  For
shapes where several strategies are eligible, every (forward strategy, backward strategy) pair must reproduce the
gradients of the all-two-pass run within float rounding — the layout of `saved` (channel by channel, row by row over
the batch: csrc/cnsn_layout.h) is exercised from both sides by every kernel family."""
import itertools

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402

# shape, dtype, kind, crop, strategies that are eligible there (forced ones fall back to two-pass when they are not)
CASES = [
    ((37, 6, 28, 28), torch.float32, "sn", "neither", ["two_pass", "resident", "auto"]),
    ((37, 6, 28, 28), torch.float32, "cnsn", "both", ["two_pass", "resident", "auto"]),
    ((40, 8, 56, 56), torch.bfloat16, "cnsn", "neither", ["two_pass", "resident", "auto"]),
    ((64, 8, 14, 14), torch.float32, "sn", "neither", ["two_pass", "resident", "local", "mono", "auto"]),
    ((64, 8, 14, 14), torch.bfloat16, "cnsn", "style", ["two_pass", "mono", "auto"]),
    ((128, 16, 7, 7), torch.bfloat16, "sn", "neither", ["two_pass", "local", "mono", "auto"]),   # mono = the wide kernels here
    ((128, 8, 8, 8), torch.float32, "sn", "neither", ["two_pass", "local", "mono", "auto"]),
    ((6, 4, 128, 96), torch.float32, "cnsn", "content", ["two_pass", "resident", "auto"]),          # split planes
]


@pytest.fixture(autouse=True)
def auto():
    yield
    cnsn_amd.set_strategy("auto")


@pytest.mark.parametrize("shape,dtype,kind,crop,strategies", CASES, ids=lambda v: str(v).replace(" ", "") if isinstance(v, tuple) else None)
def test_any_forward_feeds_any_backward(shape, dtype, kind, crop, strategies):
    n, c = shape[:2]
    g = torch.Generator(device="cuda").manual_seed(9)
    x = (torch.randn(shape, device="cuda", generator=g) * 1.3 + 0.4).to(dtype).requires_grad_()
    gy = torch.randn(shape, device="cuda", generator=g).to(dtype)
    res = {}
    for fwd, bwd in itertools.product(strategies, strategies):
        torch.manual_seed(2)
        np.random.seed(2)
        mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1) if kind != "sn" else None, cnsn_amd.SelfNorm(c)).cuda().train()
        with torch.no_grad():
            for p in mod.parameters():
                p.copy_(torch.randn_like(p) * 0.5)
        if mod.crossnorm is not None:
            mod.crossnorm.active = True
        cnsn_amd.set_strategy(fwd)
        y = mod(x)
        cnsn_amd.set_strategy(bwd)
        grads = torch.autograd.grad(y, [x] + list(mod.parameters()), gy)
        res[(fwd, bwd)] = [y.detach().float()] + [t.float() for t in grads]
    ref = res[("two_pass", "two_pass")]
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    for key, out in res.items():
        for i, (a, b) in enumerate(zip(ref, out)):
            scale = max(1.0, float(a.abs().max()))
            err = float((a - b).abs().max())
            assert err <= tol * scale, (shape, dtype, kind, crop, key, i, err, scale)


# The SECOND contract (round 6): a channels-last call without CrossNorm and with one gate keeps the SLIM record — five floats per
# plane in plane order + C doubles (csrc/cnsn_nhwc_fused_kernels.h) — which the single-launch kernels write and read directly and
# the two-pass channels-last kernels convert to and from their own.  Any forward of the class feeds any backward of the class.
CL_CASES = [((37, 16, 14, 14), torch.float32), ((24, 64, 7, 7), torch.bfloat16), ((9, 8, 28, 28), torch.float32)]


@pytest.mark.parametrize("shape,dtype", CL_CASES, ids=lambda v: str(v).replace(" ", "") if isinstance(v, tuple) else str(v))
@pytest.mark.parametrize("mode,relu", [("none", False), ("pre", True), ("post", True), ("pre", False)])
def test_channels_last_slim_record_any_forward_feeds_any_backward(shape, dtype, mode, relu):
    n, c = shape[:2]
    CL = torch.channels_last
    g = torch.Generator(device="cuda").manual_seed(19)
    x = (torch.randn(shape, device="cuda", generator=g) * 1.3 + 0.4).to(dtype).contiguous(memory_format=CL)
    b = (torch.randn(shape, device="cuda", generator=g) * 0.6).to(dtype).contiguous(memory_format=CL)
    gy = torch.randn(shape, device="cuda", generator=g).to(dtype).contiguous(memory_format=CL)
    res, paths = {}, set()
    for fwd, bwd in itertools.product(["two_pass", "auto"], ["two_pass", "auto"]):
        torch.manual_seed(2)
        mod = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(c)).cuda().train()
        with torch.no_grad():
            for p in mod.parameters():
                p.copy_(torch.randn_like(p) * 0.5)
        xg, bg = x.clone(memory_format=CL).requires_grad_(), b.clone(memory_format=CL).requires_grad_()
        cnsn_amd.set_strategy(fwd)
        paths.add((fwd, cnsn_amd.which_path(xg, cnsn_amd.FusedConfig(sn_active=True, add_mode=mode, relu=relu))))
        y = mod.forward_block(xg, bg if mode != "none" else None, add_mode=mode, relu=relu)
        cnsn_amd.set_strategy(bwd)
        grads = torch.autograd.grad(y, [xg] + ([bg] if mode != "none" else []) + list(mod.parameters()), gy)
        res[(fwd, bwd)] = [y.detach().float()] + [t.float() for t in grads]
    assert ("two_pass", "streaming") in paths and ("auto", "resident") in paths      # the two strategies really are two
    ref = res[("two_pass", "two_pass")]
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    mask = (ref[0] != 0) if relu else torch.ones_like(ref[0], dtype=torch.bool)   # (a ReLU mask may flip where the pre-activation is rounding noise)
    for key, out in res.items():
        agree = ((out[0] != 0) == (ref[0] != 0)) if relu else mask
        assert float(agree.float().mean()) > 0.999
        for i, (a, bb) in enumerate(zip(ref, out)):
            scale = max(1.0, float(a.abs().max()))
            d = (a - bb).abs()
            if d.shape == agree.shape:
                d = d * agree
            assert float(d.max()) <= tol * scale, (shape, dtype, mode, relu, key, i, float(d.max()), scale)
