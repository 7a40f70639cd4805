"""GPU parity of the residual-block epilogue: y = act(CNSN(x [+ addend]) [+ addend]) issued as
cnsn_forward_fused / cnsn_backward_fused, against the oracle's CNSN composed with torch's add / relu exactly as the
reference's blocks spell it (models/imagenet/resnet_cnsn.py:112-122, models/cifar/wideresnet_cnsn.py:93-96).

The ReLU makes the gradient discontinuous where the pre-activation is ~0, so the comparison is done in two steps:
  1. forward values as usual; the ReLU masks (y > 0) of both sides may differ only where the fp64 pre-activation is
     within rounding noise of zero;
  2. gradients against the oracle differentiated THROUGH THE SAME MASK as the device used.
Tolerances as in test_gpu_parity.py (1e-5 fp32, 1e-2 bf16)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from oracle import cnsn_oracle as orc  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402
from tests.test_gpu_parity import DEV, cond_input, seed_of, to_draws  # noqa: E402

EPILOGUES = [("pre", True), ("pre", False), ("post", True), ("post", False), ("none", True)]
KINDS = [("sn", "neither"), ("cn", "neither"), ("cnsn", "neither"), ("cnsn", "both"), ("cn", "content")]
SHAPES = [
    (4, 6, 8, 8),        # 16 lanes per plane
    (6, 5, 9, 11),       # scalar accesses, odd sizes
    (8, 16, 28, 28),     # a wave per plane; resident bucket 4
    (5, 4, 56, 56),      # north-star plane
    (2, 3, 136, 136),    # block per plane
]


@pytest.fixture(params=["two_pass", "resident", "local", "mono"])
def strategy(request):
    cnsn_amd.set_strategy(request.param)
    yield request.param
    cnsn_amd.set_strategy("auto")


def oracle_block(mod, x, b, mode, relu, mask, dtype=torch.float32):
    h = x + b if mode == "pre" else x
    if mode == "pre" and dtype != torch.float32:   # `out += identity` leaves a tensor of the activations' dtype
        h = h + (h.detach().to(dtype).to(h.dtype) - h.detach())
    h = mod(h)
    if mode == "post":
        h = h + b
    if not relu:
        return h, h
    return (torch.relu(h) if mask is None else h * mask.to(h.dtype)), h


def build(lib, kind, crop, c, seed, dtype):
    cn = lib.CrossNorm(crop, 1) if kind != "sn" else None
    sn = fill_sn(lib.SelfNorm(c), seed, dtype) if kind != "cn" else None
    return lib.CNSN(cn, sn).train()


def _arm(mod, draws):
    if mod.crossnorm is not None:
        mod.crossnorm.active = True
        mod.crossnorm.next_draws = draws


def _inputs(shape, kind, crop, dtype, seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    x64 = cond_input(shape, seed)
    b64 = cond_input(shape, seed + 1) * 0.7
    gy64 = torch.randn(shape, dtype=torch.float64)
    if dtype != torch.float32:
        x64, b64, gy64 = (v.to(dtype).double() for v in (x64, b64, gy64))
    d = orc.draw_cn(shape, crop, beta=1) if kind != "sn" else None
    return x64, b64, gy64, d


def _oracle_side(shape, kind, crop, mode, relu, dtype, seed, mask):
    """the oracle's forward with its own ReLU and its backward THROUGH THE DEVICE'S MASK, fp64 and fp32: depends on the
    case and on the mask only, not on the kernel strategy (tests/_memo.py keys it on the mask's bytes)"""
    x64, b64, gy64, d = _inputs(shape, kind, crop, dtype, seed)
    c = shape[1]
    out = {}
    for tag, odt in (("t64", torch.float64), ("o32", torch.float32)):
        ref = build(orc, kind, crop, c, seed, odt)
        _arm(ref, d)
        xr = x64.detach().clone().to(odt).requires_grad_()
        br = b64.detach().clone().to(odt).requires_grad_() if mode != "none" else None
        with torch.no_grad():                              # the oracle's own ReLU, for the forward comparison
            probe = build(orc, kind, crop, c, seed, odt)
            _arm(probe, d)
            y_own, pre = oracle_block(probe, xr.detach(), br.detach() if br is not None else None, mode, relu, None, dtype)
        out["own_" + tag] = dict(y=y_own, pre=pre)
        y, _ = oracle_block(ref, xr, br, mode, relu, mask, dtype)
        y.backward(gy64.to(odt))
        out[tag] = dict(y=y.detach(), dx=xr.grad, db=br.grad if br is not None else None,
                        pg={k: v.grad for k, v in ref.named_parameters()},
                        st={k: v for k, v in ref.state_dict().items()})
    return out


def run_case(shape, kind, crop, mode, relu, dtype, seed, channels_last=False):
    """channels_last: x / addend / grad_y handed over in torch.channels_last order (tests/test_gpu_nhwc.py) — same values, so
    the oracle side is the same"""
    from tests._memo import memo
    n, c = shape[:2]
    case = (tuple(shape), kind, crop, str(dtype), seed)
    x64, b64, gy64, d = memo(("block-in",) + case, lambda: _inputs(shape, kind, crop, dtype, seed))

    # device
    mod = build(cnsn_amd, kind, crop, c, seed, torch.float32).to(DEV)
    _arm(mod, to_draws(d) if d else None)
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    xg = x64.detach().clone().to(dtype).to(DEV).contiguous(memory_format=fmt).requires_grad_()
    bg = b64.detach().clone().to(dtype).to(DEV).contiguous(memory_format=fmt).requires_grad_() if mode != "none" else None
    yg = mod.forward_block(xg, bg, add_mode=mode, relu=relu)
    if channels_last:      # computed where the tensors lie: the output comes back in the same memory order
        assert yg.is_contiguous(memory_format=torch.channels_last) and yg.shape == xg.shape
    yg.backward(gy64.to(dtype).to(DEV).contiguous(memory_format=fmt))
    torch.cuda.synchronize()
    if channels_last:
        assert xg.grad.is_contiguous(memory_format=torch.channels_last)
    assert mod.crossnorm is None or mod.crossnorm.active is False
    hip = dict(y=yg.detach().cpu(), dx=xg.grad.cpu(), db=bg.grad.cpu() if bg is not None else None,
               pg={k: v.grad.cpu() for k, v in mod.named_parameters()},
               st={k: v.cpu() for k, v in mod.state_dict().items()})
    mask = (hip["y"] > 0) if relu else None
    mask_key = hash(mask.numpy().tobytes()) if mask is not None else None
    out = dict(memo(("block-ref",) + case + (mode, relu, mask_key),
                    lambda: _oracle_side(shape, kind, crop, mode, relu, dtype, seed, mask)))
    out["hip"] = hip
    return out


def check(out, dtype, relu, ctx):
    t64, o32, hip, own, own32 = out["t64"], out["o32"], out["hip"], out["own_t64"], out["own_o32"]

    def one(name, got, truth, ref32, tol):
        got, truth, ref32 = got.double(), truth.double(), ref32.double()
        if dtype == torch.float32:
            scale = max(1.0, float(truth.abs().max()))
            e, e32 = float((got - truth).abs().max()), float((ref32 - truth).abs().max())
            assert e <= max(tol * scale, 2 * e32), f"{ctx} {name}: err {e:.3e} (oracle32 {e32:.3e}, scale {scale:.3g})"
        else:
            e = float((got - ref32).abs().max())
            assert e <= 1e-2 * max(float(ref32.abs().max()), 1e-3), f"{ctx} {name}: err {e:.3e}"

    # forward against the oracle's own ReLU
    one("y", hip["y"], own["y"], own32["y"], 1e-5)
    if relu:
        band = (1e-4 if dtype == torch.float32 else 3e-2) * max(1.0, float(own["pre"].abs().max()))
        differ = (hip["y"] > 0) != (own["pre"] > 0)
        assert not bool((differ & (own["pre"].abs() > band)).any()), f"{ctx}: ReLU mask differs away from zero"
        assert float(differ.double().mean()) < 1e-2
    one("dx", hip["dx"], t64["dx"], o32["dx"], 1e-5)
    if hip["db"] is not None:
        one("d_addend", hip["db"], t64["db"], o32["db"], 1e-5)
    for k in t64["pg"]:
        one(f"grad {k}", hip["pg"][k], t64["pg"][k], o32["pg"][k], 1e-4)
    for k in t64["st"]:
        if "num_batches" in k:
            assert int(hip["st"][k]) == int(t64["st"][k])
        else:
            one(f"state {k}", hip["st"][k], t64["st"][k], o32["st"][k], 1e-5)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("kind,crop", KINDS)
@pytest.mark.parametrize("mode,relu", EPILOGUES)
def test_fused_block_fp32(strategy, shape, kind, crop, mode, relu):
    seed = seed_of(shape, kind, crop, mode, relu)
    check(run_case(shape, kind, crop, mode, relu, torch.float32, seed), torch.float32, relu,
          (shape, kind, crop, mode, relu, strategy))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(8, 16, 28, 28), (4, 6, 56, 56), (4, 8, 7, 7)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("kind,crop", [("sn", "neither"), ("cnsn", "both")])
@pytest.mark.parametrize("mode,relu", [("pre", True), ("post", True), ("pre", False)])
def test_fused_block_16bit(strategy, dtype, shape, kind, crop, mode, relu):
    seed = seed_of(shape, kind, crop, mode, relu, str(dtype))
    check(run_case(shape, kind, crop, mode, relu, dtype, seed), dtype, relu, (shape, kind, crop, mode, relu, dtype, strategy))


def test_fused_block_matches_unfused_ops_full_size():
    """(96,256,56,56) bf16 — the ResNet-50 layer-1 site of BASELINE configs[3]: the fused block equals
    relu(cnsn(out + identity)) evaluated with this library's own CNSN and torch's add / relu."""
    shape, dt = (96, 256, 56, 56), torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(shape, device=DEV, dtype=dt, generator=g).requires_grad_()
    b = (torch.randn(shape, device=DEV, dtype=dt, generator=g) * 0.5).requires_grad_()
    gy = torch.randn(shape, device=DEV, dtype=dt, generator=g)
    m1 = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(256), 7, torch.float32)).to(DEV).train()
    m2 = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(256), 7, torch.float32)).to(DEV).train()
    y1 = m1.forward_block(x, b, add_mode="pre", relu=True)
    y1.backward(gy)
    dx1, db1 = x.grad.clone(), b.grad.clone()
    x.grad = b.grad = None
    y2 = torch.relu(m2(x + b))
    y2.backward(gy)
    assert float((y1.float() - y2.float()).abs().max()) <= 2e-2 * float(y2.float().abs().max())
    # the 16-bit masks may differ on values that round to zero: compare gradients where both are open or both shut
    same = (y1 > 0) == (y2 > 0)
    assert float(same.float().mean()) > 0.999
    err = ((dx1.float() - x.grad.float()).abs() * same).max()
    assert float(err) <= 2e-2 * float(x.grad.float().abs().max())
    assert torch.equal(dx1, db1)
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert float((p1.grad - p2.grad).abs().max()) <= 2e-2 * max(float(p2.grad.abs().max()), 1e-3), k


def test_fused_block_falls_back_to_plain_ops_when_idle():
    """A CNSN with an idle CrossNorm and no SelfNorm is the identity: forward_block is add + relu."""
    m = cnsn_amd.CNSN(cnsn_amd.CrossNorm("neither", 1), None).to(DEV).train()
    x = torch.randn(4, 3, 8, 8, device=DEV)
    b = torch.randn(4, 3, 8, 8, device=DEV)
    assert torch.equal(m.forward_block(x, b, add_mode="pre", relu=True), torch.relu(x + b))
    assert torch.equal(m.forward_block(x, b, add_mode="post", relu=False), x + b)


def test_fused_abi_argument_errors():
    import ctypes as C
    from cnsn_amd import _ffi
    lib = cnsn_amd.lib()
    x = torch.randn(4, 4, 8, 8, device=DEV)
    y = torch.empty_like(x)
    prob = cnsn_amd.functional._problem(x, cnsn_amd.FusedConfig(sn_active=False, cn_active=False))
    ws_bytes = lib.cnsn_workspace_bytes(C.byref(prob))
    ws = torch.empty(ws_bytes // 4 + 4, device=DEV)
    epi = _ffi.Epilogue(C.sizeof(_ffi.Epilogue), _ffi.ADD_PRE, 1, 0, None)
    args = (C.c_void_p(x.data_ptr()), None, None, None, None, C.c_void_p(y.data_ptr()), None,
            C.c_void_p(ws.data_ptr()), ws_bytes, None)
    assert lib.cnsn_forward_fused(C.byref(prob), C.byref(epi), *args) == -1       # addend NULL
    epi.struct_bytes = 3
    assert lib.cnsn_forward_fused(C.byref(prob), C.byref(epi), *args) == -8       # struct size
    epi = _ffi.Epilogue(C.sizeof(_ffi.Epilogue), 7, 0, 0, x.data_ptr())
    assert lib.cnsn_forward_fused(C.byref(prob), C.byref(epi), *args) == -9       # unknown add mode
    # epilogue NULL == plain cnsn_forward
    assert lib.cnsn_forward_fused(C.byref(prob), None, *args) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, x)
