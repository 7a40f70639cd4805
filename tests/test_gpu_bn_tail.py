"""SURVEY §8 f1, second half: WideResNet's NEXT-block `relu(bn1(.))` (models/cifar/wideresnet_cnsn.py:76-77 / :69-70 /
:222) evaluated in the CNSN launch that ends the previous block (`torch.add(x, out)` -> cnsn, :93-96):
cnsn_forward_bnrelu / cnsn_backward_bnrelu through `CNSN.forward_block_bn`.

Oracle: the oracle's SelfNorm composed with torch.nn.BatchNorm2d and relu exactly as the reference spells it, in fp64
(truth) and fp32 (noise floor).  The ReLU makes the gradient discontinuous at 0, so gradients are compared with the
oracle differentiated through the DEVICE's mask (as tests/test_gpu_fused_block.py does)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from cnsn_amd.functional import FusedConfig  # noqa: E402
from oracle import cnsn_oracle as orc  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402

DEV = torch.device("cuda:0")
WRN_SITES = [(128, 64, 16, 16), (128, 128, 8, 8), (128, 32, 32, 32)]      # BASELINE configs[1], bs 128


def fill_bn(bn, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(bn.num_features, generator=g) + 0.5)
        bn.bias.copy_(torch.rand(bn.num_features, generator=g) - 0.5)
        bn.running_mean.copy_(torch.randn(bn.num_features, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(bn.num_features, generator=g) + 0.5)
    return bn.to(dtype)


def run_case(shape, dtype, add, want_y, training=True, seed=3):
    torch.manual_seed(seed)
    np.random.seed(seed)
    n, c = shape[:2]
    g = torch.Generator().manual_seed(seed)
    x64 = (torch.randn(shape, generator=g) * (torch.rand(n, c, 1, 1, generator=g) * 1.5 + 0.5) + torch.randn(n, c, 1, 1, generator=g)).double()
    b64 = (torch.randn(shape, generator=g) * 0.7).double()
    gy64 = torch.randn(shape, generator=g).double()
    gz64 = torch.randn(shape, generator=g).double()
    if dtype != torch.float32:
        x64, b64, gy64, gz64 = (v.to(dtype).double() for v in (x64, b64, gy64, gz64))

    # device
    mod = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(c), seed, torch.float32)).to(DEV)
    bn = fill_bn(torch.nn.BatchNorm2d(c), seed + 1, torch.float32).to(DEV)
    mod.train(training)
    bn.train(training)
    xg = x64.to(dtype).to(DEV).requires_grad_()
    bg = b64.to(dtype).to(DEV).requires_grad_() if add else None
    y, z = mod.forward_block_bn(xg, bg, "pre" if add else "none", bn, want_y=want_y)
    assert (y is not None) == want_y
    loss = (z.float() * gz64.float().to(DEV)).sum()
    if want_y:
        loss = loss + (y.float() * gy64.float().to(DEV)).sum()
    if training:
        loss.backward()
    torch.cuda.synchronize()
    hip = dict(y=y.detach().cpu() if want_y else None, z=z.detach().cpu(), dx=xg.grad.cpu() if training else None,
               db=bg.grad.cpu() if (add and training) else None,
               pg={k: v.grad.cpu() for k, v in list(mod.named_parameters()) + [("bn." + k, v) for k, v in bn.named_parameters()]} if training else {},
               st={**{k: v.cpu() for k, v in mod.state_dict().items()}, **{"bn." + k: v.cpu() for k, v in bn.state_dict().items()}})
    mask = hip["z"] > 0

    out = {"hip": hip}
    for tag, odt in (("t64", torch.float64), ("o32", torch.float32)):
        ref = orc.CNSN(None, fill_sn(orc.SelfNorm(c), seed, odt))
        rbn = fill_bn(torch.nn.BatchNorm2d(c), seed + 1, odt)
        ref.train(training)
        rbn.train(training)
        xr = x64.detach().clone().to(odt).requires_grad_()
        br = b64.detach().clone().to(odt).requires_grad_() if add else None
        h = xr + br if add else xr
        if add and dtype != torch.float32:          # `torch.add(x, out)` leaves a tensor of the activations' dtype
            h = h + (h.detach().to(dtype).to(odt) - h.detach())
        yr = ref(h)
        pre = rbn(yr)
        zr = pre * mask.to(odt)                    # relu through the device's mask
        loss = (zr * gz64.to(odt)).sum()
        if want_y:
            loss = loss + (yr * gy64.to(odt)).sum()
        if training:
            loss.backward()
        out[tag] = dict(y=yr.detach(), z=zr.detach(), pre=pre.detach(), dx=xr.grad, db=br.grad if add and training else None,
                        pg={k: v.grad for k, v in list(ref.named_parameters()) + [("bn." + k, v) for k, v in rbn.named_parameters()]} if training else {},
                        st={**dict(ref.state_dict()), **{"bn." + k: v for k, v in rbn.state_dict().items()}})
    return out


def check(out, dtype, ctx):
    t64, o32, hip = out["t64"], out["o32"], out["hip"]

    def one(name, got, truth, ref32, tol):
        got, truth, ref32 = got.double(), truth.double(), ref32.double()
        if dtype == torch.float32:
            scale = max(1.0, float(truth.abs().max()))
            e, e32 = float((got - truth).abs().max()), float((ref32 - truth).abs().max())
            assert e <= max(tol * scale, 2 * e32), f"{ctx} {name}: err {e:.3e} (oracle32 {e32:.3e}, scale {scale:.3g})"
        else:
            # 16-bit: 1e-2 on the tensors; parameter gradients are sums of products of 16-bit roundings (the merged
            # gradient H = Gy + gamma2*rstd2*Gz' is itself a 16-bit tensor, like the reference's own bf16 gradients): 3e-2
            e = float((got - ref32).abs().max())
            rel = 3e-2 if name.startswith("grad ") else 1e-2
            assert e <= rel * max(float(ref32.abs().max()), 1e-3), f"{ctx} {name}: err {e:.3e}"

    # the device's mask may differ from the fp64 pre-activation's sign only inside rounding noise of 0
    pre = t64["pre"]
    flipped = (hip["z"] > 0) != (pre > 0)
    tol0 = (1e-5 if dtype == torch.float32 else 2e-2) * max(1.0, float(pre.abs().max()))
    assert float(pre[flipped].abs().max()) <= tol0 if flipped.any() else True, f"{ctx}: ReLU mask differs away from 0"
    if hip["y"] is not None:
        one("y", hip["y"], t64["y"], o32["y"], 1e-5)
    one("z", hip["z"], t64["z"], o32["z"], 1e-5)
    if hip["dx"] is not None:
        one("dx", hip["dx"], t64["dx"], o32["dx"], 1e-5)
        if hip["db"] is not None:
            one("d_addend", hip["db"], t64["db"], o32["db"], 1e-5)
        for k in hip["pg"]:
            one("grad " + k, hip["pg"][k], t64["pg"][k], o32["pg"][k], 1e-4)
    for k in hip["st"]:
        if "num_batches" in k:
            assert int(hip["st"][k]) == int(t64["st"][k]), f"{ctx} {k}"
        else:
            one("buffer " + k, hip["st"][k], t64["st"][k], o32["st"][k], 1e-5)


@pytest.mark.parametrize("shape", WRN_SITES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("add", [True, False], ids=["add", "noadd"])
@pytest.mark.parametrize("want_y", [True, False], ids=["y+z", "z"])
def test_wideresnet_sites(shape, add, want_y):
    x = torch.empty(shape, device=DEV)
    fused = cnsn_amd.functional.bnrelu_plan(x, FusedConfig(sn_active=True, add_mode="pre" if add else "none"))
    assert fused == (shape[2] <= 16), "the 16x16 and 8x8 sites run the fused kernel; 32x32 composes the three steps"
    check(run_case(shape, torch.float32, add, want_y), torch.float32, f"{shape} add={add} want_y={want_y}")


@pytest.mark.parametrize("shape,dtype", [((37, 5, 16, 16), torch.float32), ((64, 6, 8, 8), torch.bfloat16), ((128, 4, 14, 14), torch.bfloat16),
                                         ((16, 3, 12, 12), torch.float16), ((100, 7, 16, 16), torch.float32)],
                         ids=lambda v: str(v))
def test_other_shapes_and_types(shape, dtype):
    check(run_case(shape, dtype, True, True), dtype, f"{shape} {dtype}")
    check(run_case(shape, dtype, False, False, seed=5), dtype, f"{shape} {dtype} z only")


@pytest.mark.parametrize("shape", [(128, 64, 16, 16), (128, 128, 8, 8)], ids=lambda s: "x".join(map(str, s)))
def test_eval_mode(shape):
    """net.eval(): SelfNorm's BatchNorm1d and the BatchNorm2d normalise with their running statistics"""
    with torch.no_grad():
        check(run_case(shape, torch.float32, True, True, training=False), torch.float32, f"{shape} eval")


def test_network_with_and_without_the_tail():
    """WideResNet-16-2 (same block structure as WRN-40-2): logits, every gradient and every buffer agree between the
    fused tail (default) and CNSN_FUSE_TAIL=0-style separate bn1 / relu1 calls; state_dict keys are identical"""
    from cnsn_amd.callers import WideResNetCNSN, _sites

    def run(tail):
        old = _sites.FUSE_TAIL
        _sites.FUSE_TAIL = tail
        try:
            torch.manual_seed(5)
            np.random.seed(5)
            net = WideResNetCNSN(16, 10, 2, active_num=2, pos="post", beta=1, crop="both", cnsn_type="cnsn").to(DEV).train()
            g = torch.Generator(device=DEV).manual_seed(1)
            x = torch.randn(128, 3, 32, 32, device=DEV, generator=g)
            outs = []
            for aug in (False, True):
                np.random.seed(9)
                torch.manual_seed(9)
                logits = net(x, aug=aug)
                net.zero_grad()
                logits.square().mean().backward()
                outs.append((logits.detach().clone(), [p.grad.clone() for p in net.parameters()]))
            return outs, {k: v.clone() for k, v in net.state_dict().items()}
        finally:
            _sites.FUSE_TAIL = old

    (a_idle, a_armed), sd_a = run(True)
    (b_idle, b_armed), sd_b = run(False)
    assert list(sd_a) == list(sd_b)
    for (la, ga), (lb, gb) in ((a_idle, b_idle), (a_armed, b_armed)):
        assert float((la - lb).abs().max()) <= 2e-3 * float(lb.abs().max())
        for u, v in zip(ga, gb):
            assert float((u - v).abs().max()) <= 2e-2 * max(float(v.abs().max()), 1e-6)
    for k in sd_a:
        assert float((sd_a[k].double() - sd_b[k].double()).abs().max()) <= 1e-3 * max(float(sd_b[k].double().abs().max()), 1e-6), k
