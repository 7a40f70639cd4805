"""Round 4: the backward of un-boxed CrossNorm + SelfNorm in the partial-moment cluster kernels
(`resident_sn_bwd_kernel<..., CN = true>`, csrc/cnsn_resident_sn_kernels.h) — reference models/cnsn.py:58-91 with
crop='neither' in front of :130-150; the style source is not detached (:66-68), so a plane's gradient has a contribution
from the plane that borrowed its statistics.

Every instantiated class (register bucket x element type) forced with CNSN_SNXCN=2, batch sizes below / above one
workgroup's planes, the permutation as a launch argument and as a device array, exchange through the persistent context
and through the workspace:
  * against the oracle (fp64 truth + fp32), BASELINE.json's tolerances (tests/test_gpu_parity.py);
  * against the general cluster kernels on the same inputs (CNSN_SNXCN=0): same algebra functions, another order of the
    batch sums — tight;
  * the forward is the general kernels' in both cases: `saved` is one contract."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402
from tests.test_gpu_parity import assert_parity, run_pair  # noqa: E402
from tests.test_gpu_resident_instantiations import DT, PLANES_F32, PLANES_V16  # noqa: E402

DEV = torch.device("cuda:0")
CASES = [("f32", nv, hw) for nv, hw in PLANES_F32.items() if nv >= 7] + \
        [(t, nv, hw) for nv, hw in PLANES_V16.items() if nv >= 7 for t in ("bf16", "f16")]


@pytest.fixture
def forced():
    old = {k: os.environ.get(k) for k in ("CNSN_SNXCN", "CNSN_CONTEXT")}
    os.environ["CNSN_SNXCN"] = "2"
    cnsn_amd.set_strategy("resident")
    yield
    cnsn_amd.set_strategy("auto")
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _takes(shape, dtype, crop="neither"):
    x = torch.empty(shape, dtype=dtype, device=DEV)
    cfg = cnsn_amd.FusedConfig(cn_active=True, sn_active=True,
                               content_box=(1, 1, 5, 5) if crop in ("both", "content") else None,
                               style_box=(0, 0, 4, 4) if crop in ("both", "style") else None)
    return cnsn_amd.sn_cluster(x, cfg, backward=True) and not cnsn_amd.sn_cluster(x, cfg, backward=False)


@pytest.mark.parametrize("tag,nv,hw", CASES, ids=lambda v: str(v).replace(" ", ""))
@pytest.mark.parametrize("n", [5, 37])
@pytest.mark.parametrize("crop", ["neither", "both", "style", "content"])
def test_against_the_oracle(forced, tag, nv, hw, n, crop):
    """crop != 'neither' (round 4, BOXED): four sums per plane, the 11-coefficient piecewise dx, bit masks per lane and slot"""
    if crop in ("style", "content") and n == 5:
        pytest.skip("one box alone: at N = 37 only")
    shape = (n, 4, *hw)
    assert _takes(shape, DT[tag], crop), (shape, tag, crop)
    out = run_pair(shape, crop, "cnsn", DT[tag], 1200 + nv + n, training=True)
    assert_parity(out, DT[tag], ("cn-partial", tag, nv, shape, crop))


def _both(shape, dtype, seed, lam=None, device_perm=False, context=None, crop="neither"):
    torch.manual_seed(seed)
    np.random.seed(seed)
    g = torch.Generator(device=DEV).manual_seed(seed)
    n, c = shape[:2]
    x = (torch.randn(shape, device=DEV, generator=g) * (torch.rand(n, c, 1, 1, device=DEV, generator=g) + 0.5)
         + torch.randn(n, c, 1, 1, device=DEV, generator=g)).to(dtype)
    gy = torch.randn(shape, device=DEV, generator=g).to(dtype)
    d = cnsn_amd.draw_cn(shape, crop, 1)
    perm = d.perm
    res = {}
    for mode in ("2", "0"):
        os.environ["CNSN_SNXCN"] = mode
        if context is not None:
            os.environ["CNSN_CONTEXT"] = context
        sn = fill_sn(cnsn_amd.SelfNorm(c), seed, torch.float32).to(DEV).train()
        kw, gp, _ = sn._fused_args()
        cfg = cnsn_amd.FusedConfig(cn_active=True, lam=lam, content_box=d.content_box, style_box=d.style_box, **kw)
        assert cnsn_amd.sn_cluster(x, cfg, backward=True) == (mode == "2")
        xg = x.clone().requires_grad_()
        y = cnsn_amd.fused_cnsn(xg, cfg, perm=perm.to(DEV) if device_perm else perm, g=gp)
        y.backward(gy)
        torch.cuda.synchronize()
        res[mode] = [y.detach().float(), xg.grad.float()] + [p.grad.float() for p in sn.parameters()]
    return res


@pytest.mark.parametrize("shape,dtype", [((37, 6, 40, 40), torch.float32), ((5, 4, 56, 56), torch.float32),
                                         ((70, 4, 56, 56), torch.float32), ((70, 4, 56, 56), torch.bfloat16),
                                         ((300, 2, 56, 56), torch.float32), ((19, 3, 60, 64), torch.float16)],
                         ids=lambda v: str(v).replace(" ", "").replace("torch.", ""))
@pytest.mark.parametrize("variant", ["inline", "device_perm", "workspace", "lam", "boxed", "boxed_workspace", "boxed_lam"])
def test_against_the_general_cluster_kernels(forced, shape, dtype, variant):
    res = _both(shape, dtype, 77, lam=0.3 if variant.endswith("lam") else None, device_perm=variant == "device_perm",
                context="0" if variant.endswith("workspace") else None, crop="both" if variant.startswith("boxed") else "neither")
    a, b = res["2"], res["0"]
    assert torch.equal(a[0], b[0])                                        # the same forward kernels ran
    for i, (u, v) in enumerate(zip(a[1:], b[1:])):
        scale = max(1.0, float(v.abs().max()))
        tol = (2e-6 if i == 0 else 2e-5) if dtype == torch.float32 else (1e-2 if i == 0 else 1e-4)
        err = float((u - v).abs().max())
        assert err <= tol * scale, (shape, dtype, variant, i, err, scale)


def test_many_channels_and_auto(forced):
    """the persistent loop over many channels (several items per cluster), and what AUTO takes at the north-star shape"""
    res = _both((37, 700, 40, 40), torch.float32, 5)
    assert float((res["2"][1] - res["0"][1]).abs().max()) <= 2e-6 * max(1.0, float(res["0"][1].abs().max()))
    os.environ.pop("CNSN_SNXCN", None)
    cnsn_amd.set_strategy("auto")
    cfg = cnsn_amd.FusedConfig(cn_active=True, sn_active=True)
    assert cnsn_amd.sn_cluster(torch.empty(256, 256, 56, 56, device=DEV), cfg, backward=True)
    assert cnsn_amd.sn_cluster(torch.empty(256, 256, 56, 56, device=DEV, dtype=torch.bfloat16), cfg, backward=True)
    boxed = cnsn_amd.FusedConfig(cn_active=True, sn_active=True, style_box=(0, 0, 9, 9), content_box=(2, 2, 30, 30))
    assert cnsn_amd.sn_cluster(torch.empty(256, 256, 56, 56, device=DEV), boxed, backward=True)       # crop boxes, fp32: every batch
    assert cnsn_amd.sn_cluster(torch.empty(256, 256, 56, 56, device=DEV, dtype=torch.bfloat16), boxed, backward=True)
    assert cnsn_amd.sn_cluster(torch.empty(96, 256, 56, 56, device=DEV, dtype=torch.bfloat16), boxed, backward=True)   # ... 16-bit 56x56 too
    assert not cnsn_amd.sn_cluster(torch.empty(96, 256, 64, 64, device=DEV, dtype=torch.bfloat16), boxed, backward=True)  # 8 slots: general kernels
    assert not cnsn_amd.sn_cluster(torch.empty(256, 512, 28, 28, device=DEV), cfg, backward=True)      # below 7 slots: not built


_GIVE_UP = r'''
import os, sys, torch, numpy as np
sys.path.insert(0, %r)
import cnsn_amd
from cnsn_amd import _ffi
cnsn_amd.follow_environ()
from tests.golden.gen_golden_fill import fill_sn
dev = torch.device("cuda:0")
torch.manual_seed(0); np.random.seed(0)
x0 = torch.randn(64, 8, 56, 56, device=dev)
gy = torch.randn(64, 8, 56, 56, device=dev)
perm = torch.randperm(64)
def run(strategy, inject=False):
    cnsn_amd.set_strategy(strategy)
    mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm("neither", 1), fill_sn(cnsn_amd.SelfNorm(8), 3, torch.float32)).to(dev).train()
    mod.crossnorm.active = True
    mod.crossnorm.next_draws = cnsn_amd.CNDraws(perm)
    x = x0.clone().requires_grad_()
    y = mod(x)
    torch.cuda.synchronize()
    os.environ["CNSN_FAULT_INJECT"] = "1" if inject else "0"
    y.backward(gy)
    torch.cuda.synchronize()
    os.environ["CNSN_FAULT_INJECT"] = "0"
    return x.grad
cfg = cnsn_amd.FusedConfig(cn_active=True, sn_active=True)
assert cnsn_amd.sn_cluster(x0, cfg, backward=True)
ref = run("two_pass")
good = run("auto")
assert float((good - ref).abs().max()) < 1e-5 * float(ref.abs().max())
bad = run("auto", inject=True)                   # a member never publishes its partial: the launch gives up, no trap
assert _ffi.lib().cnsn_resident_timeouts() == 1 and torch.isnan(bad).any()
try:
    run("auto")
    raise SystemExit("the time-out was not reported")
except cnsn_amd.CnsnError as e:
    assert "timed out" in str(e)
assert not cnsn_amd.sn_cluster(x0, cfg, backward=True)
assert torch.equal(run("auto"), ref)
print("CN-GIVE-UP-OK")
'''


def test_a_launch_that_cannot_complete_gives_up_without_trapping():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for ctx in ("1", "0"):
        env = dict(os.environ, CNSN_WAIT_MS="200", CNSN_CONTEXT=ctx)
        env.pop("CNSN_SNXCN", None)
        r = subprocess.run([sys.executable, "-c", _GIVE_UP % root], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "CN-GIVE-UP-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
