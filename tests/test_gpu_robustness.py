"""Behaviour at the edges of the drop-in boundary that the round-1 review found untested: tensors on a device that is
not the current one, contiguous views with a misaligned storage offset, and a cluster-resident launch that cannot
complete (its bounded wait runs out): it must give up WITHOUT trapping, tell the host, and the library must carry on
with the two-pass kernels."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(5, 3, 7, 7), (9, 6, 14, 14), (7, 4, 56, 56)], ids=lambda s: "x".join(map(str, s)))
def test_offset_views_are_accepted(shape, dtype):
    """x[1:] of a contiguous tensor is contiguous but starts 1 image into the storage: (3,7,7) fp32 planes put it at
    byte 588 — not a multiple of 16.  The reference just computes on it (models/cnsn.py:14)."""
    torch.manual_seed(1)
    np.random.seed(1)
    n, c = shape[:2]
    base = torch.randn(n + 1, *shape[1:], device=DEV).to(dtype)
    gbase = torch.randn(n + 1, *shape[1:], device=DEV).to(dtype)
    view, gview = base[1:], gbase[1:]
    res = []
    for xin, gin in ((view, gview), (view.clone(), gview.clone())):
        torch.manual_seed(2)
        np.random.seed(2)
        mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm("both", 1), fill_sn(cnsn_amd.SelfNorm(c), 3, torch.float32)).to(DEV).train()
        mod.crossnorm.active = True
        x = xin.detach().requires_grad_()
        y = mod(x)
        y.backward(gin)
        m, s = cnsn_amd.calc_ins_mean_std(xin)
        res.append((y.detach(), x.grad, mod.selfnorm.g_fc.weight.grad, m, s))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert view.data_ptr() % 16 != 0 or shape[2] * shape[3] * shape[1] * base.element_size() % 16 == 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_tensor_on_a_device_that_is_not_current():
    """model.to('cuda:1') without set_device(1): launches must go to cuda:1 (grid sizing, launch chaining and the
    stream all belong to the tensor's device)."""
    torch.cuda.set_device(0)
    d1 = torch.device("cuda:1")
    out = []
    for dev in (DEV, d1):
        torch.manual_seed(3)
        np.random.seed(3)
        mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm("neither", 1), fill_sn(cnsn_amd.SelfNorm(16), 3, torch.float32)).to(dev).train()
        mod.crossnorm.active = True
        x = torch.randn(32, 16, 56, 56, generator=torch.Generator().manual_seed(4)).to(dev).requires_grad_()
        y = mod(x)
        y.backward(torch.ones_like(y))
        torch.cuda.synchronize(dev)
        assert y.device == dev and x.grad.device == dev
        out.append((y.detach().cpu(), x.grad.cpu()))
    assert torch.cuda.current_device() == 0
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


_TIMEOUT_SCRIPT = r'''
import os, sys, torch, numpy as np
sys.path.insert(0, %r)
import cnsn_amd
from cnsn_amd import _ffi
cnsn_amd.follow_environ()
from tests.golden.gen_golden_fill import fill_sn
dev = torch.device("cuda:0")
torch.manual_seed(0); np.random.seed(0)
x = torch.randn(64, 8, 56, 56, device=dev)
def run(strategy):
    cnsn_amd.set_strategy(strategy)
    sn = fill_sn(cnsn_amd.SelfNorm(8), 3, torch.float32).to(dev).train()
    y = sn(x)
    torch.cuda.synchronize()
    return y
assert cnsn_amd.which_path(x, cnsn_amd.FusedConfig(sn_active=True)) == "resident"
ref = run("two_pass")
assert _ffi.lib().cnsn_resident_timeouts() == 0
os.environ["CNSN_FAULT_INJECT"] = "1"          # the last member of channel 0's cluster never publishes
bad = run("auto")                              # gives up after CNSN_WAIT_MS, no trap: the context is still alive
os.environ["CNSN_FAULT_INJECT"] = "0"
assert _ffi.lib().cnsn_resident_timeouts() == 1, _ffi.lib().cnsn_resident_timeouts()
assert torch.isnan(bad).any()                  # the incomplete output is loud on the SAME step: NaNs in the planes still owed
try:
    run("auto")
    raise SystemExit("the time-out was not reported")
except cnsn_amd.CnsnError as e:
    assert "timed out" in str(e)
# reported once; from now on AUTO (and a forced 'resident') resolve to the two-pass kernels and results are right
assert cnsn_amd.which_path(x, cnsn_amd.FusedConfig(sn_active=True)) == "streaming"
again = run("auto")
assert torch.equal(again, ref)
forced = run("resident")
assert torch.equal(forced, ref)
print("TIMEOUT-PATH-OK")
'''


def test_resident_timeout_degrades_instead_of_trapping():
    env = dict(os.environ, CNSN_WAIT_MS="200")
    # CNSN_PIPE=2: the pipelined forward gives up the same way; CNSN_SNX=0: the general kernels instead of the
    # SelfNorm-only cluster kernels (which AUTO takes for this call)
    for glue, pipe, snx in (("0", "1", "1"), ("1", "1", "1"), ("0", "1", "0"), ("1", "1", "0"), ("0", "2", "0")):
        env["CNSN_NO_GLUE"] = glue
        env["CNSN_PIPE"] = pipe
        env["CNSN_SNX"] = snx
        r = subprocess.run([sys.executable, "-c", _TIMEOUT_SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "TIMEOUT-PATH-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_resident_can_be_switched_off():
    x = torch.randn(64, 8, 56, 56, device=DEV)
    cfg = cnsn_amd.FusedConfig(sn_active=True)
    assert cnsn_amd.which_path(x, cfg) == "resident"
    cnsn_amd.set_resident(False)
    try:
        assert cnsn_amd.which_path(x, cfg) == "streaming"
        cnsn_amd.set_strategy("resident")            # an explicit request is still honoured
        assert cnsn_amd.which_path(x, cfg) == "resident"
    finally:
        cnsn_amd.set_strategy("auto")
        cnsn_amd.set_resident(True)
    assert cnsn_amd.which_path(x, cfg) == "resident"
    r = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); import torch, cnsn_amd; "
                        "x = torch.randn(64, 8, 56, 56, device='cuda'); "
                        "print(cnsn_amd.which_path(x, cnsn_amd.FusedConfig(sn_active=True)))"],
                       capture_output=True, text=True, env=dict(os.environ, CNSN_RESIDENT="0"), timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("streaming"), (r.stdout, r.stderr[-2000:])
