"""The single-launch channels-last kernels (csrc/cnsn_nhwc_fused_kernels.h, cnsn_nhwc_bnhead_kernels.h) when a launch cannot
complete: a grid barrier whose last workgroup never arrives (CNSN_FAULT_INJECT=1) gives up after the bounded wait — no trap, no
hang —, marks the outputs it still owed with NaNs, counts the time-out; the library then runs the launch-per-pass kernels (and
`forward_bn_block` the un-fused sequence) until somebody re-arms; the first launch after a re-arm finds the barrier block of the
context short of arrivals and puts it in order; the BACKWARD of a forward that ran fused still runs after a degradation (nothing
else can read its record).  Each scenario in a process of its own (the time-out counter is process-wide)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_HEAD = r'''
import os, sys, torch, numpy as np
sys.path.insert(0, %r)
import cnsn_amd
from cnsn_amd import _ffi, functional as F_
cnsn_amd.follow_environ()
from tests.golden.gen_golden_fill import fill_sn
dev = torch.device("cuda:0")
CL = torch.channels_last
lib = _ffi.lib()
torch.manual_seed(0); np.random.seed(0)
shape = (64, 16, 28, 28)
x = (torch.randn(shape, device=dev) * 1.3 + 0.2).contiguous(memory_format=CL)
b = (torch.randn(shape, device=dev) * 0.5).contiguous(memory_format=CL)
cfg = cnsn_amd.FusedConfig(sn_active=True, add_mode="pre", relu=True)
def block(strategy, grad=False):
    cnsn_amd.set_strategy(strategy)
    m = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(16), 3, torch.float32)).to(dev).train()
    xg = x.clone(memory_format=CL).requires_grad_(grad)
    y = m.forward_block(xg, b, add_mode="pre", relu=True)
    torch.cuda.synchronize()
    return y, xg
'''

_SN_SCRIPT = _HEAD + r'''
assert cnsn_amd.which_path(x, cfg) == "resident"
ref = block("two_pass")[0]
ok = block("auto")[0]
assert float((ok - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) and lib.cnsn_resident_timeouts() == 0
os.environ["CNSN_FAULT_INJECT"] = "1"          # the last workgroup never arrives at the first grid barrier
bad = block("auto")[0]                         # gives up after CNSN_WAIT_MS
os.environ["CNSN_FAULT_INJECT"] = "0"
assert lib.cnsn_resident_timeouts() == 1 and torch.isnan(bad).any()
try:
    block("auto")
    raise SystemExit("the time-out was not reported")
except cnsn_amd.CnsnError as e:
    assert "timed out" in str(e)
assert cnsn_amd.which_path(x, cfg) == "streaming"          # degraded: a launch per pass, right results
assert torch.equal(block("auto")[0], ref)
lib.cnsn_resident_rearm()
assert cnsn_amd.which_path(x, cfg) == "resident"
for _ in range(3):                                         # the barrier block was left short of arrivals: healed by the first launch
    again = block("auto")[0]
    assert float((again - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
assert lib.cnsn_resident_timeouts() == 1
# a backward whose forward ran BEFORE a degradation: the record is the slim one either way, the launch-per-pass kernels read it
y, xg = block("auto", grad=True)
os.environ["CNSN_FAULT_INJECT"] = "1"
block("auto")
os.environ["CNSN_FAULT_INJECT"] = "0"
assert lib.cnsn_resident_timeouts() == 2
with _ffi.deferred_timeouts():
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
assert torch.isfinite(xg.grad).all()
print("NHWC-FAULT-OK")
'''

_BN_SCRIPT = _HEAD + r'''
def tail(fused_expected, grad=True):
    cnsn_amd.set_strategy("auto")
    m = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(16), 3, torch.float32)).to(dev).train()
    bn = torch.nn.BatchNorm2d(16).to(dev).train()
    xg = x.clone(memory_format=CL).requires_grad_(grad)
    y = m.forward_bn_block(xg, bn, b, relu=True)
    torch.cuda.synchronize()
    assert (type(y.grad_fn).__name__ == "FusedBnBlockBackward") == fused_expected, type(y.grad_fn).__name__
    return y, xg, bn
y0, x0, _ = tail(True)
ref = y0.detach().clone()
os.environ["CNSN_FAULT_INJECT"] = "1"
bad, _, _ = tail(True)                                     # the fused tail gives up like every other persistent launch
os.environ["CNSN_FAULT_INJECT"] = "0"
assert lib.cnsn_resident_timeouts() == 1 and torch.isnan(bad).any()
with _ffi.deferred_timeouts():
    y1, x1, _ = tail(False)                                # degraded: BatchNorm2d, then the op — the same values to rounding
    assert float((y1 - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    y0.backward(torch.ones_like(y0))                       # the fused forward of BEFORE the degradation: its backward still runs
    torch.cuda.synchronize()
    assert torch.isfinite(x0.grad).all()
    y1.backward(torch.ones_like(y1))
    torch.cuda.synchronize()
    assert float((x0.grad - x1.grad).abs().max()) <= 1e-4 * max(1.0, float(x1.grad.abs().max()))
assert _ffi.poll_timeouts() == 1                           # (what a step guard does at the step boundary)
lib.cnsn_resident_rearm()
y2, _, _ = tail(True)
assert float((y2 - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
print("NHWC-FAULT-OK")
'''


@pytest.mark.parametrize("script", [_SN_SCRIPT, _BN_SCRIPT], ids=["selfnorm-block", "bn-block"])
@pytest.mark.parametrize("glue", ["0", "1"], ids=["glue", "ctypes"])
def test_a_grid_barrier_that_cannot_complete(script, glue):
    env = dict(os.environ, CNSN_WAIT_MS="200", CNSN_NO_GLUE=glue)
    r = subprocess.run([sys.executable, "-c", script % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "NHWC-FAULT-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
