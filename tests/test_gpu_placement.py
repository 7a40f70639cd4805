"""cnsn_amd.placement: the caching allocator is left with output blocks that were measured (profiles/r04_memory_map.md)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402

DEV = torch.device("cuda:0")


_KEPT_BLOCKS = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
import cnsn_amd
DEV = torch.device("cuda:0")
x = torch.randn(64, 64, 56, 56, device=DEV)                     # 51 MB blocks
y = torch.empty_like(x)
assert cnsn_amd.placement.probe_write_ms(x, y) > 0.0
sn = cnsn_amd.SelfNorm(64).to(DEV).eval()
with torch.no_grad():                                           # the probe IS the inference launch: y holds its result
    sn.g_fc.weight.fill_(0.1)
    want = sn(x)
assert torch.allclose(y, want, rtol=1e-6, atol=1e-6)
rep = cnsn_amd.placement.prefer_fast_write_blocks(x, keep=2, candidates=10, min_gain=-1.0)   # (keep the two fastest whatever the spread)
assert rep["candidates"] == 10 and rep["kept"] == 2 and len(rep["kept_ptrs"]) == 2, rep
assert rep["probe_ms"]["min"] <= rep["probe_ms"]["median"] <= rep["probe_ms"]["max"]
a, b = torch.empty_like(x), torch.empty_like(x)
assert {hex(a.data_ptr()), hex(b.data_ptr())} == set(rep["kept_ptrs"]), (hex(a.data_ptr()), hex(b.data_ptr()), rep["kept_ptrs"])
small = torch.empty(1 << 20, device=DEV)                        # a smaller request does not carve up a kept block
del a, b
c = torch.empty_like(x)
assert hex(c.data_ptr()) in rep["kept_ptrs"]
print("ok")
"""


def test_probe_and_kept_blocks_are_what_the_allocator_hands_out_next():
    """in a process of its own: what torch's caching allocator hands out next depends on every block the process has ever
    cached (other tests' tensors), and this test is about the allocator's state after the search alone"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _KEPT_BLOCKS, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-1000:], r.stderr[-3000:])


def test_uniform_memory_leaves_the_allocator_alone():
    x = torch.randn(16, 16, 56, 56, device=DEV)
    rep = cnsn_amd.placement.prefer_fast_write_blocks(x, keep=2, candidates=6, min_gain=0.9)     # nothing is 90 % faster than the median
    assert rep["kept"] == 0 and rep["kept_ptrs"] == []


def test_results_do_not_depend_on_placement():
    torch.manual_seed(0)
    x = torch.randn(32, 16, 56, 56, device=DEV, requires_grad=True)
    gy = torch.randn_like(x)
    mod = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(16)).to(DEV).train()
    y0 = mod(x)
    g0, = torch.autograd.grad(y0, [x], gy)
    cnsn_amd.placement.prefer_fast_write_blocks(x.detach(), keep=3, candidates=8, min_gain=-1.0)
    mod2 = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(16)).to(DEV).train()
    mod2.load_state_dict({k: v for k, v in mod.state_dict().items()})
    mod2.selfnorm.g_bn.reset_running_stats()
    mod.selfnorm.g_bn.reset_running_stats()
    y1 = mod2(x)
    g1, = torch.autograd.grad(y1, [x], gy)
    assert torch.equal(mod(x), y1) and torch.allclose(g0, g1, rtol=0, atol=0)
