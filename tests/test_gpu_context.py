"""The persistent exchange context of the cluster-resident kernels (cnsn_context_init): values tagged with the launch
number instead of a fill launch in front of every resident launch.  Same bits with and without it, across shapes that
share one context, across the wrap-around of the launch counter, and under graph capture (where it is not used)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import os, sys, torch, numpy as np
sys.path.insert(0, %r)
import cnsn_amd
from cnsn_amd import functional as F
cnsn_amd.follow_environ()
from tests.golden.gen_golden_fill import fill_sn
dev = torch.device("cuda:0")
cnsn_amd.set_strategy("resident")
def run(shape, crop, seed):
    torch.manual_seed(seed); np.random.seed(seed)
    n, c = shape[:2]
    mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1), fill_sn(cnsn_amd.SelfNorm(c), 3, torch.float32)).to(dev).train()
    mod.crossnorm.active = True
    x = torch.randn(shape, device=dev, generator=torch.Generator(device=dev).manual_seed(seed)).requires_grad_()
    y = mod(x)
    y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    return [y.detach().clone(), x.grad.clone(), mod.selfnorm.g_fc.weight.grad.clone()]
SHAPES = [((37, 6, 56, 56), "neither"), ((20, 5, 28, 28), "both"), ((64, 3, 40, 40), "content"), ((37, 6, 56, 56), "both")]
outs = {}
for use in ("0", "1"):
    os.environ["CNSN_CONTEXT"] = use
    res = []
    for rep in range(3):                       # the same context serves every shape, forward and backward, repeatedly
        for i, (shape, crop) in enumerate(SHAPES):
            res.append(run(shape, crop, 10 * rep + i))
    outs[use] = res
assert len(F._contexts) == (1 if os.environ.get('CNSN_NO_GLUE') == '1' else 0), F._contexts.keys()   # (the C++ glue keeps its own)
for a, b in zip(outs["0"], outs["1"]):
    for u, v in zip(a, b):
        assert torch.equal(u, v)
assert cnsn_amd._ffi.lib().cnsn_resident_timeouts() == 0
print("CONTEXT-OK")
'''


@pytest.mark.parametrize("glue", ["0", "1"], ids=["glue", "ctypes"])
@pytest.mark.parametrize("epoch_start", [None, "0xfffffff0"], ids=["fresh", "wrap"])
def test_context_gives_the_same_bits(glue, epoch_start):
    env = dict(os.environ, CNSN_NO_GLUE=glue, CNSN_WAIT_MS="3000")
    if epoch_start:
        env["CNSN_EPOCH_START"] = epoch_start           # 16 launches before the counter wraps: the context is cleared once
    r = subprocess.run([sys.executable, "-c", _SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "CONTEXT-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_context_is_left_out_under_graph_capture():
    import cnsn_amd
    from tests.golden.gen_golden_fill import fill_sn
    dev = torch.device("cuda:0")
    cnsn_amd.set_strategy("resident")
    try:
        mod = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(8), 3, torch.float32)).to(dev).train()
        x = torch.randn(32, 8, 56, 56, device=dev)
        with torch.no_grad():
            want = mod(x).clone()              # (first call: creates the context)
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                mod(x)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                y = mod(x)
            for _ in range(3):                 # replays repeat the SAME launch: it must not depend on a launch number
                g.replay()
            torch.cuda.synchronize()
        assert torch.equal(y, want)
    finally:
        cnsn_amd.set_strategy("auto")
