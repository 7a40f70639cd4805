#!/usr/bin/env python3
"""Host cost of one op call at small shapes (the launch-bound sites of WideResNet): wall time per call of an un-synchronised
loop (the GPU work is shorter than the host work, so the loop runs at the host's pace), through the module surface with
the C++ glue, without it (CNSN_NO_GLUE=1 in a second process), and through the bare C ABI.  A measurement aid."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnsn_amd  # noqa: E402
from cnsn_amd import _ffi  # noqa: E402

dev = torch.device("cuda:0")


def per_call(fn, n=2000, warm=200):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t
    torch.cuda.synchronize()
    return dt / n * 1e6


print("glue:", _ffi.glue() is not None)
for shape in ((128, 32, 32, 32), (128, 64, 16, 16), (128, 128, 8, 8), (256, 2048, 7, 7)):
    x = torch.randn(shape, device=dev, requires_grad=True)
    gy = torch.randn(shape, device=dev)
    mod = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(shape[1])).to(dev).train()
    params = list(mod.parameters())
    f = per_call(lambda: mod(x))
    y = mod(x)
    fb = per_call(lambda: torch.autograd.grad(mod(x), [x] + params, gy))
    with torch.no_grad():
        nograd = per_call(lambda: mod(x))
    # a plain torch op of the same size for scale
    ref = per_call(lambda: torch.relu(x))
    print(f"{shape}: forward {f:.1f} us, forward+backward {fb:.1f} us, forward under no_grad {nograd:.1f} us, torch.relu {ref:.1f} us")

# the bare C ABI (ctypes call with pre-built arguments: ~2-3 us of ctypes marshalling included)
from cnsn_amd import functional as F  # noqa: E402
lib = _ffi.lib()
for shape in ((128, 32, 32, 32), (128, 64, 16, 16), (128, 128, 8, 8)):
    x = torch.randn(shape, device=dev)
    gy = torch.randn(shape, device=dev)
    sn = cnsn_amd.SelfNorm(shape[1]).to(dev).train()
    cfg = cnsn_amd.FusedConfig(sn_active=True, sn_training=True)
    prob = F._problem(x, cfg)
    F._context(prob, dev)
    g = F._GateBuffers(sn.g_fc.weight, sn.g_bn.weight, sn.g_bn.bias, sn.g_bn.running_mean, sn.g_bn.running_var)
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    saved_floats, ws_bytes = F._sizes(prob)[:2]
    saved = torch.empty(saved_floats, dtype=torch.float32, device=dev)
    ws = torch.empty(ws_bytes // 4 + 1, dtype=torch.float32, device=dev)
    stream = F._stream(x)
    args = (C.byref(prob), None, F._ptr(x), None, None, C.byref(g.c), None, F._ptr(y), F._ptr(saved), F._ptr(ws), ws_bytes, stream)
    f = per_call(lambda: lib.cnsn_forward_fused(*args))
    e = per_call(lambda: torch.empty_like(x))
    print(f"{shape}: cnsn_forward_fused through ctypes {f:.1f} us; torch.empty_like {e:.1f} us")
