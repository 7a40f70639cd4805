#!/usr/bin/env python3
"""tools/offset_probe.py: does the headline forward's time depend on WHERE its output lies relative to its input?

`bench.py` runs showed the forward launch at 0.30-0.31 ms in some processes and 0.34-0.35 ms in others on the same box, the
backward unchanged — the difference was what the caching allocator had handed out before.  Here x and y are carved out of ONE
buffer at a controlled distance (y = x + tensor bytes + delta) and the forward is called through the C ABI (ctypes,
cnsn_forward_fused; CrossNorm + SelfNorm at (256,256,56,56) fp32), HIP events around 20 launches per delta.

    python tools/offset_probe.py > profiles/rNN_offset_probe.md
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnsn_amd  # noqa: E402
from cnsn_amd import _ffi, functional as F  # noqa: E402

dev = torch.device("cuda:0")
shape = (256, 256, 56, 56)
n = 1
for s in shape:
    n *= s
nbytes = 4 * n
DELTAS = [0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20, 2 << 20, 3 << 20, 4 << 20,
          6 << 20, 8 << 20, 12544, 12544 * 256, 12544 * 256 + 4096, 3 * 4096, 5 * 4096, 7 * 8192, 100 * 4096]
MODE = sys.argv[1] if len(sys.argv) > 1 else "none"


def preroll(copies):
    a = torch.empty(n, device=dev)
    b = torch.empty(n, device=dev)
    c = torch.empty(n, device=dev)
    if copies:
        a.normal_()
        b.normal_()
        for _ in range(13):
            c.copy_(a)
            torch.add(a, b, alpha=2.0, out=c)
    torch.cuda.synchronize()
    del a, b, c


if MODE == "alloc_before":
    preroll(False)
if MODE == "copy_before":
    preroll(True)
if MODE != "none":
    DELTAS = DELTAS[:3]
pad = max(DELTAS) + (4 << 20)
big = torch.empty((2 * nbytes + pad) // 4, dtype=torch.float32, device=dev)
if MODE == "copy_after":
    preroll(True)
print("mode", MODE)
base = big.data_ptr()
x = big[:n].view(shape)
x.normal_()
sn = cnsn_amd.SelfNorm(shape[1]).to(dev).train()
cfg = cnsn_amd.FusedConfig(cn_active=True, sn_active=True, sn_training=True)
prob = F._problem(x, cfg)
F._context(prob, dev)
g = F._GateBuffers(sn.g_fc.weight, sn.g_bn.weight, sn.g_bn.bias, sn.g_bn.running_mean, sn.g_bn.running_var)
perm = torch.randperm(shape[0], device=dev)
saved_floats, ws_bytes = F._sizes(prob)[:2]
saved = torch.empty(saved_floats, dtype=torch.float32, device=dev)
ws = torch.empty(ws_bytes // 4 + 1, dtype=torch.float32, device=dev)
stream = F._stream(x)
lib = _ffi.lib()
print(f"x at {base:#x} (mod 2 MiB = {base % (2 << 20):#x}); tensor = {nbytes} B = {nbytes / (1 << 20):.3f} MiB")
print("| y - x - tensor bytes | (y - x) mod 4 KiB | mod 64 KiB | mod 2 MiB | forward ms |")
print("|---|---|---|---|---|")
for delta in DELTAS:
    off = nbytes + delta
    y = big[off // 4: off // 4 + n].view(shape)
    args = (C.byref(prob), None, F._ptr(x), F._ptr(perm), None, C.byref(g.c), None, F._ptr(y), F._ptr(saved), F._ptr(ws), ws_bytes, stream)
    for _ in range(5):
        st = lib.cnsn_forward_fused(*args)
    assert st == 0, st
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.cnsn_forward_fused(*args)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20
        best = t if best is None else min(best, t)
    d = y.data_ptr() - x.data_ptr()
    print(f"| {delta} | {d % 4096} | {d % 65536} | {d % (2 << 20)} | {best:.4f} |", flush=True)
