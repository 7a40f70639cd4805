#!/usr/bin/env python3
"""tools/placement_probe.py: the headline forward through the C ABI on EIGHT different (x, y) buffer pairs of one process
(fresh allocations, then the same blocks with the roles of x and y swapped): is the launch time a property of the process or
of WHERE the tensors lie?  (profiles/r04_placement_sensitivity.md)"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnsn_amd
from cnsn_amd import _ffi, functional as F
dev = torch.device("cuda:0")
shape = (256, 256, 56, 56)
n = 256*256*56*56
sn = cnsn_amd.SelfNorm(shape[1]).to(dev).train()
cfg = cnsn_amd.FusedConfig(cn_active=True, sn_active=True, sn_training=True)
lib = _ffi.lib()
keep = []
perm = torch.randperm(shape[0], device=dev)
def timeit(args):
    for _ in range(5): assert lib.cnsn_forward_fused(*args) == 0
    torch.cuda.synchronize()
    best=None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): lib.cnsn_forward_fused(*args)
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/20; best = t if best is None else min(best,t)
    return best
for rep in range(8):
    x = torch.empty(shape, device=dev).normal_()
    y = torch.empty(shape, device=dev)
    prob = F._problem(x, cfg); F._context(prob, dev)
    g = F._GateBuffers(sn.g_fc.weight, sn.g_bn.weight, sn.g_bn.bias, sn.g_bn.running_mean, sn.g_bn.running_var)
    saved_floats, ws_bytes = F._sizes(prob)[:2]
    saved = torch.empty(saved_floats, dtype=torch.float32, device=dev)
    ws = torch.empty(ws_bytes // 4 + 1, dtype=torch.float32, device=dev)
    stream = F._stream(x)
    args = (C.byref(prob), None, F._ptr(x), F._ptr(perm), None, C.byref(g.c), None, F._ptr(y), F._ptr(saved), F._ptr(ws), ws_bytes, stream)
    t1 = timeit(args)
    t2 = timeit(args)
    print(rep, hex(x.data_ptr()), hex(y.data_ptr()), hex(saved.data_ptr()), round(t1,4), round(t2,4), flush=True)
    keep.append((x,y,saved,ws))
    if rep == 3:
        keep.clear()   # free: the next allocations reuse cached blocks
