#!/usr/bin/env python3
"""The four ResNet-50 block sites (add + SelfNorm + ReLU, bs 256, bf16) in NCHW (cluster kernels) and channels-last (two-pass NHWC
kernels): ms per forward + backward call through the module surface, HIP events.  (measurement aid; profiles/r05_nhwc.md)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnsn_amd  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
print(f"| site | NCHW fwd / bwd ms | channels-last fwd / bwd ms | bytes x (MB) |")
print("|---|---|---|---|")
SITES = ((256, 256, 56, 56), (256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7))
only = [int(a[4:]) for a in sys.argv[2:] if a.startswith("site")]
for shape in ([SITES[i] for i in only] or SITES):
    row = []
    for fmt in ((torch.channels_last,) if "cl" in sys.argv[2:] else (torch.contiguous_format, torch.channels_last)):
        x = torch.randn(shape, device=dev).to(dt).contiguous(memory_format=fmt).requires_grad_()
        b = torch.randn(shape, device=dev).to(dt).contiguous(memory_format=fmt).requires_grad_()
        gy = torch.randn(shape, device=dev).to(dt).contiguous(memory_format=fmt)
        m = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(shape[1])).to(dev).train()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for i in range(14):
            x.grad = b.grad = None
            ev[0].record()
            y = m.forward_block(x, b, add_mode="pre", relu=True)
            ev[1].record()
            y.backward(gy)
            ev[2].record()
            torch.cuda.synchronize()
            if i >= 4:
                tf += ev[0].elapsed_time(ev[1]) / 10
                tb += ev[1].elapsed_time(ev[2]) / 10
        row.append(f"{tf:.3f} / {tb:.3f}")
        del x, b, gy, y
    if len(row) < 2:
        row.insert(0, "-")
    print(f"| {shape} | {row[0]} | {row[1]} | {shape[0] * shape[1] * shape[2] * shape[3] * (2 if dt == torch.bfloat16 else 4) / 1e6:.0f} |")
