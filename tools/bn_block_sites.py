#!/usr/bin/env python3
"""The four ResNet-50 bottleneck tails (bn3 + add + SelfNorm + ReLU, bs 256, bf16, channels-last) through
`CNSN.forward_bn_block`: fused (one launch per direction) against the un-fused sequence (MIOpen's BatchNorm2d + the op's
launches); ms per forward / backward call, HIP events.  `site<i>` selects one; `down` adds the downsample's BatchNorm2d on the
skip path.  (measurement aid; profiles/r06_bn_block.md)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnsn_amd  # noqa: E402
from cnsn_amd import functional as F_  # noqa: E402

dev = torch.device("cuda:0")
CL = torch.channels_last
SITES = ((256, 256, 56, 56), (256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7))
only = [int(a[4:]) for a in sys.argv[1:] if a.startswith("site")]
down = "down" in sys.argv[1:]
modes = [m for m in ("fused", "unfused") if m in sys.argv[1:]] or ["fused", "unfused"]
print("| site | " + " | ".join(f"{m} fwd / bwd ms" for m in modes) + " | bytes x (MB) |")
print("|---|" + "---|" * (len(modes) + 1))
for shape in ([SITES[i] for i in only] or SITES):
    row = []
    for mode in modes:
        F_._BN_BLOCK = mode == "fused"
        c = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=CL).requires_grad_()
        b = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=CL).requires_grad_()
        gy = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=CL)
        m = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(shape[1])).to(dev).train()
        bn = torch.nn.BatchNorm2d(shape[1]).to(dev).train()
        bn2 = torch.nn.BatchNorm2d(shape[1]).to(dev).train() if down else None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for i in range(14):
            c.grad = b.grad = None
            ev[0].record()
            y = m.forward_bn_block(c, bn, b, relu=True, identity_bn=bn2)
            ev[1].record()
            y.backward(gy)
            ev[2].record()
            torch.cuda.synchronize()
            if i >= 4:
                tf += ev[0].elapsed_time(ev[1]) / 10
                tb += ev[1].elapsed_time(ev[2]) / 10
        row.append(f"{tf:.3f} / {tb:.3f}")
        del c, b, gy, y
    print(f"| {shape} | " + " | ".join(row) + f" | {shape[0] * shape[1] * shape[2] * shape[3] * 2 / 1e6:.0f} |")
