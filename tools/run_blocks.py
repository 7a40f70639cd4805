#!/usr/bin/env python3
"""A few fused-block (add + SelfNorm + ReLU) and SelfNorm-only forward+backward calls at the ResNet-50 site shapes, for
`rocprofv3 --kernel-trace --stats` (kernel-only durations).  usage: run_blocks.py [bf16|f32] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnsn_amd  # noqa: E402

dev = torch.device("cuda:0")
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt]
for shape in ((256, 256, 56, 56), (256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7)):
    a = torch.randn(shape, device=dev).to(dtype).requires_grad_()
    b = torch.randn(shape, device=dev).to(dtype).requires_grad_()
    gy = torch.randn(shape, device=dev).to(dtype)
    mod = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(shape[1])).to(dev).train()
    ins = [a, b] + list(mod.parameters())
    for _ in range(reps):
        torch.autograd.grad(mod.forward_block(a, b, add_mode="pre", relu=True), ins, gy)
    for _ in range(reps):
        torch.autograd.grad(mod(a), [a] + list(mod.parameters()), gy)
    torch.cuda.synchronize()
    del a, b, gy, mod
