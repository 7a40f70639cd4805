#!/usr/bin/env python3
"""Forward+backward calls of the kernels the review named as weak, for rocprofv3 (kernel stats / PMC passes):
    run_cases.py <case> [reps]
      boxed_f32, boxed_bf16, neither_bf16, neither_f32 : CrossNorm(crop)+SelfNorm at (256,256,56,56)
      block_bf16, block_f32                             : add + SelfNorm + ReLU at the four ResNet-50 site shapes
      sn_bf16                                           : SelfNorm alone at 56x56 and 28x28"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnsn_amd  # noqa: E402

dev = torch.device("cuda:0")
case = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dtype = torch.bfloat16 if case.endswith("bf16") else torch.float32
np.random.seed(3)
torch.manual_seed(3)


def fb(mod, x, gy, b=None):
    if mod.crossnorm is not None:
        mod.crossnorm.active = True
    y = mod.forward_block(x, b, add_mode="pre", relu=True) if b is not None else mod(x)
    torch.autograd.grad(y, [x] + ([b] if b is not None else []) + list(mod.parameters()), gy)


if case.startswith(("boxed", "neither")):
    shape = (256, 256, 56, 56)
    crop = "both" if case.startswith("boxed") else "neither"
    x = torch.randn(shape, device=dev).to(dtype).requires_grad_()
    gy = torch.randn(shape, device=dev).to(dtype)
    mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1), cnsn_amd.SelfNorm(256)).to(dev).train()
    for _ in range(reps):
        fb(mod, x, gy)
else:
    shapes = ((256, 256, 56, 56), (256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7))
    if case.startswith("sn"):
        shapes = shapes[:2]
    for shape in shapes:
        x = torch.randn(shape, device=dev).to(dtype).requires_grad_()
        b = torch.randn(shape, device=dev).to(dtype).requires_grad_() if case.startswith("block") else None
        gy = torch.randn(shape, device=dev).to(dtype)
        mod = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(shape[1])).to(dev).train()
        for _ in range(reps):
            fb(mod, x, gy, b)
        del x, b, gy, mod
torch.cuda.synchronize()
