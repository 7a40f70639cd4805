#!/usr/bin/env python3
"""A/B of CrossNorm(crop)+SelfNorm calls at (256,256,56,56): what AUTO runs against the pipelined cluster kernels forced
(CNSN_PIPE=2 + strategy resident), forward and backward timed separately with HIP events, same process, interleaved.
A measurement aid, not the bench contract."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnsn_amd  # noqa: E402
cnsn_amd.follow_environ()   # CNSN_* knobs are read at load: re-read after every change below
from tools.ab_sn_cluster import time_pair  # noqa: E402  (prints its own table first when imported: run with `none`)

dev = torch.device("cuda:0")
shape = (256, 256, 56, 56)
print("| dtype | crop | AUTO fwd / bwd ms | forced pipelined fwd / bwd ms | fwd | bwd |")
print("|---|---|---|---|---|---|")
for dt in ("f32", "bf16"):
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt]
    for crop in ("both", "content", "style", "neither"):
        x = torch.randn(shape, device=dev).to(dtype).requires_grad_()
        gy = torch.randn(shape, device=dev).to(dtype)
        mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1), cnsn_amd.SelfNorm(shape[1])).to(dev).train()
        mod.crossnorm.active = True
        ins = [x] + list(mod.parameters())

        def fwd():
            mod.crossnorm.active = True
            return mod(x)
        bwd = lambda y: torch.autograd.grad(y, ins, gy)  # noqa: E731
        res = {}
        for rep in range(2):
            for side, (strat, pipe) in {"0": ("auto", "1"), "1": ("resident", "2")}.items():
                cnsn_amd.set_strategy(strat)
                os.environ["CNSN_PIPE"] = pipe
                try:
                    f, b = time_pair(fwd, bwd)
                except Exception as e:  # noqa: BLE001
                    f = b = float("nan")
                    print("  (", dt, crop, side, "failed:", str(e)[:80], ")")
                if side not in res or f + b < sum(res[side]):
                    res[side] = (f, b)
        (f0, b0), (f1, b1) = res["0"], res["1"]
        print(f"| {dt} | {crop} | {f0:.4f} / {b0:.4f} | {f1:.4f} / {b1:.4f} | {(f1 / f0 - 1) * 100:+.1f} % | {(b1 / b0 - 1) * 100:+.1f} % |", flush=True)
        del x, gy, mod
cnsn_amd.set_strategy("auto")
