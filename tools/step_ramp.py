#!/usr/bin/env python3
"""tools/step_ramp.py: how the time of the headline step (fused CNSN fwd+bwd, (256,256,56,56) fp32) evolves over the first
steps of a process — chunks of `chunk` steps, each bracketed by a device synchronisation.  Answers what a 20-step timed
region after 5 warm-up steps (the driver's bench call) measures compared with a long run."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import cnsn_amd  # noqa: E402

chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 5
total = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
shape = (256, 256, 56, 56)
x = torch.randn(shape, device=dev).requires_grad_()
gy = torch.randn(shape, device=dev)
mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm("neither", 1), cnsn_amd.SelfNorm(256)).to(dev).train()


def step():
    mod.crossnorm.active = True
    x.grad = None
    for p in mod.parameters():
        p.grad = None
    mod(x).backward(gy)


torch.cuda.synchronize()
t_start = time.perf_counter()
out = []
for c in range(total // chunk):
    t0 = time.perf_counter()
    for _ in range(chunk):
        step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out.append((t0 - t_start, (t1 - t0) / chunk * 1e3))
for i, (at, ms) in enumerate(out):
    if i < 12 or i % 10 == 0:
        print(f"steps {i * chunk:4d}-{(i + 1) * chunk - 1:4d}  at {at * 1e3:7.1f} ms  {ms:.4f} ms/step")
