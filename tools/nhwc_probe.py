#!/usr/bin/env python3
"""What would a channels-last variant of the op be worth to ResNet-50?  (measurement aid; result: profiles/r05_nhwc_probe.md)

The backbone WITHOUT its CNSN units (`cnsn_type=None`: convolutions, BatchNorm2d, ReLU, add — torch / MIOpen only), bs 256, bf16
autocast, forward + backward + SGD, in torch.contiguous_format and torch.channels_last: if MIOpen's NHWC convolutions do not beat
its NCHW ones on this stack, an NHWC statistics / apply pair for the op (review item 10) has nothing to unlock; if they do, the
difference is the budget the op's 16 sites (4.6 ms in NCHW) would have to stay inside, layout changes included."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cnsn_amd.callers import ResNet50CNSN  # noqa: E402

dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for fmt_name, fmt in (("NCHW", torch.contiguous_format), ("NHWC", torch.channels_last)):
    torch.manual_seed(0)
    net = ResNet50CNSN(num_classes=1000, cnsn_type=None).to(dev).to(memory_format=fmt).train()
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    x = torch.randn(bs, 3, 224, 224, device=dev).contiguous(memory_format=fmt)
    y = torch.randint(0, 1000, (bs,), device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(net(x).float(), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    warm = time.perf_counter() - t0
    t0 = time.perf_counter()
    k = 12
    for _ in range(k):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / k * 1e3
    print(f"{fmt_name}: {ms:.2f} ms per step = {bs / ms * 1e3:.0f} img/s (bs {bs}, bf16 autocast, no CNSN units; warm-up incl. MIOpen search {warm:.0f} s)", flush=True)
    del net, opt, x, y
    torch.cuda.empty_cache()
