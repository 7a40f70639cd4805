#!/usr/bin/env python3
"""tools/mall_probe.py: what the 256 MB Infinity Cache gives the SECOND pass of a two-pass schedule (round-3 review, item 7:
"measure, don't argue, the cached second pass for image-space CrossNorm").

For image-space tensors (N,3,224,224) of growing size: the library's apply kernel (`cnsn_plane_affine`: one read of x, one
write of y, the second pass of the two-pass forward) timed with HIP events
  cold : after 2 GB of unrelated traffic (nothing of x on chip),
  warm : directly behind a full read of the same tensor with the DEFAULT cache policy (`torch.sum(x)`: what the first pass of
         the library's two-pass schedule does for tensors of at most 512 MiB, `Geom::keep`; the stand-alone statistics
         entry point reads non-temporally and would leave nothing behind).
The difference is what the cache delivers; rocprofv3's FETCH_SIZE cannot show it (Infinity-Cache hits are counted as
fetches, MI355X_MICROARCH.md).  Also the statistics kernel itself, cold, as the HBM read rate of the same box."""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnsn_amd  # noqa: E402

dev = torch.device("cuda:0")
junk_a = torch.empty(1 << 29, device=dev)          # 2 GB
junk_b = torch.empty(1 << 29, device=dev)


def evict():
    junk_b.copy_(junk_a)


def timed(fn, pre, reps=7):
    out = []
    for _ in range(reps):
        pre()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1))
    return sorted(out)[len(out) // 2]


print("| shape | dtype | MB | stats cold (1 read) | apply cold (read+write) | apply behind a default-policy read | apply warm / cold | two-pass forward, GB/s on 3*E*b |")
print("|---|---|---|---|---|---|---|---|")
for dtype in (torch.float32, torch.bfloat16):
    for n in (32, 64, 128, 192, 256, 384, 512, 768):
        shape = (n, 3, 224, 224)
        x = torch.randn(shape, device=dev).to(dtype)
        mb = x.numel() * x.element_size() / 2 ** 20
        scale = torch.rand(n, 3, 1, 1, device=dev) + 0.5
        shift = torch.randn(n, 3, 1, 1, device=dev)
        stats = lambda: cnsn_amd.calc_ins_mean_std(x)                                   # noqa: E731
        apply_ = lambda: cnsn_amd.functional.PlaneAffine.apply(x, scale, shift)        # noqa: E731
        with torch.no_grad():
            for _ in range(3):
                stats(), apply_()
            t_stats = timed(stats, evict)
            t_cold = timed(apply_, evict)
            t_warm = timed(apply_, lambda: (evict(), torch.sum(x)))
        eb = x.numel() * x.element_size()
        print(f"| {shape} | {str(dtype)[6:]} | {mb:.0f} | {t_stats * 1e3:.1f} us = {eb / t_stats / 1e6:.0f} GB/s | "
              f"{t_cold * 1e3:.1f} us = {2 * eb / t_cold / 1e6:.0f} GB/s | {t_warm * 1e3:.1f} us = {2 * eb / t_warm / 1e6:.0f} GB/s | "
              f"{t_warm / t_cold:.2f} | {3 * eb / (t_stats + t_warm) / 1e6:.0f} (cold second pass: {3 * eb / (t_stats + t_cold) / 1e6:.0f}) |", flush=True)
        del x
