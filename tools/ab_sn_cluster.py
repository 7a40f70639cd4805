#!/usr/bin/env python3
"""A/B of the SelfNorm-only cluster kernels (CNSN_SNX=1, csrc/cnsn_resident_sn_kernels.h) against the general resident
kernels (CNSN_SNX=0) on the shapes the ResNet-50 / WideResNet / segmentation configurations put through SelfNorm alone:
forward and backward timed separately with HIP events through the module surface, both sides in the same process
(interleaved).  Prints a markdown table (profiles/r03_sn_cluster.md).  A measurement aid, not the bench contract."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnsn_amd  # noqa: E402
cnsn_amd.follow_environ()   # CNSN_* knobs are read at load: re-read after every change below

dev = torch.device("cuda:0")


def cond(shape, dtype, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    n, c = shape[:2]
    x = torch.randn(shape, generator=g, device=dev)
    x.mul_(torch.rand(n, c, 1, 1, generator=g, device=dev) * 1.5 + 0.5).add_(torch.randn(n, c, 1, 1, generator=g, device=dev))
    return x.to(dtype)


def time_pair(fwd, bwd, k=30, w=6):
    """ms of fwd() and of bwd(y) separately (events around each call)"""
    for _ in range(w):
        bwd(fwd())
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(k)]
    for i in range(k):
        ev[i][0].record()
        y = fwd()
        ev[i][1].record()
        bwd(y)
        ev[i][2].record()
    torch.cuda.synchronize()
    f = sorted(a.elapsed_time(b) for a, b, _ in ev)
    b_ = sorted(b.elapsed_time(c) for _, b, c in ev)
    return f[len(f) // 2], b_[len(b_) // 2]


if __name__ == "__main__":
    CASES = [((256, 256, 56, 56), "bf16"), ((256, 512, 28, 28), "bf16"), ((96, 256, 56, 56), "bf16"), ((96, 512, 28, 28), "bf16"),
             ((256, 256, 56, 56), "f32"), ((256, 512, 28, 28), "f32"), ((128, 32, 32, 32), "f32"), ((16, 512, 64, 64), "bf16"),
             ((16, 512, 64, 64), "f32"), ((128, 256, 40, 40), "bf16")]
    if len(sys.argv) > 1 and sys.argv[1] == "short":
        CASES = CASES[:2] + CASES[4:5]
    # side "0": what AUTO runs with the family off; side "1": the family (forced where AUTO prefers another strategy)
    SIDES = {"0": ("auto", "0"), "1": ("auto", "1")}
    if len(sys.argv) > 1 and sys.argv[1] == "forced":   # the classes AUTO leaves to the general kernels, forced for comparison
        CASES = [((256, 256, 56, 56), "bf16"), ((96, 256, 56, 56), "bf16"), ((256, 512, 28, 28), "f32"), ((128, 32, 32, 32), "f32"),
                 ((16, 512, 64, 64), "f32"), ((64, 64, 112, 112), "bf16")]
        SIDES = {"0": ("auto", "0"), "1": ("resident", "2")}
    if len(sys.argv) > 1 and sys.argv[1] == "small":   # one-slot planes: against the channel-in-registers (mono) kernels
        CASES = [((256, 1024, 14, 14), "bf16"), ((256, 1024, 14, 14), "f32"), ((96, 1024, 14, 14), "bf16"),
                 ((128, 1024, 14, 14), "bf16"), ((512, 1024, 14, 14), "bf16")]
        SIDES = {"0": ("auto", "0"), "1": ("resident", "2")}

    print("| shape | dtype | call | SNX=0 fwd / bwd ms | SNX=1 fwd / bwd ms | fwd | bwd | of 8 TB/s (new, fwd / bwd) |")
    print("|---|---|---|---|---|---|---|---|")
    for shape, dt in CASES:
        dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt]
        e = 1
        for v in shape:
            e *= v
        eb = e * (2 if dt == "bf16" else 4)
        a = cond(shape, dtype, 1).requires_grad_()
        b = (cond(shape, dtype, 2) * 0.5).detach().requires_grad_()
        gy = torch.randn(shape, device=dev).to(dtype)
        mod = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(shape[1])).to(dev).train()
        for call, passes_f, passes_b in (("sn", 2, 3), ("block", 3, 4)):
            ins = [a] + ([b] if call == "block" else []) + list(mod.parameters())
            fwd = (lambda: mod.forward_block(a, b, add_mode="pre", relu=True)) if call == "block" else (lambda: mod(a))
            bwd = lambda y: torch.autograd.grad(y, ins, gy)  # noqa: E731
            res = {}
            for rep in range(2):                    # interleave the two sides twice, keep the better of each
                for snx in ("0", "1"):
                    cnsn_amd.set_strategy(SIDES[snx][0])
                    os.environ["CNSN_SNX"] = SIDES[snx][1]
                    f, bw = time_pair(fwd, bwd)
                    if snx not in res or f + bw < sum(res[snx]):
                        res[snx] = (f, bw)
            cfg = cnsn_amd.FusedConfig(sn_active=True, add_mode="pre" if call == "block" else "none", relu=call == "block")
            cnsn_amd.set_strategy(SIDES["1"][0])
            os.environ["CNSN_SNX"] = SIDES["1"][1]
            took = cnsn_amd.sn_cluster(a, cfg), cnsn_amd.sn_cluster(a, cfg, backward=True)
            (f0, b0), (f1, b1) = res["0"], res["1"]
            print(f"| {shape} | {dt} | {call}{'' if all(took) else ' (not taken: ' + str(took) + ')'} | {f0:.4f} / {b0:.4f} | {f1:.4f} / {b1:.4f} | "
                  f"{(f1 / f0 - 1) * 100:+.1f} % | {(b1 / b0 - 1) * 100:+.1f} % | "
                  f"{passes_f * eb / f1 / 1e6 / 8000:.3f} / {passes_b * eb / b1 / 1e6 / 8000:.3f} |", flush=True)
        del a, b, gy, mod
    os.environ.pop("CNSN_SNX", None)
