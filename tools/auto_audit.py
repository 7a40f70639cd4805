#!/usr/bin/env python3
"""tools/auto_audit.py: is CNSN_STRATEGY_AUTO the fastest choice ON THIS BOX?

The AUTO rules of the library (which kernel family takes a call) were derived from sweeps on the boxes of earlier rounds, and
the boxes of the pool differ by 8-20 % in what their memory system delivers (round-3 review, weak #7: "nothing verifies they
are the fastest on the driver's box").  This runs every shape of BASELINE.json's configs x dtype x mode under AUTO and under
every forced alternative (strategy, and the pipelined / SelfNorm-only cluster kernels switched off or forced on), forward +
backward through the module surface, and prints one row per case: what AUTO resolves to, its time, the best alternative and
the ratio.  Rows where an alternative beats AUTO by more than 3 % are flagged: those are the rules to revisit.

    python tools/auto_audit.py [--quick] > profiles/rNN_auto_audit.md
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnsn_amd  # noqa: E402

cnsn_amd.follow_environ()
dev = torch.device("cuda:0")

SHAPES = [(128, 32, 32, 32), (128, 64, 16, 16), (128, 128, 8, 8),                                   # configs[1] WideResNet-40-2
          (256, 256, 56, 56), (256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7),           # configs[2] ResNet-50
          (96, 256, 56, 56), (96, 512, 28, 28), (96, 1024, 14, 14), (96, 2048, 7, 7),               # configs[3] 3 x 32 views
          (16, 256, 128, 128), (16, 512, 64, 64), (16, 2048, 64, 64),                               # configs[4] segmentation
          (256, 3, 224, 224)]                                                                       # image-space CrossNorm
MODES = [("sn", "neither", False), ("sn-block", "neither", True), ("cnsn", "neither", False), ("cnsn", "both", False),
         ("cn", "style", False)]
ALTS = [("two_pass", {}), ("resident", {}), ("local", {}), ("mono", {}), ("auto", {"CNSN_PIPE": "0"}), ("auto", {"CNSN_PIPE": "2"}),
        ("resident", {"CNSN_PIPE": "2"}), ("auto", {"CNSN_SNX": "0"}), ("auto", {"CNSN_SNX": "2"}), ("auto", {"CNSN_SNXCN": "0"}),
        ("auto", {"CNSN_SNXCN": "2"})]


def timeit(fn, calls, reps):
    best = None
    for _ in range(reps):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(calls):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / calls * 1e3
        best = dt if best is None else min(best, dt)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    calls, reps = (10, 2) if args.quick else (25, 3)
    print("| shape | dtype | mode | AUTO runs | AUTO ms | best alternative | its ms | AUTO / best | |")
    print("|---|---|---|---|---|---|---|---|---|")
    flagged = 0
    rows = 0
    for shape in SHAPES:
        for dt, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            x = torch.randn(shape, device=dev).to(dtype).requires_grad_()
            idt = torch.randn(shape, device=dev).to(dtype).requires_grad_()
            gy = torch.randn(shape, device=dev).to(dtype)
            for kind, crop, block in MODES:
                if shape[1] == 3 and kind != "cn":
                    continue
                if kind == "cn" and shape[1] != 3 and shape[0] != 16:
                    continue                                   # CrossNorm alone: image space and the segmentation sites
                base = kind.split("-")[0]
                mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1) if base != "sn" else None,
                                    cnsn_amd.SelfNorm(shape[1]) if base != "cn" else None).to(dev).train()
                ins = [x] + ([idt] if block else []) + list(mod.parameters())

                def run():
                    if mod.crossnorm is not None:
                        mod.crossnorm.active = True
                    y = mod.forward_block(x, idt, add_mode="pre", relu=True) if block else mod(x)
                    torch.autograd.grad(y, ins, gy)

                cfg = cnsn_amd.FusedConfig(cn_active=base != "sn", sn_active=base != "cn",
                                           content_box=(1, 1, 3, 3) if crop == "both" else None,
                                           style_box=(0, 0, 2, 2) if crop in ("both", "style") else None,
                                           add_mode="pre" if block else "none", relu=block)

                def paths():
                    return cnsn_amd.which_path(x, cfg, False)[:3] + "/" + cnsn_amd.which_path(x, cfg, True)[:3]

                cnsn_amd.set_strategy("auto")
                auto_paths = paths()
                t_auto = timeit(run, calls, reps)
                best_name, best_t = None, None
                for strat, env in ALTS:
                    for k, v in env.items():
                        os.environ[k] = v
                    cnsn_amd.set_strategy(strat)
                    same = (paths() == auto_paths and not env)   # a forced strategy that resolves to AUTO's kernels: not an alternative
                    try:
                        if not same:
                            t = timeit(run, calls, max(1, reps - 1))
                            if best_t is None or t < best_t:
                                best_name = strat + ("".join(f" {k}={v}" for k, v in env.items())) + " " + paths()
                                best_t = t
                    finally:
                        for k in env:
                            os.environ.pop(k, None)
                        cnsn_amd.set_strategy("auto")
                ratio = t_auto / best_t if best_t else 1.0
                flag = "<- revisit" if ratio > 1.03 else ""
                flagged += bool(flag)
                rows += 1
                print(f"| {shape} | {dt} | {kind}/{crop} | {auto_paths} | {t_auto:.4f} | {best_name} | {best_t:.4f} | {ratio:.3f} | {flag} |",
                      flush=True)
            del x, idt, gy
    print(f"\n{rows} cases, {flagged} where an alternative beats AUTO by more than 3 % on this box "
          f"({torch.cuda.get_device_name(0)}; best of {reps} x {calls} calls per timing, forward + backward, host clock around a "
          "synchronised loop).")


if __name__ == "__main__":
    main()
