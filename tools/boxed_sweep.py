#!/usr/bin/env python3
"""tools/boxed_sweep.py: forward and backward times of calls WITH crop boxes under every kernel family, per direction.

tools/auto_audit.py times forward + backward together and prints the best alternative only; the AUTO rules are per direction
and per plane class, so this prints every alternative for the crop-box cases (56x56 / 64x64 / 128x128 planes over a range of
batch sizes): the table the boxed rules in csrc/cnsn_resident_host.h / cnsn_resident_pipe.hip / cnsn_resident_sn_host.h cite.

    python tools/boxed_sweep.py > profiles/rNN_boxed_sweep.md
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnsn_amd  # noqa: E402

cnsn_amd.follow_environ()
dev = torch.device("cuda:0")
SHAPES = [(256, 256, 56, 56), (192, 256, 56, 56), (128, 256, 56, 56), (96, 256, 56, 56), (32, 256, 56, 56),
          (256, 128, 64, 64), (64, 256, 64, 64), (16, 512, 64, 64), (16, 2048, 64, 64), (16, 256, 128, 128), (64, 64, 128, 128),
          (256, 512, 28, 28), (96, 512, 28, 28)]
MODES = [("cnsn", "both"), ("cnsn", "content"), ("cn", "style")]
ALTS = [("auto", {}), ("two_pass", {}), ("resident", {"CNSN_PIPE": "0"}), ("resident", {"CNSN_PIPE": "2"}),
        ("resident", {"CNSN_PIPE": "0", "CNSN_SNXCN": "0"}), ("resident", {"CNSN_PIPE": "2", "CNSN_SNXCN": "0"}),
        ("resident", {"CNSN_PIPE": "0", "CNSN_SNXCN": "2"})]


def timeit(fn, calls=20, reps=3):
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
    print("| shape | dtype | mode | " + " | ".join(s + "".join(f" {k[5:]}={v}" for k, v in e.items()) for s, e in ALTS) + " |")
    print("|---|---|---|" + "---|" * len(ALTS))
    for shape in SHAPES:
        for dt, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            x = torch.randn(shape, device=dev).to(dtype).requires_grad_()
            gy = torch.randn(shape, device=dev).to(dtype)
            for kind, crop in MODES:
                mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1), cnsn_amd.SelfNorm(shape[1]) if kind != "cn" else None).to(dev).train()
                ins = [x] + list(mod.parameters())
                cfg = cnsn_amd.FusedConfig(cn_active=True, sn_active=kind != "cn",
                                           content_box=(1, 1, 3, 3) if crop in ("both", "content") else None,
                                           style_box=(0, 0, 2, 2) if crop in ("both", "style") else None)
                cells = []
                for strat, env in ALTS:
                    for k, v in env.items():
                        os.environ[k] = v
                    cnsn_amd.set_strategy(strat)
                    try:
                        def fwd():
                            mod.crossnorm.active = True
                            with torch.no_grad():
                                return mod(x)

                        mod.crossnorm.active = True
                        y = mod(x)

                        def bwd():
                            torch.autograd.grad(y, ins, gy, retain_graph=True)

                        p = cnsn_amd.which_path(x, cfg, False)[:3] + "/" + cnsn_amd.which_path(x, cfg, True)[:3]
                        cells.append(f"{p} {timeit(fwd):.4f} / {timeit(bwd):.4f}")
                        del y
                    finally:
                        for k in env:
                            os.environ.pop(k, None)
                        cnsn_amd.set_strategy("auto")
                print(f"| {shape} | {dt} | {kind}/{crop} | " + " | ".join(cells) + " |", flush=True)
            del x, gy
    print(f"\nforward (no grad) / backward (autograd.grad on a retained graph) ms per call, best of 3 x 20 calls, host clock around a "
          f"synchronised loop; {torch.cuda.get_device_name(0)}.")


if __name__ == "__main__":
    main()
