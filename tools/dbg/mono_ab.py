"""Forward / backward times of the small-plane sites (14x14, 7x7) through the module surface, for A/B runs of tuning builds
(CNSN_LIB_PATH=tools/dbg/lib_<variant>.so): profiles/r03_sn_cluster.md section 8."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnsn_amd
from tools.ab_sn_cluster import time_pair, cond
dev = torch.device("cuda:0")
for shape, dt in (((256, 1024, 14, 14), "bf16"), ((256, 1024, 14, 14), "f32"), ((128, 1024, 14, 14), "bf16"), ((256, 2048, 7, 7), "bf16")):
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt]
    a = cond(shape, dtype, 1).requires_grad_()
    b = (cond(shape, dtype, 2) * 0.5).detach().requires_grad_()
    gy = torch.randn(shape, device=dev).to(dtype)
    mod = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(shape[1])).to(dev).train()
    for call in ("sn", "block"):
        ins = [a] + ([b] if call == "block" else []) + list(mod.parameters())
        fwd = (lambda: mod.forward_block(a, b, add_mode="pre", relu=True)) if call == "block" else (lambda: mod(a))
        bwd = lambda y: torch.autograd.grad(y, ins, gy)
        best = None
        for rep in range(3):
            f, bw = time_pair(fwd, bwd)
            if best is None or f + bw < sum(best): best = (f, bw)
        print(f"{shape} {dt} {call}: {best[0]:.4f} / {best[1]:.4f}", flush=True)
