"""fp32 (256,1024,14,14) and bf16 (256,512,16,16) backward through the mono kernels: single phase, quarters (CNSN_MONO_RELOAD16=1),
and what AUTO runs (the cluster kernels where they are preferred)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnsn_amd
cnsn_amd.follow_environ()   # CNSN_* knobs are read at load: re-read after every change below
from tools.ab_sn_cluster import time_pair, cond
dev = torch.device("cuda:0")
for shape, dt in (((256, 1024, 14, 14), "f32"), ((256, 512, 16, 16), "bf16"), ((256, 512, 16, 16), "f32")):
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt]
    a = cond(shape, dtype, 1).requires_grad_()
    b = (cond(shape, dtype, 2) * 0.5).detach().requires_grad_()
    gy = torch.randn(shape, device=dev).to(dtype)
    mod = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(shape[1])).to(dev).train()
    for call in ("sn", "block"):
        ins = [a] + ([b] if call == "block" else []) + list(mod.parameters())
        fwd = (lambda: mod.forward_block(a, b, add_mode="pre", relu=True)) if call == "block" else (lambda: mod(a))
        bwd = lambda y: torch.autograd.grad(y, ins, gy)
        out = []
        for snx, r16 in (("0", "0"), ("0", "1"), ("1", "0")):
            os.environ["CNSN_SNX"] = snx
            os.environ["CNSN_MONO_RELOAD16"] = r16
            best = None
            for rep in range(3):
                f, bw = time_pair(fwd, bwd)
                if best is None or bw < best[1]: best = (f, bw)
            cfg = cnsn_amd.FusedConfig(sn_active=True, add_mode="pre" if call == "block" else "none", relu=call == "block")
            out.append(f"{best[0]:.4f}/{best[1]:.4f} ({cnsn_amd.which_path(a, cfg, True)})")
        print(f"{shape} {dt} {call}: mono single {out[0]} | mono quarters {out[1]} | AUTO {out[2]}", flush=True)
