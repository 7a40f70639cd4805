"""Same-process forward / backward times at the north-star shape: CrossNorm+SelfNorm (general pipelined kernels), SelfNorm alone
(partial-moment cluster kernels) and SelfNorm alone through the general kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnsn_amd
cnsn_amd.follow_environ()   # CNSN_* knobs are read at load: re-read after every change below
from tools.ab_sn_cluster import time_pair, cond
dev = torch.device("cuda:0")
for dt in ("f32", "bf16"):
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt]
    shape = (256, 256, 56, 56)
    x = cond(shape, dtype, 1).requires_grad_()
    gy = torch.randn(shape, device=dev).to(dtype)
    cnsn = cnsn_amd.CNSN(cnsn_amd.CrossNorm("neither", 1), cnsn_amd.SelfNorm(256)).to(dev).train()
    sn = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(256)).to(dev).train()
    def f_cnsn():
        cnsn.crossnorm.active = True
        return cnsn(x)
    res = {}
    for rep in range(3):
        for name, fwd, mod in (("cnsn", f_cnsn, cnsn), ("sn", lambda: sn(x), sn)):
            ins = [x] + list(mod.parameters())
            f, b = time_pair(fwd, lambda y: torch.autograd.grad(y, ins, gy))
            if name not in res or f + b < sum(res[name]): res[name] = (f, b)
        os.environ["CNSN_SNX"] = "0"
        ins = [x] + list(sn.parameters())
        f, b = time_pair(lambda: sn(x), lambda y: torch.autograd.grad(y, ins, gy))
        os.environ.pop("CNSN_SNX")
        if "sn_general" not in res or f + b < sum(res["sn_general"]): res["sn_general"] = (f, b)
    for k, v in res.items():
        print(f"{dt} {k}: fwd {v[0]:.4f} bwd {v[1]:.4f}", flush=True)
