"""(256,2048,7,7) with CrossNorm armed (no crop boxes): the channel-group kernels against what ran before (CNSN_WIDE=0: the
packed two-pass kernels), forward / backward ms, SelfNorm behind CrossNorm alone and inside the residual-block epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnsn_amd
cnsn_amd.follow_environ()   # CNSN_* knobs are read at load: re-read after every change below
from tools.ab_sn_cluster import time_pair, cond
dev = torch.device("cuda:0")
for dt in ("bf16", "f32"):
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32}[dt]
    shape = (256, 2048, 7, 7)
    x = cond(shape, dtype, 1).requires_grad_()
    b = (cond(shape, dtype, 2) * 0.5).detach().requires_grad_()
    gy = torch.randn(shape, device=dev).to(dtype)
    mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm("neither", 1), cnsn_amd.SelfNorm(shape[1])).to(dev).train()
    for call in ("cnsn", "block"):
        ins = [x] + ([b] if call == "block" else []) + list(mod.parameters())
        def fwd():
            mod.crossnorm.active = True
            return mod.forward_block(x, b, add_mode="pre", relu=True) if call == "block" else mod(x)
        bwd = lambda y: torch.autograd.grad(y, ins, gy)
        res = {}
        for rep in range(2):
            for side in ("0", "1"):
                os.environ["CNSN_WIDE"] = side
                f, bw = time_pair(fwd, bwd)
                if side not in res or f + bw < sum(res[side]): res[side] = (f, bw)
        os.environ.pop("CNSN_WIDE")
        cfg = cnsn_amd.FusedConfig(cn_active=True, sn_active=True, add_mode="pre" if call == "block" else "none", relu=call == "block")
        print(f"{shape} {dt} {call}: before {res['0'][0]:.4f} / {res['0'][1]:.4f}  wide {res['1'][0]:.4f} / {res['1'][1]:.4f}  "
              f"({(res['1'][0]/res['0'][0]-1)*100:+.0f} % / {(res['1'][1]/res['0'][1]-1)*100:+.0f} %)  path now: {cnsn_amd.which_path(x, cfg)}", flush=True)
