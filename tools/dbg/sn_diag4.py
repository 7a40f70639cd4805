import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnsn_amd
from tests.golden.gen_golden_fill import fill_sn
os.environ["CNSN_SNX"] = "2"; os.environ["CNSN_DEBUG"] = "1"
cnsn_amd.set_strategy("resident")
for tag, h, w, n in (("f32", 40, 40, 5), ("f32", 56, 56, 5), ("f32", 28, 32, 5), ("f32", 56, 56, 256), ("bf16", 56, 56, 256)):
    dtype = {"f32": torch.float32, "bf16": torch.bfloat16}[tag]
    c = 4 if n < 100 else 64
    x = torch.randn(n, c, h, w, device="cuda").to(dtype).requires_grad_()
    mod = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(c), 3, torch.float32)).cuda().train()
    print("==", tag, (n, c, h, w), file=sys.stderr, flush=True)
    y = mod(x)
    torch.autograd.grad(y, [x] + list(mod.parameters()), torch.randn_like(y))
    torch.cuda.synchronize()
