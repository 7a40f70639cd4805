import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnsn_amd
from tests.golden.gen_golden_fill import fill_sn
def run(shape, dtype, fs, bs, seed=7):
    n, c = shape[:2]
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (torch.randn(shape, device="cuda", generator=g) * (torch.rand(n, c, 1, 1, device="cuda", generator=g) * 1.5 + 0.5)
         + torch.randn(n, c, 1, 1, device="cuda", generator=g)).to(dtype).requires_grad_()
    gy = torch.randn(shape, device="cuda", generator=g).to(dtype)
    mod = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(c), seed, torch.float32)).cuda().train()
    cnsn_amd.set_strategy("resident")
    os.environ["CNSN_SNX"] = fs
    y = mod(x)
    os.environ["CNSN_SNX"] = bs
    grads = torch.autograd.grad(y, [x] + list(mod.parameters()), gy)
    torch.cuda.synchronize()
    return [y.detach()] + [t.detach() for t in grads]
names = ["y", "dx", "dw", "dgamma", "dbeta"]
for tag, h, w, n in (("f32", 40, 40, 5), ("f32", 60, 32, 5), ("bf16", 56, 56, 5), ("f32", 28, 32, 5)):
    dtype = {"f32": torch.float32, "bf16": torch.bfloat16}[tag]
    shape = (n, 4, h, w)
    ref = run(shape, dtype, "0", "0")
    for fs, bs in (("2", "0"), ("0", "2"), ("2", "2")):
        out = run(shape, dtype, fs, bs)
        print(tag, shape, "fwd snx", fs, "bwd snx", bs, " ".join(f"{nm}:{float((a.double() - b.double()).abs().max()) / max(float(b.double().abs().max()), 1e-6):.1e}" for nm, a, b in zip(names, out, ref)), flush=True)
