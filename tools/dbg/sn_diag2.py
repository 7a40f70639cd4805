import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnsn_amd
from tests.golden.gen_golden_fill import fill_sn
os.environ["CNSN_SNX"] = "2"
torch.set_printoptions(linewidth=200, precision=3, sci_mode=True)
def run(shape, dtype, st, seed=7):
    n, c = shape[:2]
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (torch.randn(shape, device="cuda", generator=g) * (torch.rand(n, c, 1, 1, device="cuda", generator=g) * 1.5 + 0.5)
         + torch.randn(n, c, 1, 1, device="cuda", generator=g)).to(dtype).requires_grad_()
    gy = torch.randn(shape, device="cuda", generator=g).to(dtype)
    mod = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(c), seed, torch.float32)).cuda().train()
    cnsn_amd.set_strategy(st)
    y = mod(x)
    grads = torch.autograd.grad(y, [x] + list(mod.parameters()), gy)
    torch.cuda.synchronize()
    return [y.detach()] + [t.detach() for t in grads]
for tag, h, w, n in (("f32", 40, 40, 5), ("f32", 40, 40, 9), ("f32", 60, 32, 5)):
    dtype = {"f32": torch.float32, "bf16": torch.bfloat16}[tag]
    shape = (n, 4, h, w)
    ref = run(shape, dtype, "two_pass")
    out = run(shape, dtype, "resident")
    print(tag, shape)
    print(" dbeta ref", ref[4].tolist(), "\n dbeta out", out[4].tolist())
    print(" dgamma ref", ref[3].tolist(), "\n dgamma out", out[3].tolist())
    e = (out[1].double() - ref[1].double()).abs().amax(dim=(2, 3)) / ref[1].double().abs().amax()
    print(" dx err per plane [n][c]:\n", e)
    # within plane (0,0): error by flat index, in chunks of 64 vectors (256 floats)
    d = (out[1][0, 0].double() - ref[1][0, 0].double()).abs().flatten()
    print(" plane(0,0) err by slot:", [float(d[i:i + 256].max()) for i in range(0, d.numel(), 256)])
    # is dx an affine function of (G, x) per plane with wrong coefficients? fit dx = a*G + b*x + c on plane (0,0)
