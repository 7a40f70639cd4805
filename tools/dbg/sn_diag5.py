import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnsn_amd
from tests.golden.gen_golden_fill import fill_sn
torch.set_printoptions(linewidth=220, precision=5, sci_mode=False)
os.environ["CNSN_SNX"] = "2"
cnsn_amd.set_strategy("resident")
for tag, h, w, n in (("f32", 40, 40, 5), ("f32", 56, 56, 5)):
    shape = (n, 2, h, w)
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(shape, device="cuda", generator=g).requires_grad_()
    gy = torch.randn(shape, device="cuda", generator=g)
    mod = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(2), 7, torch.float32)).cuda().train()
    y = mod(x)
    dx, = torch.autograd.grad(y, [x], gy)
    torch.cuda.synchronize()
    dbg = dx.reshape(n, 2, -1)[:, :, :12]
    xd, gd = x.detach().double(), gy.double()
    mu = xd.mean(dim=(2, 3))
    s1 = gd.sum(dim=(2, 3)); s2 = (gd * (xd - mu[:, :, None, None].float().double())).sum(dim=(2, 3))
    gate = (y.detach().double() / xd).median(dim=3).values.median(dim=2).values
    dt = (s2 + mu * s1) * gate * (1 - gate)
    print(tag, shape)
    for c in range(2):
        print(" c", c, "kernel  s1", dbg[:, c, 0].tolist(), "\n      torch s1", s1[:, c].tolist())
        print("      kernel s2", dbg[:, c, 1].tolist(), "\n      torch s2", s2[:, c].tolist())
        print("      kernel dt", dbg[:, c, 2].tolist(), "\n      torch dt", dt[:, c].tolist())
        print("      kernel mu", dbg[:, c, 3].tolist(), " torch mu", mu[:, c].tolist())
        print("      kernel g", dbg[:, c, 4].tolist(), " torch g", gate[:, c].tolist())
        print("      gathered Sdt parts: m0", dbg[0, c, 7].item(), "m1", dbg[0, c, 8].item(), "K", dbg[0, c, 9].item(), " sum torch dt", dt[:, c].sum().item(), " dt[:4] sum", dt[:4, c].sum().item(), "dt[4]", dt[4, c].item())
