"""cProfile of 2000 forward+backward calls of one small site: where the host time of a call goes (tools/host_floor.py has the
per-call totals)."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnsn_amd
dev = torch.device("cuda:0")
shape = (128, 64, 16, 16)
x = torch.randn(shape, device=dev, requires_grad=True)
gy = torch.randn(shape, device=dev)
mod = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(shape[1])).to(dev).train()
params = list(mod.parameters())
def step():
    y = mod(x)
    torch.autograd.grad(y, [x] + params, gy)
for _ in range(300): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
