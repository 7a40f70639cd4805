import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cnsn_amd
dev = torch.device("cuda:0")
shape = (128, 128, 8, 8)
x = torch.randn(shape, device=dev, requires_grad=True)
gy = torch.randn(shape, device=dev)
mod = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(shape[1])).to(dev).train()
params = list(mod.parameters())
def fwd():
    for _ in range(3000): mod(x)
def fb():
    for _ in range(2000): torch.autograd.grad(mod(x), [x] + params, gy)
for name, fn in (("forward", fwd), ("forward+backward", fb)):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter(); fn(); torch.cuda.synchronize()
    n = 3000 if name == "forward" else 2000
    print(name, (time.perf_counter() - t) / n * 1e6, "us per call")
    pr = cProfile.Profile(); pr.enable(); fn(); pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
