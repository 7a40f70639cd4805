import os, sys, time, cProfile, pstats, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnsn_amd
from cnsn_amd.callers import WideResNetCNSN
dev = torch.device("cuda:0")
net = WideResNetCNSN(40, 100, 2, active_num=2, pos="post", beta=1, crop="both", cnsn_type="cnsn").to(dev).train()
opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True)
x = torch.randn(128, 3, 32, 32, device=dev); y = torch.randint(0, 100, (128,), device=dev)
def step(aug):
    loss = torch.nn.functional.cross_entropy(net(x, aug=aug), y)
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(15): step(False)
torch.cuda.synchronize()
for phase in ("fwd", "all"):
    t0 = time.perf_counter()
    for _ in range(40):
        if phase == "fwd":
            with torch.no_grad(): net(x, aug=False)
        else: step(False)
    torch.cuda.synchronize(); print(phase, "ms/iter", (time.perf_counter() - t0) / 40 * 1e3, flush=True)
# host time of the forward alone, no GPU wait
t0 = time.perf_counter()
for _ in range(40):
    out = net(x, aug=False)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("forward host ms (grad mode)", (t1 - t0) / 40 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step(False)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
