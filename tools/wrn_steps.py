"""Tuning aid: WideResNet-40-2+CNSN training steps (bs 128, fp32) timed by kind — idle (no CrossNorm site armed), armed, mixed —
with the device-allocation count of each phase; CNSN_FUSE_TAIL=0/1 compares the fused BatchNorm2d+ReLU tail (profiles/r03_bn_tail.md)."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnsn_amd
from cnsn_amd.callers import WideResNetCNSN
dev = torch.device("cuda:0")
net = WideResNetCNSN(40, 100, 2, active_num=2, pos="post", beta=1, crop="both", cnsn_type="cnsn").to(dev).train()
opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True)
x = torch.randn(128, 3, 32, 32, device=dev); y = torch.randint(0, 100, (128,), device=dev)
def step(aug):
    loss = torch.nn.functional.cross_entropy(net(x, aug=aug), y)
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
np.random.seed(0)
for _ in range(20): step(bool(np.random.rand() < 0.5))
torch.cuda.synchronize()
for name, fn in (("idle", lambda: step(False)), ("armed", lambda: step(True)), ("mixed", lambda: step(bool(np.random.rand() < 0.5)))):
    a0 = torch.cuda.memory_stats()["num_device_alloc"]
    t0 = time.perf_counter()
    for _ in range(60): fn()
    torch.cuda.synchronize()
    print(name, "ms/iter %.3f" % ((time.perf_counter() - t0) / 60 * 1e3), "device allocs during:", torch.cuda.memory_stats()["num_device_alloc"] - a0, flush=True)
