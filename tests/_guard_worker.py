"""Worker of tests/test_gpu_step_guard.py — one rank of a 2-rank data-parallel run (torch.distributed.run) in which ONE
rank's cluster-resident launch is made to give up (CNSN_FAULT_INJECT=1: a cluster member never publishes; the wait is
bounded by CNSN_WAIT_MS).  What must hold (round-3 review / advisor): both ranks repeat that step in lock-step — nobody
hangs in a collective —, the weights never see the poisoned gradients, BatchNorm running statistics and counters of the
failed attempt are put back, every rank stops choosing the cluster kernels, and the ranks end with identical parameters.

  >= 2 GPUs : one device per rank, nccl (= RCCL)          1 GPU : both ranks on cuda:0, gloo (RCCL refuses that);
              the healthy rank runs with the cluster kernels off there, so only ONE persistent grid is on the device"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FAULT_RANK, FAULT_STEP, STEPS = 1, 1, 4


def main(out_dir):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ngpu = torch.cuda.device_count()
    shared = ngpu < world
    dev = torch.device("cuda", 0 if shared else int(os.environ["LOCAL_RANK"]))
    torch.cuda.set_device(dev)
    if shared:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)

    import cnsn_amd
    from cnsn_amd import _ffi, data_parallel as dp
    from cnsn_amd.callers import StepGuard
    cnsn_amd.follow_environ()
    if shared and rank != FAULT_RANK:
        cnsn_amd.set_resident(False)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = torch.nn.Conv2d(3, 8, 3, padding=1, bias=False)
            self.cnsn = cnsn_amd.CNSN(cnsn_amd.CrossNorm("both", 1), cnsn_amd.SelfNorm(8))
            self.bn = torch.nn.BatchNorm2d(8)
            self.fc = torch.nn.Linear(8, 10)

        def forward(self, x, arm):
            h = self.conv(x)
            self.cnsn.crossnorm.active = arm
            h = torch.relu(self.bn(self.cnsn(h)))
            return self.fc(h.mean((2, 3)))

    torch.manual_seed(7)
    net = Net().to(dev).train()
    model = torch.nn.parallel.DistributedDataParallel(net, device_ids=None if shared else [dev.index])
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    dp.seed_rank(300, rank)
    g = torch.Generator(device=dev).manual_seed(40 + rank)
    xs = [torch.randn(64, 3, 56, 56, device=dev, generator=g) for _ in range(STEPS)]
    ys = [torch.randint(0, 10, (64,), device=dev, generator=g) for _ in range(STEPS)]
    probe = torch.empty(64, 8, 56, 56, device=dev)
    path_before = cnsn_amd.which_path(probe, cnsn_amd.FusedConfig(sn_active=True))
    guard = StepGuard(net)
    attempts, draws = 0, []
    for i in range(STEPS):
        first = [True]

        def compute_loss():
            nonlocal attempts
            attempts += 1
            inject = rank == FAULT_RANK and i == FAULT_STEP and first[0]
            first[0] = False
            os.environ["CNSN_FAULT_INJECT"] = "1" if inject else "0"
            draws.append(float(np.random.rand()))
            out = model(xs[i], arm=(i % 2 == 0))               # armed on even steps (host-drawn permutation and boxes)
            os.environ["CNSN_FAULT_INJECT"] = "0"
            return torch.nn.functional.cross_entropy(out, ys[i])

        loss = guard.run(compute_loss, opt)
        assert bool(torch.isfinite(loss)), f"rank {rank} step {i}: loss {float(loss)}"
    torch.cuda.synchronize()
    state = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    res = dict(rank=rank, world=world, backend=dist.get_backend(), shared_device=shared, attempts=attempts,
               repeats=guard.repeats, local_timeouts=guard.local_timeouts, path_before=path_before,
               path_after=cnsn_amd.which_path(probe, cnsn_amd.FusedConfig(sn_active=True)),
               library_timeouts=int(_ffi.lib().cnsn_resident_timeouts()),
               finite=all(bool(torch.isfinite(v).all()) for v in state.values()),
               counters={k: int(v) for k, v in state.items() if k.endswith("num_batches_tracked")},
               repeated_draw_equal=(len(draws) == STEPS + 1 and draws[FAULT_STEP] == draws[FAULT_STEP + 1]),
               wait_ms=int(_ffi.lib().cnsn_wait_ms()))
    torch.save({k: v for k, v in state.items() if "running" not in k and "num_batches" not in k},
               os.path.join(out_dir, f"params{rank}.pt"))
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
