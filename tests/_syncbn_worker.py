"""Worker of tests/test_gpu_sync_bn.py — one rank of a 2-rank run (torch.distributed.run).

The reference's segmentation trainer converts every BatchNorm of the model to nn.SyncBatchNorm when `sync_bn` is set
(segmentation/tool/train_cnsn.py:160) — SelfNorm's gate BatchNorm1d included, so the gate's batch statistic spans the
GLOBAL batch.  Each rank runs SelfNorm (gate = nn.SyncBatchNorm, composed path) on its half of a global batch; the
same process also runs the fused single-launch SelfNorm on the WHOLE batch.  They must agree: outputs and input
gradients on the rank's half, parameter gradients after summing over the ranks, running statistics.

  >= 2 GPUs: one device per rank, nccl (= RCCL);  1 GPU: both ranks on cuda:0, gloo."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_dir):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    shared = torch.cuda.device_count() < world
    dev = torch.device("cuda", 0 if shared else int(os.environ["LOCAL_RANK"]))
    torch.cuda.set_device(dev)
    backend = "gloo" if shared else "nccl"
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    import cnsn_amd
    from tests.golden.gen_golden_fill import fill_sn
    if shared:
        cnsn_amd.set_resident(False)               # two processes on one GPU (tests/_ddp_worker.py)
    res = dict(rank=rank, world=world, backend=backend, cases=[])
    for case, (shape, is_two) in enumerate([((12, 6, 28, 28), False), ((8, 5, 14, 14), True), ((6, 16, 56, 56), False)]):
        n_global, c = shape[0], shape[1]
        g = torch.Generator().manual_seed(40 + case)                   # the same global batch on every rank
        x_all = torch.randn(shape, generator=g) * (torch.rand(n_global, c, 1, 1, generator=g) * 1.5 + 0.5)
        x_all = (x_all + torch.randn(n_global, c, 1, 1, generator=g)).to(dev)      # per-plane scales / offsets
        gy_all = torch.randn(shape, generator=g).to(dev)
        lo, hi = rank * n_global // world, (rank + 1) * n_global // world

        whole = fill_sn(cnsn_amd.SelfNorm(c, is_two=is_two), 3 + case, torch.float32).to(dev).train()
        assert whole._fusable()
        xw = x_all.clone().requires_grad_()
        yw = whole(xw)
        yw.backward(gy_all)

        mine = fill_sn(cnsn_amd.SelfNorm(c, is_two=is_two), 3 + case, torch.float32)
        mine = torch.nn.SyncBatchNorm.convert_sync_batchnorm(mine).to(dev).train()
        assert type(mine.g_bn) is torch.nn.SyncBatchNorm and not mine._fusable()
        xm = x_all[lo:hi].clone().requires_grad_()
        ym = mine(xm)
        ym.backward(gy_all[lo:hi])
        torch.cuda.synchronize()

        def rel(a, b):
            return float((a.double() - b.double()).abs().max() / max(1.0, float(b.double().abs().max())))

        errs = dict(y=rel(ym, yw[lo:hi]), dx=rel(xm.grad, xw.grad[lo:hi]))
        for (k, pm), (_, pw) in zip(mine.named_parameters(), whole.named_parameters()):
            gsum = pm.grad.detach().clone()
            if backend == "gloo":
                h = gsum.cpu()
                dist.all_reduce(h)
                gsum = h.to(dev)
            else:
                dist.all_reduce(gsum)
            errs["grad " + k] = rel(gsum, pw.grad)
        for (k, bm), (_, bw) in zip(mine.named_buffers(), whole.named_buffers()):
            if "num_batches" in k:
                assert int(bm) == int(bw) == 1
            else:
                errs["state " + k] = rel(bm, bw)
        # the local statistic is NOT the global one: the comparison above is not vacuous
        local = fill_sn(cnsn_amd.SelfNorm(c, is_two=is_two), 3 + case, torch.float32).to(dev).train()
        differs = rel(local(x_all[lo:hi].clone()), yw[lo:hi].detach())
        res["cases"].append(dict(shape=shape, is_two=is_two, errs=errs, local_vs_global=differs))
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
