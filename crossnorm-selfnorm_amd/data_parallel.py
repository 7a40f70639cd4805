"""Data-parallel plumbing for the CNSN hot path: one process per GPU, `torch.distributed` with the
`nccl` backend (= RCCL over xGMI on MI355X).

The op itself never communicates: CrossNorm permutes inside the local minibatch (reference
models/cnsn.py:62) and SelfNorm's BatchNorm1d statistics are local (cnsn.py:121,138) — exactly what
each replica sees under the reference's `nn.DataParallel` (cifar.py:395, imagenet.py:533).  The only
exchange is the gradient all-reduce of the parameters (for a CNSN site: g_fc.weight (C,1,2),
g_bn.weight, g_bn.bias = 4C floats), and ranks must draw DIFFERENT permutations / boxes.
"""
from __future__ import annotations

from typing import Iterable, Optional

import numpy as np
import torch
import torch.distributed as dist


def seed_rank(seed: int, rank: int) -> None:
    """Seed the two host RNGs CrossNorm consumes (torch CPU generator for randperm, numpy global
    for the boxes) with a per-rank offset so that ranks draw different style partners and crops."""
    torch.manual_seed(seed + rank)
    np.random.seed((seed + rank) % (2 ** 32))


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None,
                        average: bool = True) -> None:
    """One bucketed all-reduce over the gradients of `params` (what DDP does for a small model).
    A single flat buffer per dtype: CNSN's own parameters are a few KB, so one message is optimal on
    xGMI (latency-bound); large backbones should use torch DDP's 25 MB buckets instead."""
    if not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    by_dtype = {}
    for p in params:
        if p.grad is not None:
            by_dtype.setdefault(p.grad.dtype, []).append(p.grad)
    backend = dist.get_backend(group)
    for grads in by_dtype.values():
        flat = torch.cat([g.reshape(-1) for g in grads])
        if flat.is_cuda and backend == "gloo":   # CPU-only backend (tests): stage on host
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            flat.copy_(host)
            if average:
                flat.div_(world)
        elif average and backend == "nccl" and _nccl_avg_ok(flat, group):
            pass                                 # RCCL averaged inside the collective: no separate division launch
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                flat.div_(world)
        # back into the gradients: ONE launch for all of them (three launches per step in all: cat, all-reduce, copy —
        # the step of the headline workload is 0.85 ms, every small launch is half a percent of it)
        pieces = [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)]
        torch._foreach_copy_(grads, pieces)


_avg_unsupported = False


def _nccl_avg_ok(flat: torch.Tensor, group) -> bool:
    """all_reduce(flat, AVG) if this RCCL build takes ReduceOp.AVG (it is refused before anything is sent, on every rank
    alike, when it does not); False = nothing was reduced, the caller sums and divides."""
    global _avg_unsupported
    if _avg_unsupported:
        return False
    try:
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
        return True
    except (RuntimeError, ValueError, TypeError):
        _avg_unsupported = True
        return False


def agree_to_repeat(local_timeouts: int, device: Optional[torch.device] = None,
                    group: Optional[dist.ProcessGroup] = None) -> int:
    """The one collective of the time-out protocol: MAX over the ranks of "cluster launches of mine that gave up during
    this step" (4 bytes).  Returns the same number on every rank — 0: apply the step; > 0: EVERY rank repeats it, and
    every rank stops choosing the cluster kernels (`degrade_all`), so the ranks' launch sequences and collectives stay
    in lock-step.  Without a process group the local number comes back.  Call it once per step on every rank, at the
    same point of the step (after the backward and its gradient all-reduce, before `optimizer.step()`): the reference's
    multi-GPU loops (imagenet.py:533, cifar.py:395, segmentation/tool/train_cnsn.py:175-177) have no such point because
    eager PyTorch has no launch that can give up."""
    n = int(local_timeouts)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return n
    on_device = dist.get_backend(group) == "nccl"
    if on_device and device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    flag = torch.tensor([n], dtype=torch.int32, device=device if on_device else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return int(flag.item())


def degrade_all() -> None:
    """What every rank does once ANY rank reported a time-out: CNSN_STRATEGY_AUTO stops choosing the cluster-resident
    kernels (`cnsn_resident_enable(0)`), so all ranks run the same (two-pass / single-workgroup) kernels from the repeat
    on — a rank that kept its cluster kernels would be faster than the degraded one and wait for it in every collective,
    and could be the next to time out if the cause (a shared GPU, a foreign persistent kernel) is node-wide."""
    global _degraded_here, _allowed_before
    from . import functional
    if not _degraded_here:
        _allowed_before = functional.resident_allowed()
    functional.set_resident(False)
    _degraded_here = True


_degraded_here = False       # the cluster kernels are off because degrade_all() switched them off (not the user)
_allowed_before = True       # ... and what the user's switch said at that moment (CNSN_RESIDENT=0 / set_resident(False))


def rearm_all() -> int:
    """The way back from `degrade_all`: forgive the time-outs counted so far (`cnsn_resident_rearm`) and let
    CNSN_STRATEGY_AUTO choose the cluster-resident kernels again.  What kept part of a persistent grid off the device for
    seconds — another process's kernel, a debugger, a clock event — is usually gone minutes later, and a job that runs for
    days should not pay the two-pass kernels for the rest of its life.  EVERY rank must call it at the same step: the
    callers count clean steps since the (rank-agreed) degradation, so no collective is needed (`StepGuard`).  Does nothing
    when this module did not switch them off; puts the user's switch (CNSN_RESIDENT=0, `set_resident(False)`) back as it was.  Returns how
    often this process has re-armed."""
    global _degraded_here
    from . import _ffi, functional
    if not _degraded_here:
        return 0
    n = int(_ffi.lib().cnsn_resident_rearm())
    functional.set_resident(_allowed_before)      # (back to what the switch said: a job started with CNSN_RESIDENT=0 stays off)
    _degraded_here = False
    return n


def gather_ints(value: int, device: Optional[torch.device] = None, group: Optional[dist.ProcessGroup] = None):
    """[value of rank 0, value of rank 1, ...] on every rank (bench / logging: per-rank time-out counters)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [int(value)]
    world = dist.get_world_size(group)
    on_device = dist.get_backend(group) == "nccl"
    if on_device and device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    mine = torch.tensor([int(value)], dtype=torch.int64, device=device if on_device else "cpu")
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return [int(t.item()) for t in out]


def shard_batch(n_total: int, rank: int, world: int):
    """[begin, end) of the global batch owned by `rank` (contiguous, sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)
