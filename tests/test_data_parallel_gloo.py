"""N>1 path on CPU: two processes, gloo backend.  Compute stand-in = the CPU oracle's modules (tests
may use the oracle); what is under test is the product's data-parallel plumbing: per-rank RNG
offsets, batch sharding, the bucketed gradient all-reduce — and the claim that the op needs no
data-path collective (each rank's output equals the single-process result on its own shard)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cnsn_oracle as orc
from tests.golden.gen_golden_fill import fill_sn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _local_step(rank, world, n_global=12, c=5):
    """One CNSN step on this rank's shard; returns (y, grads, perm) — pure local computation."""
    from cnsn_amd import data_parallel as dp
    g = torch.Generator().manual_seed(7)
    x_all = torch.randn(n_global, c, 6, 8, generator=g, dtype=torch.float64)
    gy_all = torch.randn(n_global, c, 6, 8, generator=g, dtype=torch.float64)
    b, e = dp.shard_batch(n_global, rank, world)
    dp.seed_rank(100, rank)
    d = orc.draw_cn((e - b, c, 6, 8), "both", beta=1)
    mod = orc.CNSN(orc.CrossNorm("both", 1), fill_sn(orc.SelfNorm(c), 3, torch.float64)).train()
    mod.crossnorm.active = True
    mod.crossnorm.next_draws = d
    x = x_all[b:e].clone().requires_grad_()
    y = mod(x)
    y.backward(gy_all[b:e])
    return mod, y.detach(), d


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cnsn_amd import _ffi
        from cnsn_amd import data_parallel as dp
        # what a rank of a process group gets WITHOUT asking (review of round 5, item 3): a 2 s bound on cluster waits and
        # persistent grids that leave RCCL's channel kernels their compute units
        before = (int(_ffi.lib().cnsn_wait_ms()), int(_ffi.lib().cnsn_headroom_cus()))
        defaults = _ffi.under_process_group_defaults()
        mod, y, d = _local_step(rank, world)
        local = [p.grad.clone() for p in mod.parameters()]
        dp.allreduce_gradients(mod.parameters())
        out[rank] = dict(y=y, perm=d.perm, box=d.content_box, local=local, before=before, defaults=defaults,
                         reduced=[p.grad.clone() for p in mod.parameters()])
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    for r in (r0, r1):
        assert r["before"] == (5000, 0) and r["defaults"] == {"wait_ms": 2000, "headroom_cus": 32}, (r["before"], r["defaults"])
    # ranks drew different permutations / boxes (per-rank seed offset)
    assert not (torch.equal(r0["perm"], r1["perm"]) and r0["box"] == r1["box"])
    # the all-reduce averaged exactly the two local gradients, identically on both ranks
    for a, b, m0, m1 in zip(r0["local"], r1["local"], r0["reduced"], r1["reduced"]):
        torch.testing.assert_close(m0, (a + b) / 2, rtol=1e-12, atol=1e-14)
        assert torch.equal(m0, m1)
    # no data-path collective: each rank's output is what a single process computes on that shard
    for rank, r in ((0, r0), (1, r1)):
        _, y_single, _ = _local_step(rank, world)
        assert torch.equal(r["y"], y_single)


def test_process_group_defaults_follow_the_environment(monkeypatch):
    from cnsn_amd import _ffi
    assert _ffi.under_process_group_defaults() == {}                 # no process group: nothing changes
    assert _ffi.default_headroom_cus() == 32
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "8")
    assert _ffi.default_headroom_cus() == 16
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "64")
    assert _ffi.default_headroom_cus() == 64                         # never more than a quarter of the part
    lib = _ffi.lib()
    try:
        lib.cnsn_set_headroom_cus(24)
        assert lib.cnsn_headroom_cus() == 24
        monkeypatch.setenv("CNSN_HEADROOM_CUS", "0")                 # the environment wins over the setter
        lib.cnsn_reload_env()
        assert lib.cnsn_headroom_cus() == 0
        monkeypatch.delenv("CNSN_HEADROOM_CUS")
        lib.cnsn_reload_env()
        assert lib.cnsn_headroom_cus() == 24
    finally:
        lib.cnsn_set_headroom_cus(0)
        lib.cnsn_reload_env()


def test_shard_batch_covers_everything():
    from cnsn_amd import data_parallel as dp
    for n in (1, 7, 8, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_batch(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
