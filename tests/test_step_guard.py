"""The time-out protocol of the training steps, without a GPU: `callers.steps.StepGuard` (snapshot / settle / agree /
repeat) and `data_parallel.agree_to_repeat` under a two-rank gloo group.  The library's counter is replaced by a scripted
one (`_ffi._timeout_count`); the poisoned attempt is imitated the way the kernels do it — NaNs in the activations of the
attempt that "gave up" — so that the test sees what the advisor's finding was about: BatchNorm running statistics,
`num_batches_tracked` and the host RNG streams of a failed attempt must not survive the repeat, and no rank may repeat
alone.  (The same protocol with a real injected fault on an MI355X: tests/test_gpu_step_guard.py.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Net(nn.Module):
    """Linear -> BatchNorm1d -> Linear; `poison` multiplies the hidden activations by NaN (what a cluster launch that
    gave up leaves in the planes it still owed)."""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 8)
        self.bn = nn.BatchNorm1d(8)
        self.b = nn.Linear(8, 3)
        self.poison = False
        self.draws = []

    def forward(self, x):
        self.draws.append((float(np.random.rand()), float(torch.rand(()))))     # the host RNG streams CrossNorm uses
        h = self.a(x)
        if self.poison:
            h = h * float("nan")
        return self.b(self.bn(h))


class _Script:
    """a scripted time-out counter: `fail_at` = indices of `compute_loss` calls during which a launch "gives up" """

    def __init__(self, net, fail_at):
        self.net, self.fail_at, self.calls, self.count = net, set(fail_at), 0, 0

    def begin_attempt(self):
        bad = self.calls in self.fail_at
        self.calls += 1
        self.net.poison = bad
        if bad:
            self.count += 1

    def __call__(self):
        return self.count


def _train(net, steps, script, group_ok=True, rearm_after=200, log=None):
    from cnsn_amd import _ffi
    from cnsn_amd.callers import StepGuard
    torch.manual_seed(11)
    xs = [torch.randn(16, 6) for _ in range(steps)]
    ys = [torch.randint(0, 3, (16,)) for _ in range(steps)]
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    guard = StepGuard(net, rearm_after=rearm_after)
    model = net
    if dist.is_initialized():
        model = nn.parallel.DistributedDataParallel(net)
    old = _ffi._timeout_count
    _ffi._timeout_count = script
    try:
        for i in range(steps):
            def compute_loss():
                script.begin_attempt()
                return nn.functional.cross_entropy(model(xs[i]), ys[i])
            loss = guard.run(compute_loss, opt)
            assert torch.isfinite(loss)
            if log is not None:
                log.append((i, guard.degraded, guard.rearms, guard.rearm_after, int(_ffi.lib().cnsn_resident_degraded())))
    finally:
        _ffi._timeout_count = old
    return guard


def _state(net):
    return {k: v.clone() for k, v in net.state_dict().items()}


@pytest.fixture
def resident_reenabled():
    yield
    import cnsn_amd
    cnsn_amd.set_resident(True)
    from cnsn_amd import _ffi
    _ffi._timeouts_reported = 0


def test_failed_attempt_leaves_no_trace(resident_reenabled):
    """one process: the attempt that times out poisons the BatchNorm statistics, moves the counter and consumes RNG
    draws; after the guarded step everything equals a run in which nothing failed"""
    torch.manual_seed(3)
    np.random.seed(3)
    clean = _Net()
    faulty = _Net()
    faulty.load_state_dict(clean.state_dict())
    torch.manual_seed(5)
    np.random.seed(5)
    g0 = _train(clean, 4, _Script(clean, ()))
    torch.manual_seed(5)
    np.random.seed(5)
    g1 = _train(faulty, 4, _Script(faulty, (1, 4)))          # second step fails once, fourth step (call index 4) once
    assert (g0.repeats, g1.repeats, g1.local_timeouts) == (0, 2, 2)
    a, b = _state(clean), _state(faulty)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert int(faulty.bn.num_batches_tracked) == 4 and torch.isfinite(faulty.bn.running_var).all()
    # the repeats drew the SAME numbers the failed attempts had drawn: the trajectory of draws is that of the clean run
    applied = [d for i, d in enumerate(faulty.draws) if i not in (1, 4)]
    assert applied == clean.draws and faulty.draws[1] == faulty.draws[2] and faulty.draws[4] == faulty.draws[5]


def test_cluster_kernels_come_back_after_a_clean_stretch(resident_reenabled):
    """round-4 review item 7b: a process that degraded once tries the cluster kernels again after `rearm_after` applied
    steps without a time-out (`data_parallel.rearm_all` -> `cnsn_resident_rearm` + `cnsn_resident_enable(1)`); a relapse
    doubles the stretch.  Steps: 0 ok | 1 fails -> degraded, its repeat is the first clean step | 2 clean -> re-armed behind
    step 2 | 3 ok | 4 fails (relapse: the way back is now 4 steps; its repeat is the first) | 5, 6, 7 clean -> re-armed behind
    step 7 | 8, 9 ok."""
    import cnsn_amd
    from cnsn_amd import data_parallel as dp
    net = _Net()
    log = []
    # call indices: step 0 -> 0, step 1 -> 1 (fails) + 2, steps 2, 3 -> 3, 4, step 4 -> 5 (fails) + 6, ...
    g = _train(net, 10, _Script(net, (1, 5)), rearm_after=2, log=log)
    assert g.repeats == 2 and g.rearms == 2 and g.rearm_after == 4 and not g.degraded
    degraded = [d for _, d, *_ in log]
    assert degraded == [False, True, False, False, True, True, True, False, False, False]
    assert [r for _, _, r, *_ in log] == [0, 0, 1, 1, 1, 1, 1, 2, 2, 2]
    assert not dp._degraded_here
    # the library's own switch follows: AUTO may choose the cluster kernels again (asked without a GPU: the plan query)
    assert cnsn_amd.lib().cnsn_resident_degraded() == 0
    # a user's own CNSN_RESIDENT=0 / set_resident(False) is never undone by the protocol
    cnsn_amd.set_resident(False)
    assert dp.rearm_all() == 0


def test_gives_up_loudly_when_every_attempt_fails(resident_reenabled):
    from cnsn_amd import CnsnError
    net = _Net()
    with pytest.raises(CnsnError, match="still time out"):
        _train(net, 1, _Script(net, range(10)))


def test_mid_step_poll_is_silent_inside_the_guard(resident_reenabled):
    from cnsn_amd import CnsnError, _ffi
    old = _ffi._timeout_count
    _ffi._timeout_count = lambda: 7
    try:
        with _ffi.deferred_timeouts():
            _ffi.check_resident_health("inside")             # no raise: the boundary poll owns the report
        assert _ffi.poll_timeouts() == 7 and _ffi.poll_timeouts() == 0
        _ffi._timeout_count = lambda: 8
        with pytest.raises(CnsnError, match="repeat"):
            _ffi.check_resident_health("outside")
    finally:
        _ffi._timeout_count = old
        _ffi._timeouts_reported = 0


def test_repeat_on_timeout_refuses_a_process_group(monkeypatch):
    from cnsn_amd import CnsnError
    from cnsn_amd.callers import steps
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda *a: 2)
    with pytest.raises(CnsnError, match="StepGuard"):
        steps.repeat_on_timeout(lambda: None)


def _worker(rank, world, port, out, fail_rank, fail_at):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cnsn_amd import data_parallel as dp
        torch.manual_seed(3)
        net = _Net()                                          # same initial weights on both ranks
        dp.seed_rank(50, rank)
        script = _Script(net, fail_at if rank == fail_rank else ())
        log = []
        guard = _train(net, 5, script, rearm_after=2, log=log)
        out[rank] = dict(state=_state(net), repeats=guard.repeats, local=guard.local_timeouts, attempts=script.calls,
                         rearm_log=[(i, d, r) for i, d, r, *_ in log], rearms=guard.rearms,
                         agreed=dp.agree_to_repeat(rank, None), gathered=dp.gather_ints(10 + rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _two_ranks(fail_rank, fail_at):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out, fail_rank, fail_at), nprocs=2, join=True)
    return out[0], out[1]


def test_two_ranks_repeat_in_lock_step():
    """rank 1's launch "gives up" in steps 2 and 4; rank 0 is healthy.  Both ranks repeat those steps (same number of
    attempts = same number of DDP all-reduces: nothing hangs), end with identical parameters AND buffers, and those are
    the parameters of a run in which nobody failed."""
    r0, r1 = _two_ranks(1, (1, 4))
    assert (r0["repeats"], r1["repeats"]) == (2, 2) and (r0["local"], r1["local"]) == (0, 2)
    assert r0["attempts"] == r1["attempts"] == 7
    # the way back is rank-agreed without a collective: both ranks re-arm behind the same step (steps 1 and 3 failed on rank
    # 1 only; two applied steps later — behind step 2 — BOTH switch the cluster kernels on again; step 3 is a relapse)
    assert r0["rearm_log"] == r1["rearm_log"] and r0["rearms"] == r1["rearms"] == 1
    assert [d for _, d, _ in r0["rearm_log"]] == [False, True, False, True, True]
    assert r0["agreed"] == r1["agreed"] == 1 and r0["gathered"] == r1["gathered"] == [10, 11]
    for k in r0["state"]:
        if "running" in k or "num_batches" in k:              # BatchNorm buffers are rank-local under DDP(broadcast off at
            assert torch.isfinite(r1["state"][k].float()).all(), k   # the last step): finite, counted once per step
        else:
            assert torch.equal(r0["state"][k], r1["state"][k]), k
    assert int(r1["state"]["bn.num_batches_tracked"]) == 5 == int(r0["state"]["bn.num_batches_tracked"])
    c0, c1 = _two_ranks(-1, ())
    assert c0["repeats"] == 0
    for k in c0["state"]:
        assert torch.equal(c0["state"][k], r0["state"][k]) and torch.equal(c1["state"][k], r1["state"][k]), k
