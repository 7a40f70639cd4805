"""The time-out protocol on an MI355X with a REAL injected fault (round-3 review, "make the N-rank path
collective-safe"): one rank's cluster launch gives up in the middle of a data-parallel run — through
`callers.steps.StepGuard` under torch DDP (tests/_guard_worker.py) and through `bench.py --gpus 2` (headline workload and
a whole-network workload).  One GPU: both ranks share cuda:0 over gloo; two or more: one device per rank over RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**kw):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), CNSN_WAIT_MS="300")
    env.pop("CNSN_FAULT_INJECT", None)
    env.update(kw)
    return env


def test_two_ranks_repeat_a_step_one_of_them_timed_out(tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_guard_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_env(), timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    res = [json.load(open(tmp_path / f"rank{k}.json")) for k in (0, 1)]
    for d in res:
        assert d["repeats"] == 1 and d["attempts"] == 5, d         # BOTH ranks ran the failed step twice: lock-step
        assert d["finite"], d                                       # no NaN in weights or BatchNorm statistics
        assert set(d["counters"].values()) == {4}, d                # every num_batches_tracked moved once per step
        assert d["path_after"] == "streaming", d                    # rank-wide degradation (the healthy rank too)
    assert res[1]["path_before"] == "resident" and res[1]["local_timeouts"] == 1 and res[1]["library_timeouts"] == 1
    assert res[0]["local_timeouts"] == 0 and res[0]["library_timeouts"] == 0
    assert res[1]["repeated_draw_equal"]                            # the repeat re-drew the same host random numbers
    p0, p1 = (torch.load(tmp_path / f"params{k}.pt") for k in (0, 1))
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k                         # identical final parameters on both ranks


@pytest.mark.parametrize("workload", ["cnsn", "wrn40"])
def test_bench_two_ranks_with_an_injected_fault(workload):
    """`python bench.py --gpus 2 [...]` completes, reports the repeat and which rank's launch gave up"""
    extra = ["--shape", "64,16,56,56", "--no-extra", "--no-cpu-baseline", "--no-ceiling"] if workload == "cnsn" else \
        ["--workload", "wrn40", "--batch", "32"]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_env(CNSN_BENCH_FAULT="1:3"), timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["resident_timeouts"] == [0, 1], d
    assert d["steps_repeated_after_a_cluster_timeout"] == 4, d       # the whole timed window, on both ranks
    assert d["value"] > 0
