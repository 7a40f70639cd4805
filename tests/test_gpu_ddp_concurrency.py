"""Two data-parallel ranks with gradient communication overlapping the backward's cluster-resident launches
(BASELINE.json configs[3]; see tests/_ddp_worker.py).  Runs on a 1-GPU box (ranks share the device, gloo) and on a
multi-GPU box (one device per rank, RCCL)."""
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


@pytest.mark.parametrize("world", [2, 1], ids=["two_ranks", "one_rank_rccl"])
def test_resident_kernels_under_data_parallel_communication(tmp_path, world):
    """world = 2: see the module docstring.  world = 1: ONE rank on its own device, so the group is a real RCCL
    communicator (backend nccl, `device_id` bound, torch DDP with its reducer hooks and streams) next to the cluster
    kernels — all that a 1-GPU box can show of the nccl path."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_ddp_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    reps = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(world)]
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):                                   # evidence for profiles/
        json.dump(reps, open(os.path.join(keep, "ddp_concurrency.json" if world == 2 else "rccl_one_rank.json"), "w"), indent=1)
    for rep in reps:
        assert rep["world"] == world and rep["sites"] == 16
        if world == 1:
            # (a group of ONE rank has no peer that would wait with it: the 5 s default stays — advisor, round 4)
            assert rep["backend"] == "nccl" and rep["mode"] == "ddp" and rep["wait_ms"] == 5000, rep
        else:
            assert rep["wait_ms"] == 2000, rep                 # peers wait in the next collective: the shorter bound
        if not rep["shared_device"]:
            assert "resident" in rep["paths"], rep             # one rank per GPU: the cluster kernels were in play
        assert rep["timeouts"] == 0, rep                       # no bounded wait ran out
        assert rep["site_outputs_bit_identical"], rep          # communication changes nothing in the op's results
        assert rep["finite"]
        if rep["mode"] == "ddp":
            assert rep["grads_bit_identical"], rep             # (a + b) / 2 either way: exact


def test_resident_kernels_next_to_a_busy_side_stream():
    """One process, one GPU: the cluster-resident kernels on the main stream while a second stream keeps the memory system
    and some compute units busy the whole time (what a gradient-reduction stream does).  Bit-identical results, no
    bounded wait running out."""
    import cnsn_amd
    from cnsn_amd import _ffi
    from tests.golden.gen_golden_fill import fill_sn
    dev = torch.device("cuda:0")
    shape = (64, 128, 56, 56)
    torch.manual_seed(5)
    x = torch.randn(shape, device=dev)
    gy = torch.randn(shape, device=dev)
    assert cnsn_amd.which_path(x, cnsn_amd.FusedConfig(sn_active=True, cn_active=True)) == "resident"

    def step():
        torch.manual_seed(6)
        mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm("neither", 1), fill_sn(cnsn_amd.SelfNorm(shape[1]), 3, torch.float32)).to(dev).train()
        outs = []
        for _ in range(6):
            mod.crossnorm.active = True
            mod.crossnorm.next_draws = cnsn_amd.CNDraws(torch.arange(shape[0]).flip(0))
            xi = x.clone().requires_grad_()
            y = mod(xi)
            y.backward(gy)
            outs += [y.detach().clone(), xi.grad.clone()]
        torch.cuda.synchronize()
        return outs

    quiet = step()
    side = torch.cuda.Stream(device=dev)
    a = torch.zeros(32 << 20, device=dev)
    b = torch.empty_like(a)
    with torch.cuda.stream(side):
        for _ in range(300):
            b.copy_(a, non_blocking=True)
            b.mul_(1.0001)
    busy = step()
    side.synchronize()
    assert all(torch.equal(u, v) for u, v in zip(quiet, busy))
    assert _ffi.lib().cnsn_resident_timeouts() == 0
