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


def test_resident_kernels_under_data_parallel_communication(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_ddp_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    reps = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):                                   # evidence for profiles/
        json.dump(reps, open(os.path.join(keep, "ddp_concurrency.json"), "w"), indent=1)
    for rep in reps:
        assert rep["world"] == 2 and rep["sites"] == 16
        assert "resident" in rep["paths"], rep                 # the cluster kernels really were in play
        assert rep["timeouts"] == 0, rep                       # no bounded wait ran out
        assert rep["site_outputs_bit_identical"], rep          # communication changes nothing in the op's results
        assert rep["finite"]
        if rep["mode"] == "ddp":
            assert rep["grads_bit_identical"], rep             # (a + b) / 2 either way: exact
