"""SelfNorm when its gate cannot be evaluated inside the fused launch (`SelfNorm._fusable`): the gate's BatchNorm1d
converted to nn.SyncBatchNorm (reference segmentation/tool/train_cnsn.py:160, `sync_bn`; SURVEY §8 f4's optional
statistic), and the two gates of `is_two` in different modes (the reference calls them as modules, models/cnsn.py:138,144).
The op is then composed from the library's building blocks (`SelfNorm._forward_composed`)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnsn_amd  # noqa: E402
from oracle import cnsn_oracle as orc  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402

DEV = torch.device("cuda:0")


def _close(got, want64, ref32, tol=1e-5, what=""):
    got, want64, ref32 = got.double().cpu(), want64.double(), ref32.double()
    scale = max(1.0, float(want64.abs().max()))
    e, e32 = float((got - want64).abs().max()), float((ref32 - want64).abs().max())
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):                       # observed error against its bound (evidence for profiles/)
        with open(os.path.join(keep, "sync_bn_margins.jsonl"), "a") as f:
            f.write(json.dumps(dict(what=what, err=e, err_oracle32=e32, scale=scale, bound=max(tol * scale, 2 * e32))) + "\n")
    assert e <= max(tol * scale, 2 * e32), (what, e, e32, scale)


def _oracle(c, is_two, seed, dt, x, gy, prepare):
    sn = fill_sn(orc.SelfNorm(c, is_two=is_two), seed, dt).train()
    prepare(sn)
    xr = x.detach().clone().to(dt).requires_grad_()
    y = sn(xr)
    y.backward(gy.to(dt))
    return y.detach(), xr.grad, {k: v.grad for k, v in sn.named_parameters() if v.grad is not None}, dict(sn.state_dict())


@pytest.mark.parametrize("shape", [(9, 5, 14, 14), (6, 4, 56, 56), (16, 8, 7, 7)], ids=str)
@pytest.mark.parametrize("case", ["sync_no_group", "f_eval", "g_eval", "sync_two"])
def test_composed_selfnorm_matches_the_oracle(shape, case):
    n, c = shape[:2]
    is_two = case != "sync_no_group"
    g = torch.Generator().manual_seed(11)
    # per-plane scales and offsets (the gate's BatchNorm over N divides by the spread of the plane statistics —
    # planes that all look alike make every implementation ill-conditioned), values exact in fp32 for both sides
    x = torch.randn(shape, generator=g, dtype=torch.float64) * (torch.rand(n, c, 1, 1, generator=g, dtype=torch.float64) * 1.5 + 0.5)
    x = (x + torch.randn(n, c, 1, 1, generator=g, dtype=torch.float64)).float().double()
    gy = torch.randn(shape, generator=g, dtype=torch.float64).float().double()

    def prepare(sn):
        if case == "f_eval":
            sn.f_bn.eval()
        elif case == "g_eval":
            sn.g_bn.eval()

    t64 = _oracle(c, is_two, 5, torch.float64, x, gy, prepare)
    o32 = _oracle(c, is_two, 5, torch.float32, x, gy, prepare)
    sn = fill_sn(cnsn_amd.SelfNorm(c, is_two=is_two), 5, torch.float32)
    if case.startswith("sync"):       # without a process group nn.SyncBatchNorm evaluates the local batch (torch semantics)
        sn = torch.nn.SyncBatchNorm.convert_sync_batchnorm(sn)
    sn = sn.to(DEV).train()
    prepare(sn)
    assert not sn._fusable()
    xg = x.detach().clone().float().to(DEV).requires_grad_()
    for wrap in (False, True):        # alone, and as the SelfNorm of a CNSN site whose CrossNorm is idle
        if wrap:
            for p in sn.parameters():
                p.grad = None
            xg.grad = None
            sn2 = fill_sn(cnsn_amd.SelfNorm(c, is_two=is_two), 5, torch.float32)
            if case.startswith("sync"):
                sn2 = torch.nn.SyncBatchNorm.convert_sync_batchnorm(sn2)
            sn = sn2.to(DEV)
            site = cnsn_amd.CNSN(cnsn_amd.CrossNorm("neither", 1), sn).to(DEV).train()
            prepare(sn)                   # (after the site's .train(): that call reaches every sub-module)
            y = site(xg)
        else:
            y = sn(xg)
        y.backward(gy.float().to(DEV))
        torch.cuda.synchronize()
        _close(y.detach(), t64[0], o32[0], what=f"{case} {shape} y")
        _close(xg.grad, t64[1], o32[1], 1e-5, what=f"{case} {shape} dx")
        grads = {k: v.grad for k, v in sn.named_parameters() if v.grad is not None}
        assert set(grads) == set(t64[2])
        for k in grads:
            _close(grads[k], t64[2][k], o32[2][k], 1e-5, what=f"{case} {shape} grad {k}")
        st = sn.state_dict()
        for k in t64[3]:
            if "num_batches" in k:
                assert int(st[k]) == int(t64[3][k]), k
            else:
                _close(st[k], t64[3][k], o32[3][k], what=f"{case} {shape} state {k}")


def test_armed_site_with_a_sync_gate_takes_the_reference_control_flow():
    """CNSN with CrossNorm armed and a SyncBatchNorm gate: CrossNorm (its own fused call), then the composed SelfNorm —
    the same draws, `active` dropped (models/cnsn.py:159-164)."""
    shape = (8, 6, 28, 28)
    torch.manual_seed(3)
    np.random.seed(3)
    x = torch.randn(shape, device=DEV)
    d = cnsn_amd.draw_cn(shape, "both", 1)
    sn = torch.nn.SyncBatchNorm.convert_sync_batchnorm(fill_sn(cnsn_amd.SelfNorm(6), 2, torch.float32)).to(DEV).train()
    site = cnsn_amd.CNSN(cnsn_amd.CrossNorm("both", 1), sn).to(DEV).train()
    site.crossnorm.active = True
    site.crossnorm.next_draws = d
    y = site(x)
    assert site.crossnorm.active is False
    fused = cnsn_amd.CNSN(cnsn_amd.CrossNorm("both", 1), fill_sn(cnsn_amd.SelfNorm(6), 2, torch.float32)).to(DEV).train()
    fused.crossnorm.active = True
    fused.crossnorm.next_draws = d
    want = fused(x)                     # one launch; without a process group the two statistics are the same
    assert float((y - want).detach().abs().max()) <= 2e-5 * float(want.detach().abs().max())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_sync_gate_statistic_spans_the_global_batch(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_syncbn_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    reps = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        json.dump(reps, open(os.path.join(keep, "sync_gate.json"), "w"), indent=1)
    for rep in reps:
        assert rep["world"] == 2 and len(rep["cases"]) == 3
        for case in rep["cases"]:
            for k, e in case["errs"].items():      # composed (fp32 gate, fp32 plane sums) against the fused launch (fp64 inside)
                assert e <= 1e-5, (case["shape"], k, e)
            assert case["local_vs_global"] > 1e-3, case      # half-batch statistics would have given something else
