"""SelfNorm-only cluster kernels (csrc/cnsn_resident_sn_kernels.h, round 3): the workgroups of a channel exchange the
PARTIAL batch moments of BatchNorm1d's input (models/cnsn.py:121,138) — 4 floats per workgroup — instead of every plane's
statistics; pipelined like the general resident kernels; optional residual-block epilogue (PRE add, ReLU:
models/imagenet/resnet_cnsn.py:117-122).

Checked here, every instantiation class (register buckets 2..16, fp32 / bf16 / fp16, with and without the epilogue):
  * against the oracle in fp32 and fp64 (same bars as tests/test_gpu_full_size.py and test_gpu_fused_block.py),
  * against the two-pass strategy on the same inputs (tight: same algebra, different summation order of the batch),
  * the `saved` contract in both directions (this forward -> another strategy's backward and vice versa),
  * partial last members (N = 5, 37), more than 64 members per channel (N = 260), pipelines that fill (384 channels),
  * replay from a HIP graph (untagged granules), the persistent context on / off, and the give-up path (NaN, counter)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from cnsn_amd.functional import FusedConfig  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402
from tests.test_gpu_full_size import check_case  # noqa: E402

DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
# (tag, H, W): one-slot planes (33..64 vectors; 8-byte vectors in 16 bits: the 14x14 class, 16 / 8 planes per wave), then the
# register buckets 2, 4, 7, 8, 13, 16 of 16-byte vectors
PLANES = [("f32", 14, 14), ("bf16", 14, 14), ("f16", 14, 14), ("f32", 16, 16), ("f32", 12, 32), ("f32", 28, 32), ("f32", 40, 40), ("f32", 60, 32), ("f32", 56, 56), ("f32", 60, 64),
          ("bf16", 12, 64), ("bf16", 28, 64), ("bf16", 56, 56), ("bf16", 60, 64), ("bf16", 96, 64), ("bf16", 120, 64),
          ("f16", 28, 28), ("f16", 56, 56), ("f16", 96, 64)]


@pytest.fixture(autouse=True)
def forced():
    old = {k: os.environ.get(k) for k in ("CNSN_SNX", "CNSN_CONTEXT", "CNSN_FAULT_INJECT", "CNSN_WAIT_MS")}
    os.environ["CNSN_SNX"] = "2"
    cnsn_amd.set_strategy("resident")
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    cnsn_amd.set_strategy("auto")


def cfg_of(mode="none", relu=False):
    return FusedConfig(sn_active=True, sn_training=True, add_mode=mode, relu=relu)


def test_the_plan_takes_what_it_should():
    x = torch.empty(37, 3, 56, 56, device="cuda")
    assert cnsn_amd.sn_cluster(x, cfg_of()) and cnsn_amd.sn_cluster(x, cfg_of(), backward=True)
    assert cnsn_amd.sn_cluster(x, cfg_of("pre", True)) and cnsn_amd.sn_cluster(x, cfg_of("pre", True), backward=True)
    assert not cnsn_amd.sn_cluster(x, cfg_of("post", True))
    assert not cnsn_amd.sn_cluster(x, FusedConfig(sn_active=True, sn_training=False))          # inference: no coupling
    assert not cnsn_amd.sn_cluster(x, FusedConfig(sn_active=True, sn_two=True))
    assert not cnsn_amd.sn_cluster(x, FusedConfig(cn_active=True, sn_active=True))
    os.environ["CNSN_SNX"] = "0"
    assert not cnsn_amd.sn_cluster(x, cfg_of())
    os.environ["CNSN_SNX"] = "2"
    cnsn_amd.set_strategy("two_pass")
    assert not cnsn_amd.sn_cluster(x, cfg_of())


@pytest.mark.parametrize("tag,h,w", PLANES, ids=lambda v: str(v))
@pytest.mark.parametrize("n", [5, 37])
def test_against_the_oracle(tag, h, w, n):
    x = torch.empty(n, 4, h, w, device="cuda", dtype=DT[tag])
    assert cnsn_amd.sn_cluster(x, cfg_of()) and cnsn_amd.sn_cluster(x, cfg_of(), backward=True)
    check_case((n, 4, h, w), DT[tag], "sn", "neither", 40 + n)
    assert cnsn_amd.lib().cnsn_resident_timeouts() == 0


def run(shape, dtype, seed, mode="none", relu=False):
    n, c = shape[:2]
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (torch.randn(shape, device="cuda", generator=g) * (torch.rand(n, c, 1, 1, device="cuda", generator=g) * 1.5 + 0.5)
         + torch.randn(n, c, 1, 1, device="cuda", generator=g)).to(dtype).requires_grad_()
    b = (torch.randn(shape, device="cuda", generator=g) * 0.7).to(dtype).requires_grad_() if mode != "none" else None
    gy = torch.randn(shape, device="cuda", generator=g).to(dtype)
    mod = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(c), seed, torch.float32)).cuda().train()
    y = mod.forward_block(x, b, add_mode=mode, relu=relu) if (mode != "none" or relu) else mod(x)
    grads = torch.autograd.grad(y, [x] + ([b] if b is not None else []) + list(mod.parameters()), gy)
    torch.cuda.synchronize()
    return [y.detach()] + [t.detach() for t in grads] + [t.clone() for t in mod.buffers()]


def close(a, b, dtype, what, params=False):
    a, b = a.double(), b.double()
    scale = max(float(b.abs().max()), 1e-3)
    # fp32: 2e-6 of the largest entry for the planes; 5e-6 for the parameter gradients — sums over the whole batch with
    # cancellation, fed by per-plane statistics that the two strategies round differently in the last bit (the second pass
    # of the cluster kernels is taken about sum * rcp(M), the two-pass kernels' about sum / M)
    tol = (5e-6 if params else 2e-6) if dtype == torch.float32 else 1e-2
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: {err:.3e} vs scale {scale:.3g}"


@pytest.mark.parametrize("tag,h,w", PLANES, ids=lambda v: str(v))
@pytest.mark.parametrize("mode,relu", [("none", False), ("pre", True), ("pre", False), ("none", True)])
def test_against_the_two_pass_kernels(tag, h, w, mode, relu):
    """same inputs through the two-pass strategy: same per-plane algebra, a different summation order over the batch.
    (With a ReLU the masks agree except where the pre-activation is within rounding of zero: compare through the
    outputs' own masks.)"""
    shape = (37, 5, h, w)
    dtype = DT[tag]
    out = run(shape, dtype, 7, mode, relu)
    cnsn_amd.set_strategy("two_pass")
    ref = run(shape, dtype, 7, mode, relu)
    assert cnsn_amd.lib().cnsn_resident_timeouts() == 0
    names = ["y", "dx"] + (["db"] if mode != "none" else []) + ["dw", "dgamma", "dbeta", "rm", "rv", "nbt"]
    if relu:   # gradient entries where the two masks differ are not comparable
        same = ((out[0] > 0) == (ref[0] > 0))
        assert float((~same).float().mean()) < 1e-3
        for i in (1, 2) if mode != "none" else (1,):
            out[i] = torch.where(same, out[i], torch.zeros_like(out[i]))
            ref[i] = torch.where(same, ref[i], torch.zeros_like(ref[i]))
    for nm, a, b in zip(names, out, ref):
        if relu and nm in ("dw", "dgamma", "dbeta") and float(((out[0] > 0) != (ref[0] > 0)).float().sum()) > 0:
            continue  # a flipped mask element moves the parameter sums by more than rounding
        close(a, b, dtype, f"{tag} {h}x{w} {mode}/{relu} {nm}", params=nm in ("dw", "dgamma", "dbeta"))


@pytest.mark.parametrize("tag,h,w", [("f32", 56, 56), ("bf16", 56, 56), ("bf16", 28, 28), ("f32", 28, 28), ("bf16", 14, 14),
                                     ("f32", 14, 14)], ids=lambda v: str(v))
@pytest.mark.parametrize("mode,relu", [("pre", True), ("none", True), ("pre", False)])
def test_block_against_the_oracle(tag, h, w, mode, relu):
    from tests.test_gpu_fused_block import check, run_case
    shape = (37, 6, h, w)
    x = torch.empty(shape, device="cuda", dtype=DT[tag])
    assert cnsn_amd.sn_cluster(x, cfg_of(mode, relu)) and cnsn_amd.sn_cluster(x, cfg_of(mode, relu), backward=True)
    out = run_case(shape, "sn", "neither", mode, relu, DT[tag], 91)
    check(out, DT[tag], relu, f"sn-cluster {shape} {tag} {mode}/{relu}")
    assert cnsn_amd.lib().cnsn_resident_timeouts() == 0


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_saved_contract_both_ways(tag):
    """this family's forward feeds the two-pass backward and vice versa (cnsn_layout.h: one contract for `saved`)"""
    shape, dtype = (37, 6, 56, 56), DT[tag]
    want = run(shape, dtype, 5, "pre", True)

    def mixed(fwd_strategy, bwd_strategy):
        n, c = shape[:2]
        g = torch.Generator(device="cuda").manual_seed(5)
        x = (torch.randn(shape, device="cuda", generator=g) * (torch.rand(n, c, 1, 1, device="cuda", generator=g) * 1.5 + 0.5)
             + torch.randn(n, c, 1, 1, device="cuda", generator=g)).to(dtype).requires_grad_()
        b = (torch.randn(shape, device="cuda", generator=g) * 0.7).to(dtype).requires_grad_()
        gy = torch.randn(shape, device="cuda", generator=g).to(dtype)
        mod = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(c), 5, torch.float32)).cuda().train()
        cnsn_amd.set_strategy(fwd_strategy)
        y = mod.forward_block(x, b, add_mode="pre", relu=True)
        cnsn_amd.set_strategy(bwd_strategy)          # (the backward plans from the strategy current at ITS call)
        import cnsn_amd.functional as F
        grads = torch.autograd.grad(y, [x, b] + list(mod.parameters()), gy)
        return [y.detach()] + list(grads)

    for fs, bs in (("resident", "two_pass"), ("two_pass", "resident")):
        got = mixed(fs, bs)
        same = (got[0] > 0) == (want[0] > 0)
        for i, (a, b) in enumerate(zip(got, want)):
            if i in (1, 2):
                a, b = torch.where(same, a, torch.zeros_like(a)), torch.where(same, b, torch.zeros_like(b))
            if i >= 3 and not bool(same.all()):
                continue
            close(a, b, dtype, f"{tag} fwd {fs} / bwd {bs} output {i}")


def test_more_than_64_members_per_channel():
    """N = 260 at one plane per wave: K = 65 members — the merge loops over the partials"""
    check_case((260, 2, 56, 56), torch.float32, "sn", "neither", 3)
    check_case((260, 2, 28, 32), torch.bfloat16, "sn", "neither", 4)


@pytest.mark.parametrize("tag", ["f32", "bf16"])
@pytest.mark.parametrize("n", [70, 260])
def test_one_slot_planes_partial_members(tag, n):
    """14x14: 64 (16-bit) / 32 (fp32) planes per workgroup, the member's partial is summed lane-parallel; N = 70, 260 leave
    a last member with 6 / 4 planes"""
    check_case((n, 3, 14, 14), DT[tag], "sn", "neither", 50 + n)
    assert cnsn_amd.lib().cnsn_resident_timeouts() == 0


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_one_slot_planes_full_pipeline(tag):
    """14x14 with several channels per cluster (the pipeline fills and the grid wraps), deterministic"""
    shape = (130, 1500, 14, 14)
    check_case(shape, DT[tag], "sn", "neither", 77)
    out = run(shape, DT[tag], 5, "pre", True)
    again = run(shape, DT[tag], 5, "pre", True)
    for a, b in zip(out, again):
        assert torch.equal(a, b)
    assert cnsn_amd.lib().cnsn_resident_timeouts() == 0


@pytest.mark.parametrize("tag,h,w", [("f32", 56, 56), ("bf16", 56, 56), ("bf16", 28, 28), ("f32", 28, 28)], ids=lambda v: str(v))
def test_full_pipeline_many_channels(tag, h, w):
    """several items per workgroup, the grid wraps around; N = 37: a partial last member in every cluster"""
    check_case((37, 384, h, w), DT[tag], "sn", "neither", 31)
    out = run((37, 384, h, w), DT[tag], 5, "pre", True)
    again = run((37, 384, h, w), DT[tag], 5, "pre", True)
    for a, b in zip(out, again):
        assert torch.equal(a, b)                     # deterministic: fixed summation orders, no atomics
    assert cnsn_amd.lib().cnsn_resident_timeouts() == 0


@pytest.mark.parametrize("tag,h,w", PLANES, ids=lambda v: str(v))
@pytest.mark.parametrize("mode,relu", [("pre", True), ("none", False)])
def test_every_class_is_deterministic(tag, h, w, mode, relu):
    """Run-to-run bit equality of every output, every instantiation class, several items per cluster.  (What it guards:
    these kernels overlap stores, LDS traffic and loads slot by slot — a store whose data registers were overwritten by the
    next VALU instruction showed up exactly here, as differences in a few lanes of one slot: DESIGN §4.2h.)"""
    shape = (37, 300, h, w) if h * w <= 1024 else (21, 96, h, w)
    a = run(shape, DT[tag], 11, mode, relu)
    for _ in range(2):
        b = run(shape, DT[tag], 11, mode, relu)
        for i, (u, v) in enumerate(zip(a, b)):
            assert torch.equal(u, v), f"{tag} {h}x{w} {mode}/{relu}: output {i} differs between two runs"
    assert cnsn_amd.lib().cnsn_resident_timeouts() == 0


def test_context_on_off_same_bits():
    """tagged granules in the persistent context and untagged pairs in the workspace carry the same floats"""
    shape = (37, 12, 56, 56)
    os.environ["CNSN_CONTEXT"] = "1"
    a = run(shape, torch.float32, 9, "pre", True)
    os.environ["CNSN_CONTEXT"] = "0"
    b = run(shape, torch.float32, 9, "pre", True)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


def test_replay_from_a_graph():
    """under capture the exchange goes through the workspace (a captured launch would replay its tag)"""
    shape = (37, 6, 56, 56)
    mod = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(6), 5, torch.float32)).cuda().train()
    ref = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(6), 5, torch.float32)).cuda().train()
    x = torch.randn(shape, device="cuda", requires_grad=True)
    b = torch.randn(shape, device="cuda", requires_grad=True)
    gy = torch.randn(shape, device="cuda")
    params = list(mod.parameters())
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            torch.autograd.grad(mod.forward_block(x, b, add_mode="pre", relu=True), [x, b] + params, gy)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        grads = torch.autograd.grad(mod.forward_block(x, b, add_mode="pre", relu=True), [x, b] + params, gy)
    for _ in range(2):
        torch.autograd.grad(ref.forward_block(x, b, add_mode="pre", relu=True), [x, b] + list(ref.parameters()), gy)
    for _ in range(3):
        g.replay()
        want = torch.autograd.grad(ref.forward_block(x, b, add_mode="pre", relu=True), [x, b] + list(ref.parameters()), gy)
    torch.cuda.synchronize()
    for a, w_ in zip(grads, want):
        assert torch.equal(a, w_)
    assert torch.equal(mod.selfnorm.g_bn.running_var, ref.selfnorm.g_bn.running_var)
    assert cnsn_amd.lib().cnsn_resident_timeouts() == 0


def test_auto_takes_the_resnet50_blocks():
    """AUTO (no forcing): the fused blocks of BASELINE configs[2] at 56x56 and 28x28 run this family, both directions"""
    os.environ.pop("CNSN_SNX", None)
    cnsn_amd.set_strategy("auto")
    for shape in ((256, 256, 56, 56), (256, 512, 28, 28), (96, 256, 56, 56), (96, 512, 28, 28)):
        x = torch.empty(shape, device="cuda", dtype=torch.bfloat16)
        for bw in (False, True):
            assert cnsn_amd.sn_cluster(x, cfg_of("pre", True), backward=bw), (shape, bw)
    # one-slot planes: ahead of the channel-in-registers kernels where measured faster (large batches; fp32 backward)
    x = torch.empty((512, 64, 14, 14), device="cuda", dtype=torch.bfloat16)
    assert cnsn_amd.sn_cluster(x, cfg_of()) and cnsn_amd.sn_cluster(x, cfg_of("pre", True), backward=True)
    x = torch.empty((256, 64, 14, 14), device="cuda", dtype=torch.float32)
    assert cnsn_amd.sn_cluster(x, cfg_of(), backward=True) and cnsn_amd.sn_cluster(x, cfg_of())   # (forward too since round 4)
    x = torch.empty((256, 64, 14, 14), device="cuda", dtype=torch.bfloat16)
    assert not cnsn_amd.sn_cluster(x, cfg_of()) and not cnsn_amd.sn_cluster(x, cfg_of(), backward=True)
