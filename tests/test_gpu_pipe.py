"""Pipelined forward of the cluster-resident strategy (csrc/cnsn_resident_pipe_kernels.h): the next item's planes are
loaded while the current item's channel is exchanged; the current item waits in LDS.  Same arithmetic as the plain
kernel: outputs, saved state (seen through the gradients) and running statistics must be BIT-IDENTICAL to the plain
kernel's (CNSN_PIPE=0), for every instantiation (CNSN_PIPE=2 lifts the profitability rules), and agree with the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402

DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
# (tag, H, W): register buckets 2, 4, 7, 8, 13, 16 of 16-byte vectors (2 and 4: four planes per wave in the forward)
PLANES = [("f32", 12, 32), ("f32", 28, 32), ("bf16", 12, 64), ("bf16", 28, 64), ("f16", 28, 64), ("f32", 40, 40), ("f32", 60, 32), ("f32", 56, 56), ("f32", 60, 64), ("bf16", 56, 56), ("bf16", 60, 64),
          ("bf16", 96, 64), ("bf16", 120, 64), ("f16", 56, 56), ("f16", 96, 64)]


@pytest.fixture(autouse=True)
def restore_env():
    old = os.environ.get("CNSN_PIPE")
    old_snx = os.environ.get("CNSN_SNX")
    os.environ["CNSN_SNX"] = "0"       # SelfNorm alone: the general kernels, not tests/test_gpu_sn_cluster.py's
    yield
    if old_snx is None:
        os.environ.pop("CNSN_SNX", None)
    else:
        os.environ["CNSN_SNX"] = old_snx
    if old is None:
        os.environ.pop("CNSN_PIPE", None)
    else:
        os.environ["CNSN_PIPE"] = old
    cnsn_amd.set_strategy("auto")


def run(shape, dtype, kind, crop, seed, training=True, is_two=False):
    n, c = shape[:2]
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (torch.randn(shape, device="cuda", generator=g) * (torch.rand(n, c, 1, 1, device="cuda", generator=g) * 1.5 + 0.5)
         + torch.randn(n, c, 1, 1, device="cuda", generator=g)).to(dtype).requires_grad_()
    gy = torch.randn(shape, device="cuda", generator=g).to(dtype)
    cn = cnsn_amd.CrossNorm(crop, 1) if kind != "sn" else None
    sn = cnsn_amd.SelfNorm(c, is_two=is_two) if kind != "cn" else None
    mod = (cnsn_amd.CNSN(cn, sn) if sn is not None else cn).cuda()
    mod.train(training)
    torch.manual_seed(seed)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(torch.randn_like(p) * 0.5)
    if cn is not None:
        cn.active = True
    np.random.seed(seed)
    torch.manual_seed(seed)
    y = mod(x)
    grads = torch.autograd.grad(y, [x] + list(mod.parameters()), gy)
    stats = [b.clone() for b in mod.buffers()]
    return [y.detach()] + list(grads) + stats


@pytest.mark.parametrize("tag,h,w", PLANES, ids=lambda v: str(v))
@pytest.mark.parametrize("kind,crop", [("sn", "neither"), ("cnsn", "neither"), ("cnsn", "both"), ("cn", "style")])
@pytest.mark.parametrize("n", [5, 37])
def test_bit_identical_to_the_plain_kernel(tag, h, w, kind, crop, n):
    cnsn_amd.set_strategy("resident")
    shape = (n, 3, h, w)
    os.environ["CNSN_PIPE"] = "0"
    ref = run(shape, DT[tag], kind, crop, 11 + n)
    os.environ["CNSN_PIPE"] = "2"
    out = run(shape, DT[tag], kind, crop, 11 + n)
    assert cnsn_amd.lib().cnsn_resident_timeouts() == 0
    for i, (a, b) in enumerate(zip(ref, out)):
        assert torch.isfinite(b.float()).all()
        assert torch.equal(a, b), (tag, shape, kind, crop, i, (a.float() - b.float()).abs().max().item())


def test_two_gate_selfnorm_bit_identical():
    """SelfNorm(is_two=True) (cnsn.py:121-150, both gates): the second gate's rows travel through the pipelined kernels too."""
    cnsn_amd.set_strategy("resident")
    for tag, h, w in (("f32", 56, 56), ("bf16", 56, 56), ("f32", 28, 32)):
        for kind, crop in (("sn", "neither"), ("cnsn", "both")):
            os.environ["CNSN_PIPE"] = "0"
            ref = run((37, 3, h, w), DT[tag], kind, crop, 21, is_two=True)
            os.environ["CNSN_PIPE"] = "2"
            out = run((37, 3, h, w), DT[tag], kind, crop, 21, is_two=True)
            for a, b in zip(ref, out):
                assert torch.equal(a, b), (tag, h, w, kind, crop)


# many channels: the pipeline fills (several items per workgroup, the grid wraps around); checked against the oracle's
# eager ops and against the plain kernel; N = 37: a partial last member in every cluster
@pytest.mark.parametrize("tag,h,w", [("f32", 56, 56), ("f32", 40, 40), ("bf16", 56, 56)], ids=lambda v: str(v))
@pytest.mark.parametrize("kind,crop", [("sn", "neither"), ("cnsn", "neither"), ("cnsn", "both")])
def test_full_pipeline_many_channels(tag, h, w, kind, crop):
    from tests.test_gpu_full_size import check_case
    os.environ["CNSN_PIPE"] = "2"
    cnsn_amd.set_strategy("resident")
    check_case((37, 384, h, w), DT[tag], kind, crop, 31)
    shape = (37, 384, h, w)
    out = run(shape, DT[tag], kind, crop, 5)
    os.environ["CNSN_PIPE"] = "0"
    ref = run(shape, DT[tag], kind, crop, 5)
    for a, b in zip(ref, out):
        assert torch.equal(a, b)


def test_default_rule_takes_the_headline_shape():
    """AUTO: the un-boxed north-star call runs the pipelined kernel (checked through its LDS footprint: the launch
    asks for more than 48 KB of dynamic LDS, which only the pipelined kernel does) — indirectly: results equal the
    plain kernel's and the call is faster is not asserted here; this checks the plumbing does not fall over at
    full size with the persistent context in use."""
    os.environ.pop("CNSN_PIPE", None)
    cnsn_amd.set_strategy("auto")
    shape = (256, 256, 56, 56)
    out = run(shape, torch.float32, "sn", "neither", 3)
    os.environ["CNSN_PIPE"] = "0"
    ref = run(shape, torch.float32, "sn", "neither", 3)
    for a, b in zip(ref, out):
        assert torch.equal(a, b)
    assert cnsn_amd.lib().cnsn_resident_timeouts() == 0


def test_pipelined_kernels_replay_from_a_graph():
    """Under stream capture the launch-tagged exchange context is not used (a captured launch would replay its tag): the
    pipelined kernels then gather UNTAGGED granules through the scalar path.  A captured forward+backward of the
    13-slot class (both directions pipelined under CNSN_PIPE=2) replays bit-identically to the eager plain kernels."""
    from tests.golden.gen_golden_fill import fill_sn
    os.environ["CNSN_PIPE"] = "2"
    cnsn_amd.set_strategy("resident")
    shape = (37, 6, 56, 56)
    mod = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(6), 5, torch.float32)).cuda().train()
    ref = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(6), 5, torch.float32)).cuda().train()
    x = torch.randn(shape, device="cuda", requires_grad=True)
    gy = torch.randn(shape, device="cuda")
    params = list(mod.parameters())
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            torch.autograd.grad(mod(x), [x] + params, gy)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        grads = torch.autograd.grad(mod(x), [x] + params, gy)
    os.environ["CNSN_PIPE"] = "0"
    for _ in range(2):                                   # the reference module caught up on the two warm-up steps
        torch.autograd.grad(ref(x), [x] + list(ref.parameters()), gy)
    for _ in range(3):
        g.replay()
        want = torch.autograd.grad(ref(x), [x] + list(ref.parameters()), gy)
    torch.cuda.synchronize()
    for a, b in zip(grads, want):
        assert torch.equal(a, b)
    assert torch.equal(mod.selfnorm.g_bn.running_var, ref.selfnorm.g_bn.running_var)
    assert cnsn_amd.lib().cnsn_resident_timeouts() == 0


# ---- granule regions instead of the fill launch (CNSN_PONG: 0 never, 2 also for tensors under 64 MiB) --------------
_PONG_SEQ = [  # (shape, dtype tag, kind, crop): extents that grow and shrink, boxed and not, the split-plane and the SelfNorm-only kernels in between
    ((37, 3, 56, 56), "f32", "cnsn", "neither"), ((37, 3, 56, 56), "f32", "cnsn", "neither"), ((64, 6, 28, 32), "f32", "cnsn", "both"),
    ((5, 3, 56, 56), "bf16", "cnsn", "neither"), ((16, 4, 128, 128), "f32", "cnsn", "style"), ((37, 3, 56, 56), "f32", "sn", "neither"),
    ((64, 6, 28, 32), "bf16", "cn", "style"), ((37, 8, 40, 40), "f32", "cnsn", "content"), ((37, 3, 56, 56), "f32", "cnsn", "neither"),
]


def _pong_sequence(pipe):
    os.environ["CNSN_PIPE"] = pipe
    outs = []
    for rep in range(2):
        for i, (shape, tag, kind, crop) in enumerate(_PONG_SEQ):
            outs.append(run(shape, DT[tag], kind, crop, 100 + 10 * rep + i))
    torch.cuda.synchronize()
    return outs


@pytest.mark.parametrize("pipe", ["2", "0"], ids=["pipelined", "plain"])
def test_granule_regions_give_the_same_bits_as_fill_launches(pipe):
    """The same sequence of calls — shapes whose exchange extents grow and shrink, with and without crop boxes, forward and
    backward, other kernel families in between — with a fill launch per cluster launch (CNSN_PONG=0) and through the
    context's two granule regions (CNSN_PONG=2: small tensors too): every output, gradient and running statistic
    bit-identical, no bounded wait running out; then once more — on the ctypes path after the context has been re-created
    (cnsn_context_init forgets the regions' state), with the C++ glue simply continuing on the same context."""
    cnsn_amd.set_strategy("resident")
    old = os.environ.get("CNSN_PONG")
    try:
        os.environ["CNSN_PONG"] = "0"
        ref = _pong_sequence(pipe)
        os.environ["CNSN_PONG"] = "2"
        got = _pong_sequence(pipe)
        from cnsn_amd import functional as F
        F._contexts.clear()                       # the Python layer's context: a new buffer, cnsn_context_init again
        again = _pong_sequence(pipe)
    finally:
        if old is None:
            os.environ.pop("CNSN_PONG", None)
        else:
            os.environ["CNSN_PONG"] = old
    assert cnsn_amd.lib().cnsn_resident_timeouts() == 0
    for seq in (got, again):
        assert len(seq) == len(ref)
        for a, b in zip(ref, seq):
            for u, v in zip(a, b):
                assert torch.equal(u, v)
