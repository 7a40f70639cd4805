"""ABI 5: what used to be launches of their own in front of the op now travels WITH the op's launch.

  * `nn.BatchNorm1d.num_batches_tracked += 1` of SelfNorm's gate (models/cnsn.py:121,138 call the module, which counts in
    training mode) — done by the forward kernel through `cnsn_gate_t.num_batches_tracked`, whichever kernel family runs;
  * the batch permutation of CrossNorm (models/cnsn.py:62, `torch.randperm(N)` on the host) — handed to the
    cluster-resident kernels as a LAUNCH ARGUMENT (`cnsn_problem_t.perm_host`) instead of a host-to-device copy.

Both must be invisible in the results: same bits as the device-array path, same counters as torch's modules."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from cnsn_amd import functional as F_  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402

DEV = torch.device("cuda:0")


@pytest.fixture(autouse=True)
def auto():
    yield
    cnsn_amd.set_strategy("auto")


def _case(shape, dtype, crop, seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = (torch.randn(shape, device=DEV, generator=g) * 1.2 + 0.3).to(dtype)
    gy = torch.randn(shape, device=DEV, generator=g).to(dtype)
    return x, gy, cnsn_amd.draw_cn(shape, crop, 1)


def _run(x, gy, draws, kind, c, ctypes_path=False):
    sn = fill_sn(cnsn_amd.SelfNorm(c), 5, torch.float32).to(DEV).train() if kind == "cnsn" else None
    xg = x.clone().requires_grad_()
    if ctypes_path:                                   # the autograd.Function of functional.py, not the C++ glue
        kw, g, f = sn._fused_args() if sn is not None else ({}, None, None)
        cfg = cnsn_amd.FusedConfig(cn_active=True, content_box=draws.content_box, style_box=draws.style_box, **kw)
        ga = (g.fc_weight, g.bn_weight, g.bn_bias, g.running_mean, g.running_var) if g else (None,) * 5
        y = F_.FusedCNSN.apply(xg, cfg, draws.perm, None, *ga, *(None,) * 5, None, g.num_batches_tracked if g else None, None)
    else:
        mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm("both", 1), sn).to(DEV).train()
        mod.crossnorm.active = True
        mod.crossnorm.next_draws = draws
        y = mod(xg)
    y.backward(gy)
    torch.cuda.synchronize()
    out = [y.detach(), xg.grad]
    if sn is not None:
        out += [p.grad for p in sn.parameters()] + [sn.g_bn.running_mean, sn.g_bn.running_var, sn.g_bn.num_batches_tracked]
    return out


CASES = [((37, 6, 28, 28), torch.float32, "neither", "cnsn"), ((37, 6, 28, 28), torch.float32, "both", "cnsn"),
         ((40, 8, 56, 56), torch.bfloat16, "neither", "cnsn"), ((40, 8, 56, 56), torch.float32, "style", "cn"),
         ((9, 4, 128, 128), torch.float32, "style", "cn"),               # split planes (one plane per workgroup)
         ((9, 1024, 56, 56), torch.float32, "neither", "cnsn"),          # enough items for the pipelined kernels
         ((300, 2, 28, 28), torch.float32, "neither", "cnsn")]           # N > 256: a second stride of the index copy


@pytest.mark.parametrize("ctypes_path", [False, True], ids=["glue", "ctypes"])
@pytest.mark.parametrize("shape,dtype,crop,kind", CASES, ids=lambda v: str(v).replace(" ", "").replace("torch.", ""))
def test_permutation_as_launch_argument_gives_the_same_bits(shape, dtype, crop, kind, ctypes_path):
    x, gy, d = _case(shape, dtype, crop, 3)
    cfg = cnsn_amd.FusedConfig(cn_active=True, content_box=d.content_box, style_box=d.style_box, sn_active=kind == "cnsn")
    if shape[0] > 256 and cnsn_amd.which_path(x, cfg) != "resident":
        cnsn_amd.set_strategy("resident")                                 # (AUTO's choice at this batch size is not the point)
    assert cnsn_amd.which_path(x, cfg) == "resident" and cnsn_amd.which_path(x, cfg, True) == "resident"
    assert F_.perm_inline_ok(x, cfg, d.perm, None)                        # host int64 vector: rides in the launch arguments
    on_device = cnsn_amd.CNDraws(d.perm.to(DEV), d.style_box, None, d.content_box)
    assert not F_.perm_inline_ok(x, cfg, on_device.perm, None)            # a device tensor is used where it is
    a = _run(x, gy, d, kind, shape[1], ctypes_path)
    b = _run(x, gy, on_device, kind, shape[1], ctypes_path)
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), (shape, dtype, crop, kind, i)


def test_launch_argument_is_not_offered_where_the_kernels_read_the_array():
    d = cnsn_amd.draw_cn((64, 8, 14, 14), "style", 1)
    x = torch.randn(64, 8, 14, 14, device=DEV)                            # channel-in-registers kernels: device array
    cfg = cnsn_amd.FusedConfig(cn_active=True, style_box=d.style_box, sn_active=True)
    assert cnsn_amd.which_path(x, cfg) != "resident" and not F_.perm_inline_ok(x, cfg, d.perm, None)
    big = torch.randperm(2048)                                            # more indices than the launch arguments hold
    xb = torch.randn(2048, 1, 28, 28, device=DEV)
    assert not F_.perm_inline_ok(xb, cnsn_amd.FusedConfig(cn_active=True), big, None)
    y = cnsn_amd.cn_op_2ins_space_chan(xb, draws=cnsn_amd.CNDraws(big))
    ref = cnsn_amd.cn_op_2ins_space_chan(xb, draws=cnsn_amd.CNDraws(big.to(DEV)))
    assert torch.equal(y, ref)
    with pytest.raises(RuntimeError):                                     # an index outside the batch is refused, not read
        bad = torch.arange(37)
        bad[3] = 37
        cnsn_amd.cn_op_2ins_space_chan(torch.randn(37, 6, 28, 28, device=DEV), draws=cnsn_amd.CNDraws(bad))


SITES = [(37, 6, 28, 28), (64, 8, 14, 14), (128, 16, 7, 7), (128, 8, 8, 8), (5, 3, 9, 11), (6, 4, 128, 96), (3, 3, 224, 224)]


@pytest.mark.parametrize("strategy", ["auto", "two_pass", "resident", "local", "mono"])
@pytest.mark.parametrize("is_two", [False, True], ids=["one_gate", "two_gates"])
def test_counter_moves_inside_the_forward_launch(strategy, is_two):
    """num_batches_tracked after k training calls == k (each gate's own), untouched by eval calls and by backward,
    through every kernel family; and equal to what torch's BatchNorm1d modules of the oracle count"""
    cnsn_amd.set_strategy(strategy)
    for shape in SITES:
        for dtype in (torch.float32, torch.bfloat16):
            sn = cnsn_amd.SelfNorm(shape[1], is_two=is_two).to(DEV).train()
            x = torch.randn(shape, device=DEV).to(dtype).requires_grad_()
            for k in range(1, 4):
                y = sn(x)
                if k == 2:
                    y.float().sum().backward()
                assert int(sn.g_bn.num_batches_tracked) == k, (shape, dtype, strategy, k)
                if is_two:
                    assert int(sn.f_bn.num_batches_tracked) == k
            with torch.no_grad():
                sn(x)                                                     # torch counts under no_grad too
            assert int(sn.g_bn.num_batches_tracked) == 4
            sn.eval()
            sn(x)
            assert int(sn.g_bn.num_batches_tracked) == 4


def test_cumulative_average_and_frozen_counters_stay_on_the_host_path():
    """momentum=None needs the new count on the host (1 / num_batches_tracked): counted by the module layer as before;
    track_running_stats=False has no counter at all"""
    x = torch.randn(16, 4, 28, 28, device=DEV)
    sn = cnsn_amd.SelfNorm(4).to(DEV).train()
    sn.g_bn.momentum = None
    ref = torch.nn.BatchNorm1d(4, momentum=None).to(DEV).train()
    for _ in range(3):
        sn(x)
        ref(torch.randn(16, 4, 2, device=DEV))
    assert int(sn.g_bn.num_batches_tracked) == int(ref.num_batches_tracked) == 3
    free = cnsn_amd.SelfNorm(4).to(DEV).train()
    free.g_bn = torch.nn.BatchNorm1d(4, track_running_stats=False).to(DEV)
    free(x)
    assert free.g_bn.num_batches_tracked is None


def test_tail_counters():
    """the fused SelfNorm + next BatchNorm2d + ReLU launch counts for BOTH modules (cnsn_bn_tail_t.num_batches_tracked)"""
    c = 64
    mod = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(c)).to(DEV).train()
    bn = torch.nn.BatchNorm2d(c).to(DEV).train()
    x = torch.randn(128, c, 16, 16, device=DEV)
    cfg = cnsn_amd.FusedConfig(sn_active=True)
    assert F_.bnrelu_plan(x, cfg)
    for k in range(1, 4):
        mod.forward_block_bn(x, None, "none", bn)
        assert int(bn.num_batches_tracked) == k and int(mod.selfnorm.g_bn.num_batches_tracked) == k
    bn.eval()
    mod.forward_block_bn(x, None, "none", bn)
    assert int(bn.num_batches_tracked) == 3 and int(mod.selfnorm.g_bn.num_batches_tracked) == 4


@pytest.mark.parametrize("ctypes_path", [False, True], ids=["glue", "ctypes"])
def test_backward_uses_the_forward_pairing_when_the_caller_reuses_its_index_buffer(ctypes_path):
    """The launch-argument path keeps a SNAPSHOT of the host permutation for the backward (advisor, round 4): a caller
    that refills the same index tensor between forward and backward (`torch.randperm(n, out=buf)`) must get the gradient
    of the forward it ran — what the reference's device copy of `perm` (models/cnsn.py:62) guarantees by construction."""
    shape = (40, 8, 56, 56)
    x, gy, d = _case(shape, torch.float32, "neither", 11)
    cfg = cnsn_amd.FusedConfig(cn_active=True)
    assert F_.perm_inline_ok(x, cfg, d.perm, None)
    assert not F_.perm_inline_ok(x, cfg, d.perm[:-1].contiguous(), None)   # not one index per batch row: never inline

    def run(mutate):
        buf = d.perm.clone()
        xg = x.clone().requires_grad_()
        if ctypes_path:
            y = F_.FusedCNSN.apply(xg, cfg, buf, None, *(None,) * 10, None, None, None)
        else:
            y = F_.fused_cnsn(xg, cfg, perm=buf)
        if mutate:
            buf.copy_(torch.roll(buf, 1))
        y.backward(gy)
        torch.cuda.synchronize()
        return y.detach(), xg.grad

    ya, ga = run(False)
    yb, gb = run(True)
    assert torch.equal(ya, yb) and torch.equal(ga, gb)
