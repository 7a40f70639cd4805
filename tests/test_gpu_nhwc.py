"""Channels-last ("NHWC") calls are computed where the tensor lies (`cnsn_problem_t.layout = CNSN_LAYOUT_NHWC`, round 5).

The reference takes any layout through `.contiguous()` (models/cnsn.py:14,16) — values do not depend on the memory order.  So a
channels-last call must give what the oracle gives for the same VALUES (north_star's tolerances: 1e-5 fp32, 1e-2 bf16), outputs and
gradients must come back channels-last (the next convolution's layout), and calls the channels-last kernels do not take (crop
boxes, a channel count that is no whole number of vectors) must fall back to the NCHW copy silently."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from cnsn_amd import _ffi  # noqa: E402
from cnsn_amd import functional as F_  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402
from tests.test_gpu_fused_block import check, run_case  # noqa: E402
from tests.test_gpu_parity import DEV, seed_of  # noqa: E402

CL = torch.channels_last
SHAPES = [(4, 8, 8, 8),        # one column block, many rows per workgroup
          (6, 16, 9, 11),      # odd plane, pixel tail
          (5, 64, 28, 28),     # several pixel chunks
          (37, 8, 56, 56),     # the north-star plane, odd batch
          (9, 2048, 7, 7),     # stage-4 site: 512 (fp32) / 256 (16-bit) vector columns -> column blocks
          (3, 520, 6, 5)]      # a channel count that is no power of two (130 / 65 vector columns)
EPILOGUES = [("pre", True), ("pre", False), ("post", True), ("post", False), ("none", True), ("none", False)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("kind", ["sn", "cnsn", "cn"])
@pytest.mark.parametrize("mode,relu", EPILOGUES)
def test_channels_last_block_fp32(shape, kind, mode, relu):
    seed = seed_of(shape, kind, "neither", mode, relu)
    check(run_case(shape, kind, "neither", mode, relu, torch.float32, seed, channels_last=True), torch.float32, relu,
          (shape, kind, mode, relu, "channels_last"))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(8, 16, 28, 28), (5, 64, 14, 14), (9, 2048, 7, 7)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("kind", ["sn", "cnsn"])
@pytest.mark.parametrize("mode,relu", [("pre", True), ("post", True), ("none", False)])
def test_channels_last_block_16bit(dtype, shape, kind, mode, relu):
    seed = seed_of(shape, kind, "neither", mode, relu, str(dtype))
    check(run_case(shape, kind, "neither", mode, relu, dtype, seed, channels_last=True), dtype, relu,
          (shape, kind, mode, relu, dtype, "channels_last"))


@pytest.mark.parametrize("ctypes_path", [False, True], ids=["glue", "ctypes"])
def test_channels_last_equals_the_nchw_path_and_reports_its_kernels(ctypes_path):
    """the same values in both memory orders: outputs / gradients agree to rounding, parameter gradients and running statistics
    too; which_path says 'streaming' for the channels-last call; both glue paths"""
    shape = (16, 64, 28, 28)
    torch.manual_seed(2)
    x = torch.randn(shape, device=DEV) * 1.4 + 0.3
    add = torch.randn(shape, device=DEV) * 0.5
    gy = torch.randn(shape, device=DEV)

    def run(fmt):
        sn = fill_sn(cnsn_amd.SelfNorm(64), 4, torch.float32).to(DEV).train()
        xg = x.detach().clone(memory_format=fmt).requires_grad_()
        ag = add.detach().clone(memory_format=fmt).requires_grad_()
        kw, g, f = sn._fused_args()
        cfg = cnsn_amd.FusedConfig(add_mode="pre", relu=True, **kw)
        if ctypes_path:
            y = F_.FusedCNSN.apply(xg, cfg, None, None, g.fc_weight, g.bn_weight, g.bn_bias, g.running_mean, g.running_var,
                                   *(None,) * 5, ag, g.num_batches_tracked, None)
        else:
            y = F_.fused_cnsn(xg, cfg, g=g, addend=ag)
        y.backward(gy.contiguous(memory_format=fmt))
        torch.cuda.synchronize()
        return y.detach(), xg.grad, ag.grad, [p.grad for p in sn.parameters()], sn.g_bn.running_var.clone(), int(sn.g_bn.num_batches_tracked)

    a, b = run(torch.contiguous_format), run(CL)
    assert b[0].is_contiguous(memory_format=CL) and b[1].is_contiguous(memory_format=CL) and not b[0].is_contiguous()
    assert float((a[0] - b[0]).abs().max()) <= 1e-5 * max(1.0, float(a[0].abs().max()))
    same = (a[0] > 0) == (b[0] > 0)            # (the ReLU masks may differ where the pre-activation is rounding noise around zero)
    assert float(same.float().mean()) > 0.9999
    for u, v in zip(a[1:3], b[1:3]):
        assert float(((u - v).abs() * same).max()) <= 1e-5 * max(1.0, float(u.abs().max()))
    for u, v in zip(a[3], b[3]):
        assert float((u - v).abs().max()) <= 1e-4 * max(1.0, float(u.abs().max()))
    assert float((a[4] - b[4]).abs().max()) <= 1e-6 and a[5] == b[5] == 1
    cfg = cnsn_amd.FusedConfig(sn_active=True, add_mode="pre", relu=True)
    # (SelfNorm alone on a channels-last tensor: the single-launch kernels, reported as "resident"; CNSN_NHWC_FUSED=0: two-pass)
    assert cnsn_amd.which_path(x.contiguous(memory_format=CL), cfg) in ("resident", "streaming")
    assert cnsn_amd.which_path(x, cfg) != "streaming"


def test_calls_the_channels_last_kernels_do_not_take_fall_back_to_a_copy():
    shape = (6, 16, 12, 12)
    torch.manual_seed(3)
    np.random.seed(3)
    x = (torch.randn(shape, device=DEV) + 0.2).contiguous(memory_format=CL)
    d = cnsn_amd.draw_cn(shape, "both", 1)
    boxed = cnsn_amd.cn_op_2ins_space_chan(x, draws=d)                        # crop boxes: NCHW kernels on a copy
    want = cnsn_amd.cn_op_2ins_space_chan(x.contiguous(), draws=d)
    assert torch.equal(boxed.contiguous(), want)
    odd = (torch.randn(4, 6, 8, 8, device=DEV)).contiguous(memory_format=CL)   # 6 channels: no whole 16-byte vector
    sn = fill_sn(cnsn_amd.SelfNorm(6), 1, torch.float32).to(DEV).eval()
    with torch.no_grad():
        assert torch.equal(sn(odd).contiguous(), sn(odd.contiguous()))
    # the C ABI says so itself: the workspace query answers 0, the call CNSN_E_UNSUPPORTED
    prob = F_._problem(odd, cnsn_amd.FusedConfig(sn_active=True, sn_training=False))
    prob.layout = _ffi.LAYOUT_NHWC
    assert cnsn_amd.lib().cnsn_workspace_bytes(C.byref(prob)) == 0


def test_full_size_channels_last_against_the_nchw_kernels():
    """(256,256,56,56) bf16 block — BASELINE config 3's layer-1 site: the channels-last kernels against the cluster kernels on
    the same values"""
    shape, dt = (256, 256, 56, 56), torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(shape, device=DEV, dtype=dt, generator=g)
    b = torch.randn(shape, device=DEV, dtype=dt, generator=g) * 0.5
    gy = torch.randn(shape, device=DEV, dtype=dt, generator=g)
    outs = []
    for fmt in (torch.contiguous_format, CL):
        m = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(256), 7, torch.float32)).to(DEV).train()
        xg, bg = x.detach().clone(memory_format=fmt).requires_grad_(), b.detach().clone(memory_format=fmt).requires_grad_()
        y = m.forward_block(xg, bg, add_mode="pre", relu=True)
        y.backward(gy.contiguous(memory_format=fmt))
        outs.append((y.detach().float(), xg.grad.float(), [p.grad for p in m.parameters()], m.selfnorm.g_bn.running_mean.clone()))
        del xg, bg, y
    (y0, g0, p0, r0), (y1, g1, p1, r1) = outs
    assert float((y0 - y1).abs().max()) <= 1e-2 * float(y0.abs().max())
    same = (y0 > 0) == (y1 > 0)
    assert float(same.float().mean()) > 0.999
    assert float(((g0 - g1).abs() * same).max()) <= 2e-2 * float(g0.abs().max())
    for u, v in zip(p0, p1):
        assert float((u - v).abs().max()) <= 1e-3 * max(float(u.abs().max()), 1e-3)
    assert float((r0 - r1).abs().max()) <= 1e-5


def test_resnet50_in_channels_last_matches_the_nchw_model():
    """the whole ResNet-50+SN caller (models/imagenet/resnet_cnsn.py's counterpart, 16 SelfNorm sites with the fused block
    epilogue) in torch.channels_last against the same weights in NCHW: logits, loss gradient of the stem and of a SelfNorm gate,
    a running statistic — fp32, so that the comparison is about the op's two paths and not about 16-bit rounding"""
    from cnsn_amd.callers import ResNet50CNSN
    torch.manual_seed(4)
    np.random.seed(4)
    a = ResNet50CNSN(num_classes=10, cnsn_type="sn", pos="post").to(DEV).train()
    b = ResNet50CNSN(num_classes=10, cnsn_type="sn", pos="post").to(DEV)
    b.load_state_dict(a.state_dict())
    b = b.to(memory_format=CL).train()
    x = torch.randn(6, 3, 96, 96, device=DEV)
    y = torch.randint(0, 10, (6,), device=DEV)
    la = a(x)
    lb = b(x.contiguous(memory_format=CL))
    torch.nn.functional.cross_entropy(la, y).backward()
    torch.nn.functional.cross_entropy(lb, y).backward()
    torch.cuda.synchronize()
    la, lb = la.detach(), lb.detach()
    scale = max(1.0, float(la.abs().max()))
    # 53 convolutions run different MIOpen algorithms in the two layouts and ReLU masks amplify that: measured 4e-4..9e-4 on the
    # logits and 4-12 % of the largest element on the earliest gradients, WITH the sites forced through the NCHW kernels too
    # (CNSN_NHWC=0) — so the gradients are compared by direction, the logits by value
    assert float((la - lb).abs().max()) <= 3e-3 * scale

    def cos(u, v):
        return float(torch.nn.functional.cosine_similarity(u.flatten().double(), v.flatten().double(), dim=0))
    sa, sb = a.layer1[0].cnsn.selfnorm, b.layer1[0].cnsn.selfnorm
    assert cos(a.conv1.weight.grad, b.conv1.weight.grad) >= 0.99
    assert cos(a.fc.weight.grad, b.fc.weight.grad) >= 0.9999
    assert cos(sa.g_fc.weight.grad, sb.g_fc.weight.grad) >= 0.99
    assert cos(a.layer4[2].cnsn.selfnorm.g_fc.weight.grad, b.layer4[2].cnsn.selfnorm.g_fc.weight.grad) >= 0.999
    assert float((sa.g_bn.running_mean - sb.g_bn.running_mean).abs().max()) <= 1e-3
    # ... and the sites really ran the channels-last kernels: their input arrives channels-last from the convolutions
    seen = []
    h = b.layer2[0].cnsn.register_forward_hook(lambda m, i, o: seen.append((i[0].is_contiguous(memory_format=CL), o.is_contiguous(memory_format=CL))))
    with torch.no_grad():
        b(x.contiguous(memory_format=CL))
    h.remove()
    assert seen == [] or all(u and v for u, v in seen)      # (the fused block path calls forward_block, not forward: nothing to see then)


@pytest.mark.parametrize("fused", ["1", "0"], ids=["single-launch", "two-pass"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("kind", ["sn", "cnsn"])
@pytest.mark.parametrize("relu", [True, False])
def test_kept_sum_gives_the_same_bits(fused, dtype, kind, relu, monkeypatch):
    """cnsn_epilogue_t.sum_out (ABI 8): a PRE add in front of a channels-last call keeps X = x + addend for the backward, which is
    then the backward of the op WITHOUT the add on X.  Same arithmetic -> the SAME BITS as the call that saves x and the addend
    (y, dx, d_addend, parameter gradients, running statistics), under both channels-last strategies; and cnsn_keeps_sum() /
    CNSN_E_UNSUPPORTED say which calls take it."""
    import os
    shape = (12, 64, 14, 14)
    torch.manual_seed(11)
    np.random.seed(11)
    x = (torch.randn(shape, device=DEV) * 1.3 + 0.2).to(dtype)
    add = (torch.randn(shape, device=DEV) * 0.7).to(dtype)
    gy = torch.randn(shape, device=DEV).to(dtype)
    perm = torch.randperm(shape[0])
    monkeypatch.setenv("CNSN_NHWC_FUSED", fused)
    cnsn_amd.lib().cnsn_reload_env()
    try:
        outs = []
        for keep in (True, False):
            monkeypatch.setattr(F_, "_KEEP_SUM", keep)
            sn = fill_sn(cnsn_amd.SelfNorm(shape[1]), 4, torch.float32).to(DEV).train()
            xg = x.clone(memory_format=CL).requires_grad_()
            ag = add.clone(memory_format=CL).requires_grad_()
            kw, g, f = sn._fused_args()
            cfg = cnsn_amd.FusedConfig(add_mode="pre", relu=relu, cn_active=(kind == "cnsn"), **kw)
            y = F_.FusedCNSN.apply(xg, cfg, perm if kind == "cnsn" else None, None, g.fc_weight, g.bn_weight, g.bn_bias,
                                   g.running_mean, g.running_var, *(None,) * 5, ag, g.num_batches_tracked, None)
            saved_x = y.grad_fn.saved_tensors[0]
            y.backward(gy.contiguous(memory_format=CL))
            torch.cuda.synchronize()
            if keep:   # what was saved IS the rounded sum, and neither input
                assert saved_x.data_ptr() not in (xg.data_ptr(), ag.data_ptr())
                assert torch.equal(saved_x, (xg.detach() + ag.detach()))
            else:
                assert saved_x.data_ptr() == xg.data_ptr()
            outs.append((y.detach().clone(), xg.grad.clone(), ag.grad.clone(), [p.grad.clone() for p in sn.parameters()],
                         sn.g_bn.running_mean.clone(), sn.g_bn.running_var.clone()))
        a, b = outs
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        assert all(torch.equal(u, v) for u, v in zip(a[3], b[3])) and torch.equal(a[4], b[4]) and torch.equal(a[5], b[5])
    finally:
        monkeypatch.delenv("CNSN_NHWC_FUSED")
        cnsn_amd.lib().cnsn_reload_env()


def test_keeps_sum_query_and_refusals():
    x = torch.randn(4, 16, 8, 8, device=DEV)
    cfg = cnsn_amd.FusedConfig(sn_active=True, add_mode="pre", relu=True)
    lib = cnsn_amd.lib()

    def keeps(t, c, nhwc):
        prob = F_._problem(t, c)
        prob.layout = _ffi.LAYOUT_NHWC if nhwc else _ffi.LAYOUT_NCHW
        return lib.cnsn_keeps_sum(C.byref(prob), C.byref(F_._epilogue(c, None)))
    assert keeps(x, cfg, True) == 1
    assert keeps(x, cfg, False) == 0                                                        # NCHW strategies read both tensors once
    assert keeps(x, cnsn_amd.FusedConfig(sn_active=True, add_mode="post", relu=True), True) == 0
    assert keeps(torch.randn(4, 6, 8, 8, device=DEV), cfg, True) == 0                       # no whole 16-byte vector of channels
    # an NCHW call with sum_out set is refused, not silently computed without it
    prob = F_._problem(x, cfg)
    add, y, keep = torch.randn_like(x), torch.empty_like(x), torch.empty_like(x)
    epi = F_._epilogue(cfg, add, keep)
    sn = fill_sn(cnsn_amd.SelfNorm(16), 1, torch.float32).to(DEV).train()
    g = F_._GateBuffers(sn.g_fc.weight, sn.g_bn.weight, sn.g_bn.bias, sn.g_bn.running_mean, sn.g_bn.running_var)
    ws_bytes = lib.cnsn_workspace_bytes(C.byref(prob))
    ws = torch.empty(ws_bytes // 4 + 4, device=DEV)
    st = lib.cnsn_forward_fused(C.byref(prob), C.byref(epi), x.data_ptr(), None, None, C.byref(g.c), None, y.data_ptr(), None,
                                ws.data_ptr(), ws_bytes, None)
    assert st == _ffi.E_UNSUPPORTED
