"""Every instantiation class of the channel-resident kernels at least once: element type x vector width x register
bucket (1, 2, 4, 7, 8, 13, 16 vectors per lane) x boxed / un-boxed x with / without the residual-block epilogue,
forward and backward, forced with strategy='resident' (AUTO never reaches some of them — e.g. 16-bit boxed planes of
the 7/8-slot class run two-pass — but CNSN_STRATEGY_RESIDENT does).  Several of these instantiations spill registers;
one (16-bit, 8 slots, two planes per wave, boxed) was miscompiled until it was given one plane per wave."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from tests.test_gpu_fused_block import check as check_block, run_case as run_block  # noqa: E402
from tests.test_gpu_parity import assert_parity, run_pair  # noqa: E402

# (H, W) per bucket; 16-byte vectors: W % 8 == 0 for 16-bit (8 elements), W % 4 == 0 for fp32
PLANES_V16 = {1: (8, 64), 2: (12, 64), 4: (28, 64), 7: (56, 56), 8: (60, 64), 13: (96, 64), 16: (120, 64)}
PLANES_F32 = {1: (8, 32), 2: (12, 32), 4: (28, 32), 7: (40, 40), 8: (60, 32), 13: (56, 56), 16: (60, 64)}
# 16-bit planes whose width is a multiple of 4 only: 8-byte vectors
PLANES_V8 = {1: (14, 12), 2: (30, 12), 4: (36, 20), 7: (60, 28), 8: (44, 44), 13: (84, 36), 16: (92, 44)}

CASES = []
for nv, hw in PLANES_F32.items():
    CASES.append(("f32", nv, hw))
for nv, hw in PLANES_V16.items():
    CASES.append(("bf16", nv, hw))
    CASES.append(("f16", nv, hw))
for nv, hw in PLANES_V8.items():
    CASES.append(("bf16v4", nv, hw))
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16, "bf16v4": torch.bfloat16}


@pytest.fixture(autouse=True)
def resident():
    cnsn_amd.set_strategy("resident")
    yield
    cnsn_amd.set_strategy("auto")


@pytest.mark.parametrize("tag,nv,hw", CASES, ids=lambda v: str(v).replace(" ", ""))
@pytest.mark.parametrize("crop", ["neither", "both", "content", "style"])
@pytest.mark.parametrize("n", [5, 37])       # fewer / more instances than one workgroup owns
def test_op(tag, nv, hw, crop, n):
    shape = (n, 2, *hw)
    out = run_pair(shape, crop, "cnsn", DT[tag], 500 + nv + n, training=True)
    assert_parity(out, DT[tag], (tag, nv, shape, crop))


@pytest.mark.parametrize("tag,nv,hw", CASES, ids=lambda v: str(v).replace(" ", ""))
@pytest.mark.parametrize("kind,crop", [("sn", "neither"), ("cnsn", "both"), ("cnsn", "content")])
@pytest.mark.parametrize("n", [5, 37])
def test_block(tag, nv, hw, kind, crop, n):
    shape = (n, 2, *hw)
    check_block(run_block(shape, kind, crop, "pre", True, DT[tag], 700 + nv + n), DT[tag], True, (tag, nv, shape, kind, crop))


# POST add (`act(CNSN(x) + addend)`, the 'residual' position): the un-boxed calls run the POST instantiations of the
# resident kernels (addend fetched after the exchange; backward: mask from the addend, masked gradient written out)
@pytest.mark.parametrize("tag,nv,hw", CASES, ids=lambda v: str(v).replace(" ", ""))
@pytest.mark.parametrize("kind,crop,relu", [("sn", "neither", True), ("cnsn", "neither", True), ("sn", "neither", False)])
@pytest.mark.parametrize("n", [5, 37])
def test_block_post(tag, nv, hw, kind, crop, relu, n):
    shape = (n, 2, *hw)
    x = torch.empty(shape, dtype=DT[tag], device="cuda")
    cfg = cnsn_amd.FusedConfig(sn_active=True, cn_active=kind == "cnsn", add_mode="post", relu=relu)
    if not (tag == "bf16v4" and nv == 1):      # (168 elements = 21 16-byte vectors: too small a plane for the cluster kernels)
        assert cnsn_amd.which_path(x, cfg, backward=False) == "resident"
    check_block(run_block(shape, kind, crop, "post", relu, DT[tag], 800 + nv + n), DT[tag], relu, (tag, nv, shape, kind, crop, "post"))


# The persistent loop (`item += gridDim.x`): with C = 2 every workgroup handles ONE item.  Enough channels that the
# grid (at most 6 workgroups x 256 CUs, a whole number of clusters) wraps around at least once per register bucket —
# the second and later items reuse LDS, the granule area of another channel and the per-item prefetches.  Checked
# against the oracle's eager ops on the GPU (tests/test_gpu_full_size.py), N = 37: a partial last cluster member.
@pytest.mark.parametrize("tag,nv,hw", [c for c in CASES if c[0] in ("f32", "bf16")], ids=lambda v: str(v).replace(" ", ""))
@pytest.mark.parametrize("kind,crop", [("sn", "neither"), ("cnsn", "both")])
def test_persistent_loop_many_channels(tag, nv, hw, kind, crop):
    from tests.test_gpu_full_size import check_case
    c = 1024 if nv <= 2 else (512 if nv == 4 else 384)    # items = C * ceil(37 / (4 * planes per wave)) > 1536
    check_case((37, c, *hw), DT[tag], kind, crop, 900 + nv)


# Large planes (1025..4096 vectors): ONE plane per workgroup, a quarter per wave (cnsn_resident_split.hip) — segmentation
# layer 1's 128x128 sites.  Buckets of 8 / 16 slots per wave, fp32 / bf16 / fp16, boxed and not, inference, the POST
# epilogue; N = 3 and 9 members per cluster.
SPLIT_CASES = [("f32", (128, 64)), ("f32", (128, 128)), ("f32", (72, 64)), ("bf16", (128, 128)), ("bf16", (128, 192)),
               ("f16", (96, 96))]


@pytest.mark.parametrize("tag,hw", SPLIT_CASES, ids=lambda v: str(v).replace(" ", ""))
@pytest.mark.parametrize("kind,crop", [("cnsn", "neither"), ("cnsn", "both"), ("cn", "style"), ("sn", "neither")])
@pytest.mark.parametrize("n", [3, 9])
def test_split_plane_op(tag, hw, kind, crop, n):
    shape = (n, 2, *hw)
    x = torch.empty(shape, dtype=DT[tag], device="cuda")
    cfg = cnsn_amd.FusedConfig(sn_active=kind != "cn", cn_active=kind != "sn")
    assert cnsn_amd.which_path(x, cfg, backward=False) == "resident" and cnsn_amd.which_path(x, cfg, backward=True) == "resident"
    out = run_pair(shape, crop, kind, DT[tag], 950 + n + hw[0], training=True)
    assert_parity(out, DT[tag], ("split", tag, shape, kind, crop))


@pytest.mark.parametrize("tag,hw", SPLIT_CASES[:4], ids=lambda v: str(v).replace(" ", ""))
def test_split_plane_inference_and_post(tag, hw):
    shape = (5, 2, *hw)
    out = run_pair(shape, "neither", "sn", DT[tag], 970 + hw[1], training=False)                 # SOLO variant
    assert_parity(out, DT[tag], ("split eval", tag, shape))
    for mode, relu in (("post", True), ("post", False), ("none", True)):
        check_block(run_block(shape, "sn", "neither", mode, relu, DT[tag], 980 + hw[1]), DT[tag], relu, ("split", tag, shape, mode, relu))
    check_block(run_block(shape, "cnsn", "both", "none", True, DT[tag], 990 + hw[1]), DT[tag], True, ("split boxed relu", tag, shape))
