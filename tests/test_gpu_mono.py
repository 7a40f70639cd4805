"""The channel-in-registers strategy (cnsn_mono_kernels.h): one 1024-thread workgroup per channel holds every plane of
the channel in registers.  One case per geometry class — lanes per plane (16 / 64) x slot rows per wave (8 / 16) x
vector width (8 / 16 bytes) x element type — through SelfNorm alone (training and inference) and the residual-block
epilogue (PRE add, ReLU), forward and backward, against the CPU oracle in fp64; every case asserts that the mono
kernels are what ran."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from tests.test_gpu_fused_block import check as check_block, run_case as run_block  # noqa: E402
from tests.test_gpu_parity import assert_parity, run_pair  # noqa: E402

F32, BF16, F16 = torch.float32, torch.bfloat16, torch.float16
# (shape, dtype): what the geometry resolves to is noted per line (vector bytes, lanes per plane, slot rows per wave)
CASES = [
    ((37, 5, 14, 14), F32),     # 16 B, 49 of 64 lanes, R = 3           (ResNet-50 stage 3 plane)
    ((256, 3, 14, 14), F32),    # 16 B, 64 lanes, R = 16: the full-size channel, forward only (backward: other paths)
    ((256, 3, 14, 14), BF16),   # 8 B,  64 lanes, R = 16, both directions
    ((96, 4, 14, 14), BF16),    # R = 6
    ((128, 4, 16, 16), F32),    # 16 B, all 64 lanes, R = 8             (WideResNet stage 2)
    ((128, 4, 16, 16), F16),    # 16 B, 32 of 64 lanes
    ((256, 4, 16, 16), BF16),   # 16 B, R = 16: the backward walks the rows in QUARTERS (16 x 16-byte rows of G and x do not fit)
    ((250, 3, 16, 16), F16),    # ... with a partial last wave
    ((130, 3, 8, 8), F32),      # 16 B, 16 lanes per plane, 4 planes per row, R = 3
    ((1000, 2, 8, 8), F32),     # 16 lanes, R = 16: 1000 planes per channel
    ((33, 4, 8, 8), BF16),      # 8 B (16-byte vectors would leave 8 per plane), 16 lanes
    ((20, 6, 12, 12), F32),     # 36 vectors
    ((17, 3, 6, 10), F32),      # 15 vectors of 16 B -> 16 lanes, one idle
    ((64, 2, 10, 10), BF16),    # 25 vectors of 8 B
    ((18, 2, 2, 6), F32),       # tiny plane (3 vectors)
    ((40, 6, 7, 7), BF16),      # 2 B per lane, 49 of 64 lanes      (ResNet-50 stage 4 plane)
    ((256, 3, 7, 7), BF16),     # R = 16
    ((256, 3, 7, 7), F32),      # 4 B per lane
    ((33, 4, 7, 7), F16),
    ((50, 3, 6, 6), BF16),      # 36 elements: 18 vectors of 4 B
    ((21, 3, 5, 9), F32),       # 45 elements, one per lane
]
ids = lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else str(v).replace("torch.", "")  # noqa: E731


@pytest.fixture(autouse=True)
def mono():
    cnsn_amd.set_strategy("mono")
    yield
    cnsn_amd.set_strategy("auto")


def runs_mono(shape, dtype, backward, **cfg):
    x = torch.empty(shape, dtype=dtype, device="cuda")
    return cnsn_amd.which_path(x, cnsn_amd.FusedConfig(sn_active=True, **cfg), backward=backward) == "mono"


@pytest.mark.parametrize("shape,dtype", CASES, ids=ids)
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
def test_selfnorm(shape, dtype, training):
    assert runs_mono(shape, dtype, False, sn_training=training)
    out = run_pair(shape, "neither", "sn", dtype, 4000 + shape[0] + shape[2], training=training)
    assert_parity(out, dtype, ("mono", shape, dtype, training))


@pytest.mark.parametrize("shape,dtype", CASES, ids=ids)
@pytest.mark.parametrize("mode,relu", [("pre", True), ("pre", False), ("none", True)])
def test_block_epilogue(shape, dtype, mode, relu):
    assert runs_mono(shape, dtype, False, add_mode=mode, relu=relu)
    check_block(run_block(shape, "sn", "neither", mode, relu, dtype, 4100 + shape[0] + shape[3]), dtype, relu,
                ("mono", shape, dtype, mode, relu))


def test_what_mono_declines():
    FC = cnsn_amd.FusedConfig
    x = torch.empty((8, 4, 14, 14), device="cuda")
    assert cnsn_amd.which_path(x, FC(sn_active=True, cn_active=True)) == "mono"            # CrossNorm: the mono-cn kernels
    assert cnsn_amd.which_path(x, FC(sn_active=True, cn_active=True), chan_perm=True) != "mono"     # channel permutation
    assert cnsn_amd.which_path(x, FC(sn_active=True, sn_two=True)) != "mono"               # two-gate form
    assert cnsn_amd.which_path(x, FC(sn_active=True, add_mode="post")) != "mono"           # POST add
    assert cnsn_amd.which_path(torch.empty((8, 4, 7, 7), device="cuda"), FC(sn_active=True)) == "mono"     # odd plane: one element per lane
    assert cnsn_amd.which_path(torch.empty((8, 4, 9, 9), device="cuda"), FC(sn_active=True)) != "mono"     # 81 single elements
    seven = torch.empty((256, 2048, 7, 7), device="cuda")
    assert cnsn_amd.which_path(torch.empty((8, 4, 28, 28), device="cuda"), FC(sn_active=True)) != "mono"   # 196 vectors
    big = torch.empty((256, 4, 14, 14), device="cuda")
    assert cnsn_amd.which_path(big, FC(sn_active=True), backward=False) == "mono"
    assert cnsn_amd.which_path(big, FC(sn_active=True), backward=True) != "mono"          # fp32: G and x do not fit
    wide16 = torch.empty((256, 4, 16, 16), dtype=BF16, device="cuda")                       # 16 rows of 16-byte vectors in 16 bits:
    assert cnsn_amd.which_path(wide16, FC(sn_active=True), backward=True) == "mono"       # the backward takes them in quarters
    cnsn_amd.set_strategy("auto")
    assert cnsn_amd.which_path(torch.empty((256, 1024, 14, 14), dtype=BF16, device="cuda"), FC(sn_active=True), backward=True) == "mono"
    assert cnsn_amd.which_path(torch.empty((8, 4, 14, 14), device="cuda"), FC(sn_active=True)) != "mono"   # N < 16 under AUTO
    assert cnsn_amd.which_path(seven, FC(sn_active=True)) == "mono"                                        # 7x7 fp32: 4 B per lane
    assert cnsn_amd.which_path(seven.bfloat16(), FC(sn_active=True)) == "mono"                             # 7x7 bf16, N = 256: channel groups ("wide")
    assert cnsn_amd.which_path(seven.bfloat16()[:96], FC(sn_active=True)) == "local"                       # small batch: channel-local


# ------------------------------------------------------------------------------------------------
# the same frame WITH CrossNorm (cnsn_mono_cn_kernels.h): crop boxes, lam, with and without SelfNorm
# ------------------------------------------------------------------------------------------------
CN_CASES = [
    ((37, 5, 14, 14), F32), ((256, 3, 14, 14), BF16), ((128, 4, 16, 16), F32), ((130, 3, 8, 8), F32), ((33, 4, 8, 8), BF16),
    ((40, 6, 7, 7), BF16), ((64, 3, 7, 7), F32), ((20, 6, 12, 12), F16), ((17, 3, 6, 10), F32), ((300, 2, 8, 8), F32),
]


def runs_mono_cn(shape, dtype, backward, crop, sn, **cfg):
    x = torch.empty(shape, dtype=dtype, device="cuda")
    boxes = dict(content_box=(1, 1, 3, 3) if crop in ("content", "both") else None,
                 style_box=(0, 0, 2, 2) if crop in ("style", "both") else None)
    return cnsn_amd.which_path(x, cnsn_amd.FusedConfig(cn_active=True, sn_active=sn, **boxes, **cfg), backward=backward) == "mono"


@pytest.mark.parametrize("shape,dtype", CN_CASES, ids=ids)
@pytest.mark.parametrize("crop", ["neither", "style", "content", "both"])
@pytest.mark.parametrize("kind", ["cn", "cnsn"])
def test_crossnorm(shape, dtype, crop, kind):
    assert runs_mono_cn(shape, dtype, False, crop, kind == "cnsn") and runs_mono_cn(shape, dtype, True, crop, kind == "cnsn")
    out = run_pair(shape, crop, kind, dtype, 4200 + shape[0] + shape[3], training=True)
    assert_parity(out, dtype, ("mono-cn", shape, dtype, crop, kind))


@pytest.mark.parametrize("shape,dtype", CN_CASES[:6], ids=ids)
def test_crossnorm_options(shape, dtype):
    out = run_pair(shape, "both", "cnsn", dtype, 4300 + shape[0], lam=0.3, training=True)      # lam blend (cnsn.py:86-89)
    assert_parity(out, dtype, ("mono-cn lam", shape, dtype))
    out = run_pair(shape, "style", "cnsn", dtype, 4310 + shape[0], training=False)             # SelfNorm on running statistics
    assert_parity(out, dtype, ("mono-cn eval-sn", shape, dtype))


@pytest.mark.parametrize("shape,dtype", CN_CASES[:6], ids=ids)
@pytest.mark.parametrize("mode,relu", [("pre", True), ("pre", False), ("none", True)])
@pytest.mark.parametrize("kind,crop", [("cnsn", "both"), ("cn", "content")])
def test_crossnorm_block_epilogue(shape, dtype, mode, relu, kind, crop):
    check_block(run_block(shape, kind, crop, mode, relu, dtype, 4400 + shape[0] + shape[3]), dtype, relu,
                ("mono-cn", shape, dtype, kind, crop, mode, relu))
