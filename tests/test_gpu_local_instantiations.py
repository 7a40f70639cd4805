"""Every instantiation class of the channel-local kernels at least once: element type x copy-vector width (16 / 8 / 4 /
2 bytes, decided by the piece and row sizes) x 256- / 1024-thread workgroups x with / without the residual-block
epilogue x training / eval / two-gate, forced with strategy='local' and the tuning overrides CNSN_LOCAL_LB /
CNSN_LOCAL_CG; each case first checks with `which_path` that the local kernels are what actually runs."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from cnsn_amd import FusedConfig, which_path  # noqa: E402
from tests.test_gpu_fused_block import check as check_block, run_case as run_block  # noqa: E402
from tests.test_gpu_parity import assert_parity, run_pair  # noqa: E402

# (shape, dtype, forced CG) -> copy-vector width: piece = CG*M*b, row = C*M*b
CASES = [
    ((6, 4, 8, 8), torch.float32, "1"),      # 256-byte planes: 16-byte copies
    ((6, 3, 7, 7), torch.float32, "1"),      # 196-byte planes: 4-byte copies
    ((6, 4, 9, 10), torch.float32, "2"),     # piece 720, row 1440: 16-byte copies
    ((5, 6, 14, 14), torch.bfloat16, "1"),   # 392-byte planes: 8-byte copies
    ((5, 8, 7, 7), torch.bfloat16, "2"),     # piece 196: 4-byte copies
    ((5, 3, 7, 7), torch.bfloat16, "1"),     # 98-byte planes, odd C: 2-byte copies
    ((5, 8, 7, 7), torch.float16, "4"),      # piece 392: 8-byte copies
    ((5, 8, 7, 7), torch.float16, "8"),      # piece 784: 16-byte copies
    ((7, 2, 16, 16), torch.bfloat16, "2"),   # 512-byte planes
]


@pytest.fixture(params=["256", "1024"])
def forced(request):
    cnsn_amd.set_strategy("local")
    os.environ["CNSN_LOCAL_LB"] = request.param
    yield request.param
    cnsn_amd.set_strategy("auto")
    os.environ.pop("CNSN_LOCAL_LB", None)
    os.environ.pop("CNSN_LOCAL_CG", None)


def is_local(shape, dtype, **cfg):
    x = torch.empty(shape, dtype=dtype, device="meta")
    c = FusedConfig(sn_active=True, **cfg)
    return which_path(x, c, False) == "local" and which_path(x, c, True) == "local"


@pytest.mark.parametrize("shape,dtype,cg", CASES, ids=lambda v: str(v).replace(" ", "").replace("torch.", ""))
@pytest.mark.parametrize("training,is_two", [(True, False), (False, False), (True, True)])
def test_op(forced, shape, dtype, cg, training, is_two):
    os.environ["CNSN_LOCAL_CG"] = cg
    assert is_local(shape, dtype, sn_training=training, sn_two=is_two)
    out = run_pair(shape, "neither", "sn", dtype, 300 + int(cg), is_two=is_two, training=training)
    assert_parity(out, dtype, (shape, dtype, cg, forced, training, is_two))


@pytest.mark.parametrize("shape,dtype,cg", CASES, ids=lambda v: str(v).replace(" ", "").replace("torch.", ""))
@pytest.mark.parametrize("relu", [True, False])
def test_block(forced, shape, dtype, cg, relu):
    os.environ["CNSN_LOCAL_CG"] = cg
    assert is_local(shape, dtype, add_mode="pre", relu=relu)
    check_block(run_block(shape, "sn", "neither", "pre", relu, dtype, 400 + int(cg)), dtype, relu, (shape, dtype, cg, forced, relu))
