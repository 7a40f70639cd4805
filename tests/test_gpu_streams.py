"""Cluster-resident launches issued from two streams of one device.  Each persistent grid assumes it becomes fully
resident; the library chains such launches per device (a launch on another stream waits, stream-side, for the previous
one), so interleaving them must neither hang nor change results."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from cnsn_amd import FusedConfig, which_path  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402

DEV = torch.device("cuda:0")


def test_two_streams_interleaved():
    shape = (64, 96, 56, 56)                       # large enough that the grids of two launches would overlap
    assert which_path(torch.empty(shape, device="meta"), FusedConfig(sn_active=True)) == "resident"
    mods = [cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(shape[1]), 11 + i, torch.float32)).to(DEV).train() for i in range(2)]
    refs = [cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(shape[1]), 11 + i, torch.float32)).to(DEV).train() for i in range(2)]
    xs = [torch.randn(shape, device=DEV, requires_grad=True) for _ in range(2)]
    gys = [torch.randn(shape, device=DEV) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    got = [None, None]
    for it in range(6):                             # interleave launches of the two streams
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                y = mods[i](xs[i])
                got[i] = (y, torch.autograd.grad(y, [xs[i]] + list(mods[i].parameters()), gys[i]))
    torch.cuda.synchronize()
    for i in (0, 1):                                # the same six steps on the default stream
        for it in range(6):
            y = refs[i](xs[i])
            want = (y, torch.autograd.grad(y, [xs[i]] + list(refs[i].parameters()), gys[i]))
        torch.cuda.synchronize()
        assert torch.equal(got[i][0], want[0])
        for a, b in zip(got[i][1], want[1]):
            assert torch.equal(a, b)
        assert torch.equal(mods[i].selfnorm.g_bn.running_var, refs[i].selfnorm.g_bn.running_var)
