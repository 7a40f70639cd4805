"""`torch.nn.DataParallel` — the reference's REAL multi-GPU mode (cifar.py:395 `net = torch.nn.DataParallel(net).cuda()`,
imagenet.py:533) — through the module surface on a box with ONE GPU: `device_ids=[0, 0]` makes two replicas on cuda:0 that
torch drives from two host threads (`parallel_apply`), each on the current stream of its thread.  What that exercises here:
the Python layer's per-device state shared by two threads (exchange context, size cache, pinned staging ring), the library's
launch chaining (two persistent grids of one process never overlap), the arena's torch pool entered by two threads at once.
Fixed CrossNorm draws; the result must be, bit for bit, what ONE thread computes on the two halves of the batch.
(review of round 5, item 8; the two-device variant is tests/test_gpu_robustness.py::test_tensor_on_a_device_that_is_not_current)"""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from cnsn_amd import arena  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402

DEV = torch.device("cuda:0")


class TwoSites(nn.Module):
    """conv -> CNSN (CrossNorm armed by the test + SelfNorm) -> conv -> residual block epilogue (add + SelfNorm + ReLU)"""

    def __init__(self, c, crop):
        super().__init__()
        self.conv1 = nn.Conv2d(3, c, 3, 1, 1, bias=False)
        self.site1 = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1), fill_sn(cnsn_amd.SelfNorm(c), 5, torch.float32))
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=False)
        self.site2 = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(c), 6, torch.float32))
        self.fc = nn.Linear(c, 10)

    def forward(self, x):
        h = self.site1(self.conv1(x))
        h = self.site2.forward_block(self.conv2(h), h, add_mode="pre", relu=True)
        return self.fc(h.mean((2, 3)))


@pytest.mark.parametrize("crop,hw,arena_mb", [("neither", 56, 1), ("both", 56, 1), ("neither", 28, -1)])
def test_two_replicas_on_one_device_equal_one_thread(crop, hw, arena_mb):
    was = arena.min_bytes()
    if arena_mb > 0:
        arena.enable(min_mb=arena_mb)          # the outputs of both replicas come from the arena's pool
    else:
        arena.disable()
    try:
        torch.manual_seed(0)
        n, c = 64, 32
        net = TwoSites(c, crop).to(DEV).train()
        x = torch.randn(n, 3, hw, hw, device=DEV)
        target = torch.randint(0, 10, (n,), device=DEV)
        half = n // 2
        np.random.seed(3)
        torch.manual_seed(3)
        draws = cnsn_amd.draw_cn((half, c, hw, hw), crop, 1)
        state = {k: v.clone() for k, v in net.named_buffers()}

        def arm():
            with torch.no_grad():
                for k, v in net.named_buffers():        # same running statistics / counters in front of every forward
                    v.copy_(state[k])
            net.site1.crossnorm.active = True
            net.site1.crossnorm.next_draws = draws      # (replicas copy the module's __dict__: both halves use these draws)

        # one thread: the two halves one after the other
        outs = []
        net.zero_grad(set_to_none=True)
        arm()
        for part in (slice(0, half), slice(half, n)):
            net.site1.crossnorm.active = True            # (CrossNorm.forward always disarms itself, models/cnsn.py:108)
            net.site1.crossnorm.next_draws = draws
            outs.append(net(x[part]))
        loss = nn.functional.cross_entropy(torch.cat(outs), target)
        loss.backward()
        torch.cuda.synchronize()
        want_logits = torch.cat(outs).detach().clone()
        want_grads = {k: p.grad.clone() for k, p in net.named_parameters()}

        # torch.nn.DataParallel: two replicas on cuda:0, two host threads
        timeouts = cnsn_amd.lib().cnsn_resident_timeouts()
        dp = nn.DataParallel(net, device_ids=[0, 0])
        for _ in range(3):                               # (repeat: thread interleavings differ from run to run)
            arm()
            net.zero_grad(set_to_none=True)
            logits = dp(x)
            loss = nn.functional.cross_entropy(logits, target)
            loss.backward()
            torch.cuda.synchronize()
            assert cnsn_amd.lib().cnsn_resident_timeouts() == timeouts
            assert torch.equal(logits.detach(), want_logits)
            for k, p in net.named_parameters():
                if k.startswith("site"):                 # the op's own parameter gradients: bit for bit
                    assert torch.equal(p.grad, want_grads[k]), k
                else:                                    # (MIOpen's weight-gradient kernels may accumulate atomically)
                    torch.testing.assert_close(p.grad, want_grads[k], rtol=1e-4, atol=1e-6)
    finally:
        _ffi_glue = cnsn_amd._ffi.glue()
        if _ffi_glue is not None:
            _ffi_glue.arena_config(was)
