"""One Python thread per replica — what torch.nn.DataParallel, the reference's own multi-GPU mode (cifar.py:395,
imagenet.py:533), does to the module-level state of the Python layer (size cache, per-device exchange context, pinned
staging ring for the permutations) and to the library (launch chaining across streams, launch counter per context).
A box with one GPU stands in with two threads on two streams of the same device: fixed draws, so every thread's results
must be bit-identical to the same work done alone."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402

DEV = torch.device("cuda:0")
WORK = [((64, 48, 56, 56), torch.float32, "both"), ((48, 96, 28, 28), torch.bfloat16, "neither"),
        ((64, 32, 56, 56), torch.bfloat16, "style"), ((32, 64, 40, 40), torch.float32, "neither")]


def job(idx, reps, out, stream=None, barrier=None):
    shape, dtype, crop = WORK[idx % len(WORK)]
    n, c = shape[:2]
    g = torch.Generator(device=DEV).manual_seed(100 + idx)
    x = (torch.randn(shape, device=DEV, generator=g) + 0.2).to(dtype).requires_grad_()
    b = (torch.randn(shape, device=DEV, generator=g) * 0.5).to(dtype).requires_grad_()
    gy = torch.randn(shape, device=DEV, generator=g).to(dtype)
    rng = np.random.RandomState(idx)
    res = []
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1), fill_sn(cnsn_amd.SelfNorm(c), idx, torch.float32)).to(DEV).train()
        if barrier is not None:
            barrier.wait()
        for r in range(reps):
            armed = r % 2 == 0                     # alternate: CrossNorm-armed call / SelfNorm-only fused block
            if armed:
                mod.crossnorm.active = True
                perm = torch.from_numpy(rng.permutation(n))
                box = (2, 3, shape[2] - 4, shape[3] - 2)
                mod.crossnorm.next_draws = cnsn_amd.CNDraws(perm, box if crop in ("style", "both") else None, None,
                                                            box if crop in ("content", "both") else None)
                y = mod(x)
                grads = torch.autograd.grad(y, [x] + list(mod.parameters()), gy)
            else:
                y = mod.forward_block(x, b, add_mode="pre", relu=True)
                grads = torch.autograd.grad(y, [x, b] + list(mod.parameters()), gy)
            res.append([y.detach().clone()] + [t.detach().clone() for t in grads])
        if stream is not None:
            stream.synchronize()
        res.append([v.clone() for v in mod.buffers()])
    out[idx] = res


def test_two_threads_two_streams_same_bits_as_alone():
    reps = 8
    alone = {}
    for i in range(2):
        job(i, reps, alone)
    torch.cuda.synchronize()
    together, errors = {}, []
    barrier = threading.Barrier(2)
    streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]

    def run(i):
        try:
            with torch.cuda.device(DEV):
                job(i, reps, together, streams[i], barrier)
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    torch.cuda.synchronize()
    assert not errors, errors
    assert cnsn_amd.lib().cnsn_resident_timeouts() == 0
    for i in range(2):
        for step, (a, b) in enumerate(zip(alone[i], together[i])):
            for u, v in zip(a, b):
                assert torch.equal(u, v), (i, step)
