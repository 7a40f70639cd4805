"""The op enqueues work on the caller's stream and nothing else (no host synchronisation, no allocation of its
own): it can be captured into a HIP graph and replayed — the launch-overhead-free way to run small sites."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402

DEV = torch.device("cuda:0")


@pytest.mark.parametrize("shape", [(8, 64, 32, 32), (16, 32, 14, 14), (4, 8, 7, 7)], ids=lambda s: "x".join(map(str, s)))
def test_inference_forward_replays_from_a_graph(shape):
    mod = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(shape[1]), 3, torch.float32)).to(DEV).eval()
    x = torch.randn(shape, device=DEV)
    idt = torch.randn(shape, device=DEV)
    with torch.no_grad():
        want = mod.forward_block(x, idt, add_mode="pre", relu=True).clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                mod.forward_block(x, idt, add_mode="pre", relu=True)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = mod.forward_block(x, idt, add_mode="pre", relu=True)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, want)
        x.copy_(torch.randn(shape, device=DEV))                      # new input, same graph
        want2 = mod.forward_block(x, idt, add_mode="pre", relu=True).clone()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, want2)


def test_training_step_of_selfnorm_replays_from_a_graph():
    """forward + backward of SelfNorm in training mode (cluster exchange, running-statistics update) in one graph."""
    shape = (16, 32, 28, 28)
    mod = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(32), 5, torch.float32)).to(DEV).train()
    ref = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(32), 5, torch.float32)).to(DEV).train()
    x = torch.randn(shape, device=DEV, requires_grad=True)
    gy = torch.randn(shape, device=DEV)
    params = list(mod.parameters())
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):                                           # warm-up outside the graph (twice, as torch asks)
            torch.autograd.grad(mod(x), [x] + params, gy)
            torch.autograd.grad(ref(x), [x] + list(ref.parameters()), gy)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        grads = torch.autograd.grad(mod(x), [x] + params, gy)
    for _ in range(3):
        g.replay()
        want = torch.autograd.grad(ref(x), [x] + list(ref.parameters()), gy)
    torch.cuda.synchronize()
    for a, b in zip(grads, want):
        assert torch.equal(a, b)
    assert torch.equal(mod.selfnorm.g_bn.running_var, ref.selfnorm.g_bn.running_var)


def test_graphed_idle_step_matches_eager_steps():
    """callers.GraphedIdleStep: steps whose CrossNorm sites are idle are replayed from one captured graph; the sequence of
    parameters equals the eager sequence under the same seeds (armed steps are eager in both)."""
    import numpy as np
    from cnsn_amd.callers import GraphedIdleStep, WideResNetCNSN, train_step_cn

    def run(graphed, blocks=False):
        torch.manual_seed(11)
        np.random.seed(11)
        net = WideResNetCNSN(10, 10, 1, active_num=1, pos="post", beta=1, crop="both", cnsn_type="cnsn").to(DEV).train()
        opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
        g = torch.Generator(device=DEV).manual_seed(3)
        x = torch.randn(16, 3, 32, 32, device=DEV, generator=g)
        y = torch.randint(0, 10, (16,), device=DEV, generator=g)
        for _ in range(2):                      # (both variants: two eager steps first — MIOpen picks its kernels outside a capture)
            train_step_cn(net, x, y, opt, 0.5)
        stepper = GraphedIdleStep(net, opt, x, y, warmup=0, graph_blocks=blocks) if graphed else None
        np.random.seed(12)
        torch.manual_seed(12)
        for _ in range(6):
            if graphed:
                stepper.step(x, y, 0.5)
            else:
                train_step_cn(net, x, y, opt, 0.5)
        torch.cuda.synchronize()
        return [p.detach().clone() for p in net.parameters()], net.state_dict()["block1.layer.0.cnsn.selfnorm.g_bn.num_batches_tracked"].item()

    eager, n_e = run(False)
    graph, n_g = run(True)
    assert n_g == n_e                          # capturing records the step, it does not run it
    per_block, n_b = run(True, blocks=True)    # armed steps: one captured graph per idle block, the armed blocks eager
    assert n_b == n_e
    worst_b = max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(per_block, eager))
    assert worst_b < 2e-2, worst_b
    worst = max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(graph, eager))
    assert worst < 2e-2, worst                 # same arithmetic; MIOpen's convolutions are not bit-reproducible run to run
                                               # (six SGD steps amplify that: 1e-4 .. 6e-3 seen, depending on what ran before)
