"""Callers of the hot path: the WideResNet-40-2 / ResNet-50 counterparts place
CNSN where the reference does, expose the reference's state_dict keys, draw from the RNGs in the
reference's order, and reproduce the logits the imported reference models produced (G6 vectors).
CPU tests run the backbones with the oracle's CNSN modules; GPU tests with the HIP modules."""
import os

import numpy as np
import pytest
import torch

from cnsn_amd.callers import (ResNet50CNSN, WideResNetCNSN, jsd_consistency, train_step_cn,
                              train_step_cn_consistency, train_step_image_cn_views)
from oracle import cnsn_oracle as orc
from tests.golden.gen_golden_fill import fill_by_name


@pytest.fixture(scope="module")
def g6(golden_dir):
    return np.load(os.path.join(golden_dir, "g6_models.npz"))


def keys_of(m):
    return [f"{k}|{tuple(v.shape)}" for k, v in m.state_dict().items()]


def make_wrn(impl, dtype=torch.float32):
    return fill_by_name(WideResNetCNSN(40, 100, 2, active_num=2, pos="post", beta=1, crop="both", cnsn_type="cnsn",
                                       impl=impl), 1).to(dtype)


def make_r50(impl, dtype=torch.float32):
    return fill_by_name(ResNet50CNSN(impl=impl), 2).to(dtype)


def test_state_dict_keys_match_reference(g6):
    assert keys_of(make_wrn(orc)) == [str(k) for k in g6["wrn_keys"]]
    assert keys_of(make_r50(orc)) == [str(k) for k in g6["r50_keys"]]


def test_cnsn_sites_and_widths():
    seen = []
    m = make_wrn(orc).train()
    hooks = [mod.register_forward_hook(lambda _m, i, o: seen.append(tuple(o.shape)))
             for mod in m.modules() if isinstance(mod, orc.CNSN)]
    m(torch.randn(4, 3, 32, 32))
    for h in hooks:
        h.remove()
    assert seen == [(4, 32, 32, 32)] * 6 + [(4, 64, 16, 16)] * 6 + [(4, 128, 8, 8)] * 6
    assert m.cn_num == 18 and len(m.cn_modules) == 18
    r = make_r50(orc)
    widths = [mod.selfnorm.g_bn.num_features for mod in r.modules() if isinstance(mod, orc.CNSN)]
    assert widths == [256] * 3 + [512] * 4 + [1024] * 6 + [2048] * 3
    assert sum(p.numel() for p in r.parameters()) == 25617448


def run_sequence(model, x, aug_seed):
    """The sequence gen_golden_models.py ran on the reference: train, (train aug), eval."""
    out = {}
    model.train()
    with torch.no_grad():
        out["train"] = model(x)
        if aug_seed is not None:
            torch.manual_seed(aug_seed)
            np.random.seed(aug_seed)
            out["train_aug"] = model(x, aug=True)
        model.eval()
        out["eval"] = model(x)
    return out


def check(out, g6, prefix, what, tol):
    for k, v in out.items():
        t64, t32 = torch.from_numpy(g6[f"{prefix}_f64_{k}"]), torch.from_numpy(g6[f"{prefix}_f32_{k}"]).double()
        err = float((v.detach().cpu().double() - t64).abs().max())
        ref_err = float((t32 - t64).abs().max())
        scale = float(t64.abs().max())
        assert err <= max(tol * scale, 3 * ref_err), f"{what} {k}: err {err:.3e}, reference fp32 err {ref_err:.3e}"


def test_wrn_logits_match_reference_with_oracle_modules(g6):
    torch.set_num_threads(8)
    out = run_sequence(make_wrn(orc), torch.from_numpy(g6["wrn_x"]), int(g6["wrn_f32_aug_seed"]))
    check(out, g6, "wrn", "WRN-40-2 (oracle CNSN, CPU)", 1e-5)


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs an MI355X")
def test_wrn_logits_match_reference_on_gpu(g6):
    import cnsn_amd
    m = make_wrn(cnsn_amd.cnsn).cuda()
    out = run_sequence(m, torch.from_numpy(g6["wrn_x"]).cuda(), int(g6["wrn_f32_aug_seed"]))
    check(out, g6, "wrn", "WRN-40-2 (HIP CNSN)", 1e-3)     # convolutions run in MIOpen fp32
    rv = m.state_dict()["block3.layer.5.cnsn.selfnorm.g_bn.running_var"].cpu().double()
    assert float((rv - torch.from_numpy(g6["wrn_f64_rv_last"])).abs().max()) < 1e-3
    assert all(not c.active for c in m.cn_modules)          # every armed site dropped its flag


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs an MI355X")
def test_resnet50_logits_match_reference_on_gpu(g6):
    import cnsn_amd
    m = make_r50(cnsn_amd.cnsn).cuda()
    out = run_sequence(m, torch.from_numpy(g6["r50_x"]).cuda(), None)
    check(out, g6, "r50", "ResNet-50 (HIP SelfNorm)", 1e-3)
    rv = m.state_dict()["layer4.2.cnsn.selfnorm.g_bn.running_var"].cpu().double()
    assert float((rv - torch.from_numpy(g6["r50_f64_rv_last"])).abs().max()) < 1e-3


def test_jsd_properties_and_steps_on_cpu():
    from oracle import jsd_oracle
    torch.manual_seed(0)
    a, b, c = (torch.randn(8, 10) for _ in range(3))
    jsd = jsd_oracle.jsd_consistency
    assert float(jsd(a, a, a)) == pytest.approx(0.0, abs=1e-6)                  # identical views
    assert float(jsd(a, b, c)) > 0                                              # JSD >= 0
    assert float(jsd(a, b, c)) == pytest.approx(float(jsd(c, a, b)), rel=1e-5)
    with pytest.raises(Exception):                                              # the product's JSD is device-only
        jsd_consistency(a, b, c)
    # step structure on a tiny WRN-10-1 with the oracle modules: r is drawn BEFORE the forward
    net = WideResNetCNSN(10, 10, 1, active_num=1, pos="post", beta=1, crop="both", cnsn_type="cnsn", impl=orc)
    opt = torch.optim.SGD(net.parameters(), lr=0.01)
    x, y = torch.randn(4, 3, 32, 32), torch.randint(0, 10, (4,))
    np.random.seed(3)
    r_expected = np.random.rand(1)
    np.random.seed(3)
    torch.manual_seed(3)
    l1 = train_step_cn(net.train(), x, y, opt, cn_prob=0.5)
    assert np.isfinite(float(l1)) and (r_expected < 0.5) in (True, False)
    l2 = train_step_cn_consistency(net, x, y, opt, consist_wt=10.0, jsd=jsd)
    l3 = train_step_image_cn_views(net, [x, x + 0.1, x - 0.1], y, opt, cn_prob=1.0, beta=1, crop="both",
                                   cn_op=orc.cn_op_2ins_space_chan, jsd=jsd)
    assert np.isfinite(float(l2)) and np.isfinite(float(l3))
    assert all(not c.active for c in net.cn_modules)


def test_consistency_step_draw_order_matches_reference():
    """cifar.py:163-190: ONE np.random.rand(1) is drawn before anything else; r < cn_prob -> net(aug=False), then two
    net(aug=True) views + JSD; otherwise a single net(aug=False) with plain cross-entropy."""
    class Recorder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(12, 5)
            self.calls = []

        def forward(self, x, aug=False):
            self.calls.append((bool(aug), float(np.random.get_state()[1][0]), np.random.get_state()[2]))
            return self.lin(x.flatten(1))

    from oracle import jsd_oracle
    x, y = torch.randn(4, 3, 2, 2), torch.randint(0, 5, (4,))
    for seed in range(6):
        np.random.seed(seed)
        r = float(np.random.rand(1)[0])
        pos_after_one_draw = np.random.get_state()[2]
        for cn_prob in (0.0, 0.5, 1.0):
            net = Recorder()
            opt = torch.optim.SGD(net.parameters(), lr=0.0)
            np.random.seed(seed)
            train_step_cn_consistency(net, x, y, opt, consist_wt=10.0, cn_prob=cn_prob, jsd=jsd_oracle.jsd_consistency)
            flags = [c[0] for c in net.calls]
            assert flags == ([False, True, True] if r < cn_prob else [False]), (seed, cn_prob, r, flags)
            assert net.calls[0][2] == pos_after_one_draw          # exactly one draw before the first forward
            assert np.random.get_state()[2] == pos_after_one_draw  # and none after (the stub draws nothing)


# ------------------------------------------------------------------------------------------------
# segmentation backbone: dilated ResNet-50, SelfNorm at 'residual' + a separate CrossNorm at 'post'
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def g7(golden_dir):
    return np.load(os.path.join(golden_dir, "g7_seg.npz"))


def make_seg(impl, dtype=torch.float32):
    from cnsn_amd.callers import SegResNet50CNSN
    return fill_by_name(SegResNet50CNSN(impl=impl), 3).to(dtype)


def seg_sequence(model, x):
    """What gen_golden_seg.py ran on the reference: train idle, train with one site armed (seed 91), eval."""
    res = {}

    def put(key, t):
        res[key + "_pool"], res[key + "_head"] = t.mean((2, 3)), t[:, :8]

    model.train()
    with torch.no_grad():
        o = model(x)
        put("train_out", o["out"]), put("train_aux", o["aux"])
        torch.manual_seed(91)
        np.random.seed(91)
        armed = model._enable_cross_norm()
        o = model(x)
        put("aug_out", o["out"]), put("aug_aux", o["aux"])
        model.eval()
        put("eval_out", model(x)["out"])
    return res, armed


def seg_check(res, g7, tol, what):
    for k, v in res.items():
        t64, t32 = torch.from_numpy(g7[f"f64_{k}"]), torch.from_numpy(g7[f"f32_{k}"]).double()
        err = float((v.detach().cpu().double() - t64).abs().max())
        ref_err = float((t32 - t64).abs().max())
        assert err <= max(tol * float(t64.abs().max()), 3 * ref_err), f"{what} {k}: err {err:.3e} (reference fp32 {ref_err:.3e})"


def test_segmentation_backbone_matches_reference_with_oracle_modules(g7):
    torch.set_num_threads(8)
    m = make_seg(orc)
    assert keys_of(m) == [str(k) for k in g7["keys"]]
    assert m.cn_num == int(g7["cn_num"]) == 16
    res, armed = seg_sequence(m, torch.from_numpy(g7["x"]))
    assert armed == [int(v) for v in g7["f32_armed"]]
    seg_check(res, g7, 1e-5, "dilated ResNet-50 (oracle CNSN, CPU)")
    assert all(not c.active for c in m.cn_modules)


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs an MI355X")
def test_segmentation_backbone_matches_reference_on_gpu(g7):
    import cnsn_amd
    m = make_seg(cnsn_amd.cnsn).cuda()
    res, armed = seg_sequence(m, torch.from_numpy(g7["x"]).cuda())
    assert armed == [int(v) for v in g7["f32_armed"]]
    seg_check(res, g7, 1e-3, "dilated ResNet-50 (HIP CNSN)")


def test_poly_learning_rate_and_fcn_head():
    from cnsn_amd.callers import FCNHead, poly_learning_rate
    assert poly_learning_rate(0.01, 0, 100) == pytest.approx(0.01)
    assert poly_learning_rate(0.01, 50, 100) == pytest.approx(0.01 * 0.5 ** 0.9)
    assert poly_learning_rate(0.01, 100, 100) == 0.0
    head = FCNHead(2048, 19).eval()
    assert head(torch.randn(2, 2048, 8, 8)).shape == (2, 19, 8, 8)
    assert [tuple(p.shape) for p in head.parameters()] == [(512, 2048, 3, 3), (512,), (512,), (19, 512, 1, 1), (19,)]
