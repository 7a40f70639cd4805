"""Pin the CPU oracle to the reference: every golden vector in tests/golden/*.npz was produced by the
imported reference (tests/golden/gen_golden.py); the oracle must reproduce them BIT-EXACTLY."""
import os

import numpy as np
import pytest
import torch

from oracle import cnsn_oracle as orc

DT = {"f32": torch.float32, "f64": torch.float64}


@pytest.fixture(autouse=True)
def _single_thread():
    """The vectors were generated single-threaded; torch's CPU reductions change their summation
    order with the thread count, and bit-exactness is what is being asserted here."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def t(a):
    return torch.from_numpy(np.asarray(a))


def box(a):
    a = [int(v) for v in a]
    return None if a[0] < 0 else tuple(a)


def exact(a, b):
    assert a.shape == b.shape and torch.equal(a, b), float((a - b).abs().max())


def test_g1_bbox(golden_dir):
    z = load(golden_dir, "g1_bbox.npz")
    for si, size in enumerate(z["sizes"]):
        for seed in range(32):
            np.random.seed(seed)
            b1 = orc.cn_rand_bbox(tuple(size), beta=1, bbx_thres=0.1)
            b2 = orc.cn_rand_bbox(tuple(size), beta=0.5, bbx_thres=0.25)
            assert list(b1) == list(z["boxes"][si, seed, 0])
            assert list(b2) == list(z["boxes"][si, seed, 1])
            for (x1, y1, x2, y2), thr in ((b1, 0.1), (b2, 0.25)):
                assert 0 <= x1 < x2 <= size[2] and 0 <= y1 < y2 <= size[3]
                assert (x2 - x1) * (y2 - y1) / (size[2] * size[3]) > thr


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("etag,eps", [("e5", 1e-5), ("e12", 1e-12)])
def test_g2_stats(golden_dir, tag, etag, eps):
    z = load(golden_dir, "g2_stats.npz")
    x = t(z[f"x_{tag}"]).requires_grad_()
    m, s = orc.calc_ins_mean_std(x, eps=eps)
    exact(m.detach(), t(z[f"mean_{tag}_{etag}"]))
    exact(s.detach(), t(z[f"std_{tag}_{etag}"]))
    (m * t(z[f"gmean_{tag}"]) + s * t(z[f"gstd_{tag}"])).sum().backward()
    exact(x.grad, t(z[f"dx_{tag}_{etag}"]))
    # the constant plane: std = sqrt(eps) exactly as the reference gives it
    assert float(s[1, 2]) == pytest.approx(eps ** 0.5, rel=1e-6)


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_g3_cn(golden_dir, tag):
    z = load(golden_dir, "g3_cn.npz")
    ncases = len(z["case_crop"])
    assert ncases == 32
    for i in range(ncases):
        k = f"c{i}"
        crop = str(z["case_crop"][i])
        chan = bool(z["case_chan"][i])
        lam = None if z["case_lam"][i] < 0 else float(z["case_lam"][i])
        d = orc.CNDraws(perm=t(z[f"{k}_perm"]), style_box=box(z[f"{k}_sbox"]),
                        chan_perm=t(z[f"{k}_chan_perm"]) if chan else None,
                        content_box=box(z[f"{k}_cbox"]))
        x = t(z[f"{k}_x_{tag}"]).requires_grad_()
        y = orc.cn_op_2ins_space_chan(x, crop=crop, beta=1, lam=lam, chan=chan, draws=d)
        exact(y.detach(), t(z[f"{k}_y_{tag}"]))
        y.backward(t(z[f"{k}_gy_{tag}"]))
        exact(x.grad, t(z[f"{k}_dx_{tag}"]))
        # the recorded draws are what the RNG streams give in the reference's order
        seed = int(z["case_seed"][i])
        torch.manual_seed(seed)
        np.random.seed(seed)
        d2 = orc.draw_cn(x.shape, crop, beta=1, chan=chan)
        assert torch.equal(d2.perm, d.perm) and d2.style_box == d.style_box
        assert d2.content_box == d.content_box


def fill_from(z, prefix, mod, dtype):
    sd = {k: t(z[f"{prefix}_init_{k}"]) for k in [str(s) for s in z[f"{prefix}_keys"]]}
    mod.load_state_dict(sd)
    return mod.to(dtype)


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("is_two", [False, True])
def test_g4_selfnorm(golden_dir, tag, is_two):
    z = load(golden_dir, "g4_sn.npz")
    k = f"two{int(is_two)}_{tag}"
    keys = [str(s) for s in z[f"{k}_keys"]]
    m = orc.SelfNorm(5, is_two=is_two)
    assert list(m.state_dict().keys()) == keys
    fill_from(z, k, m, DT[tag]).train()
    for step in (1, 2):
        m.zero_grad()
        x = t(z[f"{k}_s{step}_x"]).requires_grad_()
        y = m(x)
        y.backward(t(z[f"{k}_s{step}_gy"]))
        exact(y.detach(), t(z[f"{k}_s{step}_y"]))
        exact(x.grad, t(z[f"{k}_s{step}_dx"]))
        for n, p in m.named_parameters():
            exact(p.grad, t(z[f"{k}_s{step}_grad_{n}"]))
        for n, v in m.state_dict().items():
            exact(v, t(z[f"{k}_s{step}_state_{n}"]))
    m.eval()
    with torch.no_grad():
        exact(m(t(z[f"{k}_eval_x"])), t(z[f"{k}_eval_y"]))


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("crop", orc.CROPS)
def test_g5_cnsn(golden_dir, tag, crop):
    z = load(golden_dir, "g5_cnsn.npz")
    zs = load(golden_dir, "g4_sn.npz")  # noqa: F841 (kept to show the two sets are independent)
    k = f"{crop}_{tag}"
    import tests.golden.gen_golden_fill as fill  # deterministic SN fill shared with the generator
    sn = fill.fill_sn(orc.SelfNorm(4), 3, DT[tag])
    m = orc.CNSN(orc.CrossNorm(crop=crop, beta=1), sn).train()
    x = t(z[f"{k}_x"])
    m.crossnorm.active = True
    m.crossnorm.next_draws = orc.CNDraws(perm=t(z[f"{k}_perm"]), style_box=box(z[f"{k}_sbox"]),
                                         content_box=box(z[f"{k}_cbox"]))
    xi = x.clone().requires_grad_()
    y = m(xi)
    assert m.crossnorm.active is False
    y.backward(t(z[f"{k}_gy"]))
    exact(y.detach(), t(z[f"{k}_armed_y"]))
    exact(xi.grad, t(z[f"{k}_armed_dx"]))
    for n, p in m.named_parameters():
        exact(p.grad, t(z[f"{k}_armed_grad_{n}"]))
    for n, v in m.state_dict().items():
        exact(v, t(z[f"{k}_armed_state_{n}"]))
    exact(m(x).detach(), t(z[f"{k}_idle_y"]))
    m.eval()
    m.crossnorm.active = True
    with torch.no_grad():
        exact(m(x), t(z[f"{k}_eval_y"]))
    assert m.crossnorm.active is False


def test_g5_cn_only(golden_dir):
    z = load(golden_dir, "g5_cnsn.npz")
    m = orc.CNSN(orc.CrossNorm(crop="neither", beta=1), None).train()
    x = t(z["cnonly_x"])
    m.crossnorm.active = True
    m.crossnorm.next_draws = orc.CNDraws(perm=t(z["cnonly_perm"]))
    exact(m(x), t(z["cnonly_y"]))
    assert bool(z["cnonly_idle_is_identity"]) and m(x) is x


# ------------------------------------------------------------------------------------------------
# G8: the Jensen-Shannon consistency arithmetic of the trainers (executed from the reference's own AST nodes)
# ------------------------------------------------------------------------------------------------
JSD_CASES = ["small", "cifar100", "imagenet", "clamped", "one_class"]


@pytest.mark.parametrize("case", JSD_CASES)
@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("f64", torch.float64)])
def test_jsd_oracle_reproduces_reference(golden_dir, case, tag, dtype):
    from oracle import jsd_oracle
    g8 = np.load(os.path.join(golden_dir, "g8_jsd.npz"))
    torch.set_num_threads(4)                      # (the generator's setting: reductions split the same way)
    zs = [torch.from_numpy(g8[f"{case}_logits{i}"]).to(dtype).requires_grad_() for i in range(3)]
    loss = jsd_oracle.jsd_consistency(*zs)
    loss.backward()
    assert np.array_equal(loss.detach().numpy(), g8[f"{case}_{tag}_loss"])
    for i, z in enumerate(zs):
        assert np.array_equal(z.grad.numpy(), g8[f"{case}_{tag}_grad{i}"]), (case, tag, i)
    assert len(g8["sources"]) >= 2 and any("imagenet.py:367" in str(s_) for s_ in g8["sources"])
