"""GPU parity: the HIP path (module surface -> C ABI -> kernels) against (1) the golden vectors the
imported reference produced and (2) the CPU oracle run in fp64 on the same seeded inputs.

Tolerances (BASELINE.json north_star: 1e-5 fp32 / 1e-2 bf16):
  fp32  |hip - truth64| <= max(1e-5 * scale, 2 * |oracle32 - truth64|)   scale = max(1, max|truth|)
        (on some inputs the reference's own fp32 noise already equals 1e-5 because
         BatchNorm1d over N divides by the tiny across-batch spread of plane statistics)
  bf16/fp16  |hip - oracle32(on the same quantised inputs)| <= 1e-2 * max|oracle32|
"""
import os

import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():  # collected everywhere, executed only with -m gpu on the box
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from oracle import cnsn_oracle as orc  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402

DEV = torch.device("cuda:0")
STRATEGIES = ["two_pass", "resident", "local", "mono"]


def t(a):
    return torch.from_numpy(np.asarray(a))


def box(a):
    a = [int(v) for v in a]
    return None if a[0] < 0 else tuple(a)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def close(got, want, tol=1e-5, what=""):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol:.0e} * {scale:.3g}"


def seed_of(*a):
    return zlib.crc32(repr(a).encode()) % 100000


def cond_input(shape, seed, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    n, c = shape[:2]
    x = torch.randn(shape, generator=g, dtype=torch.float64)
    s = torch.rand(n, c, 1, 1, generator=g, dtype=torch.float64) * 1.5 + 0.5
    m = torch.randn(n, c, 1, 1, generator=g, dtype=torch.float64)
    return (x * s + m).to(dtype)


def to_draws(d):
    return cnsn_amd.CNDraws(d.perm, d.style_box, d.chan_perm, d.content_box)


@pytest.fixture(params=STRATEGIES)
def strategy(request):
    cnsn_amd.set_strategy(request.param)
    yield request.param
    cnsn_amd.set_strategy("auto")


# ------------------------------------------------------------------------------------------------
# golden vectors (reference outputs)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("etag,eps", [("e5", 1e-5), ("e12", 1e-12)])
def test_golden_stats(golden_dir, etag, eps):
    z = load(golden_dir, "g2_stats.npz")
    x = t(z["x_f32"]).to(DEV).requires_grad_()
    m, s = cnsn_amd.calc_ins_mean_std(x, eps=eps)
    assert m.shape == (4, 6, 1, 1) and s.shape == (4, 6, 1, 1) and m.dtype == torch.float32
    close(m, t(z[f"mean_f64_{etag}"]), 1e-6, "mean")
    # the constant planes have std = sqrt(eps) (1e-6 for eps=1e-12): compare relatively
    rel = ((s.detach().cpu().double() - t(z[f"std_f64_{etag}"])) / t(z[f"std_f64_{etag}"])).abs().max()
    assert float(rel) < 2e-6
    (m * t(z["gmean_f32"]).to(DEV) + s * t(z["gstd_f32"]).to(DEV)).sum().backward()
    want = t(z[f"dx_f64_{etag}"])
    live = torch.ones(4, 6, dtype=torch.bool)
    live[1, 2] = live[3, 0] = False      # dead planes: grad ~ dstd*0/(sqrt(eps)*(M-1)); checked loosely
    close(x.grad.cpu()[live], want[live], 1e-5, "dx")
    assert torch.isfinite(x.grad).all()


def test_golden_cn(golden_dir, strategy):
    z = load(golden_dir, "g3_cn.npz")
    for i in range(len(z["case_crop"])):
        k = f"c{i}"
        crop, chan = str(z["case_crop"][i]), bool(z["case_chan"][i])
        lam = None if z["case_lam"][i] < 0 else float(z["case_lam"][i])
        d = cnsn_amd.CNDraws(t(z[f"{k}_perm"]), box(z[f"{k}_sbox"]),
                             t(z[f"{k}_chan_perm"]) if chan else None, box(z[f"{k}_cbox"]))
        x = t(z[f"{k}_x_f32"]).to(DEV).requires_grad_()
        y = cnsn_amd.cn_op_2ins_space_chan(x, crop=crop, beta=1, lam=lam, chan=chan, draws=d)
        y.backward(t(z[f"{k}_gy_f32"]).to(DEV))
        e32y = float((t(z[f"{k}_y_f32"]).double() - t(z[f"{k}_y_f64"])).abs().max())
        e32x = float((t(z[f"{k}_dx_f32"]).double() - t(z[f"{k}_dx_f64"])).abs().max())
        ty, tx = t(z[f"{k}_y_f64"]), t(z[f"{k}_dx_f64"])
        ey = float((y.detach().cpu().double() - ty).abs().max())
        ex = float((x.grad.cpu().double() - tx).abs().max())
        assert ey <= max(1e-5 * max(1, float(ty.abs().max())), 2 * e32y), (i, crop, chan, lam, ey, e32y)
        assert ex <= max(1e-5 * max(1, float(tx.abs().max())), 2 * e32x), (i, crop, chan, lam, ex, e32x)


@pytest.mark.parametrize("is_two", [False, True])
def test_golden_selfnorm(golden_dir, strategy, is_two):
    z = load(golden_dir, "g4_sn.npz")
    k32, k64 = f"two{int(is_two)}_f32", f"two{int(is_two)}_f64"
    keys = [str(s) for s in z[f"{k32}_keys"]]
    m = cnsn_amd.SelfNorm(5, is_two=is_two)
    assert list(m.state_dict().keys()) == keys            # checkpoint compatibility
    m.load_state_dict({kk: t(z[f"{k32}_init_{kk}"]) for kk in keys})
    m.to(DEV).train()
    for step in (1, 2):
        m.zero_grad()
        x = t(z[f"{k32}_s{step}_x"]).to(DEV).requires_grad_()
        y = m(x)
        y.backward(t(z[f"{k32}_s{step}_gy"]).to(DEV))

        def chk(got, name, tol=1e-5):
            t64, t32 = t(z[f"{k64}_s{step}_{name}"]), t(z[f"{k32}_s{step}_{name}"])
            e32 = float((t32.double() - t64).abs().max())
            e = float((got.detach().cpu().double() - t64).abs().max())
            assert e <= max(tol * max(1.0, float(t64.abs().max())), 2 * e32), (name, step, e, e32)

        chk(y, "y")
        chk(x.grad, "dx")
        for n, p in m.named_parameters():
            chk(p.grad, f"grad_{n}", 1e-4)
        for n, v in m.state_dict().items():
            chk(v.double() if v.dtype != torch.int64 else v, f"state_{n}")
    m.eval()
    with torch.no_grad():
        close(m(t(z[f"{k32}_eval_x"]).to(DEV)), t(z[f"{k64}_eval_y"]), 1e-5, "eval y")


@pytest.mark.parametrize("crop", orc.CROPS)
def test_golden_cnsn(golden_dir, strategy, crop):
    z = load(golden_dir, "g5_cnsn.npz")
    k = f"{crop}_f32"
    k64 = f"{crop}_f64"
    m = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop=crop, beta=1),
                      fill_sn(cnsn_amd.SelfNorm(4), 3, torch.float32)).to(DEV).train()
    x = t(z[f"{k}_x"]).to(DEV)
    m.crossnorm.active = True
    m.crossnorm.next_draws = cnsn_amd.CNDraws(t(z[f"{k}_perm"]), box(z[f"{k}_sbox"]), None, box(z[f"{k}_cbox"]))
    xi = x.clone().requires_grad_()
    y = m(xi)
    assert m.crossnorm.active is False
    y.backward(t(z[f"{k}_gy"]).to(DEV))

    def chk(got, name, tol=1e-5):
        t64, t32 = t(z[f"{k64}_{name}"]), t(z[f"{k}_{name}"])
        e32 = float((t32.double() - t64).abs().max())
        e = float((got.detach().cpu().double() - t64).abs().max())
        assert e <= max(tol * max(1.0, float(t64.abs().max())), 2 * e32), (name, e, e32)

    chk(y, "armed_y")
    chk(xi.grad, "armed_dx")
    for n, p in m.named_parameters():
        chk(p.grad, f"armed_grad_{n}", 1e-4)
    for n, v in m.state_dict().items():
        chk(v, f"armed_state_{n}")
    chk(m(x), "idle_y")
    m.eval()
    m.crossnorm.active = True
    with torch.no_grad():
        chk(m(x), "eval_y")
    assert m.crossnorm.active is False


def test_golden_cn_only_and_identity(golden_dir):
    z = load(golden_dir, "g5_cnsn.npz")
    m = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop="neither", beta=1), None).to(DEV).train()
    x = t(z["cnonly_x"]).to(DEV)
    m.crossnorm.active = True
    m.crossnorm.next_draws = cnsn_amd.CNDraws(t(z["cnonly_perm"]))
    close(m(x), t(z["cnonly_y"]), 1e-5, "cn only")
    assert m(x) is x                                   # idle CrossNorm, no SelfNorm: identity object


def test_rng_stream_matches_reference(golden_dir):
    """Seeded module-level draws reproduce the reference's perms/boxes (G3 recorded them)."""
    z = load(golden_dir, "g3_cn.npz")
    for i in range(len(z["case_crop"])):
        k = f"c{i}"
        seed = int(z["case_seed"][i])
        torch.manual_seed(seed)
        np.random.seed(seed)
        d = cnsn_amd.draw_cn(z[f"{k}_x_f32"].shape, str(z["case_crop"][i]), 1, chan=bool(z["case_chan"][i]))
        assert torch.equal(d.perm, t(z[f"{k}_perm"]))
        assert d.style_box == box(z[f"{k}_sbox"]) and d.content_box == box(z[f"{k}_cbox"])


# ------------------------------------------------------------------------------------------------
# oracle sweeps: shapes that exercise every launch shape (vector width x lanes per plane)
# ------------------------------------------------------------------------------------------------
SHAPES = [
    (8, 64, 32, 32),     # configs[0]; vec4, wave per plane
    (4, 8, 7, 7),        # 7x7 planes: scalar loads, 16 lanes per plane
    (4, 16, 14, 14),     # 14x14
    (6, 4, 56, 56),      # north-star plane size
    (3, 3, 224, 224),    # image-level CrossNorm (imagenet.py:215): block per plane
    (2, 4, 128, 96),     # segmentation-like, non-square
    (5, 6, 9, 11),       # odd everything
    (33, 7, 8, 8),       # batch not a multiple of anything, WRN last stage plane
]


def oracle_pair(shape, crop, kind, dtype, seed, lam=None, chan=False, is_two=False, training=True):
    """inputs, draws and the oracle's results (fp64 truth + fp32) of one case: independent of the kernel strategy under
    test, so computed once per case (tests/_memo.py) and shared by the strategies that run it"""
    torch.manual_seed(seed)
    np.random.seed(seed)
    n, c = shape[:2]
    x64 = cond_input(shape, seed)
    gy64 = torch.randn(shape, dtype=torch.float64)
    if dtype != torch.float32:                     # identical quantised inputs for both sides
        x64 = x64.to(dtype).double()
        gy64 = gy64.to(dtype).double()
    d = orc.draw_cn(shape, crop, beta=1, chan=chan)
    out = {"x64": x64, "gy64": gy64, "d": d}
    for tag, odt in (("t64", torch.float64), ("o32", torch.float32)):
        sn = fill_sn(orc.SelfNorm(c, is_two=is_two), seed, odt) if kind != "cn" else None
        xr = x64.detach().clone().to(odt).requires_grad_()
        u = xr
        if sn is not None:
            sn.train(training)
        if kind != "sn":
            u = orc.cn_op_2ins_space_chan(u, crop=crop, lam=lam, chan=chan, draws=d)
        y = sn(u) if sn is not None else u
        y.backward(gy64.to(odt))
        out[tag] = dict(y=y.detach(), dx=xr.grad,
                        pg={k: v.grad for k, v in sn.named_parameters()} if sn else {},
                        st={k: v for k, v in sn.state_dict().items()} if sn else {})
    return out


def run_pair(shape, crop, kind, dtype, seed, lam=None, chan=False, is_two=False, training=True):
    """Run oracle (fp64 truth + fp32) and the HIP modules on identical inputs/draws."""
    from tests._memo import memo
    key = ("pair", tuple(shape), crop, kind, str(dtype), seed, lam, chan, is_two, training)
    ora = memo(key, lambda: oracle_pair(shape, crop, kind, dtype, seed, lam, chan, is_two, training))
    x64, gy64, d = ora["x64"], ora["gy64"], ora["d"]
    n, c = shape[:2]
    out = {"t64": ora["t64"], "o32": ora["o32"]}
    sn = fill_sn(cnsn_amd.SelfNorm(c, is_two=is_two), seed, torch.float32).to(DEV) if kind != "cn" else None
    xg = x64.detach().clone().to(dtype).to(DEV).requires_grad_()
    if kind == "sn":
        sn.train(training)
        y = sn(xg)
    elif kind == "cn":
        y = cnsn_amd.cn_op_2ins_space_chan(xg, crop=crop, lam=lam, chan=chan, draws=to_draws(d))
    else:
        sn.train(training)
        if lam is None and not chan and training:
            mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1), sn).to(DEV).train()
            mod.crossnorm.active = True
            mod.crossnorm.next_draws = to_draws(d)
            y = mod(xg)
        else:
            kw, g, f = sn._fused_args()
            cfg = cnsn_amd.FusedConfig(cn_active=True, content_box=d.content_box, style_box=d.style_box,
                                       lam=lam, **kw)
            y = cnsn_amd.fused_cnsn(xg, cfg, perm=d.perm, chan_perm=d.chan_perm, g=g, f=f)
    y.backward(gy64.to(dtype).to(DEV))
    torch.cuda.synchronize()
    out["hip"] = dict(y=y.detach().cpu(), dx=xg.grad.cpu(),
                      pg={k: v.grad.cpu() for k, v in sn.named_parameters()} if sn else {},
                      st={k: v.cpu() for k, v in sn.state_dict().items()} if sn else {})
    return out


def assert_parity(out, dtype, ctx):
    t64, o32, hip = out["t64"], out["o32"], out["hip"]

    def one(name, got, truth, ref32, tol):
        got, truth, ref32 = got.double(), truth.double(), ref32.double()
        scale = max(1.0, float(truth.abs().max()))
        e = float((got - truth).abs().max())
        if dtype == torch.float32:
            e32 = float((ref32 - truth).abs().max())
            assert e <= max(tol * scale, 2 * e32), f"{ctx} {name}: err {e:.3e} (oracle32 err {e32:.3e}, scale {scale:.3g})"
        else:
            e = float((got - ref32).abs().max())
            assert e <= 1e-2 * max(float(ref32.abs().max()), 1e-3), f"{ctx} {name}: err {e:.3e} vs max {float(ref32.abs().max()):.3g}"

    one("y", hip["y"], t64["y"], o32["y"], 1e-5)
    one("dx", hip["dx"], t64["dx"], o32["dx"], 1e-5)
    for k in t64["pg"]:
        one(f"grad {k}", hip["pg"][k], t64["pg"][k], o32["pg"][k], 1e-4)
    for k in t64["st"]:
        if "num_batches" in k:
            assert int(hip["st"][k]) == int(t64["st"][k])
        else:
            one(f"state {k}", hip["st"][k], t64["st"][k], o32["st"][k], 1e-5)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("crop", orc.CROPS)
@pytest.mark.parametrize("kind", ["cn", "cnsn"])
def test_oracle_sweep_fp32(strategy, shape, crop, kind):
    seed = seed_of(shape, crop, kind)
    assert_parity(run_pair(shape, crop, kind, torch.float32, seed), torch.float32, (shape, crop, kind, strategy))


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("is_two", [False, True])
@pytest.mark.parametrize("training", [True, False])
def test_oracle_sweep_sn(strategy, shape, is_two, training):
    seed = seed_of(shape, is_two, training)
    out = run_pair(shape, "neither", "sn", torch.float32, seed, is_two=is_two, training=training)
    assert_parity(out, torch.float32, (shape, "sn", is_two, training, strategy))


# shapes on the edges of the resident kernels' launch geometry: partial last register slot, batch not a
# multiple of the planes a workgroup owns (K=1 and K>1), every register bucket (1,2,4,7,8,13,16), C=1,
# 16-bit tensors with 8- and 4-element vectors
EDGE_SHAPES = [
    (5, 3, 12, 11),     # 33 vectors: bucket 1, 33 of 64 lanes; N=5 < planes per workgroup
    (7, 2, 20, 13),     # 65 vectors: bucket 2, one lane in the last slot
    (9, 3, 30, 30),     # 225 vectors: bucket 4
    (37, 2, 36, 44),    # 396 vectors: bucket 7; N=37 -> 10 workgroups per channel, last one partial
    (4, 2, 50, 36),     # 450 vectors: bucket 8
    (3, 2, 52, 64),     # 832 vectors: bucket 13, full
    (3, 3, 60, 68),     # 1020 vectors: bucket 16, partial (N=2 would make BatchNorm1d over the batch singular)
    (6, 1, 40, 40),     # a single channel
]


@pytest.mark.parametrize("shape", EDGE_SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("crop", ["neither", "both"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_edge_geometry(strategy, shape, crop, dtype):
    seed = seed_of(shape, crop, str(dtype), "edge")
    assert_parity(run_pair(shape, crop, "cnsn", dtype, seed), dtype, (shape, crop, dtype, strategy))


@pytest.mark.parametrize("lam,chan,is_two", [(0.3, False, False), (None, True, False), (0.7, True, True)])
@pytest.mark.parametrize("crop", ["neither", "both"])
def test_oracle_options(strategy, lam, chan, is_two, crop):
    shape = (6, 5, 12, 16)
    seed = seed_of(lam, chan, is_two, crop)
    out = run_pair(shape, crop, "cnsn", torch.float32, seed, lam=lam, chan=chan, is_two=is_two)
    assert_parity(out, torch.float32, (lam, chan, is_two, crop, strategy))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(8, 64, 32, 32), (4, 8, 7, 7), (4, 16, 14, 14), (6, 4, 56, 56), (2, 3, 224, 224)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("crop,kind", [("neither", "sn"), ("neither", "cnsn"), ("both", "cnsn"), ("style", "cn")])
def test_oracle_sweep_16bit(strategy, dtype, shape, crop, kind):
    seed = seed_of(shape, crop, kind, str(dtype))
    assert_parity(run_pair(shape, crop, kind, dtype, seed), dtype, (shape, crop, kind, dtype, strategy))


# ------------------------------------------------------------------------------------------------
# stand-alone functions, edge cases, error behaviour
# ------------------------------------------------------------------------------------------------
def test_instance_norm_mix_and_stats_grad():
    torch.manual_seed(5)
    c64 = cond_input((4, 6, 10, 12), 1)
    s64 = cond_input((4, 6, 5, 7), 2)           # style may have another spatial size (cnsn.py:22)
    gy = torch.randn(4, 6, 10, 12, dtype=torch.float64)
    cr, sr = c64.clone().requires_grad_(), s64.clone().requires_grad_()
    yr = orc.instance_norm_mix(cr, sr)
    yr.backward(gy)
    cg, sg = c64.float().to(DEV).requires_grad_(), s64.float().to(DEV).requires_grad_()
    yg = cnsn_amd.instance_norm_mix(cg, sg)
    yg.backward(gy.float().to(DEV))
    close(yg, yr, 1e-5, "mix y")
    close(cg.grad, cr.grad, 1e-5, "mix dcontent")
    close(sg.grad, sr.grad, 1e-5, "mix dstyle")


def test_noncontiguous_input_and_dead_plane():
    torch.manual_seed(6)
    base = cond_input((6, 10, 9, 8), 3).float()
    base[2, 4] = 0.0                                   # dead channel plane inside the view
    view = base[:, 2:8]                                # non-contiguous (N,C,H,W) view
    sn_o = fill_sn(orc.SelfNorm(6), 9, torch.float64).train()
    sn_h = fill_sn(cnsn_amd.SelfNorm(6), 9, torch.float32).to(DEV).train()
    yr = sn_o(view.double())
    yh = sn_h(base.to(DEV)[:, 2:8])
    close(yh, yr, 1e-5, "non-contiguous")
    assert torch.isfinite(yh).all()


def test_errors():
    sn = cnsn_amd.SelfNorm(4).to(DEV).train()
    with pytest.raises(ValueError):                    # BatchNorm1d: more than 1 value per channel
        sn(torch.randn(1, 4, 8, 8, device=DEV))
    sn.eval()
    assert sn(torch.randn(1, 4, 8, 8, device=DEV)).shape == (1, 4, 8, 8)
    with pytest.raises(cnsn_amd.CnsnError):            # no CPU fallback
        cnsn_amd.SelfNorm(4).train()(torch.randn(2, 4, 8, 8))
    with pytest.raises(AssertionError):
        cnsn_amd.cn_op_2ins_space_chan(torch.randn(2, 4, 8, 8, device=DEV), crop="bogus")
    with pytest.raises(AssertionError):
        cnsn_amd.calc_ins_mean_std(torch.randn(2, 4, 8, device=DEV))


@pytest.mark.parametrize("double_params", [False, True], ids=["f32-params", "f64-params"])
def test_selfnorm_takes_float64_like_the_reference(double_params):
    """models/cnsn.py:130-150 accepts any floating dtype.  float64 activations (round 6): computed by the float32 kernels,
    returned as float64 — float32 accuracy (north_star's 1e-5) against the float64 oracle; a module made float64 by
    `.double()` keeps float64 parameters, running statistics and gradients."""
    shape = (6, 8, 9, 7)
    x64 = cond_input(shape, 21)
    gy64 = torch.randn(shape, generator=torch.Generator().manual_seed(22), dtype=torch.float64)
    sn_o = fill_sn(orc.SelfNorm(shape[1]), 9, torch.float64).train()
    xo = x64.clone().requires_grad_()
    yo = sn_o(xo)
    yo.backward(gy64)
    sn = fill_sn(cnsn_amd.SelfNorm(shape[1]), 9, torch.float64 if double_params else torch.float32).to(DEV).train()
    m = cnsn_amd.CNSN(None, sn)
    xg = x64.to(DEV).requires_grad_()
    y = m(xg)
    assert y.dtype == torch.float64
    y.backward(gy64.to(DEV))
    torch.cuda.synchronize()
    scale = max(1.0, float(yo.abs().max()))
    assert float((y.detach().cpu() - yo.detach()).abs().max()) <= 1e-5 * scale
    assert xg.grad.dtype == torch.float64
    assert float((xg.grad.cpu() - xo.grad).abs().max()) <= 1e-5 * max(1.0, float(xo.grad.abs().max()))
    for p, q in zip(sn.parameters(), sn_o.parameters()):
        assert p.grad.dtype == p.dtype
        assert float((p.grad.cpu().double() - q.grad).abs().max()) <= 1e-4 * max(1.0, float(q.grad.abs().max()))
    assert float((sn.g_bn.running_var.cpu().double() - sn_o.g_bn.running_var).abs().max()) <= 1e-5
    # the block form too
    z = m.forward_block(xg.detach(), xg.detach() * 0.5, add_mode="pre", relu=True)
    assert z.dtype == torch.float64 and z.shape == xg.shape


@pytest.mark.parametrize("crop", ["neither", "both"])
def test_float64_parameter_free_ops(crop):
    """the reference takes any floating dtype (models/cnsn.py:12-16): the parameter-free ops accept float64, compute with the
    float32 kernels and return float64 — float32 accuracy against the oracle's float64 arithmetic, gradients included"""
    shape = (6, 5, 12, 16)
    torch.manual_seed(3)
    np.random.seed(3)
    x64 = cond_input(shape, 3)
    gy64 = torch.randn(shape, dtype=torch.float64)
    d = orc.draw_cn(shape, crop, beta=1)
    xr = x64.clone().requires_grad_()
    want = orc.cn_op_2ins_space_chan(xr, crop=crop, draws=d)
    want.backward(gy64)
    xg = x64.clone().to(DEV).requires_grad_()
    got = cnsn_amd.cn_op_2ins_space_chan(xg, crop=crop, draws=d)
    assert got.dtype == torch.float64
    got.backward(gy64.to(DEV))
    assert xg.grad.dtype == torch.float64
    for a, b in ((got.detach().cpu(), want.detach()), (xg.grad.cpu(), xr.grad)):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
    m, s = cnsn_amd.calc_ins_mean_std(x64.to(DEV))
    mo, so = orc.calc_ins_mean_std(x64)
    assert m.dtype == torch.float64 and float((m.cpu() - mo).abs().max()) <= 1e-6 * max(1.0, float(mo.abs().max()))
    assert float((s.cpu() - so).abs().max()) <= 1e-5 * max(1.0, float(so.abs().max()))
    mix = cnsn_amd.instance_norm_mix(x64.to(DEV), x64.flip(0).to(DEV))
    assert mix.dtype == torch.float64
    assert float((mix.cpu() - orc.instance_norm_mix(x64, x64.flip(0))).abs().max()) <= 2e-5 * max(1.0, float(x64.abs().max()))


def test_crossnorm_flag_semantics():
    cn = cnsn_amd.CrossNorm(crop="neither", beta=1).to(DEV)
    x = cond_input((4, 3, 8, 8), 0).float().to(DEV)
    cn.train()
    assert cn(x) is x and cn.active is False           # idle: identity
    cn.active = True
    y = cn(x)
    assert y is not x and cn.active is False           # armed: applies once, flag drops
    cn.eval()
    cn.active = True
    assert cn(x) is x and cn.active is False           # eval: identity but the flag still drops
    assert len(cn.state_dict()) == 0                   # no parameters or buffers


# ------------------------------------------------------------------------------------------------
# full-size checks at BASELINE.json's north-star shape: properties + the eager torch path on GPU
# ------------------------------------------------------------------------------------------------
NORTH = (256, 256, 56, 56)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("crop", ["neither", "both"])
def test_full_size_properties(strategy, dtype, crop):
    torch.manual_seed(0)
    np.random.seed(0)
    n, c = NORTH[:2]
    x = (torch.randn(NORTH, device=DEV) * (torch.rand(n, c, 1, 1, device=DEV) * 1.5 + 0.5)
         + torch.randn(n, c, 1, 1, device=DEV)).to(dtype)
    d = cnsn_amd.draw_cn(NORTH, crop, 1)
    # (1) CrossNorm alone: every output plane carries the statistics of its style source
    y = cnsn_amd.cn_op_2ins_space_chan(x, crop=crop, draws=d)
    xs = x.float()
    if d.style_box:
        x1, y1, x2, y2 = d.style_box
        xs = xs[:, :, x1:x2, y1:y2]
    src_mean = xs.mean((2, 3))[d.perm.to(DEV)]
    src_std = (xs.var((2, 3)) + 1e-5).sqrt()[d.perm.to(DEV)]
    yc = y.float()
    if d.content_box:
        x1, y1, x2, y2 = d.content_box
        outside = torch.ones(NORTH[2:], dtype=torch.bool, device=DEV)
        outside[x1:x2, y1:y2] = False
        assert torch.equal(y[:, :, outside], x[:, :, outside])        # untouched outside the box
        yc = yc[:, :, x1:x2, y1:y2]
    tol = 2e-2 if dtype != torch.float32 else 2e-4
    assert float((yc.mean((2, 3)) - src_mean).abs().max()) < tol * 4
    assert float((yc.std((2, 3)) / src_std - 1).abs().max()) < tol
    # (2) idempotence of the statistics swap under the identity permutation (crop neither): y == x
    if crop == "neither":
        ident = cnsn_amd.CNDraws(torch.arange(n))
        yi = cnsn_amd.cn_op_2ins_space_chan(x, crop="neither", draws=ident)
        assert float((yi.float() - x.float()).abs().max()) <= (1e-5 if dtype == torch.float32 else 4e-2) * 8
    # (3) fused CN+SN equals the eager torch ops of the oracle run on the GPU (fp32 reference)
    sn_h = fill_sn(cnsn_amd.SelfNorm(c), 4, torch.float32).to(DEV).train()
    sn_o = fill_sn(orc.SelfNorm(c), 4, torch.float32).to(DEV).train()
    mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1), sn_h).to(DEV).train()
    mod.crossnorm.active = True
    mod.crossnorm.next_draws = d
    gy = torch.randn(NORTH, device=DEV).to(dtype)
    xg = x.clone().requires_grad_()
    yh = mod(xg)
    yh.backward(gy)
    xo = x.float().requires_grad_()
    od = orc.CNDraws(d.perm, d.style_box, None, d.content_box)
    yo = sn_o(orc.cn_op_2ins_space_chan(xo, crop=crop, draws=od))
    yo.backward(gy.float())
    rt = 1e-2 if dtype != torch.float32 else 2e-5
    assert float((yh.float() - yo).abs().max()) <= rt * float(yo.abs().max())
    assert float((xg.grad.float() - xo.grad).abs().max()) <= rt * float(xo.grad.abs().max())
    gw_h, gw_o = sn_h.g_fc.weight.grad, sn_o.g_fc.weight.grad
    assert float((gw_h - gw_o).abs().max()) <= (5e-2 if dtype != torch.float32 else 1e-3) * float(gw_o.abs().max())
    close(sn_h.g_bn.running_var, sn_o.g_bn.running_var, 1e-4, "running_var")


def test_ctypes_and_cpp_glue_paths_agree():
    """functional.fused_cnsn goes through the C++ autograd glue when it is built, through ctypes
    otherwise; both call the same C ABI and must give identical bits."""
    import subprocess
    import sys
    code = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r)
import cnsn_amd
from tests.golden.gen_golden_fill import fill_sn
torch.manual_seed(0); np.random.seed(0)
m = cnsn_amd.CNSN(cnsn_amd.CrossNorm("both", 1), fill_sn(cnsn_amd.SelfNorm(8), 5, torch.float32)).cuda().train()
x = torch.randn(6, 8, 20, 24).cuda().requires_grad_()
m.crossnorm.active = True
y = m(x); y.backward(torch.ones_like(y))
print(float(y.double().sum()), float(x.grad.double().abs().sum()), float(m.selfnorm.g_fc.weight.grad.double().abs().sum()),
      int(cnsn_amd._ffi.glue() is not None))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env_extra in ({}, {"CNSN_NO_GLUE": "1"}):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", code % root], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().split("\n")[-1].split())
    assert outs[1][3] == "0"                                   # the second run really took the ctypes path
    assert outs[0][:3] == outs[1][:3], outs


def _fuzz_cases(count=48):
    rng = np.random.RandomState(20260928)
    cases = []
    for i in range(count):
        n = int(rng.randint(3, 41))
        c = int(rng.randint(1, 7))
        h, w = int(rng.randint(6, 65)), int(rng.choice([8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64, 7, 9, 14, 30]))
        dtype = [torch.float32, torch.float32, torch.bfloat16, torch.float16][int(rng.randint(4))]
        crop = orc.CROPS[int(rng.randint(4))]
        kind = ["cnsn", "cnsn", "cn", "sn"][int(rng.randint(4))]
        is_two = bool(rng.rand() < 0.2) and kind != "cn"
        training = not (kind == "sn" and rng.rand() < 0.3)
        lam = 0.4 if (kind != "sn" and rng.rand() < 0.2) else None
        cases.append((i, (n, c, h, w), dtype, crop if kind != "sn" else "neither", kind, is_two, training, lam))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: f"{c[0]}-{'x'.join(map(str, c[1]))}-{str(c[2])[6:]}-{c[3]}-{c[4]}")
def test_fuzz_shapes_and_modes(strategy, case):
    """Seeded random shapes x dtypes x modes through both strategies (the launch geometry of the resident
    kernels has many corner cases: register bucket, partial slot, partial cluster, vector width)."""
    i, shape, dtype, crop, kind, is_two, training, lam = case
    out = run_pair(shape, crop, kind, dtype, 1000 + i, lam=lam, is_two=is_two, training=training)
    assert_parity(out, dtype, (case, strategy))


# many channels: the mid kernels of the two-pass path switch to tiles of 8 channels per workgroup at C >= 512
@pytest.mark.parametrize("shape", [(5, 512, 7, 7), (6, 1024, 6, 8), (34, 512, 8, 8)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("kind,crop", [("sn", "neither"), ("cnsn", "neither"), ("cnsn", "both"), ("cn", "style")])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_many_channels(strategy, shape, kind, crop, dtype):
    seed = seed_of(shape, kind, crop, str(dtype))
    is_two = kind == "sn" and shape[0] == 5
    assert_parity(run_pair(shape, crop, kind, dtype, seed, is_two=is_two), dtype, (shape, kind, crop, dtype, strategy))
