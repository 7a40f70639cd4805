#!/usr/bin/env python3
"""Generate the golden vectors in this directory by IMPORTING the reference (build container only).

    python tests/golden/gen_golden.py            # needs /root/reference, writes tests/golden/*.npz

The reference's Python never travels: what is committed are inputs, recorded random draws and the
reference's outputs / gradients (data only).  While generating, every case is also replayed
through `oracle/cnsn_oracle.py` on the same draws and must agree BIT-EXACTLY in fp32 and fp64 —
that is the pin that lets the oracle stand in for the reference on the GPU box.

Vector sets:
  G1 bbox     cn_rand_bbox for seeds 0..31 x 4 sizes
  G2 stats    calc_ins_mean_std incl. a constant plane, eps 1e-5 and 1e-12
  G3 cn       cn_op_2ins_space_chan, 4 crops x chan x lam, y and dx
  G4 sn       SelfNorm train steps 1,2 (y, dx, dw, dgamma, dbeta, running stats), eval y, is_two
  G5 cnsn     CNSN armed / idle / eval + CrossNorm.active reset semantics
"""
import os
import sys

sys.dont_write_bytecode = True   # the reference tree is read-only material: leave no __pycache__ in it

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
np.int = int  # the reference uses the removed alias (models/cnsn.py:39-40)

import models.cnsn as ref  # noqa: E402  (the reference itself)
from oracle import cnsn_oracle as orc  # noqa: E402

torch.set_num_threads(1)


def seeded(seed):
    torch.manual_seed(seed)
    np.random.seed(seed)


def conditioned_input(shape, seed, dtype):
    """x = randn*s + m with per-plane s~U(0.5,2), m~N(0,1)  (SURVEY.md §8d1)."""
    g = torch.Generator().manual_seed(seed)
    n, c = shape[:2]
    x = torch.randn(shape, generator=g, dtype=torch.float64)
    s = torch.rand(n, c, 1, 1, generator=g, dtype=torch.float64) * 1.5 + 0.5
    m = torch.randn(n, c, 1, 1, generator=g, dtype=torch.float64)
    return (x * s + m).to(dtype)


def box_arr(b):
    return np.array(b if b is not None else (-1, -1, -1, -1), dtype=np.int64)


def same(a, b):
    return a.shape == b.shape and bool(torch.equal(a, b))


# ------------------------------------------------------------------------------------------- G1
def gen_bbox():
    sizes = [(8, 64, 32, 32), (4, 3, 224, 224), (2, 8, 7, 7), (2, 4, 128, 96)]
    out = {"sizes": np.array(sizes, dtype=np.int64)}
    boxes = np.zeros((len(sizes), 32, 2, 4), dtype=np.int64)
    for si, size in enumerate(sizes):
        for seed in range(32):
            np.random.seed(seed)
            b1 = ref.cn_rand_bbox(size, beta=1, bbx_thres=0.1)
            b2 = ref.cn_rand_bbox(size, beta=0.5, bbx_thres=0.25)   # second draw continues stream
            np.random.seed(seed)
            o1 = orc.cn_rand_bbox(size, beta=1, bbx_thres=0.1)
            o2 = orc.cn_rand_bbox(size, beta=0.5, bbx_thres=0.25)
            assert tuple(int(v) for v in b1) == o1 and tuple(int(v) for v in b2) == o2
            boxes[si, seed, 0] = b1
            boxes[si, seed, 1] = b2
    out["boxes"] = boxes
    np.savez_compressed(os.path.join(HERE, "g1_bbox.npz"), **out)


# ------------------------------------------------------------------------------------------- G2
def gen_stats():
    out = {}
    for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        x = conditioned_input((4, 6, 5, 7), 11, dtype)
        x[1, 2] = 0.75          # constant plane: var = 0, std = sqrt(eps)
        x[3, 0] = 0.0           # dead (post-ReLU style) plane
        out[f"x_{tag}"] = x.numpy()
        for etag, eps in (("e5", 1e-5), ("e12", 1e-12)):
            xr = x.clone().requires_grad_(True)
            m, s = ref.calc_ins_mean_std(xr, eps=eps)
            gm = torch.linspace(-1, 1, m.numel(), dtype=dtype).view_as(m)
            gs = torch.linspace(0.5, -0.5, s.numel(), dtype=dtype).view_as(s)
            (m * gm + s * gs).sum().backward()
            xo = x.clone().requires_grad_(True)
            mo, so = orc.calc_ins_mean_std(xo, eps=eps)
            (mo * gm + so * gs).sum().backward()
            assert same(m, mo) and same(s, so) and same(xr.grad, xo.grad)
            out[f"mean_{tag}_{etag}"] = m.detach().numpy()
            out[f"std_{tag}_{etag}"] = s.detach().numpy()
            out[f"gmean_{tag}"] = gm.numpy()
            out[f"gstd_{tag}"] = gs.numpy()
            out[f"dx_{tag}_{etag}"] = xr.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g2_stats.npz"), **out)


# ------------------------------------------------------------------------------------------- G3
def gen_cn():
    out = {}
    cases = []
    idx = 0
    for shape in ((4, 3, 8, 8), (6, 5, 9, 11)):
        for crop in orc.CROPS:
            for chan in (False, True):
                for lam in (None, 0.3):
                    seed = 100 + idx
                    for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
                        x = conditioned_input(shape, seed, dtype)
                        g = torch.Generator().manual_seed(seed + 5000)
                        gy = torch.randn(shape, generator=g, dtype=torch.float64).to(dtype)
                        seeded(seed)
                        xr = x.clone().requires_grad_(True)
                        y = ref.cn_op_2ins_space_chan(xr, crop=crop, beta=1, lam=lam, chan=chan)
                        y.backward(gy)
                        seeded(seed)
                        d = orc.draw_cn(shape, crop, beta=1, bbx_thres=0.1, chan=chan)
                        xo = x.clone().requires_grad_(True)
                        yo = orc.cn_op_2ins_space_chan(xo, crop=crop, beta=1, lam=lam, chan=chan,
                                                       draws=d)
                        yo.backward(gy)
                        assert same(y, yo) and same(xr.grad, xo.grad), (shape, crop, chan, lam)
                        k = f"c{idx}"
                        out[f"{k}_x_{tag}"] = x.numpy()
                        out[f"{k}_gy_{tag}"] = gy.numpy()
                        out[f"{k}_y_{tag}"] = y.detach().numpy()
                        out[f"{k}_dx_{tag}"] = xr.grad.numpy()
                    out[f"{k}_perm"] = d.perm.numpy()
                    out[f"{k}_sbox"] = box_arr(d.style_box)
                    out[f"{k}_cbox"] = box_arr(d.content_box)
                    out[f"{k}_chan_perm"] = (d.chan_perm.numpy() if d.chan_perm is not None
                                             else np.zeros(0, dtype=np.int64))
                    cases.append((idx, crop, int(chan), -1.0 if lam is None else lam, seed))
                    idx += 1
    out["case_crop"] = np.array([c[1] for c in cases])
    out["case_chan"] = np.array([c[2] for c in cases], dtype=np.int64)
    out["case_lam"] = np.array([c[3] for c in cases], dtype=np.float64)
    out["case_seed"] = np.array([c[4] for c in cases], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "g3_cn.npz"), **out)


# ------------------------------------------------------------------------------------------- G4
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402


def sn_state(mod):
    return {k: v.detach().clone() for k, v in mod.state_dict().items()}


def gen_sn():
    out = {}
    for is_two in (False, True):
        for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
            shape = (6, 5, 9, 11)
            k = f"two{int(is_two)}_{tag}"
            rm = fill_sn(ref.SelfNorm(shape[1], is_two=is_two), 7, dtype).train()
            om = fill_sn(orc.SelfNorm(shape[1], is_two=is_two), 7, dtype).train()
            assert list(rm.state_dict().keys()) == list(om.state_dict().keys())
            out[f"{k}_keys"] = np.array(list(rm.state_dict().keys()))
            for pk, pv in sn_state(rm).items():
                out[f"{k}_init_{pk}"] = pv.numpy()
            for step in (1, 2):
                x = conditioned_input(shape, 40 + step, dtype)
                g = torch.Generator().manual_seed(900 + step)
                gy = torch.randn(shape, generator=g, dtype=torch.float64).to(dtype)
                res = []
                for m in (rm, om):
                    m.zero_grad()
                    xi = x.clone().requires_grad_(True)
                    y = m(xi)
                    y.backward(gy)
                    res.append((y.detach(), xi.grad, {n: p.grad.clone() for n, p in
                                                      m.named_parameters()}, sn_state(m)))
                (y, dx, pg, st), (yo, dxo, pgo, sto) = res
                assert same(y, yo) and same(dx, dxo)
                assert all(same(pg[n], pgo[n]) for n in pg) and all(same(st[n], sto[n]) for n in st)
                out[f"{k}_s{step}_x"] = x.numpy()
                out[f"{k}_s{step}_gy"] = gy.numpy()
                out[f"{k}_s{step}_y"] = y.numpy()
                out[f"{k}_s{step}_dx"] = dx.numpy()
                for n, v in pg.items():
                    out[f"{k}_s{step}_grad_{n}"] = v.numpy()
                for n, v in st.items():
                    out[f"{k}_s{step}_state_{n}"] = v.numpy()
            rm.eval(), om.eval()
            x = conditioned_input(shape, 43, dtype)
            with torch.no_grad():
                y, yo = rm(x), om(x)
            assert same(y, yo)
            out[f"{k}_eval_x"] = x.numpy()
            out[f"{k}_eval_y"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "g4_sn.npz"), **out)


# ------------------------------------------------------------------------------------------- G5
def gen_cnsn():
    out = {}
    shape = (6, 4, 10, 12)
    idx = 0
    for crop in orc.CROPS:
        for tag, dtype in (("f32", torch.float32), ("f64", torch.float64)):
            k = f"{crop}_{tag}"
            rm = ref.CNSN(ref.CrossNorm(crop=crop, beta=1), fill_sn(ref.SelfNorm(shape[1]), 3,
                                                                       dtype)).train()
            om = orc.CNSN(orc.CrossNorm(crop=crop, beta=1), fill_sn(orc.SelfNorm(shape[1]), 3,
                                                                       dtype)).train()
            x = conditioned_input(shape, 60 + idx, dtype)
            g = torch.Generator().manual_seed(70 + idx)
            gy = torch.randn(shape, generator=g, dtype=torch.float64).to(dtype)
            # (a) armed: CN then SN; the flag must drop afterwards
            seed = 300 + idx
            seeded(seed)
            rm.crossnorm.active = True
            xr = x.clone().requires_grad_(True)
            y = rm(xr)
            y.backward(gy)
            assert rm.crossnorm.active is False
            seeded(seed)
            d = orc.draw_cn(shape, crop, beta=1)
            om.crossnorm.active = True
            om.crossnorm.next_draws = d
            xo = x.clone().requires_grad_(True)
            yo = om(xo)
            yo.backward(gy)
            assert om.crossnorm.active is False
            assert same(y, yo) and same(xr.grad, xo.grad)
            for n, p in rm.named_parameters():
                assert same(p.grad, dict(om.named_parameters())[n].grad)
                out[f"{k}_armed_grad_{n}"] = p.grad.numpy()
            out[f"{k}_x"] = x.numpy()
            out[f"{k}_gy"] = gy.numpy()
            out[f"{k}_armed_y"] = y.detach().numpy()
            out[f"{k}_armed_dx"] = xr.grad.numpy()
            out[f"{k}_perm"] = d.perm.numpy()
            out[f"{k}_sbox"] = box_arr(d.style_box)
            out[f"{k}_cbox"] = box_arr(d.content_box)
            out[f"{k}_seed"] = np.array(seed)
            for n, v in sn_state(rm).items():
                out[f"{k}_armed_state_{n}"] = v.numpy()
            # (b) idle (flag down): SN only, second BN step
            y2, y2o = rm(x), om(x)
            assert same(y2, y2o)
            out[f"{k}_idle_y"] = y2.detach().numpy()
            # (c) eval with the flag raised: CNSN still calls CrossNorm, which is identity in
            #     eval mode but clears the flag (cnsn.py:104-108)
            rm.eval(), om.eval()
            rm.crossnorm.active = True
            om.crossnorm.active = True
            with torch.no_grad():
                y3, y3o = rm(x), om(x)
            assert same(y3, y3o) and rm.crossnorm.active is False and om.crossnorm.active is False
            out[f"{k}_eval_y"] = y3.numpy()
        idx += 1
    # CN-only and SN-absent wiring
    for tag, dtype in (("f32", torch.float32),):
        x = conditioned_input(shape, 99, dtype)
        rm = ref.CNSN(ref.CrossNorm(crop="neither", beta=1), None).train()
        seeded(77)
        rm.crossnorm.active = True
        y = rm(x)
        seeded(77)
        d = orc.draw_cn(shape, "neither", beta=1)
        assert same(y, orc.cn_op_2ins_space_chan(x, draws=d))
        out["cnonly_x"] = x.numpy()
        out["cnonly_y"] = y.numpy()
        out["cnonly_perm"] = d.perm.numpy()
        out["cnonly_idle_is_identity"] = np.array(bool(rm(x) is x))
    np.savez_compressed(os.path.join(HERE, "g5_cnsn.npz"), **out)


if __name__ == "__main__":
    gen_bbox()
    gen_stats()
    gen_cn()
    gen_sn()
    gen_cnsn()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
