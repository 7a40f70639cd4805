"""The pass-structured closed form (what the kernels evaluate) == autograd through the op-for-op
oracle, in fp64, for every mode: crop x chan x lam x {CN only, SN only, CN+SN} x is_two x train/eval."""
import itertools

import zlib

import numpy as np
import pytest
import torch

from oracle import cnsn_oracle as orc
from oracle import closed_form as cf


def seed_of(*a):
    return zlib.crc32(repr(a).encode()) % 100000


def cond_input(shape, seed):
    g = torch.Generator().manual_seed(seed)
    n, c = shape[:2]
    x = torch.randn(shape, generator=g, dtype=torch.float64)
    s = torch.rand(n, c, 1, 1, generator=g, dtype=torch.float64) * 1.5 + 0.5
    m = torch.randn(n, c, 1, 1, generator=g, dtype=torch.float64)
    return x * s + m


def make_sn(C, is_two, seed):
    g = torch.Generator().manual_seed(seed)
    def branch():
        return dict(w=(torch.rand(C, 2, generator=g, dtype=torch.float64) * 1.4 - 0.7).requires_grad_(),
                    gamma=(torch.rand(C, generator=g, dtype=torch.float64) + 0.5).requires_grad_(),
                    beta=(torch.rand(C, generator=g, dtype=torch.float64) - 0.5).requires_grad_(),
                    run_mean=torch.rand(C, generator=g, dtype=torch.float64) - 0.5,
                    run_var=torch.rand(C, generator=g, dtype=torch.float64) + 0.5)
    sn = branch()
    sn["f"] = branch() if is_two else None
    return sn


MODES = list(itertools.product(orc.CROPS, (False, True), (None, 0.3), ("cn", "sn", "cnsn"),
                               (False, True), (True, False)))


@pytest.mark.parametrize("crop,chan,lam,kind,is_two,training", MODES)
def test_closed_form_matches_autograd(crop, chan, lam, kind, is_two, training):
    if kind == "sn" and (crop != "neither" or chan or lam is not None):
        pytest.skip("CN options irrelevant for SN only")
    if kind == "cn" and (is_two or not training):
        pytest.skip("SN options irrelevant for CN only")
    shape = (6, 5, 9, 11)
    seed = seed_of(crop, chan, lam, kind, is_two, training)
    torch.manual_seed(seed)
    np.random.seed(seed)
    x = cond_input(shape, seed).requires_grad_()
    G = torch.randn(shape, dtype=torch.float64)
    d = orc.draw_cn(shape, crop, beta=1, chan=chan)
    sn = make_sn(shape[1], is_two, seed + 1) if kind != "cn" else None

    # --- oracle + autograd
    u = x
    if kind != "sn":
        u = orc.cn_op_2ins_space_chan(u, crop=crop, lam=lam, chan=chan, draws=d)
    y_ref = u
    rm = rv = None
    if sn is not None:
        rm, rv = sn["run_mean"].clone(), sn["run_var"].clone()
        fp = None
        frm = frv = None
        if is_two:
            f = sn["f"]
            frm, frv = f["run_mean"].clone(), f["run_var"].clone()
            fp = (f["w"].view(-1, 1, 2), f["gamma"], f["beta"], frm, frv)
        y_ref = orc.selfnorm_forward(u, sn["w"].view(-1, 1, 2), sn["gamma"], sn["beta"], rm, rv,
                                     training=training, f_params=fp)
    params = []
    if sn is not None:
        params = [sn["w"], sn["gamma"], sn["beta"]]
        if is_two:
            params += [sn["f"]["w"], sn["f"]["gamma"], sn["f"]["beta"]]
    grads = torch.autograd.grad(y_ref, [x] + params, G)

    # --- closed form
    cn = None
    if kind != "sn":
        cn = dict(perm=d.perm, chan_perm=d.chan_perm, cbox=d.content_box, sbox=d.style_box, lam=lam)
    snd = None
    if sn is not None:
        snd = dict(sn, training=training, eps_bn=1e-5, momentum=0.1)
    with torch.no_grad():
        y, S = cf.fused_forward(x.detach(), cn=cn, sn=snd)
        dx, pg = cf.fused_backward(G, S)
    tol = dict(rtol=1e-10, atol=1e-11)
    torch.testing.assert_close(y, y_ref.detach(), **tol)
    torch.testing.assert_close(dx, grads[0], **tol)
    if sn is not None:
        torch.testing.assert_close(pg["g_dw"], grads[1], **tol)
        torch.testing.assert_close(pg["g_dgamma"], grads[2], **tol)
        torch.testing.assert_close(pg["g_dbeta"], grads[3], **tol)
        if is_two:
            torch.testing.assert_close(pg["f_dw"], grads[4], **tol)
            torch.testing.assert_close(pg["f_dgamma"], grads[5], **tol)
            torch.testing.assert_close(pg["f_dbeta"], grads[6], **tol)
        if training:
            torch.testing.assert_close(S["new_running"][0][0], rm, **tol)
            torch.testing.assert_close(S["new_running"][0][1], rv, **tol)
