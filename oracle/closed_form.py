"""Closed-form (pass-structured) restatement of fused CrossNorm+SelfNorm.  TEST INFRASTRUCTURE ONLY.

`cnsn_oracle.py` follows the reference op for op and lets autograd derive the backward.  This file
states the SAME function the way the HIP kernels evaluate it — plane statistics, a per-plane
"mid" stage on N*C scalars, and a piecewise-affine apply — with the backward written out by hand
(extended here with `lam` and the two-gate SelfNorm).  It is vectorised
torch in whatever dtype it is handed (tests use fp64) and is checked against autograd through
`cnsn_oracle.py` in `tests/test_closed_form.py`; the kernels' device code mirrors it term by term.

Notation (per plane p=(n,c), M=H*W):
  Bc content box (whole plane without one), Mc=|Bc|, Mo=M-Mc;  Bs style box, Ms=|Bs|
  q = (perm[n], chan_perm[c]) the style source plane
  eps1=1e-5 (CrossNorm, cnsn.py:8), eps2=1e-12 (SelfNorm, cnsn.py:133), eps_bn=1e-5
"""
from __future__ import annotations

import torch


def _region_masks(H, W, box, dtype, device):
    m = torch.zeros(H, W, dtype=dtype, device=device)
    if box is None:
        m[:] = 1
    else:
        x1, y1, x2, y2 = box
        m[x1:x2, y1:y2] = 1
    return m


def _moments(x, mask):
    """count, mean, M2 of x over mask (mask broadcast over N,C)."""
    cnt = mask.sum()
    mean = (x * mask).sum((2, 3)) / cnt if cnt > 0 else torch.zeros_like(x[:, :, 0, 0])
    m2 = (((x - mean[:, :, None, None]) ** 2) * mask).sum((2, 3))
    return cnt, mean, m2


def fused_forward(x, *, cn=None, sn=None, eps1=1e-5, eps2=1e-12):
    """cn = dict(perm, chan_perm|None, cbox|None, sbox|None, lam|None) or None
    sn = dict(w (C,2), gamma, beta, run_mean, run_var, training, eps_bn, momentum,
              f=None|dict(w,gamma,beta,run_mean,run_var)) or None
    Returns y and a `saved` dict for `fused_backward`.  Running stats are NOT updated here
    (returned as saved['new_running'] for the test to compare).
    """
    N, C, H, W = x.shape
    M = H * W
    dt, dev = x.dtype, x.device
    S = {"shape": (N, C, H, W)}
    one = torch.ones((), dtype=dt, device=dev)

    if cn is not None:
        cmask = _region_masks(H, W, cn.get("cbox"), dt, dev)
        smask = _region_masks(H, W, cn.get("sbox"), dt, dev)
        omask = 1 - cmask
        Mc, mu_c, M2c = _moments(x, cmask)
        Mo, mu_o, M2o = _moments(x, omask)
        Ms, mu_s_all, M2s = _moments(x, smask)
        sig_c = (M2c / (Mc - 1) + eps1).sqrt()
        sig_s_all = (M2s / (Ms - 1) + eps1).sqrt()
        perm = cn["perm"]
        cperm = cn.get("chan_perm")
        mu_s, sig_s = mu_s_all[perm], sig_s_all[perm]
        if cperm is not None:
            mu_s, sig_s = mu_s[:, cperm], sig_s[:, cperm]
        lam = cn.get("lam")
        lam = 0.0 if lam is None else float(lam)
        a = sig_s / sig_c
        a1 = lam + (1 - lam) * a                     # slope inside Bc
        m_in = lam * mu_c + (1 - lam) * mu_s         # mean of u inside Bc  (= a1*mu_c + b1)
        # post-CN whole-plane moments by Chan's merge (no third read)
        mu_p = (Mc * m_in + Mo * mu_o) / M
        M2p = a1 ** 2 * M2c + M2o + (m_in - mu_o) ** 2 * Mc * Mo / M
        S.update(cmask=cmask, smask=smask, Mc=Mc, Mo=Mo, Ms=Ms, mu_c=mu_c, M2c=M2c, mu_o=mu_o,
                 sig_c=sig_c, mu_s_src=mu_s_all, sig_s_src=sig_s_all, a=a, a1=a1, m_in=m_in,
                 lam=lam, perm=perm, chan_perm=cperm)
    else:
        full = _region_masks(H, W, None, dt, dev)
        _, mu_p, M2p = _moments(x, full)
        a1 = one.expand(N, C)
        m_in = mu_p
        mu_c = mu_p
        S.update(cmask=full, Mc=torch.tensor(float(M), dtype=dt), Mo=torch.tensor(0.0, dtype=dt),
                 mu_c=mu_p, mu_o=torch.zeros_like(mu_p), a1=a1, m_in=m_in)

    if sn is not None:
        sig_p = (M2p / (M - 1) + eps2).sqrt()
        gates = []
        new_running = []
        for br in (sn, sn.get("f")):
            if br is None:
                gates.append(None)
                continue
            w = br["w"]
            z = w[:, 0] * mu_p + w[:, 1] * sig_p                 # (N,C)
            if sn["training"]:
                m = z.mean(0)
                v = z.var(0, unbiased=False)
                mom = sn.get("momentum", 0.1)
                new_running.append(((1 - mom) * br["run_mean"] + mom * m,
                                    (1 - mom) * br["run_var"] + mom * v * N / (N - 1)))
            else:
                m, v = br["run_mean"], br["run_var"]
            r = 1.0 / (v + sn.get("eps_bn", 1e-5)).sqrt()
            zh = (z - m) * r
            gate = torch.sigmoid(br["gamma"] * zh + br["beta"])
            gates.append(dict(gate=gate, zh=zh, r=r))
        g, f = gates[0]["gate"], (gates[1]["gate"] if gates[1] is not None else None)
        S.update(sig_p=sig_p, mu_p=mu_p, gates=gates, new_running=new_running)
    else:
        g, f = one.expand(N, C), None
        S.update(mu_p=mu_p, gates=None)

    shift = mu_p * (f - g) if f is not None else torch.zeros_like(mu_p)
    cm = S["cmask"]
    e = lambda t: t[:, :, None, None]
    y_in = e(g * a1) * (x - e(S["mu_c"])) + e(g * m_in + shift)
    y_out = e(g) * x + e(shift)
    y = cm * y_in + (1 - cm) * y_out
    S.update(g=g, f=f, x=x, cn=cn, sn=sn)
    return y, S


def fused_backward(G, S):
    """Returns dx and dict of parameter grads (dw (C,2), dgamma, dbeta [, f_*])."""
    x = S["x"]
    N, C, H, W = S["shape"]
    M = H * W
    cn, sn = S["cn"], S["sn"]
    cm = S["cmask"]
    om = 1 - cm
    e = lambda t: t[:, :, None, None]
    g, f = S["g"], S["f"]
    a1, m_in, mu_c, mu_o, mu_p = S["a1"], S["m_in"], S["mu_c"], S["mu_o"], S["mu_p"]
    Mc, Mo = S["Mc"], S["Mo"]

    # pass A': four sums per plane
    S1in = (G * cm).sum((2, 3))
    S2in = (G * (x - e(mu_c)) * cm).sum((2, 3))
    S1out = (G * om).sum((2, 3))
    S2out = (G * (x - e(mu_o)) * om).sum((2, 3))
    S1 = S1in + S1out
    Gu_dot_u = a1 * S2in + m_in * S1in + S2out + mu_o * S1out        # sum G*u

    grads = {}
    dmu_p = torch.zeros_like(mu_p)
    dsig_p = torch.zeros_like(mu_p)
    if sn is not None:
        sig_p = S["sig_p"]
        two = f is not None
        d_gate = [Gu_dot_u - mu_p * S1 if two else Gu_dot_u, mu_p * S1 if two else None]
        if two:
            dmu_p = dmu_p + (f - g) * S1
        for name, br, gi, dgate in (("g", sn, S["gates"][0], d_gate[0]),
                                    ("f", sn.get("f"), S["gates"][1], d_gate[1])):
            if br is None:
                continue
            gate, zh, r = gi["gate"], gi["zh"], gi["r"]
            dt = dgate * gate * (1 - gate)
            grads[name + "_dgamma"] = (dt * zh).sum(0)
            grads[name + "_dbeta"] = dt.sum(0)
            if sn["training"]:
                dz = br["gamma"] * r * (dt - dt.mean(0) - zh * (dt * zh).mean(0))
            else:
                dz = br["gamma"] * r * dt
            grads[name + "_dw"] = torch.stack(((dz * mu_p).sum(0), (dz * sig_p).sum(0)), 1)
            dmu_p = dmu_p + dz * br["w"][:, 0]
            dsig_p = dsig_p + dz * br["w"][:, 1]
        k = dsig_p / (sig_p * (M - 1))
    else:
        k = torch.zeros_like(mu_p)

    # dL/du = g*G + dmu_p/M + k*(u - mu_p)
    if cn is not None:
        M2c, sig_c, a, lam = S["M2c"], S["sig_c"], S["a"], S["lam"]
        Ms = S["Ms"]
        T1 = g * S1in + Mc * dmu_p / M + k * Mc * (m_in - mu_p)
        T2 = g * S2in + k * a1 * M2c
        # u_in = a1*(x-mu_c) + m_in with a1 = lam+(1-lam)a, m_in = lam*mu_c+(1-lam)*mu_s:
        #   d/d a1  = T2,  d/d m_in = T1,  d/d mu_c (explicit, through -a1*mu_c) = -a1*T1
        d_a = (1 - lam) * T2
        Dmu_c = -a1 * T1 + lam * T1                              # = -(1-lam)*a*T1
        Dsig_c = -d_a * a / sig_c
        Dmu_s_at_p = (1 - lam) * T1                              # gradient wrt mu_s[q(p)]
        Dsig_s_at_p = d_a / sig_c
        # scatter to the style source planes r = q(p)
        perm, cperm = S["perm"], S["chan_perm"]
        Dmu_s = torch.zeros_like(T1)
        Dsig_s = torch.zeros_like(T1)
        if cperm is None:
            Dmu_s[perm] = Dmu_s_at_p
            Dsig_s[perm] = Dsig_s_at_p
        else:
            tmp_m = torch.zeros_like(T1)
            tmp_s = torch.zeros_like(T1)
            tmp_m[:, cperm] = Dmu_s_at_p
            tmp_s[:, cperm] = Dsig_s_at_p
            Dmu_s[perm] = tmp_m
            Dsig_s[perm] = tmp_s
        sm = S["smask"]
        mu_s_src, sig_s_src = S["mu_s_src"], S["sig_s_src"]
        u = cm * (e(a1) * (x - e(mu_c)) + e(m_in)) + om * x
        Gu = e(g) * G + e(dmu_p) / M + e(k) * (u - e(mu_p))
        dx = (cm * e(a1) + om) * Gu
        dx = dx + cm * (e(Dmu_c) / Mc + e(Dsig_c / (sig_c * (Mc - 1))) * (x - e(mu_c)))
        dx = dx + sm * (e(Dmu_s) / Ms + e(Dsig_s / (sig_s_src * (Ms - 1))) * (x - e(mu_s_src)))
    else:
        dx = e(g) * G + e(dmu_p) / M + e(k) * (x - e(mu_p))
    return dx, grads
