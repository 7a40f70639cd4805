"""CPU oracle for the CrossNorm / SelfNorm hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, op for op, what `/root/reference/models/cnsn.py` computes, in plain eager
PyTorch on CPU tensors.  It exists to CHECK the HIP path (tests/, `__graft_entry__.smoke()`) and to
be TIMED as the CPU baseline (`bench.py`'s `cpu_baseline` leg).  Nothing under
`crossnorm-selfnorm_amd/` imports it; the product path has no CPU fallback.

Parity pin: `tests/test_oracle_golden.py` checks every function here against the vectors in
`tests/golden/*.npz`, which `tests/golden/gen_golden.py` produced by importing the reference itself
in the build container (bit-exact agreement in fp32 and fp64 on the same draws).

Differences from the reference, all deliberate and all value-preserving:
  * the random draws (batch permutation, style box, channel permutation, content box) can be handed
    in explicitly through ``draws=`` so that the HIP path and the oracle see the same ones; when
    they are not handed in they are drawn in the reference's order (cnsn.py:62,65,71,76);
  * `cn_rand_bbox` uses ``int()`` where the reference uses the removed ``np.int`` (cnsn.py:39-40);
  * SelfNorm is written with `F.conv1d` / `F.batch_norm` on explicit tensors instead of sub-modules,
    and the modules here are thin state holders around the functions.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Box = Tuple[int, int, int, int]
CROPS = ("neither", "style", "content", "both")


# ------------------------------------------------------------------------------------------------
# a1  plane statistics                                                   reference cnsn.py:8-17
# ------------------------------------------------------------------------------------------------
def calc_ins_mean_std(x: torch.Tensor, eps: float = 1e-5):
    """Per-(n,c) mean and sqrt(unbiased var + eps) over H*W  (cnsn.py:14-16).

    `var(dim=2)` is torch's default unbiased estimator; eps goes inside the square root.
    Returns two (N,C,1,1) tensors, mean first (cnsn.py:17).
    """
    assert x.dim() == 4                                            # cnsn.py:12
    n, c = x.shape[0], x.shape[1]
    # two separate flattenings, as in the reference: autograd then sums the var- and mean-path
    # gradients into x in the same order, which keeps dx bit-identical to the reference's
    std = (x.contiguous().view(n, c, -1).var(dim=2) + eps).sqrt().view(n, c, 1, 1)   # :14-15
    mean = x.contiguous().view(n, c, -1).mean(dim=2).view(n, c, 1, 1)               # :16
    return mean, std


# ------------------------------------------------------------------------------------------------
# a2  re-normalise content with style statistics                        reference cnsn.py:20-29
# ------------------------------------------------------------------------------------------------
def instance_norm_mix(content_feat: torch.Tensor, style_feat: torch.Tensor) -> torch.Tensor:
    """(content - mean_c) / std_c * std_s + mean_s, eps = 1e-5 for both (cnsn.py:24-29)."""
    assert content_feat.shape[:2] == style_feat.shape[:2]          # cnsn.py:22
    shape = content_feat.shape
    s_mean, s_std = calc_ins_mean_std(style_feat)                  # cnsn.py:24 (style first)
    c_mean, c_std = calc_ins_mean_std(content_feat)                # cnsn.py:25
    normalised = (content_feat - c_mean.expand(shape)) / c_std.expand(shape)   # cnsn.py:27-28
    return normalised * s_std.expand(shape) + s_mean.expand(shape)             # cnsn.py:29


# ------------------------------------------------------------------------------------------------
# a3  box sampler                                                        reference cnsn.py:32-55
# ------------------------------------------------------------------------------------------------
def cn_rand_bbox(size, beta, bbx_thres) -> Box:
    """Rejection-sample a box on dims (2,3) from the numpy GLOBAL RNG (cnsn.py:36-53).

    Draw order per attempt: beta(beta,beta) -> randint(size[2]) -> randint(size[3]).
    Extents are truncated (`np.int` there, `int` here), halved with floor division, clipped to the
    plane; accepted when covered fraction > bbx_thres.  Returns (x1, y1, x2, y2) with x on dim 2.
    """
    d2, d3 = int(size[2]), int(size[3])                            # cnsn.py:34-35 ("W","H")
    while True:
        ratio = np.random.beta(beta, beta)                         # cnsn.py:37
        cut = np.sqrt(ratio)                                       # cnsn.py:38
        cut2, cut3 = int(d2 * cut), int(d3 * cut)                  # cnsn.py:39-40
        c2 = np.random.randint(d2)                                 # cnsn.py:43
        c3 = np.random.randint(d3)                                 # cnsn.py:44
        x1 = int(np.clip(c2 - cut2 // 2, 0, d2))                   # cnsn.py:46
        y1 = int(np.clip(c3 - cut3 // 2, 0, d3))                   # cnsn.py:47
        x2 = int(np.clip(c2 + cut2 // 2, 0, d2))                   # cnsn.py:48
        y2 = int(np.clip(c3 + cut3 // 2, 0, d3))                   # cnsn.py:49
        if float(x2 - x1) * (y2 - y1) / (d2 * d3) > bbx_thres:     # cnsn.py:51-52
            return x1, y1, x2, y2


@dataclass
class CNDraws:
    """The random quantities one `cn_op_2ins_space_chan` call consumes, in draw order."""
    perm: torch.Tensor                     # int64 (N,)  torch.randperm(N)          cnsn.py:62
    style_box: Optional[Box] = None        # crop in {style, both}                  cnsn.py:65
    chan_perm: Optional[torch.Tensor] = None   # int64 (C,) when chan             cnsn.py:71
    content_box: Optional[Box] = None      # crop in {content, both}                cnsn.py:76


def draw_cn(size, crop: str, beta, bbx_thres: float = 0.1, chan: bool = False) -> CNDraws:
    """Consume the torch CPU generator and the numpy global RNG exactly as cnsn.py:62-76 does."""
    assert crop in CROPS                                           # cnsn.py:61
    d = CNDraws(perm=torch.randperm(int(size[0])))                 # cnsn.py:62
    if crop in ("style", "both"):
        d.style_box = cn_rand_bbox(size, beta=beta, bbx_thres=bbx_thres)    # cnsn.py:65
    if chan:
        d.chan_perm = torch.randperm(int(size[1]))                 # cnsn.py:71
    if crop in ("content", "both"):
        d.content_box = cn_rand_bbox(size, beta=beta, bbx_thres=bbx_thres)  # cnsn.py:76
    return d


# ------------------------------------------------------------------------------------------------
# a4  2-instance CrossNorm                                                reference cnsn.py:58-91
# ------------------------------------------------------------------------------------------------
def cn_op_2ins_space_chan(x, crop="neither", beta=1, bbx_thres=0.1, lam=None, chan=False,
                          draws: Optional[CNDraws] = None):
    """Swap each instance's plane statistics with those of a permuted partner (cnsn.py:58-91).

    The style source `x[perm]` stays attached to autograd (cnsn.py:66,68).  With a content box only
    the box is re-normalised (with the box's own statistics) and the rest of the plane passes
    through (cnsn.py:75-82).  `lam` blends input and result (cnsn.py:86-87).
    """
    assert crop in CROPS                                           # cnsn.py:61
    if draws is None:
        draws = draw_cn(x.shape, crop, beta, bbx_thres, chan)
    perm = draws.perm.to(x.device)

    if crop in ("style", "both"):
        sx1, sy1, sx2, sy2 = draws.style_box
        style = x[perm, :, sx1:sx2, sy1:sy2]                       # cnsn.py:66
    else:
        style = x[perm]                                            # cnsn.py:68
    if chan:
        style = style[:, draws.chan_perm.to(x.device), :, :]       # cnsn.py:72

    if crop in ("content", "both"):
        cx1, cy1, cx2, cy2 = draws.content_box
        pasted = torch.zeros_like(x)                               # cnsn.py:75
        pasted[:, :, cx1:cx2, cy1:cy2] = instance_norm_mix(
            content_feat=x[:, :, cx1:cx2, cy1:cy2], style_feat=style)      # cnsn.py:77-78
        keep = torch.ones_like(x, requires_grad=False)             # cnsn.py:80
        keep[:, :, cx1:cx2, cy1:cy2] = 0.0                         # cnsn.py:81
        out = x * keep + pasted                                    # cnsn.py:82
    else:
        out = instance_norm_mix(content_feat=x, style_feat=style)  # cnsn.py:84

    if lam is not None:
        return x * lam + out * (1 - lam)                           # cnsn.py:87
    return out                                                     # cnsn.py:89


# ------------------------------------------------------------------------------------------------
# a6  SelfNorm                                                           reference cnsn.py:113-150
# ------------------------------------------------------------------------------------------------
def _sn_gate(stats, fc_weight, bn_weight, bn_bias, run_mean, run_var, training, momentum, bn_eps):
    """Conv1d(C,C,k=2,groups=C,bias=False) -> BatchNorm1d(C) -> sigmoid   (cnsn.py:137-139)."""
    c = stats.shape[1]
    z = F.conv1d(stats, fc_weight, bias=None, groups=c)            # (N,C,1)
    z = F.batch_norm(z, run_mean, run_var, bn_weight, bn_bias, training, momentum, bn_eps)
    return torch.sigmoid(z)


def selfnorm_forward(x, g_fc_w, g_bn_w, g_bn_b, g_run_mean, g_run_var, training=True,
                     f_params=None, momentum=0.1, bn_eps=1e-5, f_training=None):
    """x * g  (or x*g + mean*(f-g) for the two-gate form)  (cnsn.py:130-150).

    Statistics use eps = 1e-12 (cnsn.py:133).  Running buffers are updated in place in training
    mode exactly as `nn.BatchNorm1d` does (momentum 0.1, unbiased running variance).
    `f_params` = (f_fc_w, f_bn_w, f_bn_b, f_run_mean, f_run_var) or None.  `f_training`: the mode of the second gate's
    BatchNorm1d when it differs from the first's (the reference calls `self.g_bn` / `self.f_bn` as modules, each with its
    own `.training`, cnsn.py:138,144); None = the same.
    """
    n, c = x.shape[0], x.shape[1]
    mean, std = calc_ins_mean_std(x, eps=1e-12)                    # cnsn.py:133
    stats = torch.cat((mean.squeeze(3), std.squeeze(3)), -1)       # cnsn.py:135  (N,C,2)
    g = _sn_gate(stats, g_fc_w, g_bn_w, g_bn_b, g_run_mean, g_run_var,
                 training, momentum, bn_eps).view(n, c, 1, 1)      # cnsn.py:137-140
    if f_params is not None:
        f_fc_w, f_bn_w, f_bn_b, f_rm, f_rv = f_params
        f = _sn_gate(stats, f_fc_w, f_bn_w, f_bn_b, f_rm, f_rv, training if f_training is None else f_training,
                     momentum, bn_eps).view(n, c, 1, 1)            # cnsn.py:143-146
        return x * g.expand_as(x) + mean.expand_as(x) * (f.expand_as(x) - g.expand_as(x))  # :148
    return x * g.expand_as(x)                                      # cnsn.py:150


# ------------------------------------------------------------------------------------------------
# a5 / a6 / a7  stateful wrappers with the reference's attribute and state_dict surface
# ------------------------------------------------------------------------------------------------
class CrossNorm(torch.nn.Module):
    """`active` flag + bound op; forward always clears the flag (cnsn.py:94-110)."""

    def __init__(self, crop=None, beta=None):
        super().__init__()
        self.active = False                                        # cnsn.py:99
        self.crop, self.beta = crop, beta
        self.next_draws: Optional[CNDraws] = None                  # test hook: explicit draws

    def cn_op(self, x):                                            # cnsn.py:100-101 (partial)
        d, self.next_draws = self.next_draws, None
        return cn_op_2ins_space_chan(x, crop=self.crop, beta=self.beta, draws=d)

    def forward(self, x):
        if self.training and self.active:                          # cnsn.py:104
            x = self.cn_op(x)                                      # cnsn.py:106
        self.active = False                                        # cnsn.py:108
        return x


class SelfNorm(torch.nn.Module):
    """Same parameters / buffers / state_dict keys as cnsn.py:113-128."""

    def __init__(self, chan_num, is_two=False):
        super().__init__()
        self.g_fc = torch.nn.Conv1d(chan_num, chan_num, kernel_size=2, bias=False, groups=chan_num)
        self.g_bn = torch.nn.BatchNorm1d(chan_num)
        if is_two is True:                                         # cnsn.py:123
            self.f_fc = torch.nn.Conv1d(chan_num, chan_num, kernel_size=2, bias=False,
                                        groups=chan_num)
            self.f_bn = torch.nn.BatchNorm1d(chan_num)
        else:
            self.f_fc = None                                       # cnsn.py:128

    @staticmethod
    def _bn_args(bn):
        # nn.BatchNorm1d bookkeeping (torch/nn/modules/batchnorm.py): count batches in training.
        if bn.training and bn.track_running_stats:
            bn.num_batches_tracked.add_(1)
        return bn.weight, bn.bias, bn.running_mean, bn.running_var

    def forward(self, x):
        gw, gb, grm, grv = self._bn_args(self.g_bn)
        f_params = None
        if self.f_fc is not None:
            fw, fb, frm, frv = self._bn_args(self.f_bn)
            f_params = (self.f_fc.weight, fw, fb, frm, frv)
        return selfnorm_forward(x, self.g_fc.weight, gw, gb, grm, grv, self.g_bn.training, f_params,
                                momentum=self.g_bn.momentum, bn_eps=self.g_bn.eps,
                                f_training=self.f_bn.training if self.f_fc is not None else None)


class CNSN(torch.nn.Module):
    """CrossNorm (only when armed) then SelfNorm; either may be None (cnsn.py:152-164)."""

    def __init__(self, crossnorm, selfnorm):
        super().__init__()
        self.crossnorm = crossnorm
        self.selfnorm = selfnorm

    def forward(self, x):
        if self.crossnorm and self.crossnorm.active:               # cnsn.py:160
            x = self.crossnorm(x)
        if self.selfnorm:                                          # cnsn.py:162
            x = self.selfnorm(x)
        return x
