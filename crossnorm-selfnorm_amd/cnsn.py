"""Drop-in for the reference's `models/cnsn.py`: the same seven names, signatures, attributes,
RNG draw order and `state_dict` keys — computed by the fused HIP kernels of libcnsn_hip.so.

    reference (models/cnsn.py)            here
    calc_ins_mean_std        :8-17        one stats kernel (cnsn_plane_stats)
    instance_norm_mix        :20-29       stats x2 + one affine kernel
    cn_rand_bbox             :32-55       host, numpy global RNG, same draw order
    cn_op_2ins_space_chan    :58-91       fused forward/backward (cnsn_forward / cnsn_backward)
    CrossNorm                :94-110      flag holder; arms the fused call
    SelfNorm                 :113-150     parameter holder (g_fc, g_bn[, f_fc, f_bn]); fused call
    CNSN                     :152-164     ONE fused call for CrossNorm+SelfNorm on the tensor

Use: replace the body of `models/cnsn.py` by `from cnsn_amd.cnsn import *` — the
model files (`from ..cnsn import CrossNorm, SelfNorm, CNSN`), `isinstance(m, CrossNorm)` site
collection, `m.active = True`, and checkpoints keep working unmodified.

HIP device tensors only; a CPU tensor raises (there is deliberately no fallback path).
"""
from __future__ import annotations

import functools
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import functional as _F
from .functional import FusedConfig, GateParams

__all__ = ["calc_ins_mean_std", "instance_norm_mix", "cn_rand_bbox", "cn_op_2ins_space_chan",
           "CrossNorm", "SelfNorm", "CNSN"]

Box = Tuple[int, int, int, int]
_CROPS = ("neither", "style", "content", "both")


def _is_f64(x) -> bool:
    return isinstance(x, torch.Tensor) and x.dtype == torch.float64


def calc_ins_mean_std(x, eps=1e-5):
    """Per-(n,c) mean and sqrt(unbiased variance + eps) over H*W, each shaped (N,C,1,1)
    (reference models/cnsn.py:8-17).  Differentiable.

    float64 (the reference takes any floating dtype, :12-16): the parameter-free ops — this one, `instance_norm_mix`,
    `cn_op_2ins_space_chan` / `CrossNorm` — accept it, compute on a float32 copy with the float32 kernels (whose plane
    statistics and scalar algebra are fp32 / fp64 as for any input) and return float64: float32 ACCURACY in a float64
    container, differentiable through the two casts.  So do `SelfNorm` and `CNSN` (round 6; models/cnsn.py:130-150 takes any
    floating dtype): float64 activations go through the float32 kernels, the gate's parameters — float32 or, after
    `module.double()`, float64 — are read as float32 and get gradients in their own dtype."""
    assert x.dim() == 4
    if _is_f64(x):
        mean, std = _F.PlaneStats.apply(x.float(), float(eps), None, True)
        return mean.double(), std.double()
    return _F.PlaneStats.apply(x, float(eps), None)


def instance_norm_mix(content_feat, style_feat):
    """Give `content_feat` the plane statistics of `style_feat` (reference models/cnsn.py:20-29)."""
    assert content_feat.size()[:2] == style_feat.size()[:2]
    if _is_f64(content_feat):
        return instance_norm_mix(content_feat.float(), style_feat.float()).double()
    s_mean, s_std = calc_ins_mean_std(style_feat)
    c_mean, c_std = calc_ins_mean_std(content_feat)
    scale = s_std.float() / c_std.float()          # (N,C,1,1) scalars per plane; negligible work
    shift = s_mean.float() - c_mean.float() * scale
    return _F.PlaneAffine.apply(content_feat, scale, shift)


def cn_rand_bbox(size, beta, bbx_thres):
    """Sample the crop box of CrossNorm (reference models/cnsn.py:32-55).

    Same stream consumption as the reference — per attempt one `np.random.beta`, then
    `np.random.randint(size[2])`, then `np.random.randint(size[3])` from numpy's global generator —
    so a seeded run draws the same boxes.  Returns (x1, y1, x2, y2), x along dim 2, y along dim 3.
    """
    ext2, ext3 = int(size[2]), int(size[3])
    plane = float(ext2 * ext3)
    while True:
        side = np.sqrt(np.random.beta(beta, beta))
        half2, half3 = int(ext2 * side) // 2, int(ext3 * side) // 2   # truncate, then floor-halve
        mid2 = np.random.randint(ext2)
        mid3 = np.random.randint(ext3)
        x1, x2 = max(mid2 - half2, 0), min(mid2 + half2, ext2)
        y1, y2 = max(mid3 - half3, 0), min(mid3 + half3, ext3)
        if (x2 - x1) * (y2 - y1) / plane > bbx_thres:
            return int(x1), int(y1), int(x2), int(y2)


@dataclass
class CNDraws:
    """What one CrossNorm application consumes from the RNGs (reference draw order, cnsn.py:62-76)."""
    perm: torch.Tensor                       # int64 (N), CPU: torch.randperm(N)
    style_box: Optional[Box] = None
    chan_perm: Optional[torch.Tensor] = None
    content_box: Optional[Box] = None


def draw_cn(size, crop, beta, bbx_thres=0.1, chan=False) -> CNDraws:
    """torch.randperm(N) [CPU generator] -> style box -> torch.randperm(C) if chan -> content box."""
    assert crop in _CROPS
    d = CNDraws(perm=torch.randperm(int(size[0])))
    if crop in ("style", "both"):
        d.style_box = cn_rand_bbox(size, beta=beta, bbx_thres=bbx_thres)
    if chan:
        d.chan_perm = torch.randperm(int(size[1]))
    if crop in ("content", "both"):
        d.content_box = cn_rand_bbox(size, beta=beta, bbx_thres=bbx_thres)
    return d


def _cn_config(d: CNDraws, lam) -> FusedConfig:
    return FusedConfig(cn_active=True, content_box=d.content_box, style_box=d.style_box, lam=lam)


def cn_op_2ins_space_chan(x, crop='neither', beta=1, bbx_thres=0.1, lam=None, chan=False, draws=None):
    """2-instance CrossNorm with optional cropping (reference models/cnsn.py:58-91).

    `draws` (a CNDraws) is an extension for tests: when given, nothing is drawn from the RNGs.
    """
    assert crop in _CROPS
    if draws is None:
        draws = draw_cn(x.size(), crop, beta, bbx_thres, chan)
    if _is_f64(x):   # (see calc_ins_mean_std)
        return _F.fused_cnsn(x.float(), _cn_config(draws, lam), perm=draws.perm, chan_perm=draws.chan_perm).double()
    return _F.fused_cnsn(x, _cn_config(draws, lam), perm=draws.perm, chan_perm=draws.chan_perm)


class CrossNorm(nn.Module):
    """CrossNorm site (reference models/cnsn.py:94-110): stateless apart from the `active` flag
    that the network raises on randomly chosen sites before a forward (`_enable_cross_norm`)."""

    def __init__(self, crop=None, beta=None):
        super().__init__()
        self.active = False
        self.crop, self.beta = crop, beta
        self.cn_op = functools.partial(cn_op_2ins_space_chan, crop=crop, beta=beta)
        self.next_draws: Optional[CNDraws] = None      # test hook: use these instead of drawing

    def _take_draws(self, x) -> CNDraws:
        d, self.next_draws = self.next_draws, None
        return d if d is not None else draw_cn(x.size(), self.crop, self.beta)

    def forward(self, x):
        if self.training and self.active:
            x = self.cn_op(x, draws=self._take_draws(x))
        self.active = False                            # always dropped, also in eval (cnsn.py:108)
        return x


class SelfNorm(nn.Module):
    """SelfNorm (reference models/cnsn.py:113-150).  `g_fc`/`g_bn` (and `f_fc`/`f_bn`) are kept as
    the same nn.Conv1d / nn.BatchNorm1d sub-modules so parameters, buffers and state_dict keys are
    identical; they are never called — their tensors feed the fused kernel."""

    def __init__(self, chan_num, is_two=False):
        super().__init__()
        self.g_fc = nn.Conv1d(chan_num, chan_num, kernel_size=2, bias=False, groups=chan_num)
        self.g_bn = nn.BatchNorm1d(chan_num)
        if is_two is True:
            self.f_fc = nn.Conv1d(chan_num, chan_num, kernel_size=2, bias=False, groups=chan_num)
            self.f_bn = nn.BatchNorm1d(chan_num)
        else:
            self.f_fc = None

    @staticmethod
    def _gate(fc, bn, counter=None) -> GateParams:
        rm, rv = bn.running_mean, bn.running_var
        if rm is None or rv is None:      # track_running_stats=False: batch statistics always; the kernel's
            w = fc.weight                 # running-buffer update goes to scratch that nobody reads
            rm = torch.zeros(w.shape[0], dtype=torch.float32, device=w.device)
            rv = torch.ones(w.shape[0], dtype=torch.float32, device=w.device)
        return GateParams(fc.weight, bn.weight, bn.bias, rm, rv, counter)

    @staticmethod
    def _bn_call_state(bn, in_kernel=False):
        """What nn.BatchNorm1d.forward decides per call (torch/nn/modules/batchnorm.py): the counter moves only
        under `training and track_running_stats`; `momentum=None` means the cumulative average 1/num_batches_tracked;
        batch statistics are used in training mode or when there are no running buffers.
        `in_kernel`: the caller hands `num_batches_tracked` to the forward launch, which adds the 1 itself
        (cnsn_gate_t.num_batches_tracked) — returned as a fourth value, None when the counter was moved here (the
        cumulative average needs its new value on the host) or does not move at all."""
        momentum = 0.0 if bn.momentum is None else float(bn.momentum)
        counter = None
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            if in_kernel and bn.momentum is not None and bn.num_batches_tracked.is_cuda and not _F.LEGACY_LAUNCHES:
                counter = bn.num_batches_tracked
            else:
                bn.num_batches_tracked.add_(1)
                if bn.momentum is None:
                    momentum = 1.0 / float(bn.num_batches_tracked)
        use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
        if in_kernel:
            return use_batch, float(bn.eps), momentum, counter
        return use_batch, float(bn.eps), momentum

    @staticmethod
    def _bn_peek(bn):
        """`_bn_call_state` without moving the counter (the cumulative-average momentum of `momentum=None` left out:
        it changes every call and both gates' counters move together)"""
        use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
        return use_batch, float(bn.eps), (None if bn.momentum is None else float(bn.momentum)), bool(
            bn.training and bn.track_running_stats)

    def _fusable(self) -> bool:
        """Can ONE launch evaluate this module's gates?  Yes for what the constructor builds — plain, affine
        nn.BatchNorm1d modules, both gates of `is_two` in one configuration.  No when
          * a gate's BatchNorm1d has become an nn.SyncBatchNorm: the reference's segmentation trainer converts every
            BatchNorm of the model when `sync_bn` is set (segmentation/tool/train_cnsn.py:160), and the gate's batch
            statistic then spans the GLOBAL batch — a collective the launch cannot make;
          * `g_bn` and `f_bn` differ in mode, eps or momentum (one of them frozen, say): the kernels evaluate both
            gates with one BatchNorm1d configuration;
          * a gate's BatchNorm1d is not affine (no weight / bias to hand to the kernel).
        `forward` then composes the op from its building blocks (`_forward_composed`)."""
        mods = self._modules                        # (plain dict reads: this runs on every forward)
        g_bn = mods["g_bn"]

        def affine(bn):
            # a replica made by nn.DataParallel (cifar.py:395) keeps its parameters as plain attributes in __dict__ and an EMPTY
            # `_parameters` (torch/nn/parallel/replicate.py): look in both places (tests/test_gpu_data_parallel_one_device.py)
            par, own = bn._parameters, bn.__dict__
            return (par.get("weight") if "weight" in par else own.get("weight")) is not None and \
                   (par.get("bias") if "bias" in par else own.get("bias")) is not None

        if type(g_bn) is not nn.BatchNorm1d or not affine(g_bn):
            return False
        if self.f_fc is None:
            return True
        f_bn = mods["f_bn"]
        if type(f_bn) is not nn.BatchNorm1d or not affine(f_bn):
            return False
        return self._bn_peek(g_bn) == self._bn_peek(f_bn)

    def _forward_composed(self, x):
        """The reference's own sequence (models/cnsn.py:130-150) on this library's building blocks: plane statistics
        (cnsn_plane_stats, differentiable), the two-tap FC and the gate's BatchNorm modules CALLED as modules in fp32 —
        so an nn.SyncBatchNorm all-reduces its (sum, sum of squares, count) over the process group forward and its two
        sums backward, exactly as in the reference —, then one per-plane affine launch (cnsn_plane_affine):
        3 + 5 tensor passes instead of 2 + 3, (N, C)-sized torch ops in between."""
        b, c = int(x.size(0)), int(x.size(1))
        mean, std = _F.PlaneStats.apply(x, 1e-12, None, True)                 # fp32 (N,C,1,1)
        with torch.autocast(device_type=x.device.type, enabled=False):
            stats = torch.cat((mean.view(b, c, 1), std.view(b, c, 1)), -1)    # (N,C,2), cnsn.py:134
            g_y = torch.sigmoid(self.g_bn(self.g_fc(stats))).view(b, c, 1, 1)
            if self.f_fc is not None:
                f_y = torch.sigmoid(self.f_bn(self.f_fc(stats))).view(b, c, 1, 1)
                shift = mean * (f_y - g_y)                                    # cnsn.py:148
            else:
                shift = torch.zeros_like(g_y)
        return _F.PlaneAffine.apply(x, g_y, shift)

    def _fused_args(self):
        """(config fields, g, f) for the fused call; does each BatchNorm1d's own per-call book-keeping."""
        *state, counter = self._bn_call_state(self.g_bn, in_kernel=True)
        state = tuple(state)
        g = self._gate(self.g_fc, self.g_bn, counter)
        f = None
        if self.f_fc is not None:
            *state_f, counter_f = self._bn_call_state(self.f_bn, in_kernel=True)
            state_f = tuple(state_f)
            if state_f != state:   # (`_fusable` has sent differing configurations to `_forward_composed`;
                raise _F._ffi.CnsnError(   # what is left: momentum=None with counters that have drifted apart)
                    "SelfNorm(is_two=True): g_bn and f_bn must share mode, eps and momentum "
                    f"(g_bn: training/eps/momentum = {state}, f_bn: {state_f})")
            f = self._gate(self.f_fc, self.f_bn, counter_f)
        use_batch, eps, momentum = state
        kw = dict(sn_active=True, sn_two=f is not None, sn_training=use_batch, eps_bn=eps, momentum=momentum)
        return kw, g, f

    def _fused_args_peek(self):
        """the configuration `_fused_args` would return, WITHOUT BatchNorm1d's per-call book-keeping (planning only)"""
        bn = self.g_bn
        use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
        kw = dict(sn_active=True, sn_two=self.f_fc is not None, sn_training=use_batch, eps_bn=float(bn.eps),
                  momentum=0.0 if bn.momentum is None else float(bn.momentum))
        return kw, None, None

    def forward(self, x):
        if _is_f64(x):     # (float32 accuracy in a float64 container: see calc_ins_mean_std)
            return self.forward(x.float()).double()
        if not self._fusable():
            return self._forward_composed(x)
        kw, g, f = self._fused_args()
        return _F.fused_cnsn(x, FusedConfig(**kw), g=g, f=f)


class CNSN(nn.Module):
    """CrossNorm then SelfNorm (reference models/cnsn.py:152-164), issued as ONE fused call when
    both apply: 2 tensor reads + 1 write forward instead of the reference's ~15 passes."""

    def __init__(self, crossnorm, selfnorm):
        super().__init__()
        self.crossnorm = crossnorm
        self.selfnorm = selfnorm

    def forward(self, x):
        if _is_f64(x):
            return self.forward(x.float()).double()
        cn, sn = self.crossnorm, self.selfnorm
        fuse = (cn is not None and sn is not None and type(cn) is CrossNorm and type(sn) is SelfNorm and sn._fusable()
                and cn.active and cn.training)
        if not fuse:                                    # literal reference control flow
            if cn and cn.active:
                x = cn(x)
            if sn:
                x = sn(x)
            return x
        d = cn._take_draws(x)
        cn.active = False
        kw, g, f = sn._fused_args()
        cfg = FusedConfig(cn_active=True, content_box=d.content_box, style_box=d.style_box, **kw)
        return _F.fused_cnsn(x, cfg, perm=d.perm, chan_perm=d.chan_perm, g=g, f=f)

    def forward_block(self, x, addend=None, add_mode="none", relu=False):
        """`act(self(x [+ addend]) [+ addend])` — the op together with the residual block's element-wise
        neighbours, in the op's own launches (cnsn_forward_fused).  What the reference spells as
          out += identity; out = self.cnsn(out); out = self.relu(out)   (imagenet/resnet_cnsn.py:117-122, pos='post')
          out = self.cnsn(out); out += identity; out = self.relu(out)   (:112-122, pos='residual'/'identity')
          out = torch.add(x, out); return self.cnsn(out)                (cifar/wideresnet_cnsn.py:93-96, pos='post')
        add_mode: 'none' | 'pre' (sum feeds the op) | 'post' (sum after the op); relu: ReLU last.
        RNG draws, `active` reset and BatchNorm1d book-keeping are those of `forward`."""
        assert add_mode in ("none", "pre", "post")
        assert (addend is None) == (add_mode == "none")
        if _is_f64(x):
            return self.forward_block(x.float(), None if addend is None else addend.float(), add_mode, relu).double()
        cn, sn = self.crossnorm, self.selfnorm
        ours = (cn is None or type(cn) is CrossNorm) and (sn is None or (type(sn) is SelfNorm and sn._fusable()))
        cn_on = cn is not None and cn.active and cn.training
        if not ours or not (cn_on or sn is not None):       # nothing of ours to fuse into: plain ops
            if add_mode == "pre":
                x = x + addend
            x = self.forward(x)
            if add_mode == "post":
                x = x + addend
            return torch.relu(x) if relu else x
        kw, g, f, perm, chan = {}, None, None, None, None
        if cn is not None and cn.active:
            if cn_on:
                d = cn._take_draws(x)
                kw.update(cn_active=True, content_box=d.content_box, style_box=d.style_box)
                perm, chan = d.perm, d.chan_perm
            cn.active = False                               # always dropped (cnsn.py:108)
        if sn is not None:
            skw, g, f = sn._fused_args()
            kw.update(skw)
        cfg = FusedConfig(add_mode=add_mode, relu=bool(relu), **kw)
        return _F.fused_cnsn(x, cfg, perm=perm, chan_perm=chan, g=g, f=f, addend=addend)

    def forward_bn_block(self, conv_out, bn, identity, relu=True, identity_bn=None):
        """`act(self(bn(conv_out) + identity))` — the tail of a ResNet bottleneck:
            out = self.bn3(out); out += identity; out = self.cnsn(out); out = self.relu(out)    (imagenet/resnet_cnsn.py:108-122)
        `identity_bn`: the skip path ends in a BatchNorm2d of its own (the block's `downsample`, :99-100) and `identity` is the
        INPUT of that layer — `act(self(bn(conv_out) + identity_bn(identity)))`.
        ONE launch per direction when the fused kernels take the call (`functional.bn_block_plan`: channels-last tensors,
        SelfNorm alone — the site's CrossNorm idle — with one gate, plain affine nn.BatchNorm2d layers, everything in training
        mode, N <= 256): 13 tensor passes per block and step instead of BatchNorm2d's 8 (16 with the downsample's) + the op's 10;
        otherwise the same steps as separate calls (the BatchNorm2d layers, then `forward_block`).  BatchNorm2d's per-call
        book-keeping (`num_batches_tracked`, `momentum=None`) is done here exactly as `nn.BatchNorm2d.forward` does it."""
        cn, sn = self.crossnorm, self.selfnorm
        armed = cn is not None and cn.active

        def plain(m):
            return type(m) is nn.BatchNorm2d and m.affine and m.track_running_stats and m.training
        fused = (plain(bn) and (identity_bn is None or plain(identity_bn)) and sn is not None
                 and type(sn) is SelfNorm and sn.f_fc is None and sn._fusable() and not armed and conv_out.is_cuda
                 and (cn is None or type(cn) is CrossNorm) and identity.shape == conv_out.shape
                 and identity.dtype == conv_out.dtype and not torch.cuda.is_current_stream_capturing())
        if fused:
            kw, _, _ = sn._fused_args_peek()
            fused = kw["sn_training"] and _F.bn_block_plan(conv_out, FusedConfig(add_mode="pre", relu=bool(relu), **kw))
        if not fused:
            skip = identity if identity_bn is None else identity_bn(identity)
            return self.forward_block(bn(conv_out), skip, add_mode="pre", relu=relu)
        kw, g, _ = sn._fused_args()
        _, bn_eps, bn_mom, bn_counter = SelfNorm._bn_call_state(bn, in_kernel=True)
        cfg = FusedConfig(add_mode="pre", relu=bool(relu), **kw)
        extra = ()
        if identity_bn is not None:
            _, e2, m2, c2 = SelfNorm._bn_call_state(identity_bn, in_kernel=True)
            extra = (identity_bn.weight, identity_bn.bias, identity_bn.running_mean, identity_bn.running_var, e2, m2, c2)
        return _F.FusedBnBlock.apply(conv_out, identity, cfg, g.fc_weight, g.bn_weight, g.bn_bias, g.running_mean, g.running_var,
                                     bn.weight, bn.bias, bn.running_mean, bn.running_var, bn_eps, bn_mom, g.num_batches_tracked,
                                     bn_counter, *extra)

    def forward_block_bn(self, x, addend, add_mode, bn, want_y=True):
        """`y = self(x [+ addend]); z = relu(bn(y))` — the end of one WideResNet block together with the NEXT block's
        `relu1(bn1(.))` (cifar/wideresnet_cnsn.py:93-96 then :76-77 / :69-70 / :222).  Returns `(y, z)`; `y` is None
        when `want_y` is False (nothing but the BatchNorm consumes it).  One launch per direction when a fused kernel
        covers the call (`functional.bnrelu_plan`: SelfNorm alone — the site's CrossNorm idle —, training or eval,
        planes of at most 64 vectors with the whole channel in one workgroup's registers); otherwise the same three
        steps as separate calls: `forward_block`, `bn`, ReLU.  BatchNorm2d's per-call book-keeping
        (`num_batches_tracked`, `momentum=None`) is done here exactly as `nn.BatchNorm2d.forward` does it."""
        assert add_mode in ("none", "pre")
        cn, sn = self.crossnorm, self.selfnorm
        armed = cn is not None and cn.active
        fused = (type(bn) is nn.BatchNorm2d and bn.affine and bn.track_running_stats and sn is not None
                 and type(sn) is SelfNorm and sn.f_fc is None and sn._fusable() and not armed and x.is_cuda
                 and (cn is None or type(cn) is CrossNorm))
        if fused:
            kw, g, _ = sn._fused_args_peek()
            cfg = FusedConfig(add_mode=add_mode, relu=False, **kw)
            fused = _F.bnrelu_plan_cached(x, cfg, torch.is_grad_enabled())
        if not fused:
            y = self.forward_block(x, addend, add_mode=add_mode, relu=False) if add_mode != "none" else self.forward(x)
            return (y if want_y else None), torch.relu(bn(y))
        kw, g, _ = sn._fused_args()
        bn_batch, bn_eps, bn_mom, bn_counter = SelfNorm._bn_call_state(bn, in_kernel=True)
        cfg = FusedConfig(add_mode=add_mode, relu=False, **kw)
        return _F.fused_cnsn_tail(x, cfg, addend, bool(want_y), g, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                  bn_batch, bn_eps, bn_mom, bn_counter)

