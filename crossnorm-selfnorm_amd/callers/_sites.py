"""What the two backbones share: building a CNSN unit from the reference's option strings and arming
CrossNorm sites at random before a forward."""
import os

import numpy as np
import torch

# Fold the block's `sum of the two branches -> CNSN -> ReLU` into the op's own launches (CNSN.forward_block,
# cnsn_forward_fused).  CNSN_FUSE_BLOCK=0 (or setting this to False) keeps the three separate ops.
FUSE_BLOCK = os.environ.get("CNSN_FUSE_BLOCK", "1") != "0"
# WideResNet: also hand the NEXT block's relu(bn1(.)) out of the op's launch (CNSN.forward_block_bn, cnsn_forward_bnrelu;
# SURVEY §8 f1, second half).  CNSN_FUSE_TAIL=0 keeps bn1 / relu1 as separate ops.
FUSE_TAIL = FUSE_BLOCK and os.environ.get("CNSN_FUSE_TAIL", "1") != "0"


def residual_sum(cnsn, pos, residual, skip, relu, skip_first=False):
    """`residual + skip` (or `skip + residual` when `skip_first`, the order WideResNet writes) with CNSN at
    `pos` in {'post', 'residual', 'identity', other = none here} and the block's final ReLU if `relu`:
        resnet_cnsn.py:112-122      out(+=)identity, pos in {'residual','identity','post'}, relu
        wideresnet_cnsn.py:86-96    torch.add(x, out), same positions, no relu
    One fused call when the CNSN unit offers it (this library's, on device tensors); the literal ops otherwise."""
    fb = getattr(cnsn, "forward_block", None) if (FUSE_BLOCK and cnsn is not None) else None
    fusable = (fb is not None and pos in ("post", "residual", "identity") and residual.is_cuda
               and residual.shape == skip.shape and residual.dtype == skip.dtype)
    if fusable:
        if pos == "post":
            a, b = (skip, residual) if skip_first else (residual, skip)
            return fb(a, b, add_mode="pre", relu=relu)
        if pos == "residual":
            return fb(residual, skip, add_mode="post", relu=relu)
        return fb(skip, residual, add_mode="post", relu=relu)
    if pos == "residual":
        residual = cnsn(residual)
    elif pos == "identity":
        skip = cnsn(skip)
    y = torch.add(skip, residual) if skip_first else residual + skip
    if pos == "post":
        y = cnsn(y)
    return torch.relu(y) if relu else y


def make_cnsn(impl, cnsn_type, crop, beta, width):
    """CNSN(crossnorm?, selfnorm?) from `cnsn_type` in {'sn','cn','cnsn'}
    (reference models/cifar/wideresnet_cnsn.py:42-60, models/imagenet/resnet_cnsn.py:62-82)."""
    assert cnsn_type in ("sn", "cn", "cnsn")
    cross = impl.CrossNorm(crop=crop, beta=beta) if "cn" in cnsn_type else None
    selfn = impl.SelfNorm(width) if "sn" in cnsn_type else None
    return impl.CNSN(crossnorm=cross, selfnorm=selfn)


class CrossNormSites:
    """Mixin: collect the CrossNorm modules in registration order and raise `active` on a random
    subset (reference `_enable_cross_norm`, wideresnet_cnsn.py:199-203, resnet_cnsn.py:242-247)."""

    def _collect_sites(self, impl, cnsn_type, active_num):
        # a plain list, not registered as sub-modules a second time (as in the reference, :178-189)
        object.__setattr__(self, "cn_modules", [m for m in self.modules() if isinstance(m, impl.CrossNorm)])
        if cnsn_type is not None and "cn" in cnsn_type:
            self.cn_num = len(self.cn_modules)
            self.active_num = active_num
            assert self.cn_num > 0 and self.active_num > 0

    def _enable_cross_norm(self):
        picked = np.random.choice(self.cn_num, self.active_num, replace=False).tolist()  # numpy global RNG
        for i in picked:
            self.cn_modules[i].active = True
        return picked
