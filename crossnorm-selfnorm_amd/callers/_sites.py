"""What the two backbones share: building a CNSN unit from the reference's option strings and arming
CrossNorm sites at random before a forward."""
import numpy as np


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
