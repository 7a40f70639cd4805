"""Deterministic SelfNorm parameter fill shared by gen_golden.py and the tests (no reference import)."""
import torch


def fill_sn(mod, seed, dtype):
    """g_fc.weight ~ U(-0.7,0.7), bn.weight ~ U(0.5,1.5), bn.bias ~ U(-0.5,0.5) from `seed`."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for fc, bn in ((mod.g_fc, mod.g_bn), (mod.f_fc, getattr(mod, "f_bn", None))):
            if fc is None:
                continue
            c = fc.weight.shape[0]
            fc.weight.copy_(torch.rand(c, 1, 2, generator=g, dtype=torch.float64) * 1.4 - 0.7)
            bn.weight.copy_(torch.rand(c, generator=g, dtype=torch.float64) + 0.5)
            bn.bias.copy_(torch.rand(c, generator=g, dtype=torch.float64) - 0.5)
    return mod.to(dtype)
