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


def fill_by_name(model, seed=0):
    """Deterministic, name-seeded fill of EVERY parameter and buffer, so two models with the same
    state_dict keys (the reference's and this repository's) end up with identical values without
    committing a multi-MB checkpoint.  Conv/linear weights ~ N(0, 1/sqrt(fan_in)); norm weights ~ U(0.5,1.5);
    biases ~ U(-0.1,0.1); running_mean ~ N(0,0.1); running_var ~ U(0.5,1.5); SelfNorm fc ~ U(-0.7,0.7)."""
    import zlib
    with torch.no_grad():
        for name, t in model.state_dict().items():
            if not t.dtype.is_floating_point:
                t.zero_()
                continue
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()) + seed)
            shape = tuple(t.shape)
            if name.endswith("running_var"):
                v = torch.rand(shape, generator=g, dtype=torch.float64) + 0.5
            elif name.endswith("running_mean"):
                v = torch.randn(shape, generator=g, dtype=torch.float64) * 0.1
            elif "_fc.weight" in name:                      # SelfNorm g_fc / f_fc (C,1,2)
                v = torch.rand(shape, generator=g, dtype=torch.float64) * 1.4 - 0.7
            elif name.endswith("weight") and t.dim() >= 2:  # conv / linear
                fan_in = t[0].numel()
                v = torch.randn(shape, generator=g, dtype=torch.float64) / fan_in ** 0.5
            elif name.endswith("weight"):                   # norm scale
                v = torch.rand(shape, generator=g, dtype=torch.float64) + 0.5
            else:                                           # biases
                v = torch.rand(shape, generator=g, dtype=torch.float64) * 0.2 - 0.1
            t.copy_(v.to(t.dtype))
    return model
