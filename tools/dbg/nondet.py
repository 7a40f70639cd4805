"""Run-to-run bit equality of every output of the SelfNorm cluster kernels, with the lane / dword pattern of any difference
(how the gfx950 store hazard of DESIGN.md 4.2h was found)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["CNSN_SNX"] = "2"
import cnsn_amd
cnsn_amd.follow_environ()   # CNSN_* knobs are read at load: re-read after every change below
from tests.golden.gen_golden_fill import fill_sn
cnsn_amd.set_strategy("resident")
def run(shape, dtype, seed, mode, relu):
    n, c = shape[:2]
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (torch.randn(shape, device="cuda", generator=g) * (torch.rand(n, c, 1, 1, device="cuda", generator=g) * 1.5 + 0.5)
         + torch.randn(n, c, 1, 1, device="cuda", generator=g)).to(dtype).requires_grad_()
    b = (torch.randn(shape, device="cuda", generator=g) * 0.7).to(dtype).requires_grad_() if mode != "none" else None
    gy = torch.randn(shape, device="cuda", generator=g).to(dtype)
    mod = cnsn_amd.CNSN(None, fill_sn(cnsn_amd.SelfNorm(c), seed, torch.float32)).cuda().train()
    y = mod.forward_block(x, b, add_mode=mode, relu=relu) if (mode != "none" or relu) else mod(x)
    grads = torch.autograd.grad(y, [x] + ([b] if b is not None else []) + list(mod.parameters()), gy)
    torch.cuda.synchronize()
    return [y.detach()] + [t.detach() for t in grads] + [t.clone() for t in mod.buffers()]
for shape, dt in (((37, 384, 28, 28), torch.bfloat16),):
    for mode, relu in (("pre", True), ("none", False)):
        ref = run(shape, dt, 5, mode, relu)
        for rep in range(3):
            out = run(shape, dt, 5, mode, relu)
            for i, (a, b) in enumerate(zip(out, ref)):
                if not torch.equal(a, b):
                    if a.dim() == 4:
                        bad = (a != b).flatten(2).any(2)
                        idx = bad.nonzero()
                        print(shape, dt, mode, relu, "rep", rep, "output", i, "planes differing:", idx.shape[0], idx[:12].tolist(),
                              "maxdiff", float((a.float() - b.float()).abs().max()))
                        n0, c0 = idx[0].tolist()
                        d = (a[n0, c0] != b[n0, c0]).flatten().nonzero().flatten()
                        print("   first plane: elements differing", d.numel(), d[:8].tolist(), d[-3:].tolist())
                    else:
                        print(shape, dt, mode, relu, "rep", rep, "output", i, "differs", float((a.float() - b.float()).abs().max()))
print("timeouts", cnsn_amd.lib().cnsn_resident_timeouts())
# lane / dword statistics of the differing elements of y
shape, dt = (37, 384, 28, 28), torch.bfloat16
ref = run(shape, dt, 5, "pre", True)[0]
out = run(shape, dt, 5, "pre", True)[0]
diff = (ref != out)
print("planes by n:", diff.flatten(2).any(2).sum(1).tolist())
el = diff.flatten(2).any(0).any(0).nonzero().flatten()
print("elements (vector index, element in vector):", sorted(set((int(e) // 8, int(e) % 8) for e in el))[:80])
nz = diff.nonzero()
import collections
print("by channel round:", collections.Counter((int(c) // 256) for c in nz[:, 1].tolist()))
print("values: ref", ref[diff][:8].tolist(), "out", out[diff][:8].tolist())
