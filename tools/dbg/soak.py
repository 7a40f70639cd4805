"""Soak: thousands of forward+backward launches of the cluster kernels, every output compared bit for bit with the first
launch's (rare hazards and races show up as run-to-run differences), the time-out counter checked at the end."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnsn_amd
cnsn_amd.follow_environ()   # CNSN_* knobs are read at load: re-read after every change below
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for shape, dt, call in (((37, 384, 28, 28), torch.bfloat16, "block"), ((64, 256, 56, 56), torch.bfloat16, "block"),
                        ((64, 256, 56, 56), torch.float32, "sn"), ((130, 512, 14, 14), torch.bfloat16, "sn_forced"),
                        ((64, 128, 56, 56), torch.float32, "cnsn"), ((128, 512, 7, 7), torch.bfloat16, "cnsn_block")):
    if call == "sn_forced":
        os.environ["CNSN_SNX"] = "2"; cnsn_amd.set_strategy("resident")
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(shape, device=dev, generator=g).to(dt).requires_grad_()
    b = torch.randn(shape, device=dev, generator=g).to(dt).requires_grad_()
    gy = torch.randn(shape, device=dev, generator=g).to(dt)
    cn = cnsn_amd.CrossNorm("neither", 1) if call.startswith("cnsn") else None
    mod = cnsn_amd.CNSN(cn, cnsn_amd.SelfNorm(shape[1])).to(dev).train()
    perm = torch.randperm(shape[0])
    def run():
        if cn is not None:
            cn.active = True
            cn.next_draws = cnsn_amd.CNDraws(perm, None, None, None)
        blk = call in ("block", "cnsn_block")
        y = mod.forward_block(x, b, add_mode="pre", relu=True) if blk else mod(x)
        grads = torch.autograd.grad(y, [x] + ([b] if blk else []) + list(mod.parameters()), gy)
        return [y] + list(grads)
    ref = [t.clone() for t in run()]
    bad = 0
    t0 = time.time()
    for i in range(iters):
        out = run()
        if i % 50 == 0:
            for a, r in zip(out, ref):
                if not torch.equal(a, r):
                    bad += 1
                    break
    torch.cuda.synchronize()
    print(f"{shape} {dt} {call}: {iters} launches in {time.time() - t0:.1f} s, {bad} of {iters // 50 + 1} checked launches differ", flush=True)
    os.environ.pop("CNSN_SNX", None); cnsn_amd.set_strategy("auto")
print("time-outs:", cnsn_amd.lib().cnsn_resident_timeouts())
