import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnsn_amd
from cnsn_amd import arena, _ffi
from tests.golden.gen_golden_fill import fill_sn
DEV = torch.device("cuda:0")
shape = (40, 16, 56, 56)
torch.manual_seed(5); np.random.seed(5)
x = (torch.randn(shape, device=DEV) * 1.3 + 0.2)
gy = torch.randn(shape, device=DEV)
d = cnsn_amd.draw_cn(shape, "both", 1)
print("draws", d.content_box, d.style_box)
cfg = cnsn_amd.FusedConfig(cn_active=True, content_box=d.content_box, style_box=d.style_box, sn_active=True)
print("paths", cnsn_amd.which_path(x, cfg), cnsn_amd.which_path(x, cfg, True))
def run():
    sn = fill_sn(cnsn_amd.SelfNorm(16), 7, torch.float32).to(DEV).train()
    mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm("both", 1), sn).to(DEV).train()
    mod.crossnorm.active = True; mod.crossnorm.next_draws = d
    xg = x.clone().requires_grad_()
    y = mod(xg); y.backward(gy); torch.cuda.synchronize()
    return y.detach().clone(), xg.grad.clone(), y.data_ptr()
arena.disable()
b0, g0, _ = run()
for chunk in (0, 2, 2, 8, 0, 2):
    arena.set_chunk_mb(chunk); arena.enable(min_mb=1)
    for rep in range(3):
        y, g, p = run()
        bad = (y != b0); badg = (g != g0)
        msg = f"chunk {chunk} rep {rep} ptr {p:#x} owned {cnsn_amd.lib().cnsn_arena_owns(C.c_void_p(p))}: y mismatches {int(bad.sum())} dx mismatches {int(badg.sum())}"
        if bad.any():
            idx = bad.flatten().nonzero().flatten()
            msg += f" first {int(idx[0])} last {int(idx[-1])} zeros_there {int((y.flatten()[idx]==0).sum())}"
            pl = torch.unique(idx // 3136)
            msg += f" planes {pl[:12].tolist()} n={len(pl)}"
        print(msg, flush=True)
