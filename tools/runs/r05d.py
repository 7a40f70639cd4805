import os, sys, gc, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnsn_amd
from cnsn_amd import arena, _ffi
from cnsn_amd import functional as F_
from tests.golden.gen_golden_fill import fill_sn
DEV = torch.device("cuda:0")
TRIM = os.environ.get("NO_TRIM") != "1"
def case(dtype, crop):
    shape = (40, 16, 56, 56)
    torch.manual_seed(5); np.random.seed(5)
    x = (torch.randn(shape, device=DEV) * 1.3 + 0.2).to(dtype)
    gy = torch.randn(shape, device=DEV).to(dtype)
    d = cnsn_amd.draw_cn(shape, crop, 1)
    def run(ctypes_path):
        sn = fill_sn(cnsn_amd.SelfNorm(shape[1]), 7, torch.float32).to(DEV).train()
        xg = x.clone().requires_grad_()
        if ctypes_path:
            kw, g, f = sn._fused_args()
            cfg = cnsn_amd.FusedConfig(cn_active=True, content_box=d.content_box, style_box=d.style_box, **kw)
            y = F_.FusedCNSN.apply(xg, cfg, d.perm, None, g.fc_weight, g.bn_weight, g.bn_bias, g.running_mean, g.running_var,
                                   *(None,) * 5, None, g.num_batches_tracked, None)
        else:
            mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1), sn).to(DEV).train()
            mod.crossnorm.active = True; mod.crossnorm.next_draws = d
            y = mod(xg)
        y.backward(gy); torch.cuda.synchronize()
        return [y.detach().clone(), xg.grad.clone()], y.data_ptr(), xg.grad.data_ptr()
    arena.disable()
    base, _, _ = run(False)
    for chunk in (0, 2):
        arena.set_chunk_mb(chunk); arena.enable(min_mb=1)
        for cp in (False, True):
            got, py, pg = run(cp)
            for i, (a, b) in enumerate(zip(base, got)):
                bad = (a != b).flatten().nonzero().flatten()
                msg = f"{dtype} {crop} chunk {chunk} ctypes {cp} out {i} y@{py:#x} dx@{pg:#x}: mismatches {bad.numel()}"
                if bad.numel():
                    es = a.element_size()
                    msg += f" first byte {int(bad[0])*es:#x} last byte {int(bad[-1])*es:#x} zeros {int((b.flatten()[bad]==0).sum())} nan {int(torch.isnan(b.flatten()[bad].float()).sum())}"
                    msg += " got " + str(b.flatten()[bad][:4].tolist()) + " want " + str(a.flatten()[bad][:4].tolist())
                print(msg, flush=True)
    was = arena.min_bytes()
    arena.set_chunk_mb(0); gc.collect()
    if TRIM:
        print("trim", arena.trim(), flush=True)
for crop in ("neither", "both"):
    for dt in (torch.float32, torch.bfloat16):
        case(dt, crop)
