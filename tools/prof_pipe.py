#!/usr/bin/env python3
"""Tuning aid: per-phase time stamps of the pipelined cluster forward (needs a -DCNSN_PROF build of cnsn_resident_pipe.hip,
CNSN_LIB_PATH pointing at it).  usage: prof_pipe.py dtype N C H W [crop]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CNSN_PROF"] = "1"
import cnsn_amd  # noqa: E402
from cnsn_amd import _ffi  # noqa: E402
from cnsn_amd.functional import FusedConfig, _problem, _context  # noqa: E402

lib = cnsn_amd.lib()
dev = torch.device("cuda:0")
dtype = {"f32": torch.float32, "bf16": torch.bfloat16}[sys.argv[1]]
N, Cn, H, W = (int(v) for v in sys.argv[2:6])
crop = sys.argv[6] if len(sys.argv) > 6 else "neither"
x = torch.randn(N, Cn, H, W, device=dev).to(dtype)
y = torch.empty_like(x)
d = cnsn_amd.draw_cn((N, Cn, H, W), crop, 1)
cfg = FusedConfig(cn_active=True, content_box=d.content_box, style_box=d.style_box, sn_active=True, sn_training=True)
prob = _problem(x, cfg)
_context(prob, dev)
perm = d.perm.to(dev)
w = torch.rand(Cn, 2, device=dev) - 0.5
gam, bet, rm, rv = torch.ones(Cn, device=dev), torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev), torch.ones(Cn, device=dev)
g = _ffi.Gate(w.data_ptr(), gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), None)
saved = torch.empty(lib.cnsn_saved_floats(C.byref(prob)), dtype=torch.float32, device=dev)
wsb = max(lib.cnsn_workspace_bytes(C.byref(prob)), (4 << 20) + 64 * 16 * 8 * 8 + 1024)
ws = torch.zeros(wsb // 4 + 4, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
print("path", cnsn_amd.which_path(x, cfg))
for _ in range(4):
    ws.zero_()
    r = lib.cnsn_forward_fused(C.byref(prob), None, x.data_ptr(), perm.data_ptr(), None, C.byref(g), None, y.data_ptr(),
                               saved.data_ptr(), ws.data_ptr(), wsb, st)
    assert r == 0, r
    torch.cuda.synchronize()
raw = ws.view(torch.int64)[(4 << 20) // 8:(4 << 20) // 8 + 64 * 16 * 8].cpu().numpy().reshape(64, 16, 8)
t = raw[:, :, :6].astype(np.float64)          # stamps 0..5: (0 = kernel start, first iteration only) 1 top, 2 gathered, 3 algebra done, 4 t+1 published, 5 applied
ok = (t[:, :, 5] > 0) & (t[:, :, 1] > 0)
names = ["gather (+params)", "algebra + coefs + saved", "stats of t+1 + publish", "apply t / park t+1 / issue t+2"]
dd = np.diff(t[:, :, 1:], axis=2) * 0.01
print(f"== {sys.argv[1]} ({N},{Cn},{H},{W}) crop={crop} pipelined forward: iterations recorded per WG {ok.sum(1).min()}-{ok.sum(1).max()}, "
      f"gather passes mean {raw[:, :, 6][ok].mean():.2f} p90 {np.percentile(raw[:, :, 6][ok], 90):.0f}")
for i, nm in enumerate(names):
    v = dd[:, :, i][ok]
    print(f"  {nm:34s} mean {v.mean():7.2f} us  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}")
okc = ok[:, 1:] & ok[:, :-1]
cyc = (t[:, 1:, 1] - t[:, :-1, 1])[okc] * 0.01
print(f"  {'full cycle':34s} mean {cyc.mean():7.2f} us")
