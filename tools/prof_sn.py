#!/usr/bin/env python3
"""Tuning aid: per-phase time stamps of the SelfNorm-only cluster kernels (needs a -DCNSN_PROF build of the two
cnsn_resident_sn translation units, CNSN_LIB_PATH pointing at it).  usage: prof_sn.py dtype N C H W [block]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CNSN_PROF"] = "1"
import cnsn_amd  # noqa: E402
cnsn_amd.follow_environ()   # CNSN_* knobs are read at load: re-read after every change below
from cnsn_amd import _ffi  # noqa: E402
from cnsn_amd.functional import FusedConfig, _epilogue, _problem  # noqa: E402

lib = cnsn_amd.lib()
dev = torch.device("cuda:0")
dtype = {"f32": torch.float32, "bf16": torch.bfloat16}[sys.argv[1]]
N, Cn, H, W = (int(v) for v in sys.argv[2:6])
block = len(sys.argv) > 6 and sys.argv[6] == "block"
x = torch.randn(N, Cn, H, W, device=dev).to(dtype)
b = torch.randn(N, Cn, H, W, device=dev).to(dtype)
gy = torch.randn(N, Cn, H, W, device=dev).to(dtype)
y = torch.empty_like(x)
dx = torch.empty_like(x)
cfg = FusedConfig(sn_active=True, sn_training=True, add_mode="pre" if block else "none", relu=block)
prob = _problem(x, cfg)
epi = _epilogue(cfg, b if block else None)
w = torch.rand(Cn, 2, device=dev) - 0.5
gam, bet, rm, rv = torch.ones(Cn, device=dev), torch.zeros(Cn, device=dev), torch.zeros(Cn, device=dev), torch.ones(Cn, device=dev)
g = _ffi.Gate(w.data_ptr(), gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr())
dw, dg_, db = torch.empty(Cn, 2, device=dev), torch.empty(Cn, device=dev), torch.empty(Cn, device=dev)
gg = _ffi.GateGrad(dw.data_ptr(), dg_.data_ptr(), db.data_ptr())
saved = torch.empty(lib.cnsn_saved_floats(C.byref(prob)), dtype=torch.float32, device=dev)
wsb = max(lib.cnsn_workspace_bytes(C.byref(prob)), (4 << 20) + 64 * 16 * 8 * 8 + 1024)
ws = torch.zeros(wsb // 4 + 4, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
assert lib.cnsn_sn_cluster_plan(C.byref(prob), C.byref(epi), 0) == 1 and lib.cnsn_sn_cluster_plan(C.byref(prob), C.byref(epi), 1) == 1


def run(which):
    ws.zero_()
    for _ in range(3):
        if which == "fwd":
            r = lib.cnsn_forward_fused(C.byref(prob), C.byref(epi), x.data_ptr(), None, None, C.byref(g), None, y.data_ptr(),
                                       saved.data_ptr(), ws.data_ptr(), wsb, st)
        else:
            r = lib.cnsn_backward_fused(C.byref(prob), C.byref(epi), gy.data_ptr(), x.data_ptr(), None, None, C.byref(g), None,
                                        saved.data_ptr(), dx.data_ptr(), None, C.byref(gg), None, ws.data_ptr(), wsb, st)
        assert r == 0, r
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    raw = ws.view(torch.int64)[(4 << 20) // 8:(4 << 20) // 8 + 64 * 16 * 8].cpu().numpy().reshape(64, 16, 8)
    t = raw[:, :, :5].astype(np.float64)
    ok = (t[:, :, 4] > 0) & (t[:, :, 0] > 0)
    names = ["gather (+params)", "merge + gates/coefs", "stats/sums of t+1 + publish", "apply t / park t+1 / issue t+2"]
    d = np.diff(t, axis=2) * 0.01          # 100 MHz wall clock -> microseconds
    print(f"== {sys.argv[1]} ({N},{Cn},{H},{W}) {'block' if block else 'sn'} {which}: iterations recorded per WG {ok.sum(1).min()}-{ok.sum(1).max()}, "
          f"gather passes mean {raw[:, :, 6][ok].mean():.2f} p90 {np.percentile(raw[:, :, 6][ok], 90):.0f}")
    for i, nm in enumerate(names):
        v = d[:, :, i][ok]
        print(f"  {nm:34s} mean {v.mean():7.2f} us  p10 {np.percentile(v, 10):7.2f}  p90 {np.percentile(v, 90):7.2f}")
    okc = ok[:, 1:] & ok[:, :-1]
    cyc = (t[:, 1:, 0] - t[:, :-1, 0])[okc] * 0.01
    print(f"  {'full cycle':34s} mean {cyc.mean():7.2f} us")
    gap = (t[:, 1:, 0] - t[:, :-1, 4])[okc] * 0.01
    print(f"  {'end of apply -> next top':34s} mean {gap.mean():7.2f} us")
    if os.environ.get("PROF_PER_WG"):
        print("  per workgroup: iterations, mean us of the four phases")
        for wg in range(64):
            m = ok[wg]
            if m.any():
                print(f"    wg {wg:2d} k={wg % int(os.environ['PROF_PER_WG']):2d}: its {int(m.sum()):2d}  " + "  ".join(f"{d[wg, :, i][m].mean():8.1f}" for i in range(4))
                      + "   max " + "  ".join(f"{d[wg, :, i][m].max():8.1f}" for i in range(4)))
    # timeline of the recorded workgroups (chip-wide clock): when each iteration starts / ends relative to the first stamp
    t0 = t[:, :, 0][ok].min()
    its = int(ok.sum(1).max())
    print("  iteration: start (mean over WGs, us since the first stamp) / cycle (us)")
    for i in range(its):
        m = ok[:, i]
        if not m.any():
            continue
        st_ = (t[:, i, 0][m] - t0) * 0.01
        en_ = (t[:, i, 4][m] - t0) * 0.01
        print(f"    it {i:2d}: start {st_.mean():7.1f} (min {st_.min():6.1f} max {st_.max():6.1f})  end {en_.mean():7.1f}  WGs {int(m.sum())}")


run("fwd")
if not os.environ.get("PROF_FWD_ONLY"):
    run("bwd")
