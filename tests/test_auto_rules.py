"""What CNSN_STRATEGY_AUTO resolves to for the shapes of the scope contract — `cnsn_which_path` is a
pure function of the problem, so these run without a GPU.  The table is the measured outcome recorded in
profiles/r01_resident_tuning.md and r01_small_planes.md; a change of a rule has to change this file with it."""
import pytest
import torch

import cnsn_amd
from cnsn_amd import FusedConfig as FC
from cnsn_amd import which_path

SN = dict(sn_active=True)
CN = dict(cn_active=True)
BOTH = dict(cn_active=True, content_box=(1, 1, 5, 5), style_box=(0, 0, 4, 4))
BLOCK = dict(sn_active=True, add_mode="pre", relu=True)
F32, BF16 = torch.float32, torch.bfloat16

TABLE = [
    # shape, dtype, config, forward, backward
    ((256, 256, 56, 56), F32, FC(**SN, **CN), "resident", "resident"),         # the north-star workload
    ((256, 256, 56, 56), F32, FC(**SN, **BOTH), "resident", "resident"),
    ((256, 256, 56, 56), BF16, FC(**SN, **CN), "resident", "resident"),
    ((256, 256, 56, 56), BF16, FC(**SN, **BOTH), "resident", "resident"),      # 16-bit boxed 56x56: pipelined forward since the region select is branch-free; backward: partial-moment cluster kernel (round 4)
    ((32, 256, 56, 56), BF16, FC(**SN, **BOTH), "resident", "resident"),       # ... at every batch size
    ((256, 128, 64, 64), BF16, FC(**SN, **BOTH), "resident", "resident"),      # 16-bit boxed 64x64 WITH SelfNorm: single-touch forward too since round 5 (config 5's mode)
    ((16, 2048, 64, 64), BF16, FC(**SN, **BOTH), "resident", "resident"),
    ((256, 128, 64, 64), BF16, FC(**CN, style_box=(0, 0, 32, 32)), "resident", "resident"),  # ... CrossNorm alone does not
    ((96, 256, 56, 56), BF16, FC(**SN, **BOTH), "resident", "resident"),       # ... (until the branch-free region select: two-pass both ways at N < 192)
    ((256, 512, 28, 28), BF16, FC(**SN, **BOTH), "resident", "resident"),      # <= 4 slots: always resident
    ((256, 1024, 14, 14), BF16, FC(**SN), "mono", "mono"),                     # a channel = 100 KiB: one workgroup's registers
    ((256, 1024, 14, 14), BF16, FC(**BLOCK), "mono", "mono"),
    ((256, 1024, 14, 14), F32, FC(**SN), "resident", "resident"),              # fp32 N = 256: the one-slot cluster kernels, both directions (round 4 audit)
    ((96, 1024, 14, 14), BF16, FC(**SN), "mono", "mono"),
    ((256, 1024, 14, 14), BF16, FC(**SN, **CN), "mono", "mono"),               # CrossNorm inside the channel's workgroup
    ((64, 1024, 14, 14), BF16, FC(**SN, **CN), "resident", "resident"),        # fewer than 96 planes per channel, un-boxed: the cluster kernels
    ((96, 1024, 14, 14), BF16, FC(**SN, **CN), "resident", "resident"),        # below 128 planes per channel (round 4 audit: -19 %)
    ((128, 1024, 14, 14), BF16, FC(**SN, **CN), "mono", "mono"),
    ((16, 512, 64, 64), F32, FC(**SN, **BOTH), "resident", "resident"),        # fp32 64x64 with crop boxes at N <= 32: cluster kernels again (branch-free region select), partial-moment backward (-26 %)
    ((16, 512, 64, 64), F32, FC(**CN, style_box=(0, 0, 32, 32)), "resident", "resident"),    # ... CrossNorm alone too
    ((64, 512, 64, 64), F32, FC(**SN, **BOTH), "resident", "resident"),
    ((64, 1024, 14, 14), BF16, FC(**SN, **BOTH), "mono", "mono"),              # with crop boxes the channel-in-registers kernels keep it
    ((256, 2048, 7, 7), F32, FC(**SN, **CN), "mono", "mono"),                  # CrossNorm without boxes: the channel-group kernels (round 3)
    ((256, 1024, 14, 14), BF16, FC(**SN, **BOTH), "mono", "mono"),             # (was packed two-pass: vectors straddle rows)
    ((128, 128, 8, 8), F32, FC(**SN, **BOTH), "mono", "mono"),                 # WideResNet stage 3, armed site
    ((256, 2048, 7, 7), BF16, FC(**SN, **BOTH), "mono", "mono"),               # 7x7 bf16 with boxes: one element per lane
    ((256, 2048, 7, 7), BF16, FC(**SN), "mono", "mono"),                       # 98-byte planes: 8 / 4 adjacent channels per workgroup ("wide")
    ((256, 2048, 7, 7), BF16, FC(**BLOCK), "mono", "mono"),
    ((96, 2048, 7, 7), BF16, FC(**SN), "local", "local"),                      # small batch: the channel-local kernels
    ((256, 2048, 7, 7), BF16, FC(**SN, **CN), "mono", "mono"),                 # CrossNorm without boxes: channel groups in registers ("wide", round 3)
    ((256, 2048, 7, 7), BF16, FC(**CN), "mono", "mono"),                       # CrossNorm ALONE: the same kernels without a gate (round 4; was packed two-pass)
    ((96, 2048, 7, 7), BF16, FC(**CN), "packed", "packed"),                    # ... below N = 128 the step is host-bound either way
    ((256, 2048, 7, 7), F32, FC(**SN), "mono", "mono"),                        # fp32 7x7: one element (4 B) per lane
    ((16, 512, 64, 64), F32, FC(sn_active=True, add_mode="post", relu=True), "resident", "resident"),   # segmentation: SN at 'residual'
    ((16, 2048, 64, 64), BF16, FC(sn_active=True, add_mode="post", relu=True), "resident", "resident"),
    ((16, 256, 128, 128), F32, FC(sn_active=True, add_mode="post", relu=True), "resident", "resident"),  # a plane per workgroup (split)
    ((16, 256, 128, 128), BF16, FC(**CN, style_box=(0, 0, 64, 64)), "resident", "resident"),            # segmentation's separate CrossNorm, 16-bit boxed: cluster kernels since the region select is branch-free
    ((16, 256, 128, 128), BF16, FC(**SN, **BOTH), "streaming", "resident"),                               # ... with SelfNorm: the backward only
    ((16, 256, 128, 128), F32, FC(**CN, style_box=(0, 0, 64, 64)), "resident", "resident"),
    ((128, 128, 8, 8), F32, FC(**SN), "mono", "mono"),                        # 256-byte planes, 16 lanes each
    ((128, 64, 16, 16), F32, FC(**SN), "mono", "mono"),                        # WideResNet stage 2
    ((128, 64, 16, 16), F32, FC(**SN, **CN), "resident", "resident"),          # ... armed site, fp32: 64 channels leave 3 CUs in 4 idle with one workgroup per channel (round 5, three audits)
    ((128, 64, 16, 16), F32, FC(**SN, **BOTH), "resident", "resident"),        # ... with crop boxes too
    ((128, 64, 16, 16), BF16, FC(**SN, **BOTH), "mono", "mono"),               # ... in 16 bits the two are level: unchanged
    ((256, 64, 16, 16), F32, FC(**SN, **CN), "mono", "resident"),              # ... above N = 128 the forward stays with the channel-in-registers kernels (the backward never fitted them)
    ((128, 32, 32, 32), F32, FC(**SN, **BOTH), "resident", "resident"),        # WideResNet stage 1
    ((768, 3, 224, 224), F32, FC(**CN), "streaming", "streaming"),             # image-level CrossNorm: 12544 vectors
    ((16, 256, 128, 128), F32, FC(**SN), "resident", "resident"),              # 4096 vectors: split over the workgroup's waves
    ((16, 256, 72, 64), F32, FC(**SN), "streaming", "streaming"),              # 5 of 8 slots per wave used: not worth it
    ((256, 256, 56, 56), F32, FC(sn_active=True, sn_training=False), "resident", "resident"),   # inference (SOLO)
    ((64, 128, 88, 88), BF16, FC(**SN, **CN), "streaming", "resident"),        # 16 slots, 16-bit: un-boxed backward only
]


@pytest.mark.parametrize("shape,dtype,cfg,fwd,bwd", TABLE, ids=lambda v: str(v)[:28].replace(" ", ""))
def test_auto_resolves_as_measured(shape, dtype, cfg, fwd, bwd):
    x = torch.empty(shape, dtype=dtype, device="meta")
    cnsn_amd.set_strategy("auto")
    assert which_path(x, cfg, backward=False) == fwd
    assert which_path(x, cfg, backward=True) == bwd


def test_forced_strategies_fall_back():
    x = torch.empty((8, 4, 224, 224), dtype=F32, device="meta")
    try:
        for name in ("two_pass", "resident", "local", "mono"):
            cnsn_amd.set_strategy(name)
            assert which_path(x, FC(**SN)) == "streaming"          # too large for any single-touch strategy
        cnsn_amd.set_strategy("two_pass")
        assert which_path(torch.empty((8, 4, 7, 7), dtype=BF16, device="meta"), FC(**SN)) == "packed"
        cnsn_amd.set_strategy("local")
        assert which_path(torch.empty((8, 4, 7, 7), dtype=BF16, device="meta"), FC(**SN)) == "local"
        assert which_path(torch.empty((8, 4, 7, 7), dtype=BF16, device="meta"), FC(**SN, **CN)) == "packed"
    finally:
        cnsn_amd.set_strategy("auto")


def test_environment_knobs_are_read_at_load_not_per_call():
    """csrc/cnsn_env.h: CNSN_* is snapshotted when the library is loaded; a later change takes effect only through
    cnsn_reload_env() — run in a child process so that the session's `follow_environ` hook is not in the way."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import os, sys
sys.path.insert(0, {root!r})
import torch, cnsn_amd
from cnsn_amd import FusedConfig as FC, which_path
x = torch.empty((256, 1024, 14, 14), dtype=torch.bfloat16, device="meta")
cfg = FC(sn_active=True)
assert which_path(x, cfg) == "mono"
os.environ["CNSN_MONO"] = "0"                       # after the load: not seen
assert which_path(x, cfg) == "mono"
cnsn_amd.reload_env()                               # ... until the table is re-read
assert which_path(x, cfg) != "mono"
cnsn_amd.follow_environ()                           # tests / tools: every CNSN_* change re-reads
del os.environ["CNSN_MONO"]
assert which_path(x, cfg) == "mono"
os.environ["CNSN_MONO"] = "0"
assert which_path(x, cfg) != "mono"
print("KNOBS-OK")
"""
    env = {k: v for k, v in os.environ.items() if not k.startswith("CNSN_")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "KNOBS-OK" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
    probe = code.split("assert which_path")[0] + 'print(which_path(x, cfg))\n'
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, env=dict(env, CNSN_MONO="0"), timeout=300)
    assert r.returncode == 0 and r.stdout.strip() != "mono", (r.stdout, r.stderr[-2000:])
