#!/usr/bin/env python3
"""Benchmark of the CrossNorm/SelfNorm hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one fused CNSN forward + backward (CrossNorm armed, SelfNorm in training mode) over one
synthetic activation batch that is already resident in HBM: BASELINE.json's north-star workload
N=256, C=256, H=W=56 (the layer-1 site of ResNet-50 at batch 256).  Every rank owns its own batch
(the op has no cross-GPU exchange: the permutation and the BatchNorm1d statistics are local to the
minibatch); with N>1 the SelfNorm parameter gradients (4C floats) are all-reduced
over RCCL each step like DDP would.  `value` = all ranks' algorithmic bytes (8*E*b per step:
3 tensor passes forward, 5 backward — SURVEY.md §8d3) / wall time, in GB/s.

`--gpus N` with N > 1 and no torchrun environment re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` (one process per GPU, RCCL); under the driver's own torchrun
launch it just joins the group.  With more ranks than devices (a 1-GPU box) the ranks share a device and the
collectives go over gloo — RCCL refuses two ranks on one device — which still exercises the launcher end to end.

One JSON line is printed by rank 0; it also carries
  roofline     — the backward launch (dominant kernel) timed with HIP events on the launch stream, priced on the
                 bytes that launch HAS to move (single touch: 3*E*b) against the 8 TB/s peak; the same interval
                 priced on SURVEY §8(d3)'s two-pass byte count and the in-process copy / triad ceilings ride along
  cpu_baseline — the CPU oracle (a port of the reference's eager PyTorch path) timed on this
                 host's cores on a bounded slice of the same workload (rank 0, N=1 only), plus one thread
  arena        — y and dx come from the library's output arena (cnsn_amd.arena: blocks of its own, a stable home; a NEW block
                 is the fastest of 4 candidates timed when it is created — during the warm-up here).  `ms_per_step` is that default; `ms_per_step_plain_allocator` is the same
                 K steps re-timed in the same process with the arena off (torch's caching allocator places y and dx).
                 MI355X's memory has regions that take plane-strided writes 15-20 % faster (profiles/r04_memory_map.md):
                 `--prospect N` lets the arena time N more candidate blocks and keep the fastest (cnsn_arena_prospect) and
                 adds `ms_per_step_prospected` — an explicit, bounded search that is OFF in the default line
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--shape", type=str, default="256,256,56,56")
    ap.add_argument("--dtype", type=str, default="f32", choices=["f32", "bf16", "f16"])
    ap.add_argument("--crop", type=str, default="neither", choices=["neither", "style", "content", "both"])
    ap.add_argument("--kind", type=str, default="cnsn", choices=["cnsn", "cn", "sn"])
    ap.add_argument("--strategy", type=str, default="auto", choices=["auto", "two_pass", "resident", "local"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads (crop=both, bf16)")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the copy / triad / access-order ceilings (child runs)")
    ap.add_argument("--no-arena", action="store_true", help="y / dx from torch's caching allocator (cnsn_amd.arena off)")
    ap.add_argument("--prospect", type=int, default=0,
                    help="let the output arena time this many candidate blocks of the input's size and keep the 4 fastest "
                         "(cnsn_arena_prospect); adds ms_per_step_prospected to the line (never the headline).  0: no search")
    ap.add_argument("--no-alt", action="store_true",
                    help="only the contract's W + K steps: no second window with the other allocator, no prospecting (rocprofv3 "
                         "passes: the kernel averages then belong to ONE placement)")
    ap.add_argument("--workload", type=str, default="cnsn", choices=["cnsn", "resnet50", "resnet50_jsd", "wrn40", "seg"],
                    help="cnsn: the fused op at the north-star shape (headline); resnet50 / wrn40: whole "
                         "training steps of the caller backbones (images/s)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch of the model workloads (0 = config default)")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="budget of the CPU baseline leg")
    ap.add_argument("--block-graphs", action="store_true",
                    help="wrn40: armed steps replay one captured graph per idle block (measured SLOWER on ROCm 7.2: 8.45 vs "
                         "7.11 ms per step; off by default)")
    ap.add_argument("--no-graph", action="store_true",
                    help="wrn40: run the steps whose CrossNorm sites are idle eagerly instead of replaying them from a HIP graph")
    ap.add_argument("--channels-last", action="store_true",
                    help="model workloads: the network and its input in torch.channels_last (MIOpen's NHWC convolutions; the op is "
                         "computed where the activations lie: cnsn_problem_t.layout = CNSN_LAYOUT_NHWC).  The default of the "
                         "ResNet-50 workloads since round 5 (4 057 -> 5 408 img/s, profiles/r05_nhwc.md)")
    ap.add_argument("--nchw", action="store_true", help="ResNet-50 workloads in torch's default NCHW order (what rounds 1-4 measured)")
    ap.add_argument("--sweep", action="store_true", help="print the SURVEY d1 shape sweep as a markdown table and exit")
    return ap.parse_args()


def conditioned(shape, device, dtype, seed):
    """x = randn*s + m with per-plane s~U(0.5,2), m~N(0,1): well conditioned for SelfNorm's BN."""
    g = torch.Generator(device=device).manual_seed(seed)
    n, c = shape[:2]
    x = torch.randn(shape, generator=g, device=device)
    x.mul_(torch.rand(n, c, 1, 1, generator=g, device=device) * 1.5 + 0.5)
    x.add_(torch.randn(n, c, 1, 1, generator=g, device=device))
    return x.to(dtype)


def _time_oracle(sshape, crop, kind, threads, budget_s, max_iters, min_iters=2):
    import numpy as np
    from oracle import cnsn_oracle as orc
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    np.random.seed(0)
    c = sshape[1]
    x = conditioned(sshape, "cpu", torch.float32, 0).requires_grad_()
    gy = torch.randn(sshape)
    mod = orc.CNSN(orc.CrossNorm(crop, 1) if kind != "sn" else None,
                   orc.SelfNorm(c) if kind != "cn" else None).train()
    times = []
    t_end = time.perf_counter() + budget_s
    it = 0
    while it < min_iters or (time.perf_counter() < t_end and it < max_iters):
        if mod.crossnorm is not None:
            mod.crossnorm.active = True
        x.grad = None
        t0 = time.perf_counter()
        y = mod(x)
        y.backward(gy)
        times.append(time.perf_counter() - t0)
        it += 1
    times = sorted(times[1:]) if len(times) > 1 else times          # (the first iteration pays the allocations)
    return times[len(times) // 2], len(times)


def cpu_baseline(shape, crop, kind, budget_s):
    """Time the CPU oracle (op-for-op restatement of the reference's eager path, `kind: "port"`) on this host's cores:
    the FULL batch of the workload (SURVEY §8 d4 asks for the same inputs; round-3 review: a 1/8 slice hides that
    BatchNorm1d runs over N = 32 there) at the fastest of 32 / 64 / 128 threads (swept in the line, ~50 s in all),
    the 1/8 slice of the earlier rounds beside it, and ONE thread on a 4-instance slice (d4 asks for both)."""
    n, c, h, w = shape
    # Which thread count is the host's best?  Rounds 1-4 used 32 (eager torch on the 2x EPYC 9575F host peaked at 16-32 threads
    # and collapsed beyond 64 when it was measured by hand); round-4 review: settle it in the line.  One discarded + one timed
    # iteration of the FULL batch at 32 / 64 / 128 threads (whatever the host has), then the rest of the budget on the best.
    host = os.cpu_count() or 1
    e = n * c * h * w
    sweep = {}
    for th in sorted({min(t, host) for t in (32, 64, 128)}):
        sweep[th] = _time_oracle(shape, crop, kind, th, 0.0, 2, min_iters=2)[0]
    threads = min(sweep, key=sweep.get)
    med, iters = _time_oracle(shape, crop, kind, threads, budget_s * 0.3, 3, min_iters=3)
    if sweep[threads] < med:                         # (the sweep's own iteration counts: same inputs, same code)
        med = sorted([sweep[threads], med])[0]
    ns = max(2, min(n, 32))                          # 1/8 of the north-star batch (what rounds 1-3 reported)
    meds, iterss = _time_oracle((ns, c, h, w), crop, kind, threads, budget_s * 0.2, 12)
    n1 = max(2, min(n, 4))                           # one thread: a 4-instance slice (~1 s / iteration)
    med1, iters1 = _time_oracle((n1, c, h, w), crop, kind, 1, budget_s * 0.2, 6)
    torch.set_num_threads(threads)
    es, e1 = ns * c * h * w, n1 * c * h * w
    return {"value": round(8 * e * 4 / med / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": f"oracle CNSN fwd+bwd fp32 on ({n},{c},{h},{w}) = the FULL batch of the workload, "
                      f"best of the sweep's and the median of {iters} more iters after one discarded, {med * 1e3:.1f} ms/iter, {n / med:.1f} img/s",
            "slice": {"value": round(8 * es * 4 / meds / 1e9, 3), "unit": "GB/s", "cores": threads,
                      "sample": f"({ns},{c},{h},{w}) = {ns}/{n} of the batch, median of {iterss} iters, {meds * 1e3:.1f} ms/iter"},
            "single_thread": {"value": round(8 * e1 * 4 / med1 / 1e9, 3), "unit": "GB/s", "cores": 1,
                              "sample": f"({n1},{c},{h},{w}), median of {iters1} iters, {med1 * 1e3:.1f} ms/iter"},
            "thread_sweep_ms_per_iter": {str(k): round(v * 1e3, 1) for k, v in sweep.items()},
            "thread_sweep_note": "full batch, one timed iteration after one discarded per setting; `cores` is the fastest",
            "host_threads": os.cpu_count(), "cpu": _cpu_model()}


def copy_triad_ceiling(dev, nbytes=822083584):
    """What plain streaming kernels reach on THIS box, in this process: torch copy (read + write)
    and triad c = a + 2b (two reads + one write) over tensors of the north-star size, HIP events."""
    n = nbytes // 4
    a = torch.empty(n, device=dev).normal_()
    b = torch.empty_like(a).normal_()
    c = torch.empty_like(a)

    def timeit(fn, k=10, w=3):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k

    t_copy = timeit(lambda: c.copy_(a))
    t_triad = timeit(lambda: torch.add(a, b, alpha=2.0, out=c))
    del a, b, c
    out = {"copy_GBps": round(2 * nbytes / t_copy / 1e6, 1), "triad_GBps": round(3 * nbytes / t_triad / 1e6, 1),
           "note": "torch copy_ / add(alpha=2) on 822 MB fp32 tensors, same process and device"}
    out.update(access_pattern_ceiling())
    return out


def access_pattern_ceiling():
    """What the single-touch ACCESS PATTERN allows on this box: tools/pattern_bench (built by __graft_entry__.build)
    copies 56x56 fp32 planes in the item order of the cluster-resident kernels — a channel's 256 planes are 3.2 MB
    apart — with no exchange and no arithmetic; forward shape (one plane in, one out) and backward shape (two in, one
    out).  profiles/r02_access_pattern.md has the full table (linear order reaches 10 % more)."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "pattern_bench")
    if not os.path.exists(exe):
        return {}
    try:
        r = subprocess.run([exe, "256", "256", "brief"], capture_output=True, text=True, timeout=120)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"resident_order_copy_GBps": d["column_copy_GBps"], "resident_order_triad_GBps": d["column_triad_GBps"],
                "resident_order_note": "tools/pattern_bench: plane copy / two-in-one-out in the resident kernels' item order, "
                                       "no exchange, no arithmetic — the ceiling of the forward / backward launch on this box"}
    except Exception as e:  # the tool is an aid: never fail the bench over it
        return {"resident_order_error": str(e)[:200]}


def secondary_workloads(cnsn_amd, shape, dev, args):
    """Same shape, other modes (not the headline): CrossNorm with both crop boxes, and bf16 I/O."""
    n, c, h, w = shape
    res = {}
    for tag, dtype, crop in (("f32_crop_both", torch.float32, "both"), ("f32_sn_only", torch.float32, None),
                             ("bf16_crop_neither", torch.bfloat16, "neither"), ("bf16_crop_both", torch.bfloat16, "both")):
        x = conditioned(shape, dev, dtype, 31).requires_grad_()
        gy = torch.randn(shape, device=dev).to(dtype)
        mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1) if crop else None, cnsn_amd.SelfNorm(c)).to(dev).train()

        def one():
            if mod.crossnorm is not None:
                mod.crossnorm.active = True
            x.grad = None
            for p in mod.parameters():
                p.grad = None
            mod(x).backward(gy)

        for _ in range(10):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 30
        for _ in range(k):
            one()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / k * 1e3
        b = 4 if dtype == torch.float32 else 2
        res[tag] = {"ms_per_step": round(ms, 4), "GBps_algorithmic": round(8 * n * c * h * w * b / ms / 1e6, 1)}
        del x, gy, mod
    return res


def residual_block_workloads(cnsn_amd, shape, dev):
    """The op inside the reference's residual block (resnet_cnsn.py:117-122, pos='post'):
    `out += identity; out = cnsn(out); out = relu(out)` forward+backward — as three separate ops
    (torch add, this library's CNSN, torch relu) and as ONE fused call (CNSN.forward_block)."""
    n, c, h, w = shape
    res = {}
    for tag, dtype in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        a = conditioned(shape, dev, dtype, 41).requires_grad_()
        idt = (conditioned(shape, dev, dtype, 42) * 0.5).detach().requires_grad_()
        gy = torch.randn(shape, device=dev).to(dtype)
        mod = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(c)).to(dev).train()

        ins = [a, idt] + list(mod.parameters())

        def run(fused):
            if fused:
                y = mod.forward_block(a, idt, add_mode="pre", relu=True)
            else:
                y = torch.relu(mod(a + idt))
            # autograd.grad, not .backward(): inside a network neither tensor is a leaf; a leaf's AccumulateGrad
            # would deep-copy the gradient tensor that `out` and `identity` share
            torch.autograd.grad(y, ins, gy)

        for fused in (False, True):
            for _ in range(5):
                run(fused)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k = 20
            for _ in range(k):
                run(fused)
            torch.cuda.synchronize()
            res[f"{tag}_{'fused' if fused else 'three_ops'}_ms"] = round((time.perf_counter() - t0) / k * 1e3, 4)
        del a, idt, gy, mod
    return res


def inference_workloads(cnsn_amd, shape, dev):
    """Serving path: SelfNorm in eval mode (running statistics; CrossNorm is idle in eval), forward only under
    no_grad — nothing couples the planes, so the op is one streaming launch over 2*E*b bytes."""
    n, c, h, w = shape
    res = {}
    for tag, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        x = conditioned(shape, dev, dtype, 51)
        idt = conditioned(shape, dev, dtype, 52) * 0.5
        mod = cnsn_amd.CNSN(None, cnsn_amd.SelfNorm(c)).to(dev).eval()
        b = 4 if dtype == torch.float32 else 2
        for name, fn, passes in (("sn_eval_forward", lambda: mod(x), 2),
                                 ("block_eval_forward", lambda: mod.forward_block(x, idt, add_mode="pre", relu=True), 3)):
            with torch.no_grad():
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    fn()
                e1.record()
                torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 30
            res[f"{tag}_{name}"] = {"ms": round(ms, 4), "GBps_moved": round(passes * n * c * h * w * b / ms / 1e6, 1)}
        del x, idt, mod
    return res


def roofline_bf16(cnsn_amd, dev):
    """BASELINE.json configs[2] runs the op in bf16 inside ResNet-50's residual blocks (resnet_cnsn.py:117-122, pos='post':
    add + SelfNorm + ReLU as ONE call).  Same physical definition as `roofline`: bytes the launch HAS to move (forward
    x, identity -> y = 3*E*b; backward G, x, identity -> dx = 4*E*b; the un-fused op 2 / 3) over the HIP-event
    interval of the call, against the 8 TB/s peak — at the two sites that hold most of the 16-site sum."""
    res = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel_us_source": "HIP events around each call, median of 30"}
    # (`_channels_last`: the same block in torch.channels_last — what the sites of the channels-last ResNet-50 run since round 5:
    # ONE persistent launch per direction since round 6, two tensor touches, 5 + 5 passes with the kept sum; `frac` stays on the
    # bytes a call HAS to move, `achieved_on_bytes_moved` is the rate on the passes the launch makes)
    CL = torch.channels_last
    for name, shape, block, fmt in (("block_256x256x56x56", (256, 256, 56, 56), True, None), ("block_256x512x28x28", (256, 512, 28, 28), True, None),
                                    ("cnsn_neither_256x256x56x56", (256, 256, 56, 56), False, None),
                                    ("block_256x256x56x56_channels_last", (256, 256, 56, 56), True, CL),
                                    ("block_256x512x28x28_channels_last", (256, 512, 28, 28), True, CL),
                                    ("block_256x1024x14x14_channels_last", (256, 1024, 14, 14), True, CL),
                                    ("block_256x2048x7x7_channels_last", (256, 2048, 7, 7), True, CL),
                                    # the whole bottleneck tail bn3 + add + SelfNorm + ReLU in ONE launch per direction (round 6,
                                    # cnsn_forward_bn_block): conv_out, identity -> y = 3 passes; G, conv_out, identity -> two gradients = 5
                                    ("bn_block_256x256x56x56_channels_last", (256, 256, 56, 56), "bn", CL),
                                    ("bn_block_256x1024x14x14_channels_last", (256, 1024, 14, 14), "bn", CL)):
        n, c, h, w = shape
        eb = n * c * h * w * 2
        a = conditioned(shape, dev, torch.bfloat16, 61)
        idt = conditioned(shape, dev, torch.bfloat16, 62) * 0.5
        gy = torch.randn(shape, device=dev).to(torch.bfloat16)
        if fmt is not None:
            a, idt, gy = a.contiguous(memory_format=fmt), idt.contiguous(memory_format=fmt), gy.contiguous(memory_format=fmt)
        a, idt = a.detach().requires_grad_(), idt.detach().requires_grad_()
        mod = cnsn_amd.CNSN(None if block else cnsn_amd.CrossNorm("neither", 1), cnsn_amd.SelfNorm(c)).to(dev).train()
        bn3 = torch.nn.BatchNorm2d(c).to(dev).train() if block == "bn" else None
        ins = [a] + ([idt] if block else []) + list(mod.parameters()) + (list(bn3.parameters()) if bn3 is not None else [])

        def fwd():
            if mod.crossnorm is not None:
                mod.crossnorm.active = True
            if bn3 is not None:
                return mod.forward_bn_block(a, bn3, idt, relu=True)
            return mod.forward_block(a, idt, add_mode="pre", relu=True) if block else mod(a)

        for _ in range(6):
            torch.autograd.grad(fwd(), ins, gy)
        torch.cuda.synchronize()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(30)]
        for e in ev:
            e[0].record()
            y = fwd()
            e[1].record()
            torch.autograd.grad(y, ins, gy)
            e[2].record()
        torch.cuda.synchronize()
        tf = sorted(e[0].elapsed_time(e[1]) for e in ev)[15] * 1e-3
        tb = sorted(e[1].elapsed_time(e[2]) for e in ev)[15] * 1e-3
        pf, pb = (3, 5) if block == "bn" else ((3, 4) if block else (2, 3))
        cfg = cnsn_amd.FusedConfig(cn_active=not block, sn_active=True, add_mode="pre" if block else "none", relu=bool(block))
        res[name] = {"forward": {"kernel_us": round(tf * 1e6, 1), "bytes": pf * eb, "achieved": round(pf * eb / tf / 1e9, 1),
                                 "frac": round(pf * eb / tf / 1e9 / HBM_PEAK_GBS, 4)},
                     "backward": {"kernel_us": round(tb * 1e6, 1), "bytes": pb * eb, "achieved": round(pb * eb / tb / 1e9, 1),
                                  "frac": round(pb * eb / tb / 1e9 / HBM_PEAK_GBS, 4)},
                     "sn_cluster_kernels": [bool(cnsn_amd.sn_cluster(a, cfg)), bool(cnsn_amd.sn_cluster(a, cfg, backward=True))]}
        if fmt is not None:
            res[name].pop("sn_cluster_kernels")
            res[name]["kernels"] = cnsn_amd.which_path(a, cfg)       # 'resident' = the single-launch kernels
            mv = (5, 8) if block == "bn" else (5, 5)
            res[name]["passes_moved"] = list(mv)
            res[name]["forward"]["achieved_on_bytes_moved"] = round(mv[0] * eb / tf / 1e9, 1)
            res[name]["backward"]["achieved_on_bytes_moved"] = round(mv[1] * eb / tb / 1e9, 1)
            if block == "bn":
                res[name]["unfused_passes"] = [8, 10]     # MIOpen's BatchNorm2d 3 + 5, the op's launches 5 + 5
        del a, idt, gy, mod
    return res


def model_line(workload, steps, warmup, timeout_s, extra_args=()):
    """The images/s line of a caller backbone (`bench.py --workload ...`), run as a child process with a time limit so that
    the default bench still finishes in minutes (MIOpen's first-run kernel search is the unknown); None fields on time-out."""
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(steps), "--warmup", str(warmup),
           *extra_args]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"images_per_s": d["value"], "ms_per_step": d["ms_per_step"], "config": d["config"]["workload"],
                "per_gpu_batch": d["config"]["per_gpu_batch"], "steps": steps, "warmup": warmup,
                "wall_s_incl_startup": round(time.perf_counter() - t0, 1)}
    except Exception as e:  # noqa: BLE001  (time-out, no JSON line: report, never fail the headline over it)
        return {"images_per_s": None, "error": f"{type(e).__name__}: {str(e)[:160]}", "wall_s": round(time.perf_counter() - t0, 1)}


def pmc_means(directory, counter):
    """{"fwd": KiB, "bwd": KiB}: the mean of `counter` per dispatch of the library's forward / backward kernels in the
    `*counter_collection.csv` files rocprofv3 left under `directory` (a two-pass path has several kernels per launch: their
    means are added)."""
    import csv
    import glob
    per = {}                                               # kernel name -> values of its dispatches
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "cnsn::" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    out = {}
    for direction in ("fwd", "bwd"):
        vals = [sum(v) / len(v) for k, v in per.items() if f"_{direction}_" in k or f"{direction}_kernel" in k]
        if vals:
            out[direction] = sum(vals)
    return out


def live_traffic(args, timeout_s=45):
    """`roofline.traffic` measured NOW instead of replayed from profiles/: this bench re-run as a child process under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE`, then `--pmc WRITE_SIZE` — counters in passes of their own next to the
    kernel trace only, as MI355X_MICROARCH.md prescribes — a few steps each, same shape / dtype / crop / strategy.  HBM bytes
    per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
    stream (calibrated against kernels of known traffic in profiles/r01_pmc_traffic.txt).  Returns
    ({"fwd": bytes, "bwd": bytes}, source) or (None, reason): a box without rocprofv3 or counters never fails the headline."""
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ):
        return None, "this process is itself being profiled: no nested rocprofv3 passes"
    child = [sys.executable, os.path.abspath(__file__), "--steps", "4", "--warmup", "2", "--no-extra", "--no-cpu-baseline",
             "--no-ceiling", "--no-alt", "--shape", args.shape, "--dtype", args.dtype, "--crop", args.crop, "--kind", args.kind,
             "--strategy", args.strategy]
    mean = {}                                              # (direction, counter) -> KiB per launch
    t0 = time.perf_counter()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="cnsn_pmc_", dir="/tmp")
        try:
            subprocess.run([rocprof, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", *child],
                           capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            for direction, kib in pmc_means(d, ctr).items():
                mean[(direction, ctr)] = kib
        except Exception as e:  # noqa: BLE001
            return None, f"live collection failed ({type(e).__name__}: {str(e)[:120]})"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for direction in ("fwd", "bwd"):
        if (direction, "FETCH_SIZE") in mean and (direction, "WRITE_SIZE") in mean:
            out[direction] = int((2 * mean[(direction, "FETCH_SIZE")] + mean[(direction, "WRITE_SIZE")]) * 1024)
    if "bwd" not in out:
        return None, "live collection failed (no counter rows for the library's kernels)"
    return out, (f"live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of this command (4 steps) run as child "
                 f"processes just now ({time.perf_counter() - t0:.0f} s); bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, "
                 "the gfx950 correction of MI355X_MICROARCH.md calibrated in profiles/r01_pmc_traffic.txt")


SWEEP_SHAPES = [(8, 64, 32, 32), (128, 32, 32, 32), (128, 64, 16, 16), (128, 128, 8, 8), (256, 256, 56, 56),
                (256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7), (96, 256, 56, 56), (96, 512, 28, 28),
                (96, 1024, 14, 14), (96, 2048, 7, 7), (768, 3, 224, 224), (16, 256, 128, 128), (16, 2048, 64, 64)]


def sweep(cnsn_amd, dev):
    """`python bench.py --sweep`: every shape of SURVEY §8 d1 x dtype x mode — ms per forward+backward (HIP events),
    (best of 3 x 20 calls), algorithmic GB/s (8*E*b / t) and the kernels AUTO resolved to (forward/backward) — as a markdown table
    (profiles/r02_shape_sweep.md).  Not the one-line contract: a measurement aid."""
    def timeit(fn, k=20, w=5):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k

    a = torch.empty(205520896, device=dev)
    b = torch.empty_like(a)
    c = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    print(f"copy fp32 822 MB: {2 * a.numel() * 4 / t / 1e6:.0f} GB/s (read+write)")
    t = timeit(lambda: torch.add(a, b, alpha=2.0, out=c))
    print(f"triad fp32: {3 * a.numel() * 4 / t / 1e6:.0f} GB/s\n")
    del a, b, c
    modes = [("sn", "neither", True), ("sn", "neither", False), ("cn", "neither", True), ("cn", "both", True),
             ("cnsn", "neither", True), ("cnsn", "both", True)]
    print("| shape | dtype | " + " | ".join(f"{k}{'' if k == 'sn' else '/' + cr}{'' if tr else ' eval'}" for k, cr, tr in modes) + " |")
    print("|---|---|" + "---|" * len(modes))
    short = {"streaming": "S", "packed": "P", "resident": "R", "local": "L", "mono": "M"}
    for shape in SWEEP_SHAPES:
        for dt, dtype in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            x = conditioned(shape, dev, dtype, 1).requires_grad_()
            gy = torch.randn(shape, device=dev).to(dtype)
            e = shape[0] * shape[1] * shape[2] * shape[3]
            eb = e * (4 if dtype == torch.float32 else 2)
            cells = []
            for kind, crop, train in modes:
                mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1) if kind != "sn" else None,
                                    cnsn_amd.SelfNorm(shape[1]) if kind != "cn" else None).to(dev)
                mod.train(train)
                if kind != "sn":
                    mod.crossnorm.train(True)
                ins = [x] + list(mod.parameters())

                def run():
                    if mod.crossnorm is not None:
                        mod.crossnorm.active = True
                    torch.autograd.grad(mod(x), ins, gy)

                cfg = cnsn_amd.FusedConfig(cn_active=kind != "sn", sn_active=kind != "cn", sn_training=train,
                                           content_box=(1, 1, 3, 3) if crop == "both" else None,
                                           style_box=(0, 0, 2, 2) if crop == "both" else None)
                path = short[cnsn_amd.which_path(x, cfg, False)] + short[cnsn_amd.which_path(x, cfg, True)]
                # best of three timings: calls of a few tens of microseconds leave the GPU mostly idle, and its clocks
                # then wander between power states from one timing to the next (0.055 vs 0.09 ms for the same call)
                t = min(timeit(run) for _ in range(3))
                cells.append(f"{t:.3f} ms {8 * eb / t / 1e6:.0f} {path}")
            print(f"| {shape} | {dt} | " + " | ".join(cells) + " |", flush=True)


def model_workload(args, dist, world, rank, dev):
    """Whole training steps (forward, CE [+ image-space CrossNorm], backward, SGD) of the caller
    backbones on synthetic data — BASELINE.json configs[1] (WRN-40-2+CNSN, bs128, fp32, 32x32) and
    configs[2] (ResNet-50+SN, image-space CrossNorm as imagenet-scripts/run-cnsn.sh, bs256, bf16)."""
    import numpy as np
    import cnsn_amd
    from cnsn_amd import data_parallel as dp
    from cnsn_amd.callers import ResNet50CNSN, StepGuard, WideResNetCNSN, image_space_crossnorm, jsd_consistency
    dp.seed_rank(4321, rank)
    views = 1
    if args.workload == "resnet50_jsd":          # BASELINE.json configs[3]: 3 views x 32 per GPU, CE + 12 * JSD
        bs, hw, ncls, amp, views = args.batch or 32, 224, 1000, torch.bfloat16, 3
        net = ResNet50CNSN(num_classes=ncls, cnsn_type="sn", pos="post").to(dev)
        name = ("ResNet-50+SN(post), 3 views, ONE image-space CrossNorm(p=0.5) on the concatenated batch, "
                "CE(clean) + 12*JSD (imagenet.py:337-406), bf16 autocast")
    elif args.workload == "resnet50":
        bs, hw, ncls, amp = args.batch or 256, 224, 1000, torch.bfloat16
        net = ResNet50CNSN(num_classes=ncls, cnsn_type="sn", pos="post").to(dev)
        name = "ResNet-50+SN(post) + image-space CrossNorm(p=0.5, crop=neither), bf16 autocast"
    elif args.workload == "seg":
        # BASELINE.json configs[4]: dilated FCN-ResNet50 + CNSN (segmentation/config/gtav/gtav_fcn50_cnsn.yaml:34-43: SelfNorm at
        # 'residual', a separate CrossNorm 'post' with crop='style', 1 of 16 sites armed with mix_prob 0.5), 512x512 crops,
        # bs 16, CE + 0.4 * aux CE (tool/train_cnsn.py:295-321, model/fcn.py:38-50), SGD(0.01, 0.9, 1e-4)
        from cnsn_amd.callers import FCNHead, SegResNet50CNSN
        bs, hw, ncls, amp = args.batch or 16, 512, 19, (torch.bfloat16 if args.dtype == "bf16" else None)

        class _FCN(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.backbone = SegResNet50CNSN(block_idxs="1_2_3_4", active_num=1, pos="residual", beta=1, crop="style",
                                                cnsn_type="cnsn", cn_pos="post")
                self.classifier, self.aux_classifier = FCNHead(2048, ncls), FCNHead(1024, ncls)

            def forward(self, x):
                f = self.backbone(x)
                up = lambda t: torch.nn.functional.interpolate(t, size=x.shape[-2:], mode="bilinear", align_corners=False)  # noqa: E731
                return up(self.classifier(f["out"])), up(self.aux_classifier(f["aux"]))

        net = _FCN().to(dev)
        name = ("FCN-ResNet50 (dilated) + SN(residual) + CrossNorm(post, crop=style, 1 of 16 sites armed with p=0.5), "
                f"512x512, CE + 0.4*aux CE, {'bf16 autocast' if amp else 'fp32'}")
    else:
        bs, hw, ncls, amp = args.batch or 128, 32, 100, None
        net = WideResNetCNSN(40, ncls, 2, active_num=2, pos="post", beta=1, crop="both", cnsn_type="cnsn").to(dev)
        name = "WideResNet-40-2+CNSN(post, crop=both, 2 of 18 sites armed with p=0.5), fp32"
    net.train()
    if args.workload in ("resnet50", "resnet50_jsd") and not args.nchw:
        args.channels_last = True
    if args.channels_last:
        net = net.to(memory_format=torch.channels_last)
        name += "; channels-last"
        if args.workload in ("resnet50", "resnet50_jsd"):
            from cnsn_amd import functional as _F
            from cnsn_amd.callers import _sites
            if _F._BN_BLOCK and _sites.FUSE_BLOCK:   # (round 6: profiles/r06_bn_block.md; CNSN_BN_BLOCK=0 calls the BatchNorm2d modules)
                name += ", bn3 / downsample BatchNorm2d inside the op's launch"
    model = net
    if dist is not None:
        if dist.get_backend() == "nccl":
            model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[dev.index], broadcast_buffers=True,
                                                              bucket_cap_mb=25)
        else:   # ranks share a device (gloo): same bucketing, reduction staged by gloo's own device support
            model = torch.nn.parallel.DistributedDataParallel(net, broadcast_buffers=True, bucket_cap_mb=25)
    opt = torch.optim.SGD(net.parameters(), lr=0.01 if hw == 512 else 0.1, momentum=0.9,
                          weight_decay=1e-4 if hw in (224, 512) else 5e-4, nesterov=(hw == 32))
    g = torch.Generator(device=dev).manual_seed(5 + rank)
    x = torch.randn(bs, 3, hw, hw, device=dev, generator=g)
    y = torch.randint(0, ncls, (bs,), device=dev, generator=g)
    if args.workload == "seg":                                     # per-pixel labels, a tenth of them 'ignore' (255)
        y = torch.randint(0, ncls, (bs, hw, hw), device=dev, generator=g)
        y[torch.rand(bs, hw, hw, device=dev, generator=g) < 0.1] = 255

    if views == 3:
        x = torch.cat([x, x + 0.1 * torch.randn_like(x), x + 0.1 * torch.randn_like(x)], 0)   # clean + two "augmented"
    cl = (lambda t: t.contiguous(memory_format=torch.channels_last)) if args.channels_last else (lambda t: t)
    if args.workload not in ("resnet50", "resnet50_jsd"):
        x = cl(x)            # (the image-space CrossNorm of the ResNet workloads runs on the loader's NCHW batch first)

    fault = FaultPlan(rank)

    def compute_loss():
        fault.before_step()
        xb = x
        if views == 3:
            xb = cl(image_space_crossnorm(x, 0.5, 1, "neither", cnsn_amd.cn_op_2ins_space_chan))   # imagenet.py:352-358
            with torch.autocast("cuda", dtype=amp):
                logits = model(xb).float()
            l_clean, l_a1, l_a2 = torch.split(logits, bs)
            return torch.nn.functional.cross_entropy(l_clean, y) + 12.0 * jsd_consistency(l_clean, l_a1, l_a2)
        if args.workload == "resnet50":
            xb = cl(image_space_crossnorm(x, 0.5, 1, "neither", cnsn_amd.cn_op_2ins_space_chan))   # imagenet.py:211-215
            with torch.autocast("cuda", dtype=amp):
                return torch.nn.functional.cross_entropy(model(xb).float(), y)
        if args.workload == "seg":                                                               # train_cnsn.py:302-313
            r = np.random.rand(1)
            armed = bool(r < 0.5)
            if armed:
                net.backbone._enable_cross_norm()
            with torch.autocast("cuda", dtype=amp, enabled=amp is not None):
                out, aux = model(xb)
            if armed:
                net.backbone._disable_cross_norm()
            ce = torch.nn.functional.cross_entropy
            return ce(out.float(), y, ignore_index=255) + 0.4 * ce(aux.float(), y, ignore_index=255)
        r = np.random.rand(1)                                                                    # cifar.py:127-131
        return torch.nn.functional.cross_entropy(model(xb, aug=bool(r < 0.5)), y)

    # The time-out protocol per WINDOW of steps (warm-up window, timed window), as in the headline workload: inside a
    # window nothing is raised and nothing synchronises the stream (a per-step `StepGuard.run` costs a stream
    # synchronisation per step — 25 % of a launch-bound WideResNet step); every rank runs the same number of steps and
    # DDP all-reduces; behind the window ONE 4-byte MAX all-reduce says whether any rank's cluster launch gave up, and then
    # EVERY rank puts parameters, buffers, optimizer state and RNG streams back to the window's start (callers.steps.StepGuard
    # with `optimizer=`), switches the cluster kernels off and runs the window again.  (`callers.steps.train_step_*` keep the
    # per-step form: a training loop cannot replay a window of data.)
    guard = StepGuard(net, optimizer=opt)

    def step():
        loss = compute_loss()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    graphed = None
    if args.workload == "wrn40" and dist is None and not args.no_graph:
        # launch-bound network: the steps with idle CrossNorm sites (half of them at cn_prob 0.5) replay a captured
        # step — forward, loss, backward, SGD — instead of ~700 eager launches; armed steps stay eager
        from cnsn_amd.callers import GraphedIdleStep
        graphed = GraphedIdleStep(net, opt, x, y, graph_blocks=args.block_graphs)
        name += "; idle-site steps replayed from a HIP graph" + (", armed steps from one graph per idle block"
                                                                 if args.block_graphs else "")

        def step():                                                                              # noqa: F811
            graphed.step(x, y, 0.5)

    def window(k):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        with cnsn_amd._ffi.deferred_timeouts():
            for _ in range(k):
                step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t

    def settled_window(k):
        for _ in range(3):
            if graphed is None:
                guard.save()
            t = window(k)
            new = cnsn_amd._ffi.poll_timeouts()               # (the stream is idle: every launch of the window has run)
            guard.local_timeouts += new
            if dp.agree_to_repeat(new, dev) == 0:
                guard.step_applied(k)                             # (the way back: re-armed after a clean stretch, rank-agreed)
                return t
            guard.repeats += k
            if new:
                print(f"[bench] rank {rank}: {new} cluster launch(es) gave up; all ranks repeat the window of {k} steps",
                      file=sys.stderr)
            guard.degrade()
            if graphed is None:
                guard.restore()
        raise cnsn_amd.CnsnError("bench: cluster launches still time out after three windows")

    if graphed is None:
        step()                                                # (creates the optimizer's state tensors: part of every snapshot)
    settled_window(args.warmup)
    dt = settled_window(args.steps)
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    per_rank_timeouts = dp.gather_ints(guard.local_timeouts, dev)
    if rank == 0:
        print(json.dumps({
            "steps_repeated_after_a_cluster_timeout": guard.repeats, "resident_timeouts": per_rank_timeouts,
            "resident_rearmed": [guard.rearms] * world,   # (degradations are rank-agreed: every rank re-arms at the same step)
            "metric": ("ResNet-50+CNSN images/sec" if args.workload.startswith("resnet50") else
                       "FCN-ResNet50+CNSN 512x512 images/sec" if args.workload == "seg" else "WideResNet-40-2+CNSN images/sec"),
            "value": round(world * bs * views * args.steps / dt, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if amp else "f32", "data": "synthetic",
            "config": {"workload": name, "per_gpu_batch": bs * views, "global_batch": bs * views * world, "image": hw,
                       "parallelism": f"ddp{world} ({'RCCL' if dist is None or dist.get_backend() == 'nccl' else dist.get_backend()} gradient all-reduce, 25 MB buckets)",
                       "world_size": world, "devices": torch.cuda.device_count(),
                       "backend": None if dist is None else dist.get_backend()}}), flush=True)


class FaultPlan:
    """Tests of the time-out protocol (tests/test_gpu_step_guard.py): CNSN_BENCH_FAULT="RANK:CALL" makes that rank's
    CALL-th step (counted over warm-up and timed steps, repeats included) run with CNSN_FAULT_INJECT=1 — one member of a
    cluster never publishes, the launch gives up after CNSN_WAIT_MS.  Unset: does nothing."""

    def __init__(self, rank):
        spec = os.environ.get("CNSN_BENCH_FAULT", "")
        self.rank, self.call = (int(v) for v in spec.split(":")) if spec else (-1, -1)
        self.mine = self.rank == rank
        self.calls, self.on = 0, False

    def before_step(self):
        if not self.mine:
            return
        import cnsn_amd
        want = self.calls == self.call
        self.calls += 1
        if want != self.on:
            os.environ["CNSN_FAULT_INJECT"] = "1" if want else "0"
            cnsn_amd.reload_env()
            self.on = want


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launcher_command(n, argv, port=None):
    """The command `bench.py --gpus n` re-executes itself as when it was started without a torchrun environment:
    one process per GPU on this node (the reference's multi-GPU entry is the single line
    `net = torch.nn.DataParallel(net).cuda()`, imagenet.py:533 / cifar.py:395)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()),
            os.path.abspath(__file__), *argv]


def pick_backend(world, n_devices):
    """nccl (= RCCL over xGMI) with one device per rank; when ranks have to SHARE a device (more ranks than GPUs on
    the box) RCCL refuses ("Duplicate GPU detected") and the collectives go over gloo."""
    forced = os.environ.get("CNSN_BENCH_BACKEND")
    if forced:
        return forced
    return "nccl" if n_devices >= world else "gloo"


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as plain `python bench.py --gpus N`: become the launcher of N ranks and relay their exit code
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = launcher_command(args.gpus, sys.argv[1:])
        if os.environ.get("CNSN_BENCH_DRY_LAUNCH") == "1":     # (tests: show the command, start nothing)
            print(json.dumps({"launch": cmd}))
            return
        sys.exit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: using the launched world size", file=sys.stderr)
    ngpu = torch.cuda.device_count()
    backend = None
    if world > 1:
        import torch.distributed as dist
        backend = pick_backend(world, ngpu)
        local = local % max(ngpu, 1)          # (more ranks than devices: ranks share GPUs)
        if ngpu:
            torch.cuda.set_device(local)
        if backend == "nccl":                                                    # nccl == RCCL on ROCm
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:                                                                    # gloo: shared device / plumbing test
            dist.init_process_group(backend)
    else:
        dist = None
    if os.environ.get("CNSN_BENCH_PLUMBING") == "1":
        # launcher / rendezvous check without a GPU (tests): every rank reports, rank 0 prints the world it saw
        t = torch.tensor([float(rank)])
        if dist is not None:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"plumbing": True, "n_gpus": world, "backend": backend, "rank_sum": float(t.item())}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    assert torch.cuda.is_available(), "bench.py needs an MI355X (use gpurun)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import cnsn_amd
    from cnsn_amd import data_parallel as dp
    cnsn_amd.lib()                                    # fail loudly now if the .so is missing
    cnsn_amd._ffi.under_process_group_defaults()      # (a shorter bound on cluster waits when peers would wait with us)
    cnsn_amd.set_strategy(args.strategy)
    if world > ngpu:
        # Ranks SHARE a device (a 1-GPU box running the N-rank launcher).  The cluster-resident kernels need the GPU to
        # themselves: the persistent grids of two PROCESSES can each hold the slots the other's cluster members are
        # waiting for, XCD by XCD (observed: a 5 s time-out -> the library degrades and reports it).  One rank per
        # GPU — the configuration this bench is for — is not affected.
        if not FaultPlan(rank).mine:                  # (a fault-injection test keeps them on ONE rank: no second grid to wait for)
            cnsn_amd.set_resident(False)
    import numpy as np
    if args.sweep:
        sweep(cnsn_amd, dev)
        return
    if args.workload != "cnsn":
        model_workload(args, dist, world, rank, dev)
        if dist is not None:
            dist.destroy_process_group()
        return

    shape = tuple(int(v) for v in args.shape.split(","))
    n, c, h, w = shape
    dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    b = 4 if dtype == torch.float32 else 2
    e = n * c * h * w

    dp.seed_rank(1234, rank)                          # ranks draw different perms / boxes
    x = conditioned(shape, dev, dtype, 10 + rank).requires_grad_()
    gy = torch.randn(shape, device=dev, generator=torch.Generator(device=dev).manual_seed(20 + rank)).to(dtype)
    mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(args.crop, 1) if args.kind != "sn" else None,
                        cnsn_amd.SelfNorm(c) if args.kind != "cn" else None).to(dev).train()
    params = [p for p in mod.parameters()]

    # HIP events around the forward and the backward call of every `ev_every`-th timed step.  An event record is a packet of
    # its own on the queue: it keeps the GPU idle for about 4 us between two dependent launches (rocprofv3 timeline,
    # profiles/r04_launches_per_step.md), so three records in EVERY step cost the timed region 1.5 % of a step that consists
    # of two launches.  Sampling every fourth step keeps the per-launch intervals live and leaves the other steps alone.
    ev_every = 1 if args.steps < 12 else 4
    ev = {i: [torch.cuda.Event(enable_timing=True) for _ in range(3)] for i in range(0, args.steps, ev_every)}

    fault = FaultPlan(rank)

    def step(i=None):
        fault.before_step()
        if mod.crossnorm is not None:
            mod.crossnorm.active = True               # what _enable_cross_norm does before a forward
        x.grad = None
        for p in params:
            p.grad = None
        rec = ev.get(i)
        if rec is not None:
            rec[0].record()
        y = mod(x)
        if rec is not None:
            rec[1].record()
        y.backward(gy)
        if rec is not None:
            rec[2].record()
        if dist is not None and params:               # DDP-style gradient all-reduce (RCCL over xGMI)
            dp.allreduce_gradients(params)

    # A cluster launch that gave up (the GPU was shared with something that kept part of a persistent grid off the device
    # for seconds) has marked its outputs with NaNs and bumped the library's counter.  Nothing is raised in the middle of a
    # step (`deferred_timeouts`): a rank that left a step early would miss the gradient all-reduce its peers are in.  The
    # ranks run a whole WINDOW of steps — each with exactly one all-reduce —, then agree with one 4-byte MAX all-reduce
    # whether any rank's launch gave up (data_parallel.agree_to_repeat); if so EVERY rank switches the cluster kernels off
    # and repeats the whole window, so the reported time is that of a window in which every launch completed.
    from cnsn_amd.callers import StepGuard
    state = StepGuard(mod, restore_rng=False)         # SelfNorm's running statistics: NaN after a failed window
    repeats, local_timeouts = 0, 0

    def window(k, record):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        with cnsn_amd._ffi.deferred_timeouts():
            for i in range(k):
                step(i if record else None)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t

    def settled_window(k, record):
        nonlocal repeats, local_timeouts
        for _ in range(3):
            state.save()
            t = window(k, record)
            new = cnsn_amd._ffi.poll_timeouts()           # (the stream is idle: every launch of the window has run)
            local_timeouts += new
            if dp.agree_to_repeat(new, dev) == 0:
                state.step_applied(k)                             # (the way back: re-armed after a clean stretch, rank-agreed)
                return t
            repeats += k
            if new:
                print(f"[bench] rank {rank}: {new} cluster launch(es) gave up; all ranks repeat the window of {k} steps "
                      "on the two-pass kernels", file=sys.stderr)
            state.degrade()
            state.restore()
        raise cnsn_amd.CnsnError("bench: cluster launches still time out after three windows")

    # The box's own ceilings (roofline.ceiling) are measured BEFORE the warm-up: 0.3 s of plain copies in this process.  They
    # belong to the line either way; ahead of the timed region they also take the device out of its idle clocks — a fresh
    # process needs ~25 headline steps to settle (profiles/r04_launches_per_step.md: 0.852 ms at steps 5-9, 0.79 from step 25),
    # more than the W = 5 the driver passes.  W warm-up steps and exactly K timed steps follow as the contract says.
    ceil = copy_triad_ceiling(dev) if (world == 1 and not args.no_ceiling) else None
    # Where y and dx live.  Default: the library's output arena (cnsn_amd.arena) — blocks of its own, nothing probed, nothing
    # timed.  --no-arena: torch's caching allocator.  After the contract's W + K steps the same K steps are timed again with
    # the OTHER allocator (one warm-up window in front) so that the line carries both; --prospect N adds a third window
    # after the arena has looked at N candidate blocks (profiles/r04_memory_map.md: where a block lies physically decides how
    # fast the launches write it).
    from cnsn_amd import arena as _arena
    if args.no_arena:
        _arena.disable()
    settled_window(args.warmup, False)
    dt = settled_window(args.steps, True)
    if dist is not None:
        tt = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    per_rank_timeouts = dp.gather_ints(local_timeouts, dev)

    # ---- the same K steps with the other allocator / after a bounded search (outside the contract's timed region)
    alt = {}
    if world == 1 and cnsn_amd._ffi.glue() is not None and not args.no_alt:
        arena_stats = _arena.stats(dev) if not args.no_arena else None
        if args.no_arena:
            _arena.enable()
        else:
            _arena.disable()
        x.grad = None
        settled_window(max(3, args.warmup), False)
        alt["ms_per_step_arena" if args.no_arena else "ms_per_step_plain_allocator"] = round(
            settled_window(args.steps, False) / args.steps * 1e3, 4)
        _arena.enable()
        if args.prospect > 0:
            x.grad = None
            torch.cuda.synchronize()
            alt["prospect"] = _arena.prospect(x, keep=4, candidates=args.prospect)
            settled_window(max(3, args.warmup), False)
            alt["ms_per_step_prospected"] = round(settled_window(args.steps, False) / args.steps * 1e3, 4)
        if args.no_arena:
            _arena.disable()
        alt["arena"] = arena_stats if arena_stats is not None else _arena.stats(dev)

    fwd_ms = sum(a.elapsed_time(m_) for a, m_, _ in ev.values()) / len(ev)
    bwd_ms = sum(m_.elapsed_time(z) for _, m_, z in ev.values()) / len(ev)
    ms_per_step = dt / args.steps * 1e3
    step_bytes = 8 * e * b
    value = world * step_bytes / (dt / args.steps) / 1e9

    if rank == 0:
        # Bytes of the dominant launch (the backward).  SURVEY §8(d3) prices the op as a two-pass algorithm: forward
        # 3*E*b, backward 5*E*b (`value` keeps that accounting: it is the metric BASELINE.json quotes).  The roofline
        # object prices the kernel on the bytes a launch HAS to move — single touch: read G, read x, write dx = 3*E*b
        # (forward: read x, write y = 2*E*b) — so that `frac` is a physical fraction of the HBM peak.
        cfg_path = cnsn_amd.FusedConfig(cn_active=args.kind != "sn", sn_active=args.kind != "cn", sn_training=True,
                                        content_box=(1, 1, 3, 3) if args.crop in ("content", "both") else None,
                                        style_box=(0, 0, 2, 2) if args.crop in ("style", "both") else None)
        path_f, path_b = cnsn_amd.which_path(x, cfg_path, False), cnsn_amd.which_path(x, cfg_path, True)
        single = {"resident", "local", "mono"}   # one launch, every plane read once
        moved_f = (2 if path_f in single else 3) * e * b       # what the kernels of this path read + write
        moved_b = (3 if path_b in single else 5) * e * b
        need_f, need_b = 2 * e * b, 3 * e * b                  # the least any implementation moves
        traffic = None
        traffic_source = None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(f"{args.dtype}_{args.crop}_{args.kind}_bwd_bytes")
                if traffic is not None:
                    traffic_source = ("profiles/traffic_latest.json (replayed: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                      "passes of an earlier run of this build, corrected per MI355X_MICROARCH.md; "
                                      "counters cannot be collected inside this process)")
            except (OSError, ValueError):
                traffic = None
        bwd_s, fwd_s = bwd_ms * 1e-3, fwd_ms * 1e-3
        out = {
            "metric": "fused CNSN fwd+bwd GB/s vs HBM roofline",
            "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"fused CrossNorm(crop={args.crop})+SelfNorm fwd+bwd, "
                                   f"NCHW ({n},{c},{h},{w}) per GPU, kind={args.kind}",
                       "shape": list(shape), "crop": args.crop, "kind": args.kind,
                       "strategy": args.strategy, "paths": {"forward": path_f, "backward": path_b},
                       "algorithmic_bytes_per_step": step_bytes,
                       "parallelism": f"dp{world} (independent minibatches; grads of 4C SN params all-reduced)",
                       "world_size": world, "devices": ngpu, "backend": backend},
            # READ THIS ONE FIRST: the whole step on the bytes it physically has to move (5*E*b: x in + y out, G and x in + dx out)
            "frac_of_hbm_peak_bytes_needed": round((need_f + need_b) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
            "value_note": "value = 8*E*b per step / time — SURVEY §8(d3) prices the op as a two-pass algorithm (BASELINE's metric); "
                          "this implementation touches every tensor once (5*E*b), so value can exceed the 8 TB/s peak without "
                          "anything being skipped: frac_of_hbm_peak_bytes_needed is the physical fraction",
            "frac_of_hbm_peak": round(value / world / HBM_PEAK_GBS, 4),
            "fwd_ms": round(fwd_ms, 4), "bwd_ms": round(bwd_ms, 4),
            "images_per_s": round(world * n / (dt / args.steps), 1),
            "steps_repeated_after_a_cluster_timeout": repeats, "resident_timeouts": per_rank_timeouts,
            "resident_rearmed": [state.rearms] * world,   # (degradations are rank-agreed: every rank re-arms at the same step)
            "roofline": {"bound": "hbm",
                         "kernel": f"cnsn_backward launch ({path_b} path: reads G and x, writes dx)",
                         "bytes": need_b, "bytes_moved": moved_b,
                         "achieved": round(need_b / bwd_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(need_b / bwd_s / 1e9 / HBM_PEAK_GBS, 4),
                         "frac_moved": round(moved_b / bwd_s / 1e9 / HBM_PEAK_GBS, 4),
                         "kernel_us": round(bwd_ms * 1e3, 1),
                         "kernel_us_source": f"HIP events on the launch stream around the cnsn_backward call of every "
                                             f"{'timed step' if ev_every == 1 else f'{ev_every}th timed step ({len(ev)} samples)'} "
                                             "(includes the launch gap; rocprofv3 kernel-only averages are under profiles/)",
                         "traffic": traffic, "traffic_source": traffic_source,
                         "survey_d3": {"bytes": 5 * e * b, "achieved": round(5 * e * b / bwd_s / 1e9, 1),
                                       "frac": round(5 * e * b / bwd_s / 1e9 / HBM_PEAK_GBS, 4),
                                       "note": "two-pass accounting of SURVEY §8(d3) (the basis of `value`); can "
                                               "exceed 1 for a single-touch kernel"},
                         "forward": {"kernel_us": round(fwd_ms * 1e3, 1), "bytes": need_f, "bytes_moved": moved_f,
                                     "achieved": round(need_f / fwd_s / 1e9, 1),
                                     "frac": round(need_f / fwd_s / 1e9 / HBM_PEAK_GBS, 4),
                                     "survey_d3_frac": round(3 * e * b / fwd_s / 1e9 / HBM_PEAK_GBS, 4)}},
        }
        if world == 1 and not args.no_extra:
            live, why = live_traffic(args)
            if live is not None:
                out["roofline"]["traffic"], out["roofline"]["traffic_source"] = live["bwd"], why
                out["roofline"]["traffic_over_bytes"] = round(live["bwd"] / need_b, 4)
                if "fwd" in live:
                    out["roofline"]["forward"]["traffic"] = live["fwd"]
                    out["roofline"]["forward"]["traffic_over_bytes"] = round(live["fwd"] / need_f, 4)
            elif traffic_source is not None:
                out["roofline"]["traffic_source"] = traffic_source + "; " + why
        out.update({k: v for k, v in alt.items() if k.startswith("ms_per_step")})
        if alt:
            out["arena"] = {"on": not args.no_arena, "min_bytes": _arena.min_bytes(), **(alt.get("arena") or {}),
                            "note": "y / dx are tensors over blocks of the library's output arena (cnsn_amd.arena, C ABI "
                                    "cnsn_arena_*): a stable home; every NEW block of 384 MiB or more is the fastest of `tries` candidates timed "
                                    "with a plane-strided fill when it is created (the allocator's standing policy, during the "
                                    "warm-up here); ms_per_step_plain_allocator = the same K steps in this process with torch's "
                                    "caching allocator placing them"}
            if "prospect" in alt:
                out["arena"]["prospect"] = alt["prospect"]
                if "ms_per_step_prospected" in alt:
                    out["frac_of_hbm_peak_bytes_needed_prospected"] = round(
                        (need_f + need_b) / (alt["ms_per_step_prospected"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if ceil is not None:
            out["roofline"]["ceiling"] = ceil
            if "resident_order_triad_GBps" in ceil and list(shape) == [256, 256, 56, 56] and b == 4:
                # how close the two launches are to what their access pattern allows on THIS box
                out["roofline"]["frac_of_resident_order_ceiling"] = round(
                    out["roofline"]["achieved"] / ceil["resident_order_triad_GBps"], 4)
                out["roofline"]["forward"]["frac_of_resident_order_ceiling"] = round(
                    out["roofline"]["forward"]["achieved"] / ceil["resident_order_copy_GBps"], 4)
        if world == 1 and not args.no_extra:
            out["extra"] = secondary_workloads(cnsn_amd, shape, dev, args)
            out["extra"]["residual_block_add_cnsn_relu"] = residual_block_workloads(cnsn_amd, shape, dev)
            out["extra"]["inference"] = inference_workloads(cnsn_amd, shape, dev)
            out["roofline_bf16"] = roofline_bf16(cnsn_amd, dev)
            # (channels-last, the workload's default since round 5; `--workload resnet50 --nchw`: 4 030-4 074 img/s, profiles/r05_nhwc.md)
            out["extra"]["resnet50_bs256_bf16"] = model_line("resnet50", 12, 4, 200)
            out["extra"]["seg_bs16_512"] = {"bf16": model_line("seg", 5, 2, 150, ("--dtype", "bf16")),
                                               "f32_note": "fp32: 102-106 img/s (`--workload seg`, profiles/r05h_seg_f32.json); left out of the "
                                                           "default line to keep it within minutes (a child process per model line)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(shape, args.crop, args.kind, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
