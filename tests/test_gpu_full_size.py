"""Full-size parity at EVERY shape BASELINE.json's configs put through the op (round-1 verdict, weak #1: the shapes
other than (256,256,56,56) were only checked at toy N / C, while N drives the LDS carve, the cluster size K = N / own
and the "does not fit at N = 256" fall-backs).  The checker is the oracle's eager torch ops run ON THE GPU — fp32 on the
same (quantised) inputs, and fp64 as the truth that prices the oracle's own fp32 noise — so a full-size case takes well
under a second.  AUTO strategy (what a user gets), plus the forced strategies where they apply.

  configs[2]  ResNet-50 + SN, bs 256:  (256,256,56,56) (256,512,28,28) (256,1024,14,14) (256,2048,7,7)   bf16 / fp32
  configs[3]  3 views x 32 per GPU:    the same planes at N = 96
  configs[4]  segmentation, bs 16:     (16,256,128,128) ... (16,2048,64,64); SN at 'residual', CN(crop=style) at 'post'
  image-space CrossNorm (imagenet.py:215,358): (768,3,224,224)

Tolerances (BASELINE.json north_star): fp32 |hip - truth64| <= max(1e-5 * scale, 2 * |oracle32 - truth64|) for EVERY output
(round 3: parameter gradients too — they were held to 1e-4); bf16 |hip - oracle32 on the same quantised input| <= 1e-2 *
max|oracle32| for y / dx and 1e-3 for the fp32-accumulated parameter gradients and running statistics (round 3: from 5e-2).
Every comparison records its observed error next to its bound (gpurun_out/parity_margins.jsonl ->
profiles/r03_parity_margins.md): the worst fp32 error seen is 1.0e-6 of the scale (a tenth of the bound; the
2 x oracle-noise alternative is never the binding one at these shapes), bf16 y / dx 3.5e-3, bf16 parameter gradients 2e-6."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import cnsn_amd  # noqa: E402
from oracle import cnsn_oracle as orc  # noqa: E402
from tests.golden.gen_golden_fill import fill_sn  # noqa: E402

DEV = torch.device("cuda:0")


def make_input(shape, dtype, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    n, c = shape[:2]
    x = torch.randn(shape, generator=g, device=DEV)
    x.mul_(torch.rand(n, c, 1, 1, generator=g, device=DEV) * 1.5 + 0.5).add_(torch.randn(n, c, 1, 1, generator=g, device=DEV))
    gy = torch.randn(shape, generator=g, device=DEV)
    return x.to(dtype), gy.to(dtype)


def oracle_run(x, gy, kind, crop, draws, c, dtype64, device=DEV):
    """the oracle's eager ops in fp32 or fp64 on the SAME (already quantised) x / gy — on the GPU (ATen-on-ROCm: fast
    enough for every case) or, `device='cpu'`, literally "the reference PyTorch CPU path on identical inputs" of
    BASELINE.json's north_star (seconds per case at full size: one case per config, test_full_size_against_the_cpu_path)"""
    dt = torch.float64 if dtype64 else torch.float32
    sn = fill_sn(orc.SelfNorm(c), 4, dt).to(device).train() if kind != "cn" else None
    xo = x.detach().to(device=device, dtype=dt).clone().requires_grad_()
    gy = gy.to(device)
    u = xo
    if kind != "sn":
        u = orc.cn_op_2ins_space_chan(u, crop=crop, draws=orc.CNDraws(draws.perm, draws.style_box, None, draws.content_box))
    y = sn(u) if sn is not None else u
    y.backward(gy.detach().to(dt))
    res = {"y": y.detach(), "dx": xo.grad}
    if sn is not None:
        res["dw"] = sn.g_fc.weight.grad
        res["dgamma"] = sn.g_bn.weight.grad
        res["dbeta"] = sn.g_bn.bias.grad
        res["rm"] = sn.g_bn.running_mean
        res["rv"] = sn.g_bn.running_var
    return {k: v.to(DEV) for k, v in res.items()}


def hip_run(x, gy, kind, crop, draws, c):
    sn = fill_sn(cnsn_amd.SelfNorm(c), 4, torch.float32).to(DEV).train() if kind != "cn" else None
    mod = cnsn_amd.CNSN(cnsn_amd.CrossNorm(crop, 1) if kind != "sn" else None, sn).to(DEV).train()
    if mod.crossnorm is not None:
        mod.crossnorm.active = True
        mod.crossnorm.next_draws = draws
    xg = x.detach().clone().requires_grad_()
    y = mod(xg)
    y.backward(gy)
    torch.cuda.synchronize()
    res = {"y": y.detach(), "dx": xg.grad}
    if sn is not None:
        res.update(dw=sn.g_fc.weight.grad, dgamma=sn.g_bn.weight.grad, dbeta=sn.g_bn.bias.grad,
                   rm=sn.g_bn.running_mean, rv=sn.g_bn.running_var)
    return res


import json  # noqa: E402
import os  # noqa: E402

_MARGINS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_margins.jsonl")


def _record(row):
    """observed error next to its bound, one JSON line per (case, output): profiles/parity_table.py turns the file into
    the table of profiles/r03_parity_margins.md (round-2 review: nobody knew how much of the slack is used)"""
    try:
        os.makedirs(os.path.dirname(_MARGINS), exist_ok=True)
        with open(_MARGINS, "a") as f:
            f.write(json.dumps(row) + "\n")
    except OSError:
        pass


def check_case(shape, dtype, kind, crop, seed, oracle32_on="gpu"):
    torch.manual_seed(seed)
    np.random.seed(seed)
    c = shape[1]
    x, gy = make_input(shape, dtype, seed)
    draws = cnsn_amd.draw_cn(shape, crop, 1) if kind != "sn" else None
    got = hip_run(x, gy, kind, crop, draws, c)
    o32 = oracle_run(x, gy, kind, crop, draws, c, False, DEV if oracle32_on == "gpu" else torch.device("cpu"))
    o64 = oracle_run(x, gy, kind, crop, draws, c, True)
    for k, truth in o64.items():
        g_, r32 = got[k].double(), o32[k].double()
        scale = max(1.0, float(truth.abs().max()))
        noise = float((r32 - truth).abs().max())
        if dtype == torch.float32:
            rel = 1e-5        # north_star's fp32 bar, parameter gradients (sums over N*M products) included
            err = float((g_ - truth).abs().max())
            _record({"shape": list(shape), "dtype": "fp32", "kind": kind, "crop": crop, "out": k, "err": err, "scale": scale,
                     "oracle32_noise": noise, "bound": max(rel * scale, 2 * noise), "rel_tol": rel,
                     "strategy": cnsn_amd.functional._strategy, "oracle32_on": oracle32_on})
            assert err <= max(rel * scale, 2 * noise), f"{shape} fp32 {kind}/{crop} {k}: err {err:.3e}, oracle32 noise {noise:.3e}, scale {scale:.3g}"
        else:
            err = float((g_ - r32).abs().max())
            rel = 1e-2 if k in ("y", "dx") else 1e-3          # (parameter gradients / running statistics accumulate in fp32)
            _record({"shape": list(shape), "dtype": str(dtype).replace("torch.", ""), "kind": kind, "crop": crop, "out": k,
                     "err": err, "scale": max(float(r32.abs().max()), 1e-6),
                     "bound": rel * max(float(r32.abs().max()), 1e-6) + (0 if k in ("y", "dx") else 1e-5), "rel_tol": rel,
                     "strategy": cnsn_amd.functional._strategy, "oracle32_on": oracle32_on})
            assert err <= rel * max(float(r32.abs().max()), 1e-6) + (0 if k in ("y", "dx") else 1e-5), \
                f"{shape} {dtype} {kind}/{crop} {k}: err {err:.3e} vs max {float(r32.abs().max()):.3e}"
    del got, o32, o64
    torch.cuda.empty_cache()


R50 = [(256, 256, 56, 56), (256, 512, 28, 28), (256, 1024, 14, 14), (256, 2048, 7, 7)]
R50_96 = [(96, 256, 56, 56), (96, 512, 28, 28), (96, 1024, 14, 14), (96, 2048, 7, 7)]
SEG = [(16, 256, 128, 128), (16, 512, 64, 64), (16, 1024, 64, 64), (16, 2048, 64, 64)]
WRN = [(128, 32, 32, 32), (128, 64, 16, 16), (128, 128, 8, 8)]
ids = lambda s: "x".join(map(str, s)) if isinstance(s, tuple) else str(s)  # noqa: E731


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("kind,crop", [("sn", "neither"), ("cnsn", "neither"), ("cnsn", "both")])
@pytest.mark.parametrize("shape", R50 + R50_96, ids=ids)
def test_resnet50_sites_full_size(shape, dtype, kind, crop):
    check_case(shape, dtype, kind, crop, 11)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("kind,crop", [("sn", "neither"), ("cn", "style"), ("cnsn", "style")])
@pytest.mark.parametrize("shape", SEG, ids=ids)
def test_segmentation_sites_full_size(shape, dtype, kind, crop):
    check_case(shape, dtype, kind, crop, 12)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("crop", ["neither", "both"])
def test_image_space_crossnorm_full_size(dtype, crop):
    check_case((768, 3, 224, 224), dtype, "cn", crop, 13)       # imagenet.py:358 on the 3-view batch of 3 x 256
    check_case((256, 3, 224, 224), dtype, "cn", crop, 14)       # imagenet.py:215


@pytest.mark.parametrize("kind,crop", [("sn", "neither"), ("cnsn", "both"), ("cn", "content")])
@pytest.mark.parametrize("shape", WRN, ids=ids)
def test_wideresnet_sites_full_size(shape, kind, crop):
    check_case(shape, torch.float32, kind, crop, 15)


# north_star: "correctness is checked against the reference PyTorch CPU path on identical inputs within 1e-5 fp32 / 1e-2
# bf16".  Everything above runs the oracle's fp32 ops on the GPU (same ops, ATen-on-ROCm); here the fp32 oracle runs on
# CPU tensors, at FULL size, one case per BASELINE.json config (2-6 s of host time each at 32 threads) — the bound is the
# same, with the fp32 noise priced from the CPU result itself.
CPU_CASES = [((8, 64, 32, 32), torch.float32, "cnsn", "both"),         # configs[0]
             ((128, 32, 32, 32), torch.float32, "cnsn", "both"),       # configs[1]: WideResNet-40-2 site, armed
             ((256, 256, 56, 56), torch.bfloat16, "sn", "neither"),    # configs[2]: ResNet-50 layer-1 site, bf16
             ((96, 256, 56, 56), torch.bfloat16, "cnsn", "neither"),   # configs[3]: 3 x 32 views per GPU
             ((16, 256, 128, 128), torch.float32, "cn", "style"),      # configs[4]: segmentation layer 1, crop=style
             ((256, 256, 56, 56), torch.float32, "cnsn", "neither")]   # the headline workload itself


@pytest.mark.parametrize("shape,dtype,kind,crop", CPU_CASES, ids=lambda v: ids(v) if isinstance(v, tuple) else str(v).replace("torch.", ""))
def test_full_size_against_the_cpu_path(shape, dtype, kind, crop):
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    try:
        check_case(shape, dtype, kind, crop, 18, oracle32_on="cpu")
    finally:
        torch.set_num_threads(threads)


# the forced strategies at the full-size small-plane sites (AUTO picks one of them per direction; the others are the
# fall-backs a different N or a CrossNorm-armed step lands on)
@pytest.mark.parametrize("strategy", ["two_pass", "resident", "local", "mono"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("shape", [(256, 1024, 14, 14), (256, 2048, 7, 7), (256, 512, 28, 28)], ids=ids)
def test_small_plane_sites_every_strategy(shape, dtype, strategy):
    cnsn_amd.set_strategy(strategy)
    try:
        check_case(shape, dtype, "sn", "neither", 16)
        check_case(shape, dtype, "cnsn", "both", 17)
    finally:
        cnsn_amd.set_strategy("auto")
