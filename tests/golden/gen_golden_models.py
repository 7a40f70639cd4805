#!/usr/bin/env python3
"""G6: model-level golden vectors from the IMPORTED reference backbones (build container only).

    python tests/golden/gen_golden_models.py     # writes tests/golden/g6_models.npz

WideResNet-40-2 (+CNSN, pos=post, crop=both, 18 sites) on (4,3,32,32) and ResNet-50 (+SN, pos=post)
on (4,3,224,224) with the name-seeded deterministic fill of gen_golden_fill.fill_by_name: logits in
train mode (aug False / aug True from a recorded seed) and eval mode, fp32 and fp64.  Inputs, seeds
and logits only — no reference source, no checkpoint."""
import contextlib
import io
import os
import sys

sys.dont_write_bytecode = True   # the reference tree is read-only material: leave no __pycache__ in it

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
np.int = int

from tests.golden.gen_golden_fill import fill_by_name  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):          # the reference prints per site
    from models.cifar.wideresnet_cnsn import WideResNet  # noqa: E402
    from models.imagenet.resnet_cnsn import resnet50     # noqa: E402

torch.set_num_threads(8)


def seeded(s):
    torch.manual_seed(s)
    np.random.seed(s)


def run(model, x, out, key, aug_seed):
    model.train()
    with torch.no_grad():
        out[f"{key}_train"] = model(x).numpy()
        if aug_seed is not None:
            seeded(aug_seed)
            out[f"{key}_train_aug"] = model(x, aug=True).numpy()
            out[f"{key}_aug_seed"] = np.array(aug_seed)
        model.eval()
        out[f"{key}_eval"] = model(x).numpy()


def main():
    out = {}
    g = torch.Generator().manual_seed(2024)
    xw = torch.randn(4, 3, 32, 32, generator=g, dtype=torch.float64)
    xr = torch.randn(4, 3, 224, 224, generator=g, dtype=torch.float64)
    out["wrn_x"] = xw.float().numpy()
    out["r50_x"] = xr.float().numpy()
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        with contextlib.redirect_stdout(io.StringIO()):
            wrn = WideResNet(40, 100, 2, active_num=2, pos="post", beta=1, crop="both", cnsn_type="cnsn")

            class Cfg:
                active_num, pos, beta, crop, cnsn_type = 1, "post", None, None, "sn"
            r50 = resnet50(Cfg)
        out["wrn_keys"] = np.array([f"{k}|{tuple(v.shape)}" for k, v in wrn.state_dict().items()])
        out["r50_keys"] = np.array([f"{k}|{tuple(v.shape)}" for k, v in r50.state_dict().items()])
        fill_by_name(wrn, 1).to(dt)
        fill_by_name(r50, 2).to(dt)
        run(wrn, xw.float().to(dt), out, f"wrn_{tag}", aug_seed=77)
        run(r50, xr.float().to(dt), out, f"r50_{tag}", aug_seed=None)
        # running statistics after those forwards (BatchNorm2d and SelfNorm's BatchNorm1d both moved)
        out[f"wrn_{tag}_rv_last"] = wrn.state_dict()["block3.layer.5.cnsn.selfnorm.g_bn.running_var"].double().numpy()
        out[f"r50_{tag}_rv_last"] = r50.state_dict()["layer4.2.cnsn.selfnorm.g_bn.running_var"].double().numpy()
    np.savez_compressed(os.path.join(HERE, "g6_models.npz"), **out)
    print("g6_models.npz", os.path.getsize(os.path.join(HERE, "g6_models.npz")), "bytes")
    for k in sorted(out):
        if "x" not in k and "keys" not in k:
            print(k, out[k].shape, float(np.abs(out[k]).max()))


if __name__ == "__main__":
    main()
