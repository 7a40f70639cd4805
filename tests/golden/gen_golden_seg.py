#!/usr/bin/env python3
"""G7: golden vectors of the IMPORTED segmentation backbone (build container only).

    python tests/golden/gen_golden_seg.py     # writes tests/golden/g7_seg.npz

`segmentation/model/cnsn_resnet.py::resnet50` in the configuration of config/gtav/gtav_fcn50_cnsn.yaml (dilated
layers 3/4, SelfNorm at 'residual', a separate CrossNorm at 'post' with crop='style', all four stages) on
(4,3,64,64) with the name-seeded fill of gen_golden_fill.fill_by_name: 'out' / 'aux' feature maps in train mode with
the sites idle and with one site armed from a recorded seed, and in eval mode; fp32 and fp64; each map stored as its
per-(n,c) spatial mean plus its first 8 channels.  Inputs, seeds and outputs only."""
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
sys.path.insert(0, "/root/reference/segmentation")
np.int = int

from tests.golden.gen_golden_fill import fill_by_name  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    import model.cnsn_resnet as ref  # noqa: E402

torch.set_num_threads(8)
KW = dict(replace_stride_with_dilation=[False, True, True], block_idxs="1_2_3_4", active_num=1, pos="residual", beta=1,
          crop="style", cnsn_type="cnsn", cn_pos="post")


def put(out, key, t):
    """a feature map is kept as its per-(n,c) spatial mean plus the first 8 channels in full (small fixture)"""
    out[key + "_pool"] = t.mean((2, 3)).numpy()
    out[key + "_head"] = t[:, :8].contiguous().numpy()


def main():
    out = {}
    g = torch.Generator().manual_seed(4242)
    x = torch.randn(4, 3, 64, 64, generator=g, dtype=torch.float64)
    out["x"] = x.float().numpy()
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        with contextlib.redirect_stdout(io.StringIO()):
            net = ref.resnet50(pretrained=False, **KW)
        out["keys"] = np.array([f"{k}|{tuple(v.shape)}" for k, v in net.state_dict().items()])
        out["cn_num"] = np.array(net.cn_num)
        fill_by_name(net, 3).to(dt)
        xi = x.float().to(dt)
        net.train()
        with torch.no_grad():
            o = net(xi)
            put(out, f"{tag}_train_out", o["out"])
            put(out, f"{tag}_train_aux", o["aux"])
            torch.manual_seed(91)
            np.random.seed(91)
            net._enable_cross_norm()
            out[f"{tag}_armed"] = np.array([i for i, m in enumerate(net.cn_modules) if m.active])
            o = net(xi)
            put(out, f"{tag}_aug_out", o["out"])
            put(out, f"{tag}_aug_aux", o["aux"])
            net.eval()
            o = net(xi)
            put(out, f"{tag}_eval_out", o["out"])
    np.savez_compressed(os.path.join(HERE, "g7_seg.npz"), **out)
    print("g7_seg.npz", os.path.getsize(os.path.join(HERE, "g7_seg.npz")), "bytes; sites", int(out["cn_num"]),
          "armed", out["f32_armed"])


if __name__ == "__main__":
    main()
