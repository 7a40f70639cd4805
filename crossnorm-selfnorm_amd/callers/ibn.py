"""Instance / Instance-Batch normalisation of the reference's IBN backbones on the plane-statistics
kernels of this library: `nn.InstanceNorm2d(C, affine=True)` and `IBN` (models/imagenet/resnet_ibn_cnsn.py:24-44,
:63) with the same attribute names and `state_dict` keys (`IN.weight`, `IN.bias`, `BN.*`).

InstanceNorm2d here = one plane-statistics launch (cnsn_plane_stats) + one per-plane affine launch
(cnsn_plane_affine) forward, and a dedicated backward (functional.InstanceNorm): per-plane sums of G and G*(x - mean)
(cnsn_plane_dot_shifted) + ONE apply launch (cnsn_plane_combine) — 3 + 5 tensor passes; (N, C)-sized torch ops in
between.  HIP device tensors only."""
import torch
import torch.nn as nn

from .. import functional as _F


class InstanceNorm2d(nn.Module):
    """nn.InstanceNorm2d(num_features, eps, affine=True, track_running_stats=False): per-(n,c) plane
    `(x - mean) / sqrt(biased var + eps) * weight[c] + bias[c]`."""

    def __init__(self, num_features, eps=1e-5, affine=True):
        super().__init__()
        self.num_features, self.eps, self.affine = num_features, eps, affine
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)

    def forward(self, x):
        assert x.dim() == 4 and x.size(1) == self.num_features
        return _F.InstanceNorm.apply(x, self.weight if self.affine else None, self.bias if self.affine else None, self.eps)


class IBN(nn.Module):
    """Half the channels through InstanceNorm2d, the rest through BatchNorm2d (resnet_ibn_cnsn.py:24-44)."""

    def __init__(self, planes, ratio=0.5):
        super().__init__()
        self.half = int(planes * ratio)
        self.IN = InstanceNorm2d(self.half, affine=True)
        self.BN = nn.BatchNorm2d(planes - self.half)

    def forward(self, x):
        # the first `half` channels are instance-normalised, the others batch-normalised, order kept
        # (`.narrow` views are strided: both normalisations want dense tensors, as the reference's split does)
        rest = x.size(1) - self.half
        y_in = self.IN(x.narrow(1, 0, self.half).contiguous())
        y_bn = self.BN(x.narrow(1, self.half, rest).contiguous())
        return torch.cat([y_in, y_bn], dim=1)
