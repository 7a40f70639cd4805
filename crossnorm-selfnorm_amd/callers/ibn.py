"""Instance / Instance-Batch normalisation of the reference's IBN backbones on the plane-statistics
kernels of this library: `nn.InstanceNorm2d(C, affine=True)` and `IBN` (models/imagenet/resnet_ibn_cnsn.py:24-44,
:63) with the same attribute names and `state_dict` keys (`IN.weight`, `IN.bias`, `BN.*`).

InstanceNorm2d here = one plane-statistics launch (cnsn_plane_stats) + one per-plane affine launch
(cnsn_plane_affine); the gradient flows through both custom functions (cnsn_plane_stats_backward,
cnsn_plane_dot) and a handful of (N, C)-sized torch ops.  HIP device tensors only."""
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
        m = x.size(2) * x.size(3)
        assert m > 1, "InstanceNorm2d needs more than one value per plane"
        # the kernel returns sqrt(unbiased var + e): with e = eps*M/(M-1),  biased var + eps = std_u^2 * (M-1)/M
        mean, std_u = _F.PlaneStats.apply(x, self.eps * m / (m - 1.0), None, True)
        scale = 1.0 / (std_u * ((m - 1.0) / m) ** 0.5)
        if self.affine:
            scale = scale * self.weight.float().view(1, -1, 1, 1)
            shift = self.bias.float().view(1, -1, 1, 1) - mean * scale
        else:
            shift = -mean * scale
        return _F.PlaneAffine.apply(x, scale, shift)


class IBN(nn.Module):
    """Half the channels through InstanceNorm2d, the rest through BatchNorm2d (resnet_ibn_cnsn.py:24-44)."""

    def __init__(self, planes, ratio=0.5):
        super().__init__()
        self.half = int(planes * ratio)
        self.IN = InstanceNorm2d(self.half, affine=True)
        self.BN = nn.BatchNorm2d(planes - self.half)

    def forward(self, x):
        split = torch.split(x, self.half, 1)
        out1 = self.IN(split[0].contiguous())
        out2 = self.BN(split[1].contiguous())
        return torch.cat((out1, out2), 1)
