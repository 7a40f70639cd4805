"""ResNet-50 (v1.5 bottlenecks: stride on the 3x3) with one CNSN unit per bottleneck — counterpart of
the reference's `models/imagenet/resnet_cnsn.py` (BottleneckCustom :37-124, ResNet :127-270,
resnet50 :309-323), same sub-module names / `state_dict` keys.

SelfNorm sites per forward at batch B, pos='post' (SURVEY.md §3.2): (B,256,56,56) x3, (B,512,28,28) x4,
(B,1024,14,14) x6, (B,2048,7,7) x3."""
import torch
import torch.nn as nn

from . import _sites
from ._sites import CrossNormSites, make_cnsn, residual_sum


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, impl, c_in, planes, stride, downsample, pos, beta, crop, cnsn_type):
        super().__init__()
        c_out = planes * self.expansion
        self.conv1 = nn.Conv2d(c_in, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, c_out, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(c_out)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        if cnsn_type is not None:                      # None: CrossNorm only in image space (:62)
            assert pos in ("residual", "pre", "post", "identity")
            self.cnsn = make_cnsn(impl, cnsn_type, crop, beta, c_in if pos == "pre" else c_out)   # :73-80
        self.pos = pos if cnsn_type is not None else None

    def forward(self, x):
        h = self.cnsn(x) if self.pos == "pre" else x
        h = self.relu(self.bn1(self.conv1(h)))
        h = self.relu(self.bn2(self.conv2(h)))
        h = self.conv3(h)
        # :99-122.  pos='post': bn3 (and the downsample's BatchNorm2d), the add, the CNSN unit and the ReLU are ONE call when the
        # unit offers it (this library's `CNSN.forward_bn_block`: one launch per direction on channels-last tensors, the
        # un-fused sequence otherwise)
        fb = getattr(getattr(self, "cnsn", None), "forward_bn_block", None) if (_sites.FUSE_BLOCK and self.pos == "post") else None
        if fb is not None and h.is_cuda:
            if self.downsample is None:
                return fb(h, self.bn3, x, relu=True)
            if len(self.downsample) == 2 and isinstance(self.downsample[1], nn.BatchNorm2d):
                return fb(h, self.bn3, self.downsample[0](x), relu=True, identity_bn=self.downsample[1])
            return fb(h, self.bn3, self.downsample(x), relu=True)
        skip = x if self.downsample is None else self.downsample(x)
        return residual_sum(getattr(self, "cnsn", None), self.pos, self.bn3(h), skip, relu=True)


class ResNet50CNSN(nn.Module, CrossNormSites):
    def __init__(self, num_classes=1000, layers=(3, 4, 6, 3), active_num=1, pos="post", beta=None, crop=None,
                 cnsn_type="sn", impl=None):
        super().__init__()
        if impl is None:
            from .. import cnsn as impl
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        kw = dict(pos=pos, beta=beta, crop=crop, cnsn_type=cnsn_type)
        c_in = 64
        for i, (planes, blocks) in enumerate(zip((64, 128, 256, 512), layers)):
            stride = 1 if i == 0 else 2
            units = []
            for b in range(blocks):
                down = None
                if b == 0 and (stride != 1 or c_in != planes * 4):
                    down = nn.Sequential(nn.Conv2d(c_in, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
                units.append(_Bottleneck(impl, c_in, planes, stride if b == 0 else 1, down, **kw))
                c_in = planes * 4
            setattr(self, f"layer{i + 1}", nn.Sequential(*units))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(c_in, num_classes)
        for m in self.modules():                                   # initialisation as :182-187
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._collect_sites(impl, cnsn_type, active_num)

    def forward(self, x, aug=False):
        if aug:
            self._enable_cross_norm()
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))
