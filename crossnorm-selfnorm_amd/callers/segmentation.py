"""Dilated ResNet-50 backbone of the reference's segmentation branch — counterpart of
`segmentation/model/cnsn_resnet.py` (BottleneckCustom :214-309, ResNet :313-468, resnet50 :503-512): v1.5
bottlenecks with dilation, SelfNorm (through a CNSN unit) at `pos` and, when `cn_pos` is given, a SEPARATE
CrossNorm after the block's ReLU (`real_cn`, :261-262, :305-306) — the configuration of
`config/gtav/gtav_fcn50_cnsn.yaml`: pos='residual', cn_pos='post', crop='style', block_idxs='1_2_3_4'.
Same sub-module names and `state_dict` keys; returns {'out': layer4, 'aux': layer3} like `_forward_impl` (:453-467).
Also the poly learning-rate rule of `segmentation/util/util.py:102-105`.

The FCN head itself is torchvision's `FCNHead` in the reference (model/fcn.py:24-37); `FCNHead` below restates its
five layers so that the backbone can be exercised end to end without torchvision."""
import numpy as np
import torch
import torch.nn as nn

from ._sites import residual_sum


class _SegBottleneck(nn.Module):
    expansion = 4

    def __init__(self, impl, c_in, planes, stride, downsample, dilation, custom, pos, cn_pos, beta, crop, cnsn_type):
        super().__init__()
        c_out = planes * self.expansion
        self.conv1 = nn.Conv2d(c_in, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, c_out, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(c_out)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.custom = custom
        self.pos = self.cn_pos = None
        if custom:
            assert cnsn_type in ("sn", "cn", "cnsn")
            cross = impl.CrossNorm(crop=crop, beta=beta) if ("cn" in cnsn_type and cn_pos is None) else None   # :238-241
            selfn = impl.SelfNorm(c_out) if "sn" in cnsn_type else None      # (pos='pre' uses `in_planes`, undefined there)
            self.cnsn = impl.CNSN(selfnorm=selfn, crossnorm=cross)
            if "cn" in cnsn_type and cn_pos is not None:
                self.real_cn = impl.CrossNorm(beta=beta, crop=crop)          # :261-262
            self.cn_pos = cn_pos
            self.pos = pos
            assert pos in ("residual", "identity", "pre", "post")

    def forward(self, x):
        h = self.cnsn(x) if (self.custom and self.pos == "pre") else x
        h = self.relu(self.bn1(self.conv1(h)))
        h = self.relu(self.bn2(self.conv2(h)))
        h = self.bn3(self.conv3(h))
        skip = x if self.downsample is None else self.downsample(x)
        if self.custom and self.pos == "residual":
            out = residual_sum(self.cnsn, "residual", h, skip, relu=True)   # cnsn(out); out += identity; relu  (:296-303)
        else:
            if self.custom and self.pos == "identity":
                skip = self.cnsn(h)                                          # (sic, :298-299: applied to `out`)
            out = torch.relu(h + skip)
        if self.custom:
            if self.pos == "post":
                out = self.cnsn(out)                                         # after the ReLU here (:304-305)
            if self.cn_pos == "post":
                out = self.real_cn(out)
        return out


class SegResNet50CNSN(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), replace_stride_with_dilation=(False, True, True), block_idxs="1_2_3_4",
                 active_num=1, pos="residual", beta=1, crop="style", cnsn_type="cnsn", cn_pos="post", num_classes=1000,
                 impl=None):
        super().__init__()
        if impl is None:
            from .. import cnsn as impl
        idxs = [int(v) for v in block_idxs.split("_")] if block_idxs else []
        self.block_idxs = idxs
        if 0 in idxs:
            self.img_cn = impl.CrossNorm(crop=crop, beta=beta)               # :352-353
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        c_in, dilation = 64, 1
        kw = dict(pos=pos, cn_pos=cn_pos, beta=beta, crop=crop, cnsn_type=cnsn_type)
        for i, (planes, blocks) in enumerate(zip((64, 128, 256, 512), layers)):
            stride = 1 if i == 0 else 2
            prev_dilation = dilation
            if i > 0 and replace_stride_with_dilation[i - 1]:                # :411-414
                dilation *= stride
                stride = 1
            custom = (i + 1) in idxs
            units = []
            for b in range(blocks):
                down = None
                if b == 0 and (stride != 1 or c_in != planes * 4):
                    down = nn.Sequential(nn.Conv2d(c_in, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
                units.append(_SegBottleneck(impl, c_in, planes, stride if b == 0 else 1, down,
                                            prev_dilation if b == 0 else dilation, custom, **kw))
                c_in = planes * 4
            setattr(self, f"layer{i + 1}", nn.Sequential(*units))
        self.fc = nn.Linear(c_in, num_classes)
        self.cn_modules = []
        for m in self.modules():                                             # :384-394
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.data.zero_()
            elif isinstance(m, impl.CrossNorm):
                self.cn_modules.append(m)
        if "cn" in cnsn_type or cn_pos is not None:
            self.cn_num = len(self.cn_modules)
            self.active_num = active_num
            assert self.cn_num > 0 and self.active_num > 0

    def _enable_cross_norm(self):                                            # :439-447
        picked = np.random.choice(self.cn_num, self.active_num, replace=False).tolist()
        for i in picked:
            self.cn_modules[i].active = True
        if 0 in self.block_idxs:
            self.img_cn.active = True
        return picked

    def _disable_cross_norm(self):                                           # :449-451
        for m in self.cn_modules:
            m.active = False

    def forward(self, x, aug=False):                                         # :453-471 (sites are armed by the trainer,
        if 0 in self.block_idxs:                                             #  tool/train_cnsn.py:305-310)
            x = self.img_cn(x)
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer3(self.layer2(self.layer1(x)))
        aux = x
        return {"out": self.layer4(x), "aux": aux}


class FCNHead(nn.Sequential):
    """torchvision.models.segmentation.fcn.FCNHead: 3x3 conv to C/4, BN, ReLU, Dropout(0.1), 1x1 conv to classes."""

    def __init__(self, in_channels, channels):
        inter = in_channels // 4
        super().__init__(nn.Conv2d(in_channels, inter, 3, padding=1, bias=False), nn.BatchNorm2d(inter), nn.ReLU(),
                         nn.Dropout(0.1), nn.Conv2d(inter, channels, 1))


def poly_learning_rate(base_lr, curr_iter, max_iter, power=0.9):
    """segmentation/util/util.py:102-105."""
    return base_lr * (1 - float(curr_iter) / max_iter) ** power
