"""WideResNet-d-k with one CNSN unit per basic block — counterpart of the reference's
`models/cifar/wideresnet_cnsn.py` (BasicBlockCustom :12-98, WideResNet :137-227) with the same
sub-module names, so a reference checkpoint's `state_dict` maps 1:1.

WRN-40-2 at 32x32 has 18 CNSN sites: (B,32,32,32) x6, (B,64,16,16) x6, (B,128,8,8) x6 for pos='post'."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _sites
from ._sites import CrossNormSites, make_cnsn, residual_sum


class _Block(nn.Module):
    """Pre-activation basic block: bn-relu-conv3x3-bn-relu-conv3x3 (+1x1 shortcut when widths differ);
    the CNSN unit sits at `pos` in {'pre','residual','identity','post'} (wideresnet_cnsn.py:66-98)."""

    def __init__(self, impl, c_in, c_out, stride, pos, beta, crop, cnsn_type, drop_rate):
        super().__init__()
        assert pos in ("residual", "identity", "pre", "post")
        self.bn1 = nn.BatchNorm2d(c_in)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(c_in, c_out, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(c_out)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(c_out, c_out, 3, 1, 1, bias=False)
        self.same_width = c_in == c_out
        self.conv_shortcut = None if self.same_width else nn.Conv2d(c_in, c_out, 1, stride, 0, bias=False)
        width = c_in if (pos == "pre" and not self.same_width) else c_out      # :51-54
        self.cnsn = make_cnsn(impl, cnsn_type, crop, beta, width)
        self.pos, self.drop_rate = pos, drop_rate

    def cnsn_tail_ok(self):
        return self.pos == "post" and hasattr(self.cnsn, "forward_block_bn")

    def forward(self, x, pre=None, next_bn=None, want_y=True):
        """`pre`: relu1(bn1(x)) when the PREVIOUS block's CNSN launch already produced it (second half);
        `next_bn`: the bn1 of the block that follows (the network's last bn1 after the last block) — then this block
        returns `(y, z)` with z = relu(next_bn(y)) from its own CNSN launch (`CNSN.forward_block_bn`), y = None when
        `want_y` is False (the next block only consumes z: its widths differ, wideresnet_cnsn.py:69-70)."""
        if not self.same_width:
            x = pre if pre is not None else self.relu1(self.bn1(x))
        h = self.cnsn(x) if self.pos == "pre" else x
        if self.same_width:
            h = pre if (pre is not None and self.pos != "pre") else self.relu1(self.bn1(h))
        h = self.relu2(self.bn2(self.conv1(h)))
        if self.drop_rate > 0:
            h = F.dropout(h, p=self.drop_rate, training=self.training)
        h = self.conv2(h)
        skip = x if self.same_width else self.conv_shortcut(x)
        if next_bn is not None and self.pos == "post" and hasattr(self.cnsn, "forward_block_bn") and h.is_cuda:
            return self.cnsn.forward_block_bn(skip, h, "pre", next_bn, want_y=want_y)         # torch.add(x, out) -> cnsn -> bn1 -> relu
        y = residual_sum(self.cnsn, self.pos, h, skip, relu=False, skip_first=True)             # :86-96
        return y if next_bn is None else (y, None)


class _Stage(nn.Module):
    def __init__(self, impl, n, c_in, c_out, stride, **kw):
        super().__init__()
        self.layer = nn.Sequential(*[_Block(impl, c_in if i == 0 else c_out, c_out, stride if i == 0 else 1, **kw)
                                     for i in range(n)])

    def forward(self, x):
        return self.layer(x)


class WideResNetCNSN(nn.Module, CrossNormSites):
    def __init__(self, depth=40, num_classes=100, widen_factor=2, drop_rate=0.0, active_num=None, pos="post",
                 beta=1, crop="both", cnsn_type="cnsn", impl=None):
        super().__init__()
        if impl is None:
            from .. import cnsn as impl
        assert (depth - 4) % 6 == 0
        n = (depth - 4) // 6
        w = [16, 16 * widen_factor, 32 * widen_factor, 64 * widen_factor]
        kw = dict(pos=pos, beta=beta, crop=crop, cnsn_type=cnsn_type, drop_rate=drop_rate)
        self.conv1 = nn.Conv2d(3, w[0], 3, 1, 1, bias=False)
        self.block1 = _Stage(impl, n, w[0], w[1], 1, **kw)
        self.block2 = _Stage(impl, n, w[1], w[2], 2, **kw)
        self.block3 = _Stage(impl, n, w[2], w[3], 2, **kw)
        self.bn1 = nn.BatchNorm2d(w[3])
        self.relu = nn.ReLU(inplace=True)
        self.fc = nn.Linear(w[3], num_classes)
        self.n_channels = w[3]
        self.pos = pos
        for m in self.modules():                                   # initialisation as :179-187
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, math.sqrt(2.0 / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear) and m.bias is not None:
                m.bias.data.zero_()
        self._collect_sites(impl, cnsn_type, active_num)

    def forward(self, x, aug=False):
        if aug:
            self._enable_cross_norm()
        if not (_sites.FUSE_TAIL and self.pos == "post"):
            h = self.block3(self.block2(self.block1(self.conv1(x))))
            h = F.avg_pool2d(self.relu(self.bn1(h)), 8)
            return self.fc(h.view(h.size(0), -1))
        # every block hands `relu(bn1(.))` of the block that follows (:76-77 / :69-70; after the last one: :222) out of
        # its own CNSN launch, next to its output; module ownership — and with it every state_dict key — is unchanged
        blocks = [b for stage in (self.block1, self.block2, self.block3) for b in stage.layer]
        h, pre = self.conv1(x), None
        for i, blk in enumerate(blocks):
            last = i + 1 == len(blocks)
            nxt = self.bn1 if last else blocks[i + 1].bn1
            want_y = (not last) and blocks[i + 1].same_width and blocks[i + 1].pos != "pre"   # y is the next shortcut (:93)
            y, z = blk(h, pre, next_bn=nxt, want_y=want_y or not blk.cnsn_tail_ok())
            if z is None:                     # (no tail offered at this site this step: the next block does it itself)
                h, pre = y, None
                if last:
                    z = self.relu(self.bn1(y))
            else:
                h, pre = (y if y is not None else z), z   # (y None: the next block never looks at h itself)
        return self.fc(F.avg_pool2d(z, 8).view(z.size(0), -1))
