"""Callers of the CNSN hot path: minimal counterparts of the two backbones
the reference's configs name — WideResNet-40-2 (CIFAR) and ResNet-50 (ImageNet) — with a CNSN unit at
exactly the sites, widths and `state_dict` keys of the reference model files, plus the training-step
structure (random CrossNorm site activation, 3-view JSD consistency).  Stock `nn.Conv2d` /
`nn.BatchNorm2d` (MIOpen) everywhere else: only the CNSN path is this repository's own kernels."""
from .ibn import IBN, InstanceNorm2d
from .resnet import ResNet50CNSN
from .segmentation import FCNHead, SegResNet50CNSN, poly_learning_rate
from .steps import (GraphedIdleStep, StepGuard, image_space_crossnorm, jsd_consistency, train_step_cn, train_step_cn_consistency,
                    train_step_image_cn_views)
from .wideresnet import WideResNetCNSN

__all__ = ["WideResNetCNSN", "ResNet50CNSN", "SegResNet50CNSN", "FCNHead", "poly_learning_rate", "IBN", "InstanceNorm2d", "jsd_consistency", "train_step_cn", "train_step_cn_consistency",
           "image_space_crossnorm", "train_step_image_cn_views", "GraphedIdleStep", "StepGuard"]
