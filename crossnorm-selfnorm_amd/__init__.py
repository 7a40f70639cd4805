"""MI355X-native CrossNorm / SelfNorm (the hot path of amazon-science/crossnorm-selfnorm,
`models/cnsn.py`) — hand-written HIP kernels behind a C ABI (`include/cnsn_hip.h`,
`libcnsn_hip.so`) and torch.nn.Module drop-ins with the reference's names and semantics.

Import as `cnsn_amd` (see /cnsn_amd.py; the directory name carries a hyphen).
"""
from .cnsn import (CNSN, CNDraws, CrossNorm, SelfNorm, calc_ins_mean_std, cn_op_2ins_space_chan,
                   cn_rand_bbox, draw_cn, instance_norm_mix)
from .functional import FusedConfig, GateParams, fused_cnsn, set_resident, set_strategy, sn_cluster, which_path
from ._ffi import LIB_PATH, CnsnError, follow_environ, lib, reload_env
from . import arena

__all__ = ["CNSN", "CrossNorm", "SelfNorm", "calc_ins_mean_std", "instance_norm_mix", "cn_rand_bbox",
           "cn_op_2ins_space_chan", "CNDraws", "draw_cn", "FusedConfig", "GateParams", "fused_cnsn",
           "set_strategy", "set_resident", "which_path", "sn_cluster", "lib", "LIB_PATH", "CnsnError", "reload_env", "follow_environ", "arena"]
