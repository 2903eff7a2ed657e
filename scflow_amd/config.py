"""The reference's SCFlow model config, restated as data.

Values follow configs/refine_models/scflow.py:16-113 of the reference (the
``model=`` dict).  Loss / renderer / init sub-configs are kept as opaque keys --
the hot path accepts and ignores them -- so that the very dict the reference's
``build_refiner`` takes also builds the HIP refiner.
"""
from __future__ import annotations

import copy

__all__ = ['scflow_model_cfg', 'raft_model_cfg']

_MODEL = dict(
    type='SCFlowRefiner',
    cxt_channels=128,
    h_channels=128,
    seperate_encoder=False,
    max_flow=400.,
    filter_invalid_flow=True,
    encoder=dict(type='RAFTEncoder', in_channels=3, out_channels=256, net_type='Basic',
                 norm_cfg=dict(type='IN'), init_cfg=None),
    cxt_encoder=dict(type='RAFTEncoder', in_channels=3, out_channels=256, net_type='Basic',
                     norm_cfg=dict(type='BN'), init_cfg=None),
    decoder=dict(
        type='SCFlowDecoder', net_type='Basic', num_levels=4, radius=4, iters=8,
        detach_flow=True, detach_mask=True, detach_pose=True, detach_depth_for_xy=True,
        mask_flow=False, mask_corr=False,
        pose_head_cfg=dict(type='MultiClassPoseHead', num_class=21, in_channels=224,
                           net_type='Basic', rotation_mode='ortho6d',
                           norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                           act_cfg=dict(type='ReLU')),
        corr_lookup_cfg=dict(align_corners=True), gru_type='SeqConv',
        act_cfg=dict(type='ReLU')),
    flow_loss_cfg=dict(type='SequenceLoss'),
    pose_loss_cfg=dict(type='SequenceLoss'),
    mask_loss_cfg=dict(type='SequenceLoss'),
    freeze_bn=False,
    freeze_encoder=False,
    train_cfg=dict(),
    test_cfg=dict(iters=8),
    init_cfg=None,
)


def scflow_model_cfg(iters: int = 8) -> dict:
    cfg = copy.deepcopy(_MODEL)
    cfg['decoder']['iters'] = iters
    cfg['test_cfg'] = dict(iters=iters)
    return cfg


# configs/refine_models/raft.py:4-80 (the pose-free RAFT refiner: the only route that accepts
# non-256x256 crops, i.e. BASELINE configs[4]); loss / init keys kept opaque as above.
_RAFT_MODEL = dict(
    type='RAFTRefinerFlowMask',
    cxt_channels=128,
    h_channels=128,
    seperate_encoder=False,
    max_flow=400.,
    filter_invalid_flow_by_mask=True,
    filter_invalid_flow_by_depth=False,
    encoder=dict(type='RAFTEncoder', in_channels=3, out_channels=256, net_type='Basic',
                 norm_cfg=dict(type='IN'), init_cfg=None),
    cxt_encoder=dict(type='RAFTEncoder', in_channels=3, out_channels=256, net_type='Basic',
                     norm_cfg=dict(type='BN'), init_cfg=None),
    decoder=dict(type='RAFTDecoderMask', net_type='Basic', num_levels=4, radius=4, iters=12,
                 corr_lookup_cfg=dict(align_corners=True), gru_type='SeqConv',
                 act_cfg=dict(type='ReLU')),
    flow_loss_cfg=dict(type='SequenceLoss'),
    occlusion_loss_cfg=dict(type='SequenceLoss'),
    freeze_bn=False,
    train_cfg=dict(),
    test_cfg=dict(iters=12),
    init_cfg=None,
)


def raft_model_cfg(iters: int = 12) -> dict:
    cfg = copy.deepcopy(_RAFT_MODEL)
    cfg['decoder']['iters'] = iters
    cfg['test_cfg'] = dict(iters=iters)
    return cfg
