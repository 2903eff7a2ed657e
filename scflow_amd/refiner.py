"""``SCFlowRefiner`` -- the drop-in boundary of the hot path.

Mirrors models/refiner/scflow_refiner.py:18-179 (+ base_refiner.py:17-64):
same registry name, constructor keys (``configs/refine_models/scflow.py:16-113``
applies unchanged), attribute names, ``extract_feat`` / ``get_pose`` /
``forward_single_pass`` signatures and return structure, same ``state_dict``
keys.  Renderer, losses, data formatting and PnP re-mapping are outside the hot
path: their config keys are accepted and ignored, and
``forward_single_pass`` consumes an already formatted ``data`` dict (what
``BaseRefiner.format_data_test`` produces, base_refiner.py:79-133).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple, Union

import torch

from . import ops
from .modules import HipModule, raft_encoder_pair
from .ops import ACT_RELU, ACT_TANH, small_work
from .registry import REFINERS, build_decoder, build_encoder

Tensor = torch.Tensor


@REFINERS.register_module()
class SCFlowRefiner(HipModule):
    def __init__(self, seperate_encoder: bool, cxt_channels: int, h_channels: int,
                 cxt_encoder: dict, encoder: dict, decoder: dict, renderer: Optional[dict] = None,
                 pose_loss_cfg: Optional[dict] = None, flow_loss_cfg: Optional[dict] = None,
                 mask_loss_cfg: Optional[dict] = None, max_flow: float = 400,
                 render_augmentations: Optional[list] = None, filter_invalid_flow: bool = True,
                 freeze_encoder: bool = False, freeze_bn: bool = False,
                 train_cfg: Optional[dict] = None, test_cfg: Optional[dict] = None,
                 init_cfg: Optional[Union[list, dict]] = None) -> None:
        super().__init__()
        self.seperate_encoder = seperate_encoder
        if seperate_encoder:
            self.render_encoder = build_encoder(encoder)
            self.real_encoder = build_encoder(encoder)
        else:                                   # base_refiner.py:36-39: one module, two names
            enc = build_encoder(encoder)
            self.render_encoder = enc
            self.real_encoder = enc
        self.decoder = build_decoder(decoder)
        self.context = build_encoder(cxt_encoder)
        self.renderer = None                    # pytorch3d renderer: out of scope
        self.max_flow = max_flow
        self.train_cfg = train_cfg or {}
        self.test_cfg = test_cfg or {}
        self.h_channels, self.cxt_channels = h_channels, cxt_channels
        assert self.h_channels == self.decoder.h_channels
        assert self.cxt_channels == self.decoder.cxt_channels
        assert self.h_channels + self.cxt_channels == self.context.out_channels
        self.filter_invalid_flow = filter_invalid_flow
        self.test_by_flow = self.test_cfg.get('by_flow', False)
        self.test_iter_num = self.test_cfg.get('iters', self.decoder.iters)
        self.eval()

    # -------------------------------------------------------------- features
    def extract_feat(self, render_images: Tensor, real_images: Tensor
                     ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        """scflow_refiner.py:88-110 -> (render_feat, real_feat, h_feat, cxt_feat).

        The shared feature encoder runs ONCE on the 2N stacked images (InstanceNorm is per
        sample, so this equals two separate passes); the context encoder's 1x1 head writes
        tanh(h) | relu(cxt) straight into the first 256 channels of the GRU input buffer."""
        n, _, H, W = render_images.shape
        dev = render_images.device
        ops._dev(render_images, 'render_images')        # GPU fp32 on the current device, or raise
        ops._dev(real_images, 'real_images')
        hc, cc = self.h_channels, self.cxt_channels
        sc = int(round(1 / self.context.scale))
        # allocated BEFORE the fork: the side branch writes it while the main stream keeps going, so
        # its memory must not be a block the allocator recycles from main-stream temporaries that
        # are enqueued after the fork (they could still be running when the side branch writes)
        hx = torch.empty((n, hc + cc + 128, H // sc, W // sc), dtype=torch.float32, device=dev)
        # ... and so is everything the side branch READS: a non-contiguous render_images is
        # materialised here, on the main stream, before the fork event (a copy enqueued after it
        # would not be ordered before the side branch's first read)
        rend = render_images.contiguous()
        ov_ctx = small_work(n, H, W, 'context')
        fork = ops.fork_point() if ov_ctx else None     # the context encoder may start from here
        if not self.seperate_encoder and ops.branch_mode(n, H, W, 'context') == 2 and ops._CONV_EVENTS is None:
            # r6: the context encoder's launches ride in the feature encoder's (modules.raft_encoder_pair)
            both = torch.empty((2 * n, 3, H, W), dtype=torch.float32, device=dev)
            ops.copy_channels(rend, both[:n])
            ops.copy_channels(real_images.contiguous(), both[n:])
            feats, _ = raft_encoder_pair(self.render_encoder, both, self.context, rend, out_c=hx[:, :hc + cc],
                                         head_act=ACT_TANH, head_act2=ACT_RELU, head_split=hc)
            return feats[:n], feats[n:], hx[:, :hc], hx[:, hc:hc + cc]
        if self.seperate_encoder:
            render_feat = self.render_encoder(rend)
            real_feat = self.real_encoder(real_images.contiguous())
        else:
            both = torch.empty((2 * n, 3, H, W), dtype=torch.float32, device=dev)
            ops.copy_channels(rend, both[:n])
            ops.copy_channels(real_images.contiguous(), both[n:])
            feats = self.render_encoder(both)
            render_feat, real_feat = feats[:n], feats[n:]
        br = ops.side_stream(ov_ctx, after=fork)        # small batches: next to the feature encoder
        with br:
            self.context(rend, out=hx[:, :hc + cc], head_act=ACT_TANH,
                         head_act2=ACT_RELU, head_split=hc)
        br.join()
        return render_feat, real_feat, hx[:, :hc], hx[:, hc:hc + cc]

    # ------------------------------------------------------------------ pose
    def get_pose(self, render_images: Tensor, real_images: Tensor, ref_rotation: Tensor,
                 ref_translation: Tensor, depth: Tensor, internel_k: Tensor, label: Tensor,
                 init_flow: Optional[Tensor] = None):
        """scflow_refiner.py:112-142 -> 7-tuple of length-``iters`` lists
        (flow_from_pose, flow_from_pred, rotation_preds, translation_preds, mask_preds,
        delta_rotation_preds, delta_translation_preds)."""
        feat_render, feat_real, h_feat, cxt_feat = self.extract_feat(render_images, real_images)
        if init_flow is None:
            n, _, H, W = real_images.shape
            init_flow = ops.constant((n, 2, H, W), 0.0, feat_render.device)      # read-only, filled once
        return self.decoder(feat_render, feat_real, h_feat, cxt_feat, ref_rotation,
                            ref_translation, depth.contiguous(), internel_k.contiguous(),
                            label=label, init_flow=init_flow, invalid_flow_num=0.,
                            _consume_state=True)      # h / cxt are ours: update them in place

    def forward_single_pass(self, data: Dict, data_batch: Optional[Dict] = None,
                            return_loss: bool = False) -> Dict:
        """scflow_refiner.py:146-179 minus ``remap_pose_to_origin_resoluaion`` (identity for the
        'adapt_intrinsic' pipeline of the config; cv2 EPnP otherwise -- out of scope)."""
        labels = data['labels']
        per_img = data['per_img_patch_num']
        # the reference's index_select (pose_head.py:209) raises on an out-of-range class id;
        # the pose-update kernel cannot raise (it clamps), so the check lives at this entry
        nc = self.decoder.pose_pred.num_class
        # (one reduction, one device->host transfer; skipped while the stream is being captured
        # into a hipGraph, where a synchronising read is illegal: validate before capturing)
        if labels.numel() and not (labels.is_cuda and torch.cuda.is_current_stream_capturing()):
            lo, hi = torch.stack(torch.aminmax(labels)).tolist()
            if lo < 0 or hi >= nc:
                raise IndexError(f'label out of range [0, {nc}): min {lo}, max {hi}')
        iters = self.decoder.iters
        self.decoder.iters = self.test_iter_num
        try:
            outs = self.get_pose(data['rendered_images'], data['real_images'],
                                 data['ref_rotations'], data['ref_translations'],
                                 data['rendered_depths'], data['internel_k'], labels)
        finally:
            self.decoder.iters = iters
        rot, trans = outs[2][-1], outs[3][-1]
        return dict(rotations=torch.split(rot, per_img), translations=torch.split(trans, per_img),
                    labels=torch.split(labels, per_img),
                    scores=torch.split(torch.ones_like(labels, dtype=torch.float32), per_img))

    def forward(self, data, data_batch=None, return_loss=False):
        if return_loss:
            raise NotImplementedError('training is outside the hot path (SURVEY.md section 2)')
        if self.test_cfg.get('cycles', 1) > 1:
            # base_refiner.py:250-258: every further cycle RE-RENDERS the object at the updated pose (update_data ->
            # pytorch3d renderer): outside the hot path.  Refuse instead of silently running one cycle.
            raise NotImplementedError("test_cfg['cycles'] > 1 needs the renderer between cycles (base_refiner.py:250-258); "
                                      'render outside and call forward_single_pass / get_pose once per cycle')
        return self.forward_single_pass(data, data_batch)


class _FlowRefinerBase(HipModule):
    """feature extraction + ``get_flow`` of the pose-free RAFT refiners
    (models/refiner/raft_refiner_flow_mask.py:88-133, raft_refiner_flow.py).  Their pose step
    is cv2 RANSAC-PnP on the CPU (models/utils/pose.py:203-249): out of scope, raises."""

    def __init__(self, seperate_encoder: bool, cxt_channels: int, h_channels: int,
                 cxt_encoder: dict, encoder: dict, decoder: dict, test_cfg: Optional[dict] = None,
                 **ignored) -> None:
        super().__init__()
        self.seperate_encoder = seperate_encoder
        if seperate_encoder:
            self.render_encoder = build_encoder(encoder)
            self.real_encoder = build_encoder(encoder)
        else:
            enc = build_encoder(encoder)
            self.render_encoder = enc
            self.real_encoder = enc
        self.decoder = build_decoder(decoder)
        self.context = build_encoder(cxt_encoder)
        self.h_channels, self.cxt_channels = h_channels, cxt_channels
        self.test_cfg = test_cfg or {}
        self.test_iter_num = self.test_cfg.get('iters', self.decoder.iters)
        self.eval()

    extract_feat = SCFlowRefiner.extract_feat

    def get_flow(self, render_images: Tensor, real_images: Tensor,
                 init_flow: Optional[Tensor] = None):
        """raft_refiner_flow_mask.py:120-133: init_flow defaults to zeros at 1/8 resolution."""
        feat_render, feat_real, h_feat, cxt_feat = self.extract_feat(render_images, real_images)
        if init_flow is None:
            b, _, h, w = feat_real.shape
            init_flow = ops.constant((b, 2, h, w), 0.0, feat_real.device)
        return self.decoder(feat_render, feat_real, init_flow, h_feat, cxt_feat, _consume_state=True)

    def solve_pose(self, *a, **k):
        raise NotImplementedError('RANSAC-PnP (cv2) pose solve is outside the HIP hot path')


@REFINERS.register_module()
class RAFTRefinerFlowMask(_FlowRefinerBase):
    """configs/refine_models/raft.py: RAFTDecoderMask -> (flows, occlusions)."""


@REFINERS.register_module()
class RAFTRefinerFlow(_FlowRefinerBase):
    """RAFTDecoder -> flows."""
