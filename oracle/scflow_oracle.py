"""Plain-torch fp32 CPU restatement of the SCFlow refinement hot path.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py`` for who may import this.

Every function cites the reference lines (relative to the reference checkout)
whose arithmetic it restates.  The restatement is functional: weights come
from a flat ``state_dict`` (reference key layout), there
is no ``nn.Module``, no registry and no mmcv.  Operation ORDER follows the
reference so results agree to fp32 round-off with the reference run on the
same inputs (checked by tests/test_oracle_golden.py against
tests/golden/*.npz).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

__all__ = [
    'conv_act', 'raft_encoder', 'correlation_pyramid', 'corr_lookup',
    'motion_encoder', 'sepconv_gru', 'xhead', 'multiclass_pose_head',
    'rotation_from_ortho6d', 'pose_from_delta_pose', 'unproject_depth',
    'flow_from_pose_and_points', 'scflow_decoder', 'extract_feat',
    'get_pose', 'end_point_error', 'convex_upsample', 'raft_decoder', 'raft_decoder_mask',
    'cal_epe', 'flow_from_delta_pose_and_depth', 'coords_grid', 'filter_flow_by_mask',
    'eval_pose_error', 'eval_rot_error', 'eval_tran_error',
]


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def _act(x: Tensor, act: str | None) -> Tensor:
    if act is None:
        return x
    if act == 'relu':
        return torch.relu(x)
    if act == 'sigmoid':
        return torch.sigmoid(x)
    if act == 'tanh':
        return torch.tanh(x)
    raise ValueError(act)


def conv_act(x: Tensor, sd: SD, prefix: str, *, stride=1, padding=0,
             act: str | None = 'relu') -> Tensor:
    """conv -> (bias) -> activation: mmcv ``ConvModule`` without a norm layer
    as instantiated at raft_decoder.py:141-148 (MotionEncoder), :202-221
    (ConvGRU), :273-277 (XHead) and scflow_decoder.py:113-121.  Order
    conv->act, bias present because no norm (mmcv bias='auto')."""
    w = sd[prefix + '.conv.weight']
    b = sd.get(prefix + '.conv.bias')
    return _act(F.conv2d(x, w, b, stride=stride, padding=padding), act)


def _norm(x: Tensor, sd: SD, prefix: str, kind: str) -> Tensor:
    """norm layer of the RAFT encoder.  'IN' -> InstanceNorm2d(eps=1e-5,
    affine=False, no running stats) (scflow.py:27), 'BN' -> BatchNorm2d in
    eval mode (scflow.py:42; inference path)."""
    if kind == 'IN':
        return F.instance_norm(x, eps=1e-5)
    if kind == 'BN':
        return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                            sd[prefix + '.weight'], sd[prefix + '.bias'],
                            training=False, eps=1e-5)
    raise ValueError(kind)


def _basic_block(x: Tensor, sd: SD, p: str, kind: str, stride: int) -> Tensor:
    """backbone/resnet.py:67-94 ``BasicBlock.forward``; both 3x3 convs carry a
    bias (:36-48); downsample = 1x1 strided conv WITH bias + norm
    (resnet.py:721-730)."""
    tag = 'in' if kind == 'IN' else 'bn'
    out = F.conv2d(x, sd[p + '.conv1.weight'], sd[p + '.conv1.bias'], stride=stride, padding=1)
    out = torch.relu(_norm(out, sd, f'{p}.{tag}1', kind))
    out = F.conv2d(out, sd[p + '.conv2.weight'], sd[p + '.conv2.bias'], padding=1)
    out = _norm(out, sd, f'{p}.{tag}2', kind)
    if (p + '.downsample.0.weight') in sd:
        idt = F.conv2d(x, sd[p + '.downsample.0.weight'], sd[p + '.downsample.0.bias'], stride=stride)
        idt = _norm(idt, sd, p + '.downsample.1', kind)
    else:
        idt = x
    return torch.relu(out + idt)


def raft_encoder(x: Tensor, sd: SD, prefix: str, kind: str) -> Tensor:
    """encoder/raft_encoder.py:286-314 for net_type='Basic': 7x7 s2 stem
    (:210-219) -> norm -> relu -> three ResLayers of two BasicBlocks with
    strides (1,2,2) (:68-78, :134-160) -> 1x1 conv to out_channels
    (:162-168)."""
    tag = 'in' if kind == 'IN' else 'bn'
    p = prefix
    x = F.conv2d(x, sd[p + 'conv1.weight'], sd[p + 'conv1.bias'], stride=2, padding=3)
    x = torch.relu(_norm(x, sd, f'{p}{tag}1', kind))
    for li, stride in enumerate((1, 2, 2), start=1):
        x = _basic_block(x, sd, f'{p}res_layer{li}.0', kind, stride)
        x = _basic_block(x, sd, f'{p}res_layer{li}.1', kind, 1)
    return F.conv2d(x, sd[p + 'conv2.weight'], sd[p + 'conv2.bias'])


# --------------------------------------------------------------------------
# correlation volume + lookup
# --------------------------------------------------------------------------
def correlation_pyramid(feat1: Tensor, feat2: Tensor, num_levels: int = 4) -> List[Tensor]:
    """decoder/raft_decoder.py:35-58: all-pairs dot product over channels,
    divided by sqrt(C); view (N*h*w, 1, h, w); (num_levels-1) cascaded 2x2
    average pools over the TARGET dims."""
    n, c, h, w = feat1.shape
    a = feat1.reshape(n, c, h * w).transpose(1, 2)
    b = feat2.reshape(n, c, h * w)
    corr = torch.matmul(a, b).reshape(n * h * w, 1, h, w) / torch.sqrt(torch.tensor(float(c)))
    pyr = [corr]
    for _ in range(num_levels - 1):
        pyr.append(F.avg_pool2d(pyr[-1], kernel_size=2, stride=2))
    return pyr


def corr_lookup(pyramid: Sequence[Tensor], flow: Tensor, radius: int = 4) -> Tensor:
    """utils/corr_lookup.py:102-136 with ``bilinear_sample`` :31-67.

    Per level l the window centre is (x+flow_x, y+flow_y)/2**l (:127); the
    (2r+1)^2 integer offsets come from ``stack(meshgrid(dy, dx), -1)`` (:122)
    so offset[a, b] = (a - r, b - r) is added as (x += a-r, y += b-r): the
    SLOW window index walks x.  Coordinates are normalised to [-1, 1]
    (:64-65) and sampled with grid_sample(bilinear, zeros,
    align_corners=True) (:67); the normalise/de-normalise fp32 round trip is
    kept on purpose.  Output channel = 81*l + 9*a + b (:131-136)."""
    n, _, h, w = flow.shape
    r = radius
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    base = torch.stack([xs, ys], dim=0).float()[None]          # (1,2,h,w) = (x, y)
    centre = (base + flow).permute(0, 2, 3, 1).reshape(n * h * w, 1, 1, 2)
    d = torch.linspace(-r, r, 2 * r + 1)
    da, db = torch.meshgrid(d, d, indexing='ij')
    delta = torch.stack([da, db], dim=-1)[None]                # (1,9,9,2): [...,0] -> x slot
    outs = []
    for lvl, corr in enumerate(pyramid):
        hl, wl = corr.shape[-2:]
        coords = centre / 2 ** lvl + delta
        gx = coords[..., 0] * 2. / max(wl - 1, 1) - 1.
        gy = coords[..., 1] * 2. / max(hl - 1, 1) - 1.
        samp = F.grid_sample(corr, torch.stack([gx, gy], dim=-1), mode='bilinear',
                             padding_mode='zeros', align_corners=True)
        outs.append(samp.view(n, h, w, -1))
    return torch.cat(outs, dim=-1).permute(0, 3, 1, 2).contiguous().float()


# --------------------------------------------------------------------------
# update block
# --------------------------------------------------------------------------
def motion_encoder(corr: Tensor, flow: Tensor, sd: SD, p: str) -> Tensor:
    """decoder/raft_decoder.py:152-166, 'Basic' shapes (:76-86): corr 1x1
    324->256, 3x3 256->192; flow 7x7 2->128, 3x3 128->64; out 3x3 256->126;
    all ReLU (scflow.py:74); result cat [out(126), flow(2)]."""
    c = conv_act(corr, sd, p + 'corr_net.0', padding=0)
    c = conv_act(c, sd, p + 'corr_net.1', padding=1)
    f = conv_act(flow, sd, p + 'flow_net.0', padding=3)
    f = conv_act(f, sd, p + 'flow_net.1', padding=1)
    o = conv_act(torch.cat([c, f], dim=1), sd, p + 'out_net.0', padding=1)
    return torch.cat([o, flow], dim=1)


def sepconv_gru(h: Tensor, x: Tensor, sd: SD, p: str) -> Tensor:
    """decoder/raft_decoder.py:235-253.  'SeqConv': pass 0 uses (1,5)/pad(0,2)
    kernels, pass 1 (5,1)/pad(2,0) (:180-181); 'Conv': one pass of 3x3/pad 1.  The
    passes and their 'same' paddings are read off the state dict (conv_z.{i} shapes)."""
    i = 0
    while f'{p}conv_z.{i}.conv.weight' in sd:
        kh, kw = sd[f'{p}conv_z.{i}.conv.weight'].shape[-2:]
        pad = (kh // 2, kw // 2)
        hx = torch.cat([h, x], dim=1)
        z = conv_act(hx, sd, f'{p}conv_z.{i}', padding=pad, act='sigmoid')
        r = conv_act(hx, sd, f'{p}conv_r.{i}', padding=pad, act='sigmoid')
        q = conv_act(torch.cat([r * h, x], dim=1), sd, f'{p}conv_q.{i}', padding=pad, act='tanh')
        h = (1 - z) * h + z * q
        i += 1
    return h


def xhead(h: Tensor, sd: SD, p: str, kind: str) -> Tensor:
    """decoder/raft_decoder.py:292-294; one 3x3 ConvModule(ReLU) to 256 ch
    (the feat_channels tuple check at scflow_decoder.py:73-74 always yields
    [256]) then predict_layer: 3x3 for 'flow', 1x1 for 'mask' (:279-287)."""
    y = conv_act(h, sd, p + 'layers.0', padding=1)
    pad = 1 if kind == 'flow' else 0
    return F.conv2d(y, sd[p + 'predict_layer.weight'], sd[p + 'predict_layer.bias'], padding=pad)


def multiclass_pose_head(x: Tensor, label: Tensor, sd: SD, p: str, num_class: int = 21,
                         rot_dim: int = 6, num_groups: int = 32, label_mode: int = 0) -> Tuple[Tensor, Tensor]:
    """head/pose_head.py:201-211.  Three 3x3 stride-2 convs (no bias) each
    followed by GroupNorm(32) + ReLU (:148-160), flatten, FC 2048->1024->256
    with ReLU (:166-172), two linear heads, view (N, num_class, .) and
    ``index_select(dim=1, index=label)[:, 0]`` (:209-210).

    Reference quirk reproduced as-is:
    index_select with an N-vector yields (N, N, .) and ``[:, 0]`` keeps the
    entry for ``label[0]`` -> every sample is decoded with class label[0].
    ``label_mode=1`` is NOT the reference: it is the per-sample selection the
    ``index_select`` was evidently meant to be (sample n -> class label[n]), the
    checker for ``MultiClassPoseHead.label_mode = 1`` of the HIP path."""
    for i in range(3):
        x = F.conv2d(x, sd[f'{p}conv_layers.{i}.conv.weight'], None, stride=2, padding=1)
        x = F.group_norm(x, num_groups, sd[f'{p}conv_layers.{i}.gn.weight'],
                         sd[f'{p}conv_layers.{i}.gn.bias'], eps=1e-5)
        x = torch.relu(x)
    x = x.flatten(1)
    for i in range(2):
        x = torch.relu(F.linear(x, sd[f'{p}fc_layers.{i}.0.weight'], sd[f'{p}fc_layers.{i}.0.bias']))
    t = F.linear(x, sd[p + 'translation_pred.weight'], sd[p + 'translation_pred.bias'])
    r = F.linear(x, sd[p + 'rotation_pred.weight'], sd[p + 'rotation_pred.bias'])
    t = t.view(-1, num_class, 3)
    r = r.view(-1, num_class, rot_dim)
    if label_mode:
        idx = torch.arange(t.shape[0])
        return r[idx, label], t[idx, label]
    t = torch.index_select(t, 1, label)[:, 0, :]
    r = torch.index_select(r, 1, label)[:, 0, :]
    return r, t


# --------------------------------------------------------------------------
# pose math
# --------------------------------------------------------------------------
def rotation_from_ortho6d(o6: Tensor) -> Tensor:
    """utils/pose.py:153-169: x = normalize(a); z = normalize(x X b);
    y = z X x; columns [x y z]."""
    a, b = o6[:, 0:3], o6[:, 3:6]
    x = F.normalize(a, p=2, dim=1)
    z = F.normalize(torch.cross(x, b, dim=1), p=2, dim=1)
    y = torch.cross(z, x, dim=1)
    return torch.stack([x, y, z], dim=2)


def pose_from_delta_pose(d_rot: Tensor, d_trans: Tensor, rot: Tensor, trans: Tensor,
                         weight: float = 10., depth_transform: str = 'exp') -> Tuple[Tensor, Tensor]:
    """utils/pose.py:124-149, ortho6d branch (scflow.py:68; decoder default
    depth_transform 'exp' scflow_decoder.py:60):
    R' = R_delta @ R; z' = z / exp(dz)  ('exp', :137-138)  or  z * (dz + 1)
    (any other value, :139-141); x' = z' * (dx / 10 + x / z), same y."""
    r_new = torch.bmm(rotation_from_ortho6d(d_rot), rot)
    if depth_transform == 'exp':
        vz = trans[:, 2] / torch.exp(d_trans[:, 2])
    else:
        vz = trans[:, 2] * (d_trans[:, 2] + 1)
    vx = vz * torch.addcdiv(d_trans[:, 0] / weight, trans[:, 0], trans[:, 2])
    vy = vz * torch.addcdiv(d_trans[:, 1] / weight, trans[:, 1], trans[:, 2])
    return r_new, torch.stack([vx, vy, vz], dim=-1)


def unproject_depth(depth: Tensor, k: Tensor, rot: Tensor, trans: Tensor) -> Tuple[Tensor, Tensor]:
    """utils/pose.py:44-64 ``cal_3d_2d_corr`` + :26-41 ``lift_2d_to_3d`` for one
    sample: foreground = depth > 0 in row-major order; P_cam = K^-1 [x y 1]^T d;
    P_obj = R^-1 (P_cam - t).  Returns (M,2) xy pixel coords and (M,3) points."""
    ys, xs = torch.nonzero(depth > 0, as_tuple=True)
    d = depth[ys, xs]
    homo = torch.stack([xs.float(), ys.float(), torch.ones_like(d)], dim=-1) * d[:, None]
    cam = torch.mm(torch.inverse(k), homo.t()).t()
    obj = torch.mm(torch.inverse(rot), (cam - trans[None]).t()).t()
    return torch.stack([xs, ys], dim=-1).float(), obj


def flow_from_pose_and_points(rot: Tensor, trans: Tensor, k: Tensor, pts2d: Sequence[Tensor],
                              pts3d: Sequence[Tensor], height: int, width: int,
                              invalid_num: float = 400.) -> Tensor:
    """utils/pose.py:66-88: p = K (R P + t); (u, v) = p_xy / p_z;
    flow[:, y, x] = (u - x, v - y); background = invalid_num; no z>0 guard."""
    n = rot.shape[0]
    flow = rot.new_full((n, 2, height, width), invalid_num)
    for i in range(n):
        p2, p3 = pts2d[i], pts3d[i]
        proj = torch.mm(k[i], torch.mm(rot[i], p3.t()) + trans[i][:, None]).t()
        u, v = proj[:, 0] / proj[:, 2], proj[:, 1] / proj[:, 2]
        yi, xi = p2[:, 1].long(), p2[:, 0].long()
        flow[i, 0, yi, xi] = u - p2[:, 0]
        flow[i, 1, yi, xi] = v - p2[:, 1]
    return flow


# --------------------------------------------------------------------------
# decoder loop and top-level entry
# --------------------------------------------------------------------------
def scflow_decoder(feat_render: Tensor, feat_real: Tensor, h_feat: Tensor, cxt_feat: Tensor,
                   ref_rotation: Tensor, ref_translation: Tensor, depth: Tensor,
                   internel_k: Tensor, label: Tensor, init_flow: Tensor, sd: SD, *,
                   prefix: str = 'decoder.', iters: int = 8, num_levels: int = 4,
                   radius: int = 4, invalid_flow_num: float = 0.,
                   mask_flow: bool = False, mask_corr: bool = False,
                   depth_transform: str = 'exp', label_mode: int = 0):
    """decoder/scflow_decoder.py:150-251 (inference: the detach_* flags only
    affect autograd).  ``label_mode=1``: see ``multiclass_pose_head`` (not the reference)."""
    p = prefix
    pyramid = correlation_pyramid(feat_render, feat_real, num_levels)           # :172
    rot, trans = ref_rotation, ref_translation
    scale = 2 ** (num_levels - 1)
    n, H, W = depth.shape
    flow = init_flow
    pts = [unproject_depth(depth[i], internel_k[i], ref_rotation[i], ref_translation[i])
           for i in range(n)]                                                   # :184-187
    pts2d, pts3d = [a for a, _ in pts], [b for _, b in pts]
    mask = F.interpolate(torch.ones((n, 1, H, W), dtype=init_flow.dtype),
                         scale_factor=(1 / scale, 1 / scale), mode='bilinear',
                         align_corners=True)                                    # :188-190
    outs = dict(flow_from_pose=[], flow_from_pred=[], rotation=[], translation=[],
                mask=[], delta_rotation=[], delta_translation=[])
    for _ in range(iters):
        flow = 1 / scale * F.interpolate(flow, scale_factor=(1 / scale, 1 / scale),
                                         mode='bilinear', align_corners=True)   # :196-197
        corr = corr_lookup(pyramid, flow, radius)                               # :198
        if mask_corr:
            corr = corr * mask
        motion = motion_encoder(corr, flow * mask if mask_flow else flow, sd, p + 'encoder.')
        h_feat = sepconv_gru(h_feat, torch.cat([cxt_feat, motion], dim=1), sd, p + 'gru.')
        d_flow = xhead(h_feat, sd, p + 'flow_pred.', 'flow')                    # :210
        mask = torch.sigmoid(xhead(h_feat, sd, p + 'mask_pred.', 'mask'))       # :212-213
        df = conv_act(d_flow, sd, p + 'delta_flow_encoder.0', padding=3)        # :216
        df = conv_act(df, sd, p + 'delta_flow_encoder.1', padding=1)
        mf = conv_act(mask, sd, p + 'mask_encoder.0', padding=1)                # :217
        mf = conv_act(mf, sd, p + 'mask_encoder.1', padding=1)
        d_rot, d_trans = multiclass_pose_head(torch.cat([h_feat, df, mf], dim=1), label,
                                              sd, p + 'pose_pred.', label_mode=label_mode)   # :218-219
        flow_pred = scale * F.interpolate(flow + d_flow, scale_factor=(scale, scale),
                                          mode='bilinear', align_corners=True)  # :222-224
        up_mask = F.interpolate(mask, scale_factor=(scale, scale), mode='bilinear',
                                align_corners=True)                             # :226-227
        rot, trans = pose_from_delta_pose(d_rot, d_trans, rot, trans,
                                          depth_transform=depth_transform)      # :230-236
        flow = flow_from_pose_and_points(rot, trans, internel_k, pts2d, pts3d, H, W,
                                         invalid_num=invalid_flow_num)          # :239-243
        outs['rotation'].append(rot)
        outs['translation'].append(trans)
        outs['delta_rotation'].append(d_rot)
        outs['delta_translation'].append(d_trans)
        outs['flow_from_pose'].append(flow)
        outs['flow_from_pred'].append(flow_pred)
        outs['mask'].append(up_mask)
    return (outs['flow_from_pose'], outs['flow_from_pred'], outs['rotation'],
            outs['translation'], outs['mask'], outs['delta_rotation'],
            outs['delta_translation'])


def extract_feat(render_images: Tensor, real_images: Tensor, sd: SD, *, h_channels=128,
                 cxt_channels=128):
    """refiner/scflow_refiner.py:88-110: shared IN encoder on real and rendered
    image (seperate_encoder=False -> same weights, base_refiner.py:36-39, key
    prefix 'render_encoder.'), BN context encoder on the rendered image, split
    -> tanh / relu."""
    real = raft_encoder(real_images, sd, 'real_encoder.', 'IN')
    rend = raft_encoder(render_images, sd, 'render_encoder.', 'IN')
    cxt = raft_encoder(render_images, sd, 'context.', 'BN')
    h, c = torch.split(cxt, [h_channels, cxt_channels], dim=1)
    return rend, real, torch.tanh(h), torch.relu(c)


def get_pose(render_images: Tensor, real_images: Tensor, ref_rotation: Tensor,
             ref_translation: Tensor, depth: Tensor, internel_k: Tensor, label: Tensor,
             sd: SD, *, iters: int = 8, init_flow: Tensor | None = None,
             mask_flow: bool = False, mask_corr: bool = False,
             depth_transform: str = 'exp', label_mode: int = 0, radius: int = 4):
    """refiner/scflow_refiner.py:112-142 ``SCFlowRefiner.get_pose``
    (invalid_flow_num = 0 at inference, :142); ``mask_flow`` / ``mask_corr``: the decoder's
    constructor switches (scflow_decoder.py:199-205, both False in configs/refine_models/scflow.py)."""
    fr, fl, h, c = extract_feat(render_images, real_images, sd)
    if init_flow is None:
        n, _, H, W = real_images.shape
        init_flow = torch.zeros((n, 2, H, W), dtype=torch.float32)
    return scflow_decoder(fr, fl, h, c, ref_rotation, ref_translation, depth, internel_k,
                          label, init_flow, sd, iters=iters, invalid_flow_num=0.,
                          mask_flow=mask_flow, mask_corr=mask_corr, radius=radius,
                          depth_transform=depth_transform, label_mode=label_mode)


def end_point_error(flow_a: Tensor, flow_b: Tensor, valid: Tensor | None = None) -> float:
    """mean L2 distance between two (N,2,H,W) flow fields over ``valid``
    pixels -- the quantity BASELINE.json's 'flow EPE within 1e-3' refers to
    (plain definition; the reference's ``cal_epe`` utils/flow.py:64-88 is a
    'next' row)."""
    d = torch.sqrt(((flow_a - flow_b) ** 2).sum(dim=1))
    if valid is not None:
        d = d[valid]
    return float(d.mean()) if d.numel() else 0.0


# --------------------------------------------------------------------------
# "next" rows: pose-free RAFT decoders, EPE metric
# --------------------------------------------------------------------------
def convex_upsample(x: Tensor, mask: Tensor, scale: int = 8, grid_size: int = 9,
                    x_mul: float = 1.0) -> Tensor:
    """decoder/raft_decoder.py:381-416 ``RAFTDecoder._upsample`` with a mask (same arithmetic
    in raft_decoder_mask.py:104-160): softmax over the ``grid_size`` logits of every
    sub-pixel, unfold(x_mul * x, 3x3, padding=1), weighted sum, pixel-shuffle to scale*H.
    (grid_size = 2*radius+1 = 9 only coincides with the 3x3 unfold because radius = 4.)"""
    n, c, h, w = x.shape
    side = int(math.sqrt(grid_size))
    m = torch.softmax(mask.view(n, 1, grid_size, scale, scale, h, w), dim=2)
    up = F.unfold(x_mul * x, [side, side], padding=1).view(n, c, grid_size, 1, 1, h, w)
    up = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(n, c, scale * h, scale * w)


def _raft_iteration(pyramid, flow, h_feat, cxt_feat, sd, p, radius):
    corr = corr_lookup(pyramid, flow, radius)
    motion = motion_encoder(corr, flow, sd, p + 'encoder.')
    h_feat = sepconv_gru(h_feat, torch.cat([cxt_feat, motion], dim=1), sd, p + 'gru.')
    return h_feat, flow + xhead(h_feat, sd, p + 'flow_pred.', 'flow')


def raft_decoder(feat1: Tensor, feat2: Tensor, flow: Tensor, h_feat: Tensor, cxt_feat: Tensor,
                 sd: SD, *, prefix: str = 'decoder.', iters: int = 12, num_levels: int = 4,
                 radius: int = 4) -> List[Tensor]:
    """decoder/raft_decoder.py:418-457 ``RAFTDecoder.forward`` ('Basic': convex up-sampling
    with mask = 0.25 * mask_pred(h))."""
    pyramid = correlation_pyramid(feat1, feat2, num_levels)
    scale = 2 ** (num_levels - 1)
    outs = []
    for _ in range(iters):
        h_feat, flow = _raft_iteration(pyramid, flow, h_feat, cxt_feat, sd, prefix, radius)
        mask = .25 * xhead(h_feat, sd, prefix + 'mask_pred.', 'mask')
        outs.append(convex_upsample(flow, mask, scale, 2 * radius + 1, x_mul=float(scale)))
    return outs


def raft_decoder_mask(feat1: Tensor, feat2: Tensor, flow: Tensor, h_feat: Tensor,
                      cxt_feat: Tensor, sd: SD, *, prefix: str = 'decoder.', iters: int = 12,
                      num_levels: int = 4, radius: int = 4):
    """decoder/raft_decoder_mask.py:163-208 ``RAFTDecoderMask.forward``: as RAFTDecoder plus an
    occlusion head (sigmoid) that is convex-up-sampled with the same mask."""
    pyramid = correlation_pyramid(feat1, feat2, num_levels)
    scale = 2 ** (num_levels - 1)
    flows, occs = [], []
    for _ in range(iters):
        h_feat, flow = _raft_iteration(pyramid, flow, h_feat, cxt_feat, sd, prefix, radius)
        occ = torch.sigmoid(xhead(h_feat, sd, prefix + 'occlusion_pred.', 'mask'))
        mask = .25 * xhead(h_feat, sd, prefix + 'mask_pred.', 'mask')
        flows.append(convex_upsample(flow, mask, scale, 2 * radius + 1, x_mul=float(scale)))
        occs.append(convex_upsample(occ, mask, scale, 2 * radius + 1))
    return flows, occs


def flow_from_delta_pose_and_depth(rotation_src: Tensor, translation_src: Tensor,
                                   rotation_dst: Tensor, translation_dst: Tensor,
                                   depth_src: Tensor, k: Tensor, invalid_num: float = 400.) -> Tensor:
    """utils/pose.py:92-121 ``get_flow_from_delta_pose_and_depth`` (ground-truth flow of the
    RAFT refiners, raft_refiner_flow_mask.py:180): un-project the source depth with the source
    pose (``cal_3d_2d_corr``), project with the destination pose, background = invalid_num."""
    n = rotation_src.shape[0]
    h, w = depth_src.shape[-2:]
    p2, p3 = [], []
    for i in range(n):
        a, b = unproject_depth(depth_src[i], k[i], rotation_src[i], translation_src[i])
        p2.append(a)
        p3.append(b)
    return flow_from_pose_and_points(rotation_dst, translation_dst, k, p2, p3, h, w, invalid_num)


def coords_grid(flow: Tensor) -> Tensor:
    """utils/warp.py:9-29: pixel grid + flow, normalised with (size-1) -> (N,H,W,2)."""
    b, _, h, w = flow.shape
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    grid = torch.stack([xx, yy], 0).float()[None].repeat(b, 1, 1, 1) + flow
    gx = grid[:, 0] * 2. / max(w - 1, 1) - 1.
    gy = grid[:, 1] * 2. / max(h - 1, 1) - 1.
    return torch.stack([gx, gy], -1)


def filter_flow_by_mask(flow: Tensor, gt_mask: Tensor, invalid_num: float = 400.,
                        mode: str = 'bilinear', align_corners: bool = False) -> Tensor:
    """utils/flow.py:6-26: a flow vector is invalid when both components are >= invalid_num or
    when the target-image mask sampled at its end point is < 0.9 (zeros padding).  NB the grid is
    normalised with (size-1) but sampled with ``align_corners=False`` by default, i.e. at
    (x+fx)*W/(W-1) - 0.5 -- restated as-is.  Returns a new tensor (the reference writes in
    place)."""
    flow = flow.clone()
    not_valid = (flow[:, 0] >= invalid_num) & (flow[:, 1] >= invalid_num)
    m = F.grid_sample(gt_mask[:, None].to(flow.dtype), coords_grid(flow), mode=mode,
                      padding_mode='zeros', align_corners=align_corners)
    not_valid = (m < 0.9) | not_valid[:, None]
    flow[not_valid.expand_as(flow)] = invalid_num
    return flow


def cal_epe(flow_tgt: Tensor, flow_pred: Tensor, mask, max_flow: float = 400,
            reduction: str = 'mean', threshs=(1, 3, 5)):
    """utils/flow.py:64-88 ``cal_epe`` restated AS-IS.  valid = |flow_tgt| < max_flow (and
    mask >= 0.5).  Known quirk kept on purpose: in the 'mean' branch the error of the VALID
    pixels is overwritten with 1e8 before the threshold ratios are taken (:79), so the
    '<t>px' entries count INVALID pixels below the threshold ('total_mean' does it right)."""
    mag = torch.sum(flow_tgt ** 2, dim=1).sqrt()
    valid = (mag < max_flow) & (mask >= 0.5) if mask is not None else (mag < max_flow)
    err = torch.sum((flow_tgt - flow_pred) ** 2, dim=1).sqrt()
    if reduction == 'none':
        return err * valid.to(err)
    acc = {}
    if reduction == 'mean':
        total = valid.sum(dim=(-1, -2)) + 1e-10
        acc['mean'] = (err * valid.to(err)).sum(dim=(-1, -2)) / total
        err = err.clone()
        err[valid] = 1e+8
        for t in threshs:
            acc[f'{t}px'] = (err < t).sum(dim=(-1, -2)) / total
    elif reduction == 'total_mean':
        total = valid.sum(dim=(-1, -2, -3)) + 1e-10
        acc['mean'] = (err * valid.to(err.dtype)).sum(dim=(-1, -2, -3)) / total
        for t in threshs:
            acc[f'{t}px'] = (err[valid] < t).sum() / total
    return acc


# --------------------------------------------------------------------------
# pose-error evaluation: numpy, like the reference
# --------------------------------------------------------------------------
def eval_pose_error(verts_list, gt_t, gt_r, pred_t, pred_r, labels, k, symmetry_types,
                    mesh_diameters):
    """datasets/base_dataset.py:378-424 ``BaseDataset.eval_pose_error`` + ``project_3d_point``
    (datasets/pose.py:18-78): per sample, with the model vertices of its class,
      3d error (ADD)   = mean_i || (R_gt v_i + t_gt) - (R_pr v_i + t_pr) ||
      3d error (ADD-S) = mean_i || gt_i - pred_{argmin_j ||gt_i - pred_j||} ||   (symmetric classes,
                         ``symmetry_types['cls_<label+1>']``)
      2d error         = mean_i || proj(gt_i) - proj(pred_i) ||, proj = K p, x / (z + 1e-8)
      normalised 3d    = 3d error / mesh diameter of the class.
    Returns (error_3d_normalized, error_2d, error_3d), float64 arrays of length N."""
    import numpy as np
    n = len(gt_t)
    e3n, e2, e3 = np.zeros(n), np.zeros(n), np.zeros(n)
    for c in np.unique(labels):
        idx = labels == c
        v = np.asarray(verts_list[c])

        def project(r, t):
            cam = np.matmul(r[idx], v.transpose()) + t[idx][..., None]          # (M,3,n)
            px = np.matmul(k[idx], cam).transpose((0, 2, 1))                   # (M,n,3)
            xy = px[..., :2] / (px[..., 2:3] + 1e-8)
            return xy, cam.transpose((0, 2, 1))

        g2, g3 = project(gt_r, gt_t)
        p2, p3 = project(pred_r, pred_t)
        if symmetry_types.get(f'cls_{c + 1}', False):
            p3 = np.stack([pp[np.argmin(np.linalg.norm(gg[:, None] - pp[None], axis=-1), axis=-1)]
                           for gg, pp in zip(g3, p3)], 0)
        err = np.linalg.norm(g3 - p3, axis=-1).mean(-1)
        e3n[idx] = err / mesh_diameters[c]
        e2[idx] = np.linalg.norm(g2 - p2, axis=-1).mean(-1)
        e3[idx] = err
    return e3n, e2, e3


def eval_rot_error(gt_r, pred_r):
    """datasets/pose.py:106-112: geodesic angle (degrees) of pred_r gt_r^-1."""
    import numpy as np
    c = 0.5 * (np.trace(np.matmul(pred_r, np.linalg.inv(gt_r)), axis1=1, axis2=2) - 1.0)
    return 180.0 * np.arccos(np.clip(c, -1.0, 1.0)) / np.pi


def eval_tran_error(gt_t, pred_t):
    """datasets/pose.py:114-119: (|dt|, |dz|, |dxy|)."""
    import numpy as np
    return (np.linalg.norm(gt_t - pred_t, axis=-1), np.abs(gt_t[:, -1] - pred_t[:, -1]),
            np.linalg.norm(gt_t[:, :2] - pred_t[:, :2], axis=-1))
